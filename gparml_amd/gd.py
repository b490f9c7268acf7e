"""Gradient descent with step doubling / halving: Python-3 restatement of the reference's second optimiser (gd.py:17-127,
``--optimiser GD`` in parallel_GPLVM.py:104-105) with the per-shard vector algebra behind a pluggable object
(``gparml_amd.resident.ResidentGD`` on the GPU, or any object with the function names of gd_local_MapReduce.py)."""
import numpy as np
from numpy.linalg import LinAlgError

_fail_count = 0
_allowed_failures = 100


def safe_f_and_grad_f(f_and_gradf, x, iteration=0, step_size=0, *optargs):
    """gd.py:17-39: f = inf and a gradient of ones when the evaluation raises one of the recoverable error classes."""
    global _fail_count
    try:
        f, gradf = f_and_gradf(x, iteration, step_size, *optargs)
        _fail_count = 0
    except (LinAlgError, ZeroDivisionError, ValueError, Warning, AssertionError):
        if _fail_count >= _allowed_failures:
            raise
        _fail_count += 1
        f = np.inf
        gradf = np.ones(x.shape)
    return f, gradf


def GD(f_and_gradf, x, ops, fixed_embeddings=False, optargs=(), maxiters=500, max_f_eval=500, display=False, xtol=None, ftol=None,
       gtol=None):
    """gd.py:41-127.  Returns (x, flog, None, status) like the reference."""
    xtol = 1e-16 if xtol is None else xtol
    ftol = 1e-6 if ftol is None else ftol
    gtol = 1e-6 if gtol is None else gtol
    step_size = 0.01
    mom_size = 0.0
    fnow, gradnow = safe_f_and_grad_f(f_and_gradf, x, 0, 0, *optargs)                 # :66-69
    flog = [fnow]
    direction = -gradnow
    if not fixed_embeddings:
        ops.embeddings_set_grads(None)
    iteration = 0
    while iteration < maxiters:
        xprop = x + step_size * direction                                                 # :76
        fproposed, gradprop = safe_f_and_grad_f(f_and_gradf, xprop, iteration, step_size, *optargs)
        if np.abs(fnow - fproposed) < ftol:                                               # :80-85
            break
        if np.abs(step_size) < xtol:
            break
        if fproposed <= fnow:                                                             # :87-107
            fnow = fproposed
            flog += [fnow]
            gradnow = gradprop
            if not fixed_embeddings:
                ops.embeddings_set_grads_update_grad_now(None)
            x = xprop
            if not fixed_embeddings:
                ops.embeddings_set_grads_update_X(None, step_size)
            direction = -(gradnow + mom_size * step_size * direction)
            if not fixed_embeddings:
                ops.embeddings_set_grads_update_d(None, mom_size * step_size)
            step_size *= 2.0
            iteration += 1
            max_abs_gradnow = np.max(np.abs(gradnow))
            if not fixed_embeddings:
                max_abs_gradnow = max(max_abs_gradnow, ops.embeddings_get_grads_max_gradnow(None))
            if max_abs_gradnow < gtol:
                break
        else:
            step_size /= 2.0                                                              # :108-109
        if display:
            current_grad = np.sum(np.abs(gradnow))
            if not fixed_embeddings:
                current_grad += ops.embeddings_get_grads_current_grad(None)
            print('%d  %12e  %12e  %12e' % (iteration, float(fnow), float(step_size), float(current_grad)))
    return x, flog, None, 'converged... NOT'
