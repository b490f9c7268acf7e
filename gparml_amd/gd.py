"""The reference's second optimiser (`--optimiser GD`, parallel_GPLVM.py:104-105; algorithm of gd.py:41-127): steepest
descent on the flat global vector and on the per-shard embeddings, with the step doubled after every accepted move and
halved after every rejected one.  The per-shard vector algebra is delegated to an ``ops`` object
(``gparml_amd.resident.ResidentGD`` keeps the vectors in HBM; any object with the function names of
gd_local_MapReduce.py works)."""
import numpy as np
from numpy.linalg import LinAlgError

RECOVERABLE = (LinAlgError, ZeroDivisionError, ValueError, Warning, AssertionError)   # gd.py:26
MAX_CONSECUTIVE_FAILURES = 100                                                          # gd.py:15-16


class GradientDescent(object):
    """State of one run: position ``x``, search direction ``d`` (global part), step length, bound history."""

    def __init__(self, objective, x0, ops, fixed_embeddings, optargs=(), ftol=1e-6, xtol=1e-16, gtol=1e-6, momentum=0.0):
        self.objective, self.ops, self.optargs = objective, ops, tuple(optargs)
        self.local = not fixed_embeddings                     # per-shard embeddings take part in the search
        self.ftol, self.xtol, self.gtol, self.momentum = ftol, xtol, gtol, momentum
        self.failures = 0
        self.step = 0.01                                      # gd.py:63
        self.x = np.asarray(x0, dtype=float)
        self.f, self.grad = self.evaluate(self.x, 0, 0)       # gd.py:66-69
        self.history = [self.f]
        self.d = -self.grad
        if self.local:
            ops.embeddings_set_grads(None)                    # grad_now = latest, d = -latest on every shard
        self.accepted = 0

    def evaluate(self, x, iteration, step):
        """gd.py:17-39: a recoverable error counts as an infinitely bad point with a gradient of ones."""
        try:
            f, g = self.objective(x, iteration, step, *self.optargs)
            self.failures = 0
            return f, g
        except RECOVERABLE:
            if self.failures >= MAX_CONSECUTIVE_FAILURES:
                raise
            self.failures += 1
            return np.inf, np.ones(x.shape)

    def gradient_size(self):
        """sum |gradient| over the global and the per-shard parts (what the reference prints, gd.py:113-116)."""
        total = float(np.sum(np.abs(self.grad)))
        return total + self.ops.embeddings_get_grads_current_grad(None) if self.local else total

    def advance(self):
        """One trial step; returns False when a stopping rule fires (gd.py:76-109)."""
        trial = self.x + self.step * self.d
        f_trial, g_trial = self.evaluate(trial, self.accepted, self.step)
        if abs(self.f - f_trial) < self.ftol or abs(self.step) < self.xtol:
            return False
        if f_trial > self.f:                                  # rejected: shorten and retry from the same point
            self.step /= 2.0
            return True
        self.f, self.grad, self.x = f_trial, g_trial, trial
        self.history.append(f_trial)
        gamma = self.momentum * self.step
        self.d = -(self.grad + gamma * self.d)
        if self.local:                                        # same order as gd.py:93-101
            self.ops.embeddings_set_grads_update_grad_now(None)
            self.ops.embeddings_set_grads_update_X(None, self.step)
            self.ops.embeddings_set_grads_update_d(None, gamma)
        self.step *= 2.0
        self.accepted += 1
        largest = float(np.max(np.abs(self.grad)))
        if self.local:
            largest = max(largest, self.ops.embeddings_get_grads_max_gradnow(None))
        return largest >= self.gtol


def GD(f_and_gradf, x, ops, fixed_embeddings=False, optargs=(), maxiters=500, max_f_eval=500, display=False, xtol=None, ftol=None,
       gtol=None):
    """Call-compatible with gd.GD (gd.py:41) except that ``ops`` replaces the embeddings folder; returns
    ``(x, flog, None, status)`` like the reference."""
    run = GradientDescent(f_and_gradf, x, ops, fixed_embeddings, optargs, ftol=1e-6 if ftol is None else ftol,
                          xtol=1e-16 if xtol is None else xtol, gtol=1e-6 if gtol is None else gtol)
    while run.accepted < maxiters and run.advance():
        if display:
            print('%d  %12e  %12e  %12e' % (run.accepted, float(run.f), float(run.step), run.gradient_size()))
    return run.x, run.history, None, 'converged... NOT'
