"""Scaled conjugate gradients with the big per-point parameters kept on the shards: Python-3 restatement of the
reference optimiser ``scg_adapted.SCG_adapted`` (scg_adapted.py:78-336) and its failure wrapper ``safe_f_and_grad_f``
(:44-76).  The optimiser logic is host Python exactly as in the reference (it is not part of the hot path); what changes
is WHERE the per-shard vector algebra runs: the reference's helper module reads and writes (2,N_s,Q) ``.grad_*.npy`` files
(scg_adapted_local_MapReduce.py:29-243); here ``ops`` is any object with the same function names -- for the GPU build
``gparml_amd.resident.ResidentCG`` whose vectors never leave HBM.
"""
import numpy as np


def safe_f_and_grad_f(f_and_gradf, x, iteration=0, step_size=0, *optargs):
    """scg_adapted.py:44-76: numerical failures become f = inf, grad = ones."""
    try:
        return f_and_gradf(x, iteration, step_size, *optargs)
    except (np.linalg.LinAlgError, ZeroDivisionError, ValueError, Warning, AssertionError, FloatingPointError):
        return np.inf, np.ones(x.shape[0])


def SCG_adapted(f_and_gradf, x, ops, fixed_embeddings=False, optargs=(), maxiters=500, max_f_eval=500, display=False,
                xtol=None, ftol=None, gtol=None, folder=None):
    xtol = 1e-6 if xtol is None else xtol
    ftol = 1e-6 if ftol is None else ftol
    gtol = 1e-5 if gtol is None else gtol
    sigma0 = 1.0e-4
    f, g = safe_f_and_grad_f(f_and_gradf, x, 0, 0, *optargs)
    assert f != float('inf')
    fold = fnow = f
    function_eval = 1
    gradnew = g
    gradold = gradnew.copy()
    d = -gradnew
    if not fixed_embeddings:
        ops.embeddings_set_grads(folder)
    current_grad = np.dot(gradnew, gradnew)
    if not fixed_embeddings:
        current_grad += ops.embeddings_get_grads_current_grad(folder)
    success, nsuccess = True, 0
    beta, betamin, betamax = 1.0, 1.0e-60, 1.0e100
    status = 'Not converged'
    flog = [fold]
    iteration = 0
    while iteration < maxiters:
        if success:
            mu = np.dot(d, gradnew)
            if not fixed_embeddings:
                mu += ops.embeddings_get_grads_mu(folder)
            if mu >= 0:
                d = -gradnew
                if not fixed_embeddings:
                    ops.embeddings_set_grads_reset_d(folder)
                mu = np.dot(d, gradnew)
                if not fixed_embeddings:
                    mu += ops.embeddings_get_grads_mu(folder)
            kappa = np.dot(d, d)
            if not fixed_embeddings:
                kappa += ops.embeddings_get_grads_kappa(folder)
            sigma = sigma0 / np.sqrt(kappa)
            xplus = x + sigma * d
            gplus = safe_f_and_grad_f(f_and_gradf, xplus, -1, sigma, *optargs)[1]
            theta = np.dot(d, gplus - gradnew)
            if not fixed_embeddings:
                theta += ops.embeddings_get_grads_theta(folder)
            theta = theta * np.sqrt(kappa) / sigma0
        delta = theta + beta * kappa
        if delta <= 0:
            delta = beta * kappa
            beta = beta - theta / kappa
        alpha = -mu / delta
        xnew = x + alpha * d
        fnew, gnew_vec = safe_f_and_grad_f(f_and_gradf, xnew, iteration + 1, alpha, *optargs)
        function_eval += 1
        if function_eval >= max_f_eval:
            status = 'Maximum number of function evaluations exceeded'
            break
        Delta = 2. * (fnew - fold) / (alpha * mu)
        if Delta >= 0.:
            success = True
            nsuccess += 1
            x = xnew
            if not fixed_embeddings:
                ops.embeddings_set_grads_update_X(folder, alpha)
            fnow = fnew
        else:
            success = False
            fnow = fold
        flog.append(fnow)
        iteration += 1
        if display:
            print(' %4d   %.6e   %.3e   %.3e' % (iteration, fnow, beta, current_grad))
        if success:
            max_alpha_d = np.max(np.abs(alpha * d))
            if not fixed_embeddings:
                max_alpha_d = max(max_alpha_d, ops.embeddings_get_grads_max_d(folder, alpha))
            if (max_alpha_d < xtol) or (np.abs(fnew - fold) < ftol):
                status = 'converged'
                break
            gradold = gradnew
            if not fixed_embeddings:
                ops.embeddings_set_grads_update_grad_old(folder)
            gradnew = gnew_vec
            if not fixed_embeddings:
                ops.embeddings_set_grads_update_grad_new(folder)
            current_grad = np.dot(gradnew, gradnew)
            if not fixed_embeddings:
                current_grad += ops.embeddings_get_grads_current_grad(folder)
            fold = fnew
            if current_grad <= gtol:
                status = 'converged'
                break
        if Delta < 0.25:
            beta = min(4.0 * beta, betamax)
        if Delta > 0.75:
            beta = max(0.5 * beta, betamin)
        if nsuccess == x.size:
            d = -gradnew
            nsuccess = 0
        elif success:
            Gamma = (np.dot(gradold, gradnew) - current_grad) / mu
            if not fixed_embeddings:
                Gamma += ops.embeddings_get_grads_gamma(folder) / mu
            d = Gamma * d - gradnew
            if not fixed_embeddings:
                ops.embeddings_set_grads_update_d(folder, Gamma)
    else:
        status = 'maxiter exceeded'
    return x, flog, function_eval, status
