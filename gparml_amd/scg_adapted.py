"""Scaled conjugate gradients over a parameter vector that is split in two: the small global part (Z, sf2, alpha, beta)
lives on the host, the big per-point part (embeddings and their variances) stays on the shards.  This is the algorithm of
the reference's ``scg_adapted.SCG_adapted`` (scg_adapted.py:78-336; failure wrapper :44-76) in Python 3 with every
inner product / update of the per-point part behind an ``ops`` object: the reference's helper module works on (2,N_s,Q)
``.grad_*.npy`` files (scg_adapted_local_MapReduce.py:29-243); ``gparml_amd.resident.ResidentCG`` offers the same
function names on vectors that never leave HBM.  The optimiser itself is host logic, as in
the reference -- it is not part of the hot path.
"""
import numpy as np

RECOVERABLE = (np.linalg.LinAlgError, ZeroDivisionError, ValueError, Warning, AssertionError, FloatingPointError)


_fail_count = 0               # consecutive failures, scg_adapted.py:44
_allowed_failures = 100       # scg_adapted.py:45


def safe_f_and_grad_f(f_and_gradf, x, iteration=0, step_size=0, *optargs):
    """scg_adapted.py:46-76: a numerical failure becomes f = inf, grad = ones -- at most ``_allowed_failures`` times in a row, after
    which the exception is re-raised ("Too many errors..."); a successful evaluation resets the count."""
    global _fail_count
    try:
        out = f_and_gradf(x, iteration, step_size, *optargs)
        _fail_count = 0
        return out
    except RECOVERABLE:
        if _fail_count >= _allowed_failures:
            raise
        _fail_count += 1
        return np.inf, np.ones(x.shape[0])


class _SplitVectors(object):
    """Inner products and updates of the (host part, shard part) vectors.  With fixed embeddings there is no shard part."""

    def __init__(self, ops, folder, local):
        self.ops, self.folder, self.local = ops, folder, local

    def _plus(self, host_value, getter, *args):
        # ``ops`` may be None when there is no shard part (fixed embeddings): look the getter up only when it is used
        return host_value + getattr(self.ops, getter)(self.folder, *args) if self.local else host_value

    def slope(self, d, g):                       # mu = d . grad_new
        return self._plus(np.dot(d, g), 'embeddings_get_grads_mu')

    def length2(self, d):                        # kappa = d . d
        return self._plus(np.dot(d, d), 'embeddings_get_grads_kappa')

    def curvature(self, d, g_probe, g):          # theta numerator = d . (grad(x + sigma d) - grad_new)
        return self._plus(np.dot(d, g_probe - g), 'embeddings_get_grads_theta')

    def norm2(self, g):                          # grad_new . grad_new
        return self._plus(np.dot(g, g), 'embeddings_get_grads_current_grad')

    def largest_move(self, alpha, d):            # max |alpha d|
        host = np.max(np.abs(alpha * d))
        return max(host, self.ops.embeddings_get_grads_max_d(self.folder, alpha)) if self.local else host

    def call(self, name, *args):
        if self.local:
            getattr(self.ops, name)(self.folder, *args)


def SCG_adapted(f_and_gradf, x, ops, fixed_embeddings=False, optargs=(), maxiters=500, max_f_eval=500, display=False,
                xtol=None, ftol=None, gtol=None, folder=None):
    """Returns (x, flog, function_eval, status) like the reference.  ``ops`` replaces the embeddings folder argument."""
    xtol = 1e-6 if xtol is None else xtol
    ftol = 1e-6 if ftol is None else ftol
    gtol = 1e-5 if gtol is None else gtol
    SIGMA0, BETA_MIN, BETA_MAX = 1.0e-4, 1.0e-60, 1.0e100
    vec = _SplitVectors(ops, folder, not fixed_embeddings)

    f_prev, g_new = safe_f_and_grad_f(f_and_gradf, x, 0, 0, *optargs)            # :108-113
    assert f_prev != float('inf')
    f_now, evaluations = f_prev, 1
    g_old = g_new.copy()
    d = -g_new
    vec.call('embeddings_set_grads')                                              # new = old = latest, d = -latest
    g_norm2 = vec.norm2(g_new)
    accepted, run_of_successes, beta = True, 0, 1.0
    status, flog, iteration = 'Not converged', [f_prev], 0
    mu = kappa = theta = 0.0

    while iteration < maxiters:
        if accepted:
            # second-order information along d from one extra gradient at x + sigma d (:131-166)
            mu = vec.slope(d, g_new)
            if mu >= 0:                                                           # not a descent direction: restart
                d = -g_new
                vec.call('embeddings_set_grads_reset_d')
                mu = vec.slope(d, g_new)
            kappa = vec.length2(d)
            sigma = SIGMA0 / np.sqrt(kappa)
            g_probe = safe_f_and_grad_f(f_and_gradf, x + sigma * d, -1, sigma, *optargs)[1]
            theta = vec.curvature(d, g_probe, g_new) * np.sqrt(kappa) / SIGMA0
        # make the local quadratic positive definite, then take its minimiser along d (:168-180)
        delta = theta + beta * kappa
        if delta <= 0:
            delta = beta * kappa
            beta = beta - theta / kappa
        alpha = -mu / delta
        x_trial = x + alpha * d
        f_trial, g_trial = safe_f_and_grad_f(f_and_gradf, x_trial, iteration + 1, alpha, *optargs)
        evaluations += 1
        if evaluations >= max_f_eval:
            status = 'Maximum number of function evaluations exceeded'
            break
        # comparison ratio of the actual and the predicted decrease (:188-205)
        Delta = 2. * (f_trial - f_prev) / (alpha * mu)
        accepted = Delta >= 0.
        if accepted:
            run_of_successes += 1
            x = x_trial
            vec.call('embeddings_set_grads_update_X', alpha)
            f_now = f_trial
        else:
            f_now = f_prev
        flog.append(f_now)
        iteration += 1
        if display:
            print(' %4d   %.6e   %.3e   %.3e' % (iteration, f_now, beta, g_norm2))
        if accepted:                                                              # :214-240
            if vec.largest_move(alpha, d) < xtol or np.abs(f_trial - f_prev) < ftol:
                status = 'converged'
                break
            g_old = g_new
            vec.call('embeddings_set_grads_update_grad_old')
            g_new = g_trial
            vec.call('embeddings_set_grads_update_grad_new')
            g_norm2 = vec.norm2(g_new)
            f_prev = f_trial
            if g_norm2 <= gtol:
                status = 'converged'
                break
        # trust-region style scale update (:242-248)
        if Delta < 0.25:
            beta = min(4.0 * beta, BETA_MAX)
        if Delta > 0.75:
            beta = max(0.5 * beta, BETA_MIN)
        # new direction: restart after n successes, else Polak-Ribiere-like update (:250-262)
        if run_of_successes == x.size:
            d = -g_new
            run_of_successes = 0
        elif accepted:
            Gamma = (np.dot(g_old, g_new) - g_norm2) / mu
            if vec.local:
                Gamma += ops.embeddings_get_grads_gamma(folder) / mu
            d = Gamma * d - g_new
            vec.call('embeddings_set_grads_update_d', Gamma)
    else:
        status = 'maxiter exceeded'
    return x, flog, evaluations, status
