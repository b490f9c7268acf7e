"""Test-time latent inference on the GPU: Python-3 restatement of the reference's ``predict.py``.

``predict.likelihood_and_gradient`` (predict.py:116-144) optimises the variational mean and variance of NEW points against
the stored accumulated statistics of a trained model: the new points' local statistics are added to the stored global
sums, and the bound and ``grad_X_mu / grad_X_S`` are evaluated with the same ``partial_terms`` class.  Here that class is
``gparml_amd.partial_terms.partial_terms`` (all numbers from the HIP library); the optimiser is
``gparml_amd.scg_adapted.SCG_adapted`` with ``fixed_embeddings=True`` exactly as predict.py:82 does ("the globals are now
the embeddings").
"""
import numpy

from .driver import transform, transform_back, transform_grad
from .partial_terms import partial_terms
from .scg_adapted import SCG_adapted


class Predictor(object):
    def __init__(self, global_statistics, accumulated_statistics, N_train, D, device=0):
        """global_statistics: dict Z (M,Q), sf2, alpha, beta; accumulated_statistics: the five base sums of the trained model
        (the ``accumulated_statistics_*_f.npy`` files, predict.py:31-35)."""
        self.gs = global_statistics
        self.acc = accumulated_statistics
        Z = numpy.asarray(global_statistics['Z'], dtype=float)
        self.M, self.Q = Z.shape
        self.N, self.D = int(N_train), int(D)
        self.device = device
        self._pt = None

    def _partial_terms(self):
        if self._pt is None:
            g = self.gs
            f = lambda x: float(numpy.asarray(x).reshape(-1)[0])
            self._pt = partial_terms(numpy.asarray(g['Z'], dtype=float), f(g['sf2']), numpy.asarray(g['alpha'], dtype=float).reshape(-1),
                                     f(g['beta']), self.M, self.Q, self.N, self.D, update_global_statistics=False, device=self.device)
        return self._pt

    def likelihood_and_gradient(self, flat_array, iteration=0, step_size=0):
        """predict.py:116-144."""
        shape = self.shape
        bounds = self.bounds
        t = numpy.array([transform(b, x) for b, x in zip(bounds, flat_array)])
        half = len(t) // 2
        X_mu, X_S = t[:half].reshape(shape), t[half:].reshape(shape)
        pt = self._partial_terms()
        pt.set_data(self.Y_test, X_mu, X_S, is_set_statistics=True)
        new = pt.get_local_statistics()
        a = self.acc
        pt.set_local_statistics(a['sum_YYT'] + new['sum_YYT'], a['sum_exp_K_mi_K_im'] + new['sum_exp_K_mi_K_im'],
                                a['sum_exp_K_miY'] + new['exp_K_miY'], a['sum_exp_K_ii'] + new['sum_exp_K_ii'], a['sum_KL'] + new['KL'])
        likelihood = pt.logmarglik()
        gradient = numpy.concatenate((pt.grad_X_mu().flatten(), pt.grad_X_S().flatten()))
        gradient = numpy.array([g * transform_grad(b, x) for b, x, g in zip(bounds, flat_array, gradient)])
        return -1 * likelihood, -1 * gradient

    def test(self, Y_test, X_mu0, X_S0=None, iterations=100):
        """predict.test (predict.py:19-111) for a given initial mean (the reference takes the embedding of the nearest
        training output or a random inducing point -- host-side initialisation, out of the hot path)."""
        self.Y_test = numpy.atleast_2d(numpy.asarray(Y_test, dtype=float))
        X_mu0 = numpy.atleast_2d(numpy.asarray(X_mu0, dtype=float))
        self.shape = X_mu0.shape
        if X_S0 is None:
            X_S0 = numpy.clip(numpy.ones(self.shape) * 0.5 + 0.01 * numpy.random.randn(*self.shape), 0.001, 1)   # predict.py:71-72
        n = int(numpy.prod(self.shape))
        self.bounds = [(None, None)] * n + [(0, None)] * n
        x0 = numpy.concatenate((X_mu0.flatten(), numpy.asarray(X_S0, dtype=float).flatten()))
        x0 = numpy.array([transform_back(b, x) for b, x in zip(self.bounds, x0)])
        x, flog, nfe, status = SCG_adapted(self.likelihood_and_gradient, x0, None, fixed_embeddings=True, maxiters=iterations)
        t = numpy.array([transform(b, y) for b, y in zip(self.bounds, x)])
        return [t[:n].reshape(self.shape), t[n:].reshape(self.shape), -flog[-1]]
