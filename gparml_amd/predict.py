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
    def __init__(self, global_statistics, accumulated_statistics, N_train, D, device=0, partial_terms_class=None):
        """global_statistics: dict Z (M,Q), sf2, alpha, beta; accumulated_statistics: the five base sums of the trained model
        (the ``accumulated_statistics_*_f.npy`` files, predict.py:31-35).  ``partial_terms_class`` (tests only: a CPU class with the
        reference's constructor) replaces the GPU class so that the host logic can be checked without a device."""
        self._cls = partial_terms_class
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
            args = (numpy.asarray(g['Z'], dtype=float), f(g['sf2']), numpy.asarray(g['alpha'], dtype=float).reshape(-1), f(g['beta']),
                    self.M, self.Q, self.N, self.D)
            if self._cls is not None:
                self._pt = self._cls(*args)
            else:
                self._pt = partial_terms(*args, update_global_statistics=False, device=self.device)
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

    # ---- initialisation of the new points (predict.py:37-72) ------------------------------------------------------------------------
    @staticmethod
    def nearest_training_embeddings(Y_test, training, Q, mask=None):
        """predict.py:44-66: every new point starts at the trained embedding of the training output nearest to it (Euclidean distance over
        the output columns in ``mask``, all columns by default), searched shard by shard with a k-d tree (leaf size 100) and the
        reference's cut-off of 6: a new point farther than that from every training output keeps a zero mean.  ``training`` yields
        (Y_shard, X_shard) pairs -- one shard in memory at a time, ties between shards go to the first one as in the reference."""
        import scipy.spatial
        Y_test = numpy.atleast_2d(numpy.asarray(Y_test, dtype=float))
        cols = list(range(Y_test.shape[1])) if mask is None else list(mask)
        best = numpy.full(Y_test.shape[0], numpy.inf)
        X_mu = numpy.zeros((Y_test.shape[0], Q))
        for Y, X in training:
            Y = numpy.asarray(Y, dtype=float)
            if Y.ndim == 1:
                Y = numpy.atleast_2d(Y).T                                           # predict.py:55-56
            tree = scipy.spatial.cKDTree(Y[:, cols], leafsize=100)
            dist, ind = tree.query(Y_test[:, cols], k=1, distance_upper_bound=6)
            closer = dist < best                                                    # strict: an equally near point of a later shard does not win
            best[closer] = dist[closer]
            X_mu[closer] = numpy.asarray(X)[ind[closer]]
        return X_mu

    def _optimise(self, X_mu0, X_S0, iterations):
        x0 = numpy.concatenate((X_mu0.flatten(), X_S0.flatten()))
        x0 = numpy.array([transform_back(b, x) for b, x in zip(self.bounds, x0)])
        x, flog, nfe, status = SCG_adapted(self.likelihood_and_gradient, x0, None, fixed_embeddings=True, maxiters=iterations)
        t = numpy.array([transform(b, y) for b, y in zip(self.bounds, x)])
        n = len(t) // 2
        return [t[:n].reshape(self.shape), t[n:].reshape(self.shape), -flog[-1]]

    def test(self, Y_test, X_mu0=None, X_S0=None, iterations=100, training=None, mask=None, is_random_init=False, random_restarts=100):
        """predict.test (predict.py:19-111) -> [X_mu, X_S, likelihood] of the new points.

        Starting mean: ``X_mu0`` when given; otherwise, as the reference, the embedding of the nearest training output (``training`` =
        iterable of (Y_shard, X_shard), ``mask`` = output columns to compare; no restarts, predict.py:43) or, with ``is_random_init``, a
        random inducing point, followed by ``random_restarts`` further optimisations from other random inducing points that keep the best
        likelihood (predict.py:93-108; like the reference this branch starts ONE row of Z, so it serves a single new point).  Starting
        variance 0.5 + 0.01 randn clipped to [0.001, 1] (predict.py:71-72), drawn once; a restart begins at the best variances so far.  The global numpy
        random stream is consumed in the reference's order (index, variances, one index per restart)."""
        self.Y_test = numpy.atleast_2d(numpy.asarray(Y_test, dtype=float))
        Z = numpy.asarray(self.gs['Z'], dtype=float)
        if X_mu0 is not None:
            X_mu0, random_restarts = numpy.atleast_2d(numpy.asarray(X_mu0, dtype=float)), 0
        elif is_random_init:
            X_mu0 = numpy.atleast_2d(Z[numpy.random.randint(self.M)])               # predict.py:38-41
        else:
            if training is None:
                raise AssertionError('predict.test needs the training shards (training=...) or an initial mean (X_mu0=...)')
            random_restarts = 0                                                     # predict.py:43
            X_mu0 = self.nearest_training_embeddings(self.Y_test, training, self.Q, mask)
        self.shape = X_mu0.shape
        if X_S0 is None:
            X_S0 = numpy.clip(numpy.ones(self.shape) * 0.5 + 0.01 * numpy.random.randn(*self.shape), 0.001, 1)   # predict.py:71-72
        X_S0 = numpy.asarray(X_S0, dtype=float)
        n = int(numpy.prod(self.shape))
        self.bounds = [(None, None)] * n + [(0, None)] * n
        best = self._optimise(X_mu0, X_S0, iterations)
        for _ in range(int(random_restarts) if is_random_init else 0):              # predict.py:93-108
            # the restart begins at the variances of the best result so far, not at the first draw (predict.py:87, 97-98, 105: X_S is rebound)
            trial = self._optimise(numpy.atleast_2d(Z[numpy.random.randint(self.M)]), best[1], iterations)
            if trial[2] > best[2]:
                best = trial
        return best


def test(options_, Y_test_, mask=None, is_random_init=False, random_iterations=100, random_restarts=100, device=0, map_reduce=None):
    """The reference's entry point (predict.test, predict.py:19-111) with its signature: mean, variance and likelihood of new points given
    the trained model that ``options`` describes (its ``statistics`` directory holds global_statistics_*_f.npy and
    accumulated_statistics_*_f.npy, its ``input`` / ``embeddings`` directories the training shards).  options['N'] must be populated."""
    import os
    from . import driver
    if map_reduce is None:
        from . import gpu_MapReduce as map_reduce
    options = dict(options_)
    options['load'] = True
    options, gs = driver.init_statistics(map_reduce, options)                       # predict.py:28-29 (the load branch)
    acc = {key: map_reduce.load(options['statistics'] + '/accumulated_statistics_' + key + '_f.npy')
           for key in ('sum_YYT', 'sum_exp_K_mi_K_im', 'sum_exp_K_miY', 'sum_exp_K_ii', 'sum_KL')}
    p = Predictor(gs, acc, options['N'], options['D'], device=device)

    def training():
        for name in sorted(os.listdir(options['input'] + '/')):                     # one shard in memory at a time
            yield (map_reduce._read_csv(options['input'] + '/' + name),
                   map_reduce.load(options['embeddings'] + '/' + name + '.embedding.npy'))

    return p.test(Y_test_, iterations=random_iterations, training=None if is_random_init else training(), mask=mask,
                  is_random_init=is_random_init, random_restarts=random_restarts)
