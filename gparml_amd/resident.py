"""Device-resident model: every shard's data, embeddings, search direction and gradient vectors stay in HBM for the
whole optimisation -- the "full SCG loop that never touches the filesystem".

``ResidentModel.likelihood_and_gradient(x, iteration, step_size)`` has the optimiser callback contract of
parallel_GPLVM.py:222-279; ``ResidentCG`` offers the function names of scg_adapted_local_MapReduce.py:29-243 on the
resident vectors.  Across GPUs (one process per GPU) the local scalars are summed / maxed with torch.distributed.
"""
import numpy as np

from ._lib import JitterRetry
from .driver import positive_mask, transform_grad_vec, transform_vec
from .engine import ShardEngine


class ResidentModel(object):
    def __init__(self, shards, M, Q, D, fixed_embeddings=False, fixed_beta=False, device=0, dist_group=None, N_global=None):
        """shards: list of (Y, X_mu, X_S) held by THIS process (X_S raw = softplus-inverse space unless fixed_embeddings)."""
        self.M, self.Q, self.D = M, Q, D
        self.fixed_embeddings, self.fixed_beta = fixed_embeddings, fixed_beta
        self.engines = []
        for (Y, X_mu, X_S) in shards:
            e = ShardEngine(Y.shape[0], D, M, Q, device=device)
            e.set_timing(0)            # an optimiser does not read per-kernel device timings: no timing events on the stream
            e.upload_shard(Y, X_mu, X_S, xs_is_raw=not fixed_embeddings)
            self.engines.append(e)
        self.group = dist_group
        self._dist = None
        self.version = 0            # bumped whenever the resident vectors may have changed (ResidentCG caches its reductions on it)
        self.n_collectives = 0      # all-reduces issued so far (tests assert the per-iteration count)
        if dist_group is not None or self._dist_ready():
            import torch.distributed as dist
            self._dist = dist
        n_local = sum(e.N_s for e in self.engines)
        self.N = int(N_global) if N_global is not None else int(self._allreduce_scalar(float(n_local)))
        self.bounds = [(None, None)] * (M * Q) + [(0, None)] + [(0, None)] * Q + [(0, None)]
        self._pos = positive_mask(self.bounds)
        self._dev_tensors = None
        self._native = None         # True once the root engine reduces through its own RCCL communicator (gp_allreduce)

    @staticmethod
    def _dist_ready():
        import sys
        if 'torch.distributed' not in sys.modules:     # nobody in this process can have initialised a group: do not pay the torch import
            return False
        try:
            import torch.distributed as dist
            return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        except Exception:
            return False

    def _allreduce_scalar(self, v, op='sum'):
        return float(self._allreduce_vector([v], op)[0])

    def _allreduce_vector(self, values, op='sum'):
        """One collective for a small vector of local scalars (sum or max over ranks)."""
        values = np.asarray(values, dtype=np.float64)
        if self._dist is None:
            return values
        import torch
        # the engine's own GPU, not torch's current device (a caller need not have run torch.cuda.set_device)
        t = torch.tensor(values, dtype=torch.float64,
                         device=torch.device('cuda', self.engines[0].device) if self._dist.get_backend() == 'nccl' else 'cpu')
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM if op == 'sum' else self._dist.ReduceOp.MAX, group=self.group)
        self.n_collectives += 1
        return t.cpu().numpy()

    def _allreduce_buffers(self, which):
        if self._dist is None:
            return
        from .dist import device_tensor, init_native_comm
        import torch
        root = self.engines[0]
        if self._native is None:
            self._native = init_native_comm(root, self._dist, self.group)
        if self._native:
            root.allreduce(which)             # gp_allreduce: RCCL on the engine's stream (statistics packed inside)
            self.n_collectives += 1
            return
        if self._dev_tensors is None:         # zero-copy views of the two device buffers, made once (the pointers never change)
            p, n = root.stats_packed_buffer()
            g, m = root.grads_buffer()
            dev = torch.device('cuda', root.device)
            self._dev_tensors = {'stats': device_tensor(p, n, dev), 'grads': device_tensor(g, m, dev)}
        if which == 'stats':
            # across processes the statistics travel without padding and without Psi2's lower triangle (gp_stats_pack / gp_stats_unpack)
            root.stats_pack()
        self._dist.all_reduce(self._dev_tensors[which], op=self._dist.ReduceOp.SUM, group=self.group)
        if which == 'stats':
            root.stats_unpack()
        self.n_collectives += 1

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []

    # ---- parallel_GPLVM.likelihood_and_gradient (:222-279) on resident shards
    def likelihood_and_gradient(self, flat_array, iteration, step_size=0):
        M, Q = self.M, self.Q
        xt = transform_vec(self._pos, flat_array)
        Z = xt[:M * Q].reshape(M, Q)
        sf2, alpha, beta = xt[M * Q], xt[M * Q + 1:M * Q + 1 + Q], xt[M * Q + 1 + Q]
        want_emb = not self.fixed_embeddings
        for e in self.engines:
            e.set_globals(Z, sf2, alpha, beta, N_global=self.N, step_size=step_size)
            e.phase1()
        root = self.engines[0]
        for e in self.engines[1:]:
            root.combine(e, 'stats', 'add')
        self._allreduce_buffers('stats')
        for e in self.engines[1:]:
            e.combine(root, 'stats', 'copy')
        jitter = 0
        while True:
            for e in self.engines:
                e.global_step(sync=False, jitter=jitter)
                e.phase2(want_emb)
            for e in self.engines[1:]:
                root.combine(e, 'grads', 'add')
            self._allreduce_buffers('grads')
            try:
                res = root.finish()         # the evaluation's only host synchronisation
                break
            except JitterRetry as r:        # same reduced statistics on every rank: all ranks retry together (partial_terms.py:452-456)
                jitter = r.mask
        self.version += 1                   # grad_latest changed
        grad = np.concatenate([res['grad_Z'].ravel(), [res['grad_sf2']], res['grad_alpha'], [0.0 if self.fixed_beta else res['grad_beta']]])
        grad = grad * transform_grad_vec(self._pos, flat_array)
        return -res['F'], -grad


class ResidentCG(object):
    """The helper functions of scg_adapted_local_MapReduce.py on the resident vectors (the ``folder`` argument of the
    reference's file-based helpers is accepted and ignored)."""

    def __init__(self, model):
        self.m = model
        self._cache = None      # (model.version, the six reductions)

    def _dots(self):
        """[mu, kappa, theta, |g_new|^2, g_new.g_old, max|d|] over all shards of all ranks: one pass over the resident vectors and
        two small collectives -- a packed SUM of five scalars and one MAX (scg_adapted_local_MapReduce.py:59-155 visits every
        shard's files once per quantity) -- cached until a vector changes (any update here, or a new evaluation's grad_latest)."""
        if self._cache is not None and self._cache[0] == self.m.version:
            return self._cache[1]
        tot = np.zeros(6)
        for e in self.m.engines:
            d = e.cg_dots()
            tot[:5] += d[:5]
            tot[5] = max(tot[5], d[5])
        if self.m._dist is not None:
            tot[:5] = self.m._allreduce_vector(tot[:5], 'sum')
            tot[5] = self.m._allreduce_vector(tot[5:6], 'max')[0]
        self._cache = (self.m.version, tot)
        return tot

    def _upd(self, which, a=0.0):
        for e in self.m.engines:
            e.cg_update(which, a)
        self.m.version += 1

    def embeddings_set_grads(self, folder=None):
        self._upd(ShardEngine.CG_SET_GRADS)

    def embeddings_get_grads_mu(self, folder=None):
        return self._dots()[0]

    def embeddings_get_grads_kappa(self, folder=None):
        return self._dots()[1]

    def embeddings_get_grads_theta(self, folder=None):
        return self._dots()[2]

    def embeddings_get_grads_current_grad(self, folder=None):
        return self._dots()[3]

    def embeddings_get_grads_gamma(self, folder=None):
        return self._dots()[4]

    def embeddings_get_grads_max_d(self, folder, alpha):
        return abs(alpha) * self._dots()[5]

    def embeddings_set_grads_reset_d(self, folder=None):
        self._upd(ShardEngine.CG_RESET_D)

    def embeddings_set_grads_update_d(self, folder, gamma):
        self._upd(ShardEngine.CG_UPDATE_D, gamma)

    def embeddings_set_grads_update_X(self, folder, alpha):
        self._upd(ShardEngine.CG_UPDATE_X, alpha)

    def embeddings_set_grads_update_grad_old(self, folder=None):
        self._upd(ShardEngine.CG_GRAD_OLD)

    def embeddings_set_grads_update_grad_new(self, folder=None):
        self._upd(ShardEngine.CG_GRAD_NEW)


class ResidentGD(object):
    """The helper functions of gd_local_MapReduce.py:14-105 (the gradient-descent optimiser's vector algebra) on the
    resident vectors; ``grad_now`` is the library's grad_new array.  ``folder`` is accepted and ignored."""

    def __init__(self, model):
        self.m = model
        self._cache = None      # (model.version, (sum |grad_now|, max |grad_now|))

    def _upd(self, which, a=0.0):
        for e in self.m.engines:
            e.cg_update(which, a)
        self.m.version += 1

    def _abs(self):
        if self._cache is not None and self._cache[0] == self.m.version:
            return self._cache[1]
        s, mx = 0.0, 0.0
        for e in self.m.engines:
            a = e.cg_abs()
            s += a[0]
            mx = max(mx, a[1])
        if self.m._dist is not None:
            s = self.m._allreduce_scalar(s)
            mx = self.m._allreduce_scalar(mx, 'max')
        self._cache = (self.m.version, (s, mx))
        return s, mx


    def embeddings_set_grads(self, folder=None):                       # :14-32  grad_now = latest, d = -latest
        self._upd(ShardEngine.CG_SET_GRADS)

    def embeddings_get_grads_current_grad(self, folder=None):          # :38-47  sum |grad_now|
        return self._abs()[0]

    def embeddings_get_grads_max_gradnow(self, folder=None):           # :49-61  max |grad_now|
        return self._abs()[1]

    def embeddings_set_grads_update_d(self, folder, gamma):            # :63-74  d = -(grad_now + gamma d)
        self._upd(ShardEngine.CG_UPDATE_D, -gamma)

    def embeddings_set_grads_update_X(self, folder, step_size):        # :76-94  X += step d
        self._upd(ShardEngine.CG_UPDATE_X, step_size)

    def embeddings_set_grads_update_grad_now(self, folder=None):       # :96-105 grad_now = latest
        self._upd(ShardEngine.CG_GRAD_NEW)
