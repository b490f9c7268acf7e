"""Device-resident model: every shard's data, embeddings, search direction and gradient vectors stay in HBM for the
whole optimisation -- the "full SCG loop that never touches the filesystem".

``ResidentModel.likelihood_and_gradient(x, iteration, step_size)`` has the optimiser callback contract of
parallel_GPLVM.py:222-279; ``ResidentCG`` offers the function names of scg_adapted_local_MapReduce.py:29-243 on the
resident vectors.  Across GPUs (one process per GPU) the local scalars are summed / maxed with torch.distributed.
"""
import numpy as np

from .driver import transform, transform_grad
from .engine import ShardEngine


class ResidentModel(object):
    def __init__(self, shards, M, Q, D, fixed_embeddings=False, fixed_beta=False, device=0, dist_group=None, N_global=None):
        """shards: list of (Y, X_mu, X_S) held by THIS process (X_S raw = softplus-inverse space unless fixed_embeddings)."""
        self.M, self.Q, self.D = M, Q, D
        self.fixed_embeddings, self.fixed_beta = fixed_embeddings, fixed_beta
        self.engines = []
        for (Y, X_mu, X_S) in shards:
            e = ShardEngine(Y.shape[0], D, M, Q, device=device)
            e.upload_shard(Y, X_mu, X_S, xs_is_raw=not fixed_embeddings)
            self.engines.append(e)
        self.group = dist_group
        self._dist = None
        if dist_group is not None or self._dist_ready():
            import torch.distributed as dist
            self._dist = dist
        n_local = sum(e.N_s for e in self.engines)
        self.N = int(N_global) if N_global is not None else int(self._allreduce_scalar(float(n_local)))
        self.bounds = [(None, None)] * (M * Q) + [(0, None)] + [(0, None)] * Q + [(0, None)]
        self._dev_tensors = None

    @staticmethod
    def _dist_ready():
        try:
            import torch.distributed as dist
            return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        except Exception:
            return False

    def _allreduce_scalar(self, v, op='sum'):
        if self._dist is None:
            return v
        import torch
        t = torch.tensor([v], dtype=torch.float64, device='cuda' if self._dist.get_backend() == 'nccl' else 'cpu')
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM if op == 'sum' else self._dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def _allreduce_buffers(self, which):
        if self._dist is None:
            return
        from .dist import device_tensor
        import torch
        root = self.engines[0]
        p, n = root.stats_buffer() if which == 'stats' else root.grads_buffer()
        t = device_tensor(p, n, torch.device('cuda', root.device))
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []

    # ---- parallel_GPLVM.likelihood_and_gradient (:222-279) on resident shards
    def likelihood_and_gradient(self, flat_array, iteration, step_size=0):
        M, Q = self.M, self.Q
        xt = np.array([transform(b, v) for b, v in zip(self.bounds, flat_array)])
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
        for e in self.engines:
            e.global_step()
            e.phase2(want_emb)
        for e in self.engines[1:]:
            root.combine(e, 'grads', 'add')
        self._allreduce_buffers('grads')
        res = root.finish()
        grad = np.concatenate([res['grad_Z'].ravel(), [res['grad_sf2']], res['grad_alpha'], [0.0 if self.fixed_beta else res['grad_beta']]])
        grad = np.array([g * transform_grad(b, v) for b, v, g in zip(self.bounds, flat_array, grad)])
        return -res['F'], -grad


class ResidentCG(object):
    """The helper functions of scg_adapted_local_MapReduce.py on the resident vectors (the ``folder`` argument of the
    reference's file-based helpers is accepted and ignored)."""

    def __init__(self, model):
        self.m = model
        self._cache = None

    def _dots(self):
        tot = np.zeros(6)
        for e in self.m.engines:
            d = e.cg_dots()
            tot[:5] += d[:5]
            tot[5] = max(tot[5], d[5])
        if self.m._dist is not None:
            for k in range(5):
                tot[k] = self.m._allreduce_scalar(tot[k])
            tot[5] = self.m._allreduce_scalar(tot[5], 'max')
        return tot

    def _upd(self, which, a=0.0):
        for e in self.m.engines:
            e.cg_update(which, a)

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

    def _upd(self, which, a=0.0):
        for e in self.m.engines:
            e.cg_update(which, a)

    def _abs(self):
        s, mx = 0.0, 0.0
        for e in self.m.engines:
            a = e.cg_abs()
            s += a[0]
            mx = max(mx, a[1])
        if self.m._dist is not None:
            s = self.m._allreduce_scalar(s)
            mx = self.m._allreduce_scalar(mx, 'max')
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
