"""One process per GPU: the MapReduce "reduce" (local_MapReduce.py:250-277, statistics_reducer) as an
all-reduce(sum) of the packed per-shard buffers over torch.distributed (backend "nccl" = RCCL over xGMI
on the GPU box; "gloo" in the CPU tests).

Per evaluation there are exactly two collectives:
    phase 1:  [Psi2 (Mp*Mp) | Psi1^T Y (Mp*Dp) | sum_YYT, Psi0, KL, n_local]   -> global step (replicated)
    phase 2:  [grad_Z data part (M*Q) | grad_alpha data part (Q)]             -> finish
The engine object only needs the methods used below, so the CPU tests drive this protocol with an
oracle-backed stand-in while the product path uses gparml_amd.engine.ShardEngine.
"""
import numpy as np


class _DevArray(object):
    """Zero-copy view of a device buffer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<f8', 'data': (int(ptr), False), 'version': 2}


def device_tensor(ptr, n, device):
    import torch
    return torch.as_tensor(_DevArray(ptr, n), device=device)


class DistributedEvaluator(object):
    """Drives one ShardEngine per rank through phase1 -> all-reduce -> global step -> phase2 -> all-reduce."""

    def __init__(self, engine, group=None, device=None, force_collectives=False):
        import torch.distributed as dist
        self.dist = dist
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._stats_t = None
        self._grads_t = None
        self.device = device
        self.force = force_collectives and dist.is_initialized()

    def _tensors(self):
        if self._stats_t is None:
            if hasattr(self.engine, 'host_buffers'):            # CPU stand-in (tests): numpy-backed tensors
                import torch
                s, g = self.engine.host_buffers()
                self._stats_t, self._grads_t = torch.from_numpy(s), torch.from_numpy(g)
            else:
                p, n = self.engine.stats_buffer()
                self._stats_t = device_tensor(p, n, self.device)
                p, n = self.engine.grads_buffer()
                self._grads_t = device_tensor(p, n, self.device)
        return self._stats_t, self._grads_t

    def evaluate(self, want_embedding_grads=False, kept_fraction=None):
        """One bound+gradient evaluation across all shards.  ``kept_fraction`` reproduces the node drop-out
        rescale of local_MapReduce.py:263-264 (statistics divided by kept/(kept+dropped))."""
        eng = self.engine
        stats_t, grads_t = self._tensors()
        eng.phase1()
        if self.world > 1 or self.force:
            self.dist.all_reduce(stats_t, op=self.dist.ReduceOp.SUM, group=self.group)
        if kept_fraction is not None and kept_fraction != 1.0:
            eng.scale_stats(1.0 / kept_fraction)
        eng.global_step()
        eng.phase2(want_embedding_grads)
        if self.world > 1 or self.force:
            self.dist.all_reduce(grads_t, op=self.dist.ReduceOp.SUM, group=self.group)
        return eng.finish()
