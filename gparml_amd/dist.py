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

from ._lib import JitterRetry


class _DevArray(object):
    """Zero-copy view of a device buffer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<f8', 'data': (int(ptr), False), 'version': 2}


def device_tensor(ptr, n, device):
    import torch
    return torch.as_tensor(_DevArray(ptr, n), device=device)


#: why the last init_native_comm on this rank returned False ('' after a success); bench.py prints it on rank 0
native_comm_reason = ''


def _say(rank, why):
    global native_comm_reason
    native_comm_reason = why
    if rank == 0 and why:
        import sys
        sys.stderr.write('[gparml_amd.dist] reduce falls back to torch.distributed: %s\n' % why)
    return False


def _bounded(fn, seconds, what, rank):
    """Run a collective call that may wait for other ranks in a worker thread; if it has not returned after ``seconds`` the process says why
    and ends with status 75 (os._exit from this very process: nothing is exec'ed, a launcher sees a non-zero child) -- a rank that never
    arrives at ncclCommInitRank must turn into an error of the job, not into a silent hang."""
    import os
    import sys
    import threading
    box = {}

    def run():
        try:
            box['value'] = fn()
        except BaseException as e:      # noqa: BLE001  (handed back to the caller's thread)
            box['error'] = e

    t = threading.Thread(target=run, name='gparml-' + what)
    t.daemon = True
    t.start()
    t.join(seconds)
    if t.is_alive():
        sys.stderr.write('[gparml_amd.dist] rank %d: %s did not return within %.0f s (a peer never arrived?); giving up\n' % (rank, what, seconds))
        sys.stderr.flush()
        os._exit(75)
    if 'error' in box:
        raise box['error']
    return box.get('value')


def init_native_comm(engine, dist, group=None, _agree=None):
    """Give ``engine`` its own RCCL communicator (gp_comm_init): rank 0 draws the ncclUniqueId, torch.distributed carries the 128 bytes to
    the other ranks, every rank joins.  Returns True when the engine now reduces inside the library (gp_allreduce); False when the
    engine has no such entry point, the backend is not RCCL (gloo tests), GPARML_NATIVE_ALLREDUCE=0, or ANY rank cannot use RCCL --
    the caller then keeps using torch.distributed on the device pointers (``native_comm_reason`` says why; rank 0 prints it).

    Nothing collective happens on the new communicator before every rank has agreed -- over torch -- that it can take part: each rank
    probes RCCL on its own (gp_comm_available, and rank 0 also draws the id: both local), the flags are MIN-all-reduced, and only then
    is the id broadcast and ncclCommInitRank entered (with a bounded wait, GPARML_COMM_INIT_TIMEOUT seconds, default 180).  A second
    agreement right after the call covers RCCL refusing the communicator on some rank (no collective has run on it yet, so the ranks
    that did join are not left waiting inside one); only then does a one-double probe all-reduce check that the communicator really
    sums over ``world`` ranks, and a third agreement covers its outcome.

    ``_agree`` (tests only, tests/test_dist_gloo.py): a replacement for the MIN all-reduce over the torch group -- the order of agreement rounds and
    collective calls is then exercised on CPU with a scripted peer."""
    import os
    if os.environ.get('GPARML_NATIVE_ALLREDUCE', '1') == '0':
        return _say(-1, '')
    if not hasattr(engine, 'comm_init') or not dist.is_initialized():
        return _say(-1, '')
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if dist.get_backend(group) != 'nccl':
        return _say(-1, 'process group backend is %s, not nccl' % dist.get_backend(group))
    if getattr(engine, 'has_comm', False):
        return True
    import contextlib
    limit = float(os.environ.get('GPARML_COMM_INIT_TIMEOUT', '180'))
    if _agree is None:
        import torch
        dev = torch.device('cuda', int(getattr(engine, 'device', torch.cuda.current_device())))
        on_device = lambda: torch.cuda.device(dev)

        def agree(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return int(flag.item()) == 1
    else:
        dev, agree, on_device = None, _agree, contextlib.nullcontext

    uid, why = None, ''
    try:
        ok = bool(engine.comm_available())
        if ok and rank == 0:
            uid = engine.comm_unique_id()
        if not ok:
            why = 'RCCL cannot be resolved by the library on rank %d' % rank
    except Exception as e:      # noqa: BLE001
        ok, why = False, 'rank %d: %s' % (rank, e)
    with on_device():
        if not agree(ok):
            return _say(rank, why or 'a peer rank cannot use RCCL from the library')
        box = [uid]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group, device=dev)
        try:
            _bounded(lambda: engine.comm_init(box[0], world, rank), limit, 'gp_comm_init', rank)
            ok = True
        except Exception as e:      # noqa: BLE001  (communicator refused on this rank)
            ok, why = False, 'rank %d: %s' % (rank, e)

        def give_up(reason):
            if getattr(engine, 'has_comm', False):
                engine.comm_destroy()
            return _say(rank, reason)

        # agreement (over torch) that EVERY rank holds a communicator, before anything collective runs on it: a rank whose gp_comm_init
        # raised would never enter the probe all-reduce and the others would sit in ncclAllReduce until the watchdog killed them
        if not agree(ok):
            return give_up(why or 'a peer rank failed in gp_comm_init')
        try:
            info = _bounded(lambda: engine.comm_info(probe=True), limit, 'gp_comm_info(probe)', rank)
            if info['ranks'] != world or info['probe_sum'] != float(world):
                ok, why = False, 'rank %d: communicator reports %d ranks, probe sum %r, expected %d' % (rank, info['ranks'], info['probe_sum'], world)
        except Exception as e:      # noqa: BLE001
            ok, why = False, 'rank %d: %s' % (rank, e)
        # ... and on the probe's outcome: every rank takes the same path through the two reductions, torch everywhere otherwise
        if not agree(ok):
            return give_up(why or 'a peer rank saw a wrong probe sum')
    _say(rank, '')
    return True


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
        self._packed = False
        self.time_collectives = False     # bench.py: device-side time of the two all-reduces of the last evaluation
        self._cev = None
        # the reduce inside the library (gp_allreduce: RCCL on the engine's stream) when the process group is RCCL; otherwise torch.distributed
        # all-reduces the same device buffers (gloo in the CPU / one-device tests)
        self.native = (self.world > 1 or self.force) and init_native_comm(engine, dist, group)

    def _tensors(self):
        if self._stats_t is None:
            if hasattr(self.engine, 'host_buffers'):            # CPU stand-in (tests): numpy-backed tensors
                import torch
                s, g = self.engine.host_buffers()
                self._stats_t, self._grads_t = torch.from_numpy(s), torch.from_numpy(g)
            else:
                # the packed form (Psi2's upper triangle, no padding) where the engine has it: 1.46 MB instead of 2.6 MB at M=512, D=100
                self._packed = hasattr(self.engine, 'stats_packed_buffer')
                p, n = self.engine.stats_packed_buffer() if self._packed else self.engine.stats_buffer()
                self._stats_t = device_tensor(p, n, self.device)
                p, n = self.engine.grads_buffer()
                self._grads_t = device_tensor(p, n, self.device)
        return self._stats_t, self._grads_t

    def evaluate(self, want_embedding_grads=False, kept_mask=None, kept_fraction=None):
        """One bound+gradient evaluation across all shards.

        Node drop-out (local_MapReduce.py:119-129, 263-264): ``kept_mask`` is the keep/drop decision for EVERY rank, identical on all
        ranks (``draw_kept_mask`` with a shared seed).  A dropped rank contributes nothing to either reduction -- the reference
        sums all twelve statistics, the derivative sums behind grad_Z / grad_alpha included, over the kept nodes only -- and both
        reduced buffers are divided by ``kept_fraction`` = kept/(kept+dropped).  Every rank still runs phase 2 for its own embedding
        gradients (embeddings_MR visits all nodes, local_MapReduce.py:293-295)."""
        eng = self.engine
        collective = self.world > 1 or self.force
        native = collective and self.native
        stats_t, grads_t = self._tensors() if (collective and not native) else (None, None)   # zero-copy torch views of the packed device buffers
        kept_here = True
        if kept_mask is not None:
            kept_mask = [bool(k) for k in kept_mask]
            assert len(kept_mask) == self.world, 'kept_mask needs one entry per rank'
            kept_here = kept_mask[self.rank]
            if kept_fraction is None:
                kept_fraction = float(sum(kept_mask)) / len(kept_mask)
        rescale = kept_fraction is not None and kept_fraction != 1.0
        cev = self._collective_events() if (collective and self.time_collectives) else None
        eng.phase1()
        if not kept_here:
            eng.scale_buffer('stats', 0.0)
        if native:
            if cev:
                cev[0].record()
            eng.allreduce('stats')            # pack -> ncclAllReduce -> unpack on the engine's stream
            if cev:
                cev[1].record()
        elif collective:
            if self._packed:
                eng.stats_pack()
            if cev:
                cev[0].record()
            self.dist.all_reduce(stats_t, op=self.dist.ReduceOp.SUM, group=self.group)
            if cev:
                cev[1].record()
            if self._packed:
                eng.stats_unpack()
        if rescale:
            eng.scale_buffer('stats', 1.0 / kept_fraction)
        jitter = 0
        while True:
            eng.global_step(sync=False, jitter=jitter)
            eng.phase2(want_embedding_grads)
            if not kept_here:
                eng.scale_buffer('grads', 0.0)
            if collective:
                if cev:
                    cev[2].record()
                if native:
                    eng.allreduce('grads')
                else:
                    self.dist.all_reduce(grads_t, op=self.dist.ReduceOp.SUM, group=self.group)
                if cev:
                    cev[3].record()
            if rescale:
                eng.scale_buffer('grads', 1.0 / kept_fraction)
            try:
                return eng.finish()       # the evaluation's only host synchronisation
            except JitterRetry as r:
                # every rank holds the same reduced statistics and runs the same replicated global step, so all ranks take this
                # branch together: repeat the global step with the reference's 1e-7 jitter (partial_terms.py:452-456)
                jitter = r.mask


    def _collective_events(self):
        """Four timing events on torch's current stream (the stream the engine launches on and the collectives are ordered behind)."""
        if self._cev is None:
            import torch
            self._cev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if torch.cuda.is_available() else False
        return self._cev or None

    def collective_ms(self):
        """Device time of the two all-reduces of the last evaluation (0 when not timed).  With a stream-ordered backend (RCCL) the
        events bracket the collective's kernels; with gloo they bracket its host-side staging copies."""
        if not (self.time_collectives and self._cev):
            return {'allreduce_stats_ms': 0.0, 'allreduce_grads_ms': 0.0}
        import torch
        torch.cuda.synchronize()      # free here: the evaluation's own read-back (gp_finish) has already waited for everything on the stream
        return {'allreduce_stats_ms': self._cev[0].elapsed_time(self._cev[1]), 'allreduce_grads_ms': self._cev[2].elapsed_time(self._cev[3])}


def draw_kept_mask(n_nodes, drop_out_fraction, rng):
    """The keep/drop draw of statistics_MR (local_MapReduce.py:119-129) -> (kept_mask, kept_fraction).  When every node is dropped the
    reference keeps one random node but leaves ALL n nodes in its dropped list, so its divisor is 1/(n+1) (:126-128, 263-264)."""
    drop = rng.uniform(size=n_nodes) < drop_out_fraction
    kept = ~drop
    if not kept.any():
        kept = np.zeros(n_nodes, dtype=bool)
        kept[rng.randint(0, n_nodes)] = True
        return list(kept), 1.0 / (n_nodes + 1)
    return list(kept), float(kept.sum()) / n_nodes
