"""world_size-2 test of the N>1 protocol on CPU (gloo): gparml_amd.dist.DistributedEvaluator drives two ranks, each
holding its own shard, through phase1 -> all-reduce -> global step -> phase2 -> all-reduce -> finish.  On the GPU box the
engine is gparml_amd.engine.ShardEngine and the backend is RCCL; here an oracle-backed stand-in with the same method
set and the same packed-buffer protocol plays the engine, so what is tested is the reduction protocol, the packing and the
drop-out rescale -- and that 2 ranks reproduce the single-shard result on the concatenated data."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine(object):
    """Test double with ShardEngine's evaluation methods, computing with oracle/factorised.py (tests only)."""

    def __init__(self, d, sl, N_global):
        from oracle import factorised as Fz
        self.Fz = Fz
        self.d = d
        self.sl = sl
        self.Ng = N_global
        M, Q = d['Z'].shape
        D = d['Y'].shape[1]
        self.M, self.Q, self.D = M, Q, D
        self.stats = np.zeros(M * M + M * D + 8)
        self.grads = np.zeros(M * Q + Q)

    def host_buffers(self):
        return self.stats, self.grads

    def phase1(self):
        d, sl = self.d, self.sl
        s = self.Fz.phase1(d['Z'], d['sf2'], d['alpha'], d['Y'][sl], d['X_mu'][sl], d['X_S'][sl])
        M, D = self.M, self.D
        self.stats[:M * M] = s['sum_exp_K_mi_K_im'].ravel()
        self.stats[M * M:M * M + M * D] = s['exp_K_miY'].ravel()
        self.stats[M * M + M * D:M * M + M * D + 3] = [s['sum_YYT'], s['sum_exp_K_ii'], s['KL']]

    def scale_stats(self, f):
        self.stats *= f

    def scale_buffer(self, which, f):
        if which == 'stats':
            self.stats *= f
        else:
            self.grads *= f

    def global_step(self, sync=True, jitter=0):
        d, M, D = self.d, self.M, self.D
        st = dict(sum_exp_K_mi_K_im=self.stats[:M * M].reshape(M, M).copy(), exp_K_miY=self.stats[M * M:M * M + M * D].reshape(M, D).copy(),
                  sum_YYT=self.stats[M * M + M * D], sum_exp_K_ii=self.stats[M * M + M * D + 1], KL=self.stats[M * M + M * D + 2])
        self.gs = self.Fz.global_step(d['Z'], d['sf2'], d['alpha'], d['beta'], st, self.Ng, D)

    def phase2(self, want_emb):
        d, sl = self.d, self.sl
        p2 = self.Fz.phase2(d['Z'], d['sf2'], d['alpha'], d['Y'][sl], d['X_mu'][sl], d['X_S'][sl], self.gs['Abar'], self.gs['Bbar'],
                            want_embeddings=want_emb)
        self.grads[:self.M * self.Q] = p2['grad_Z_data'].ravel()
        self.grads[self.M * self.Q:] = p2['grad_alpha_data']
        self.p2 = p2

    def finish(self):
        acc = dict(grad_Z_data=self.grads[:self.M * self.Q].reshape(self.M, self.Q), grad_alpha_data=self.grads[self.M * self.Q:])
        return self.Fz.finish(self.d['Z'], self.d['sf2'], self.d['alpha'], self.gs, acc, self.Fz.is_regime_A(self.d['X_S']))


def _worker(rank, world, port, regime, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    from gparml_amd.dist import DistributedEvaluator
    from oracle import factorised as Fz
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    d = Fz.synthetic_shard(90, 4, 7, 3, regime=regime, seed=5, zseed=6, alpha_value=0.5)
    cut = [0, 37, 90]
    eng = OracleEngine(d, slice(cut[rank], cut[rank + 1]), 90)
    ev = DistributedEvaluator(eng)
    out = ev.evaluate(regime == 'B')
    out2 = ev.evaluate(regime == 'B', kept_mask=[True, False])      # node drop-out: rank 1 dropped, kept fraction 1/2
    if rank == 0:
        q.put((out['F'], out['grad_Z'], out['grad_alpha'], out['grad_sf2'], out['grad_beta'], out2['F'], out2['grad_Z'], out2['grad_alpha'],
               out2['grad_sf2'], out2['grad_beta']))
    dist.destroy_process_group()


@pytest.mark.parametrize('regime', ['A', 'B'])
def test_two_ranks_equal_one_shard(regime):
    import torch.multiprocessing as mp
    from oracle import factorised as Fz
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, regime, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    d = Fz.synthetic_shard(90, 4, 7, 3, regime=regime, seed=5, zseed=6, alpha_value=0.5)
    ref = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'])
    assert abs(res[0] - ref['F']) <= 1e-10 * abs(ref['F'])
    assert np.max(np.abs(res[1] - ref['grad_Z'])) <= 1e-9 * np.max(np.abs(ref['grad_Z']))
    assert np.max(np.abs(res[2] - ref['grad_alpha'])) <= 1e-9 * np.max(np.abs(ref['grad_alpha']))
    assert abs(res[3] - ref['grad_sf2']) <= 1e-9 * abs(ref['grad_sf2'])
    assert abs(res[4] - ref['grad_beta']) <= 1e-9 * abs(ref['grad_beta'])
    # node drop-out (local_MapReduce.py:119-129, 263-264): every statistic -- the derivative sums behind grad_Z / grad_alpha
    # included -- is summed over the kept node only (rows 0..36) and divided by kept/(kept+dropped) = 1/2
    sl = slice(0, 37)
    st = Fz.phase1(d['Z'], d['sf2'], d['alpha'], d['Y'][sl], d['X_mu'][sl], d['X_S'][sl])
    st2 = {k: 2.0 * v for k, v in st.items()}
    gs2 = Fz.global_step(d['Z'], d['sf2'], d['alpha'], d['beta'], st2, 90, 4)
    p2 = Fz.phase2(d['Z'], d['sf2'], d['alpha'], d['Y'][sl], d['X_mu'][sl], d['X_S'][sl], gs2['Abar'], gs2['Bbar'], want_embeddings=False)
    acc = dict(grad_Z_data=2.0 * p2['grad_Z_data'], grad_alpha_data=2.0 * p2['grad_alpha_data'])
    ref2 = Fz.finish(d['Z'], d['sf2'], d['alpha'], gs2, acc, Fz.is_regime_A(d['X_S']))
    assert abs(res[5] - ref2['F']) <= 1e-10 * abs(ref2['F'])
    assert np.max(np.abs(res[6] - ref2['grad_Z'])) <= 1e-9 * np.max(np.abs(ref2['grad_Z']))
    assert np.max(np.abs(res[7] - ref2['grad_alpha'])) <= 1e-9 * np.max(np.abs(ref2['grad_alpha']))
    assert abs(res[8] - ref2['grad_sf2']) <= 1e-9 * abs(ref2['grad_sf2'])
    assert abs(res[9] - ref2['grad_beta']) <= 1e-9 * abs(ref2['grad_beta'])


def test_draw_kept_mask_follows_the_reference():
    """local_MapReduce.py:119-129: uniform draw per node; when everything is dropped one random node is kept and the divisor is 1/(n+1)."""
    from gparml_amd.dist import draw_kept_mask
    mask, f = draw_kept_mask(8, 0.5, np.random.RandomState(0))
    drop = np.random.RandomState(0).uniform(size=8) < 0.5
    assert mask == list(~drop) and f == float((~drop).sum()) / 8
    mask, f = draw_kept_mask(5, 1.0, np.random.RandomState(1))
    assert sum(mask) == 1 and f == 1.0 / 6


# ---- dist.init_native_comm: the ORDER of agreement rounds and collective calls (round-5 advice: a rank whose gp_comm_init raised skipped the probe all-reduce
# while its peers sat inside it until the watchdog killed them).  A scripted peer stands in for the torch group; no GPU, no RCCL.
class _FakeDist(object):
    def __init__(self, rank, world):
        self.rank, self.world, self.log = rank, world, []

    def is_initialized(self):
        return True

    def get_rank(self, group=None):
        return self.rank

    def get_world_size(self, group=None):
        return self.world

    def get_backend(self, group=None):
        return 'nccl'

    def get_global_rank(self, group, r):
        return r

    def broadcast_object_list(self, box, src=0, group=None, device=None):
        self.log.append('broadcast_id')
        if box[0] is None:
            box[0] = b'id-from-rank-0'


class _FakeEngine(object):
    device = 0

    def __init__(self, log, init_raises=False, probe_sum=None, world=2):
        self.log, self.init_raises, self.probe_sum, self.world, self.has_comm = log, init_raises, probe_sum, world, False

    def comm_available(self):
        return True

    def comm_unique_id(self):
        return b'id-from-rank-0'

    def comm_init(self, uid, world, rank):
        self.log.append('comm_init')
        if self.init_raises:
            raise RuntimeError('ncclCommInitRank refused')
        self.has_comm = True

    def comm_info(self, probe=False):
        self.log.append('probe' if probe else 'info')
        return {'ranks': self.world, 'rank': 0, 'probe_sum': float(self.world if self.probe_sum is None else self.probe_sum)}

    def comm_destroy(self):
        self.log.append('comm_destroy')
        self.has_comm = False


def _scripted_agree(log, peer_answers):
    it = iter(peer_answers)

    def agree(ok):
        peer = next(it)
        log.append('agree(%d,%d)' % (int(ok), int(peer)))
        return bool(ok) and bool(peer)
    return agree


def test_native_comm_agreement_rounds_all_ranks_join():
    from gparml_amd import dist as gd
    d = _FakeDist(0, 2)
    eng = _FakeEngine(d.log)
    assert gd.init_native_comm(eng, d, _agree=_scripted_agree(d.log, [1, 1, 1])) is True
    # availability agreed -> id -> init -> agreement BEFORE anything collective on the new communicator -> probe -> agreement on its outcome
    assert d.log == ['agree(1,1)', 'broadcast_id', 'comm_init', 'agree(1,1)', 'probe', 'agree(1,1)'] and eng.has_comm


def test_native_comm_never_probes_when_a_rank_failed_to_join():
    from gparml_amd import dist as gd
    # this rank's gp_comm_init raises: it must still take part in the agreement round, and must not enter the probe
    d = _FakeDist(1, 2)
    eng = _FakeEngine(d.log, init_raises=True)
    assert gd.init_native_comm(eng, d, _agree=_scripted_agree(d.log, [1, 1])) is False
    assert d.log == ['agree(1,1)', 'broadcast_id', 'comm_init', 'agree(0,1)'] and 'probe' not in d.log
    # the PEER failed to join: this rank holds a communicator, learns of the failure in the agreement round, destroys it and never probes
    d = _FakeDist(0, 2)
    eng = _FakeEngine(d.log)
    assert gd.init_native_comm(eng, d, _agree=_scripted_agree(d.log, [1, 0])) is False
    assert d.log == ['agree(1,1)', 'broadcast_id', 'comm_init', 'agree(1,0)', 'comm_destroy'] and not eng.has_comm
    # a wrong probe sum (the communicator connected fewer ranks): third agreement round, communicator destroyed everywhere
    d = _FakeDist(0, 2)
    eng = _FakeEngine(d.log, probe_sum=1.0)
    assert gd.init_native_comm(eng, d, _agree=_scripted_agree(d.log, [1, 1, 1])) is False
    assert d.log == ['agree(1,1)', 'broadcast_id', 'comm_init', 'agree(1,1)', 'probe', 'agree(0,1)', 'comm_destroy']
    # RCCL not available on a peer: nothing is created at all
    d = _FakeDist(0, 2)
    eng = _FakeEngine(d.log)
    assert gd.init_native_comm(eng, d, _agree=_scripted_agree(d.log, [0])) is False and d.log == ['agree(1,0)']
