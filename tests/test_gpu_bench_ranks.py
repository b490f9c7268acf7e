"""The N>1 paths on a 1-GPU box: several ranks share device 0 and reduce over gloo (bench.py's test hooks GPARML_BENCH_*), so the
barrier / max-over-ranks timing / rank-0 JSON line and the two per-evaluation all-reduces on the library's device buffers run exactly
as under `torch.distributed.run --nproc-per-node N` with RCCL.

Round 5: ONE torchrun launch per world size (every rank imports torch once -- on a cold box that import is the 100 s item): the
rank script runs its checks under its own process group, destroys it, and then calls bench.main() in the same process."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

BENCH_TAIL = r"""
# bench.py --gpus <world> end to end in the same processes, on the process group the checks above used (a second init_process_group in the same
# processes hung at eight ranks in round 5)
os.environ['GPARML_BENCH_ONE_DEVICE'] = '1'; os.environ['GPARML_BENCH_BACKEND'] = 'gloo'
import bench
sys.argv = ['bench.py', '--gpus', str(world), '--steps', '3', '--warmup', '1'] + %(bench_shape)r
bench.main()
dist.destroy_process_group()
print('RANK_OK', rank)
"""


def _launch(tmp_path, name, text, world, timeout):
    script = tmp_path / name
    script.write_text(text)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert r.returncode == 0 and r.stdout.count('RANK_OK') == world, r.stdout[-2000:] + r.stderr[-6000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == world and res['steps'] == 3 and res['scaling'] == 'weak' and res['value'] > 0
    assert 'roofline' in res and 'cpu_baseline' not in res and 'extra' not in res            # CPU baseline / extras: rank 0 at N=1 only
    c = res['config']
    assert c['allreduce_ms']['total'] > 0 and c['global_ms'] > 0 and len(c['ms_per_step_by_rank']['all']) == world
    # the self-diagnosis fields of the N > 1 line: over gloo the library has no communicator of its own
    assert c['collective_backend'] == 'gloo' and c['comm_ranks'] == 0 and c['comm_probe_sum'] is None
    assert c['allreduce_payload_bytes']['stats'] > c['allreduce_payload_bytes']['grads'] > 0
    return res


RCCL_SCRIPT = r"""
import os, socket, sys
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from gparml_amd.dist import DistributedEvaluator
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
import numpy as np
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
N, D, M, Q = 500, 6, 40, 4
d = Fz.synthetic_shard(N, D, M, Q, regime='A', seed=3, zseed=4, alpha_value=0.4)
ref = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], want_embeddings=False)
for native in ('1', '0'):       # gp_allreduce inside the library, then torch.distributed on zero-copy views of the library's buffers
    os.environ['GPARML_NATIVE_ALLREDUCE'] = native
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    ev = DistributedEvaluator(eng, device=torch.device('cuda', 0), force_collectives=True)
    assert ev.native == (native == '1'), (native, ev.native)
    if ev.native:
        info = eng.comm_info(probe=True)                 # the library's own answer to "how many ranks did RCCL connect"
        assert info['ranks'] == 1 and info['rank'] == 0 and info['probe_sum'] == 1.0 and info['stats_bytes'] == 8 * (M * (M + 1) // 2 + M * D + 8), info
    out = ev.evaluate(False)
    st, gt = ev._tensors()
    assert st.is_cuda and st.dtype == torch.float64 and st.data_ptr() == eng.stats_packed_buffer()[0] and gt.data_ptr() == eng.grads_buffer()[0]
    assert abs(out['F'] - ref['F']) <= 1e-6 * abs(ref['F'])
    assert np.max(np.abs(out['grad_Z'] - ref['grad_Z'])) <= 1e-5 * np.max(np.abs(ref['grad_Z']))
    eng.close()
dist.destroy_process_group()
print('RCCL_OK')
"""


def test_rccl_allreduce_on_the_packed_device_buffers():
    """The N>1 path on one GPU, in a fresh process: a 1-rank RCCL group all-reduces the engine's packed device buffers
    in place and the evaluation still matches the oracle -- through the library's own communicator (gp_comm_init / gp_allreduce) and
    through torch.distributed on a zero-copy view of the library's memory (GPARML_NATIVE_ALLREDUCE=0), both in ONE process."""
    r = subprocess.run([sys.executable, '-c', RCCL_SCRIPT % {'root': ROOT}], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'RCCL_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


NATIVE_SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
N, D, M, Q = 700, 5, 33, 3
d = Fz.synthetic_shard(N, D, M, Q, regime='B', seed=5, zseed=6, alpha_value=0.4)
eng = ShardEngine(N, D, M, Q)
eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
plain = eng.evaluate(True)
try:
    eng.allreduce('stats')
    raise SystemExit('gp_allreduce without a communicator did not fail')
except RuntimeError as e:
    assert 'gp_comm_init' in str(e), e
eng.comm_init(ShardEngine.comm_unique_id(), 1, 0)          # what a C consumer does: no torch in this process
eng.phase1(); eng.allreduce('stats'); eng.global_step(sync=False); eng.phase2(True); eng.allreduce('grads')
out = eng.finish()
assert out['F'] == plain['F'] and np.array_equal(out['grad_Z'], plain['grad_Z']) and np.array_equal(out['grad_alpha'], plain['grad_alpha'])
eng.comm_destroy(); eng.close()
assert 'torch' not in sys.modules
print('NATIVE_OK')
"""


def test_library_allreduce_without_torch():
    """gp_comm_unique_id / gp_comm_init / gp_allreduce (RCCL resolved by dlopen inside the library) in a process that never imports torch: a
    one-rank communicator leaves the evaluation bit-identical; gp_allreduce before gp_comm_init is a state error."""
    r = subprocess.run([sys.executable, '-c', NATIVE_SCRIPT % {'root': ROOT}], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'NATIVE_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


RANK_SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from gparml_amd.engine import ShardEngine
from gparml_amd.dist import DistributedEvaluator
from oracle import factorised as Fz
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=rank, world_size=world)
N, D, M, Q = 1200, 6, 70, 4
for regime in ('A', 'B'):
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=21, zseed=22, alpha_value=0.5)
    cut = [0, 500, N]
    a, b = cut[rank], cut[rank + 1]
    eng = ShardEngine(b - a, D, M, Q)
    eng.upload_shard(d['Y'][a:b], d['X_mu'][a:b], d['X_S'][a:b])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'], N_global=N)
    out = DistributedEvaluator(eng, device=torch.device('cuda', 0)).evaluate(regime == 'B')
    ref = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'])
    assert abs(out['F'] - ref['F']) <= 1e-6 * abs(ref['F']), (out['F'], ref['F'])
    for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta'):
        assert np.max(np.abs(np.asarray(out[k]) - np.asarray(ref[k]))) <= 1e-5 * np.max(np.abs(ref[k])), k
    if regime == 'B':
        for k, name in (('grad_X_mu', 'GRAD_X_MU'), ('grad_X_S', 'GRAD_X_S')):
            g = eng.download(name)
            assert np.max(np.abs(g - ref[k][a:b])) <= 1e-5 * np.max(np.abs(ref[k])), k
    eng.close()
"""


SCG_RANK_SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy as np, torch, torch.distributed as dist
from gparml_amd.resident import ResidentCG, ResidentGD, ResidentModel
from gparml_amd.scg_adapted import SCG_adapted
from gparml_amd.gd import GD
for fname, optimiser, iters in (('pipe_gplvm_2shards.npz', 'scg', 2), ('gdpipe_gplvm_2shards.npz', 'gd', 9)):
    z = np.load(os.path.join(%(root)r, 'tests', 'golden', fname)); g = {k: z[k] for k in z.files}
    M, Q, D, N = int(g['M']), int(g['Q']), int(g['D']), int(g['N'])
    mine = [(g['Y_%%d' %% rank], g['call0_in_shard%%d_embedding' %% rank], g['call0_in_shard%%d_variance' %% rank])]   # one shard per rank
    model = ResidentModel(mine, M, Q, D, fixed_embeddings=False)
    assert model.N == N, (model.N, N)                     # all-reduced point count
    calls = []
    def f_and_g(x, it, step=0):
        f, grad = model.likelihood_and_gradient(x, it, step); calls.append((np.array(x), f, grad)); return f, grad
    if optimiser == 'scg':
        x_opt = SCG_adapted(f_and_g, g['call0_x'].copy(), ResidentCG(model), fixed_embeddings=False, maxiters=iters, xtol=0, ftol=0, gtol=0)[0]
    else:
        x_opt = GD(f_and_g, g['call0_x'].copy(), ResidentGD(model), fixed_embeddings=False, maxiters=iters)[0]
    f_and_g(x_opt, 'f')
    assert len(calls) == int(g['n_calls']), (optimiser, len(calls), int(g['n_calls']))
    for k, (x, f, grad) in enumerate(calls):
        assert np.max(np.abs(x - g['call%%d_x' %% k])) <= 1e-7 * np.max(np.abs(g['call%%d_x' %% k])) + 1e-12, (optimiser, k, 'x')
        assert abs(f - float(g['call%%d_f' %% k])) <= 1e-6 * abs(float(g['call%%d_f' %% k])), (optimiser, k, 'f')
        assert np.max(np.abs(grad - g['call%%d_g' %% k])) <= 2e-5 * np.max(np.abs(g['call%%d_g' %% k])), (optimiser, k, 'g')
    if optimiser == 'scg':
        # the optimiser's per-shard reductions (scg_adapted_local_MapReduce.py:59-155): all six quantities from ONE pass and TWO small
        # collectives (packed SUM of five + one MAX), cached until a resident vector changes
        cg = ResidentCG(model)
        n0 = model.n_collectives
        vals = [cg.embeddings_get_grads_mu(), cg.embeddings_get_grads_kappa(), cg.embeddings_get_grads_theta(),
                cg.embeddings_get_grads_current_grad(), cg.embeddings_get_grads_gamma(), cg.embeddings_get_grads_max_d(None, 0.5)]
        assert model.n_collectives - n0 == 2, model.n_collectives - n0
        cg.embeddings_set_grads_update_d(None, 0.1)
        assert cg.embeddings_get_grads_kappa() != vals[1] and model.n_collectives - n0 == 4
        n1 = model.n_collectives
        model.likelihood_and_gradient(x_opt, 'f')
        assert model.n_collectives - n1 == 2            # one evaluation = the two packed buffer all-reduces
    model.close()
"""


def test_two_ranks_on_one_device(tmp_path):
    """Two processes, one shard each (both on device 0, gloo), one launch: (1) bound and every gradient equal the oracle's on the
    concatenated data -- the all-reduce of the packed device buffers is the statistics_reducer (local_MapReduce.py:250-277); (2) the
    reference's 2-shard SCG and GD runs are reproduced call by call with one shard per process -- statistics / gradient all-reduces on
    the device buffers plus the optimisers' scalar sum / max reductions (scg_adapted_local_MapReduce.py:59-155); (3) bench.py --gpus 2."""
    text = (RANK_SCRIPT + SCG_RANK_SCRIPT) % {'root': ROOT} + BENCH_TAIL % {'bench_shape': ['--N', '30000', '--D', '12', '--M', '96', '--Q', '5']}
    _launch(tmp_path, 'two_rank_script.py', text, 2, 600)


EIGHT_RANK_SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from gparml_amd.engine import ShardEngine
from gparml_amd.dist import DistributedEvaluator, draw_kept_mask
from gparml_amd.resident import ResidentCG, ResidentModel
from oracle import factorised as Fz
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=rank, world_size=world)
N, D, M, Q = 2400, 5, 48, 5
cut = [int(round(i * N / float(world))) + (7 if 0 < i < world else 0) for i in range(world + 1)]      # ragged shards
a, b = cut[rank], cut[rank + 1]
for regime in ('A', 'B'):
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=41, zseed=42, alpha_value=0.5)
    eng = ShardEngine(b - a, D, M, Q)
    eng.upload_shard(d['Y'][a:b], d['X_mu'][a:b], d['X_S'][a:b])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'], N_global=N)
    ev = DistributedEvaluator(eng, device=torch.device('cuda', 0))
    out = ev.evaluate(regime == 'B')
    ref = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'])
    assert abs(out['F'] - ref['F']) <= 1e-6 * abs(ref['F']), (out['F'], ref['F'])
    for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta'):
        assert np.max(np.abs(np.asarray(out[k]) - np.asarray(ref[k]))) <= 1e-5 * np.max(np.abs(ref[k])), k
    # node drop-out at world_size 8: every rank draws the same mask from the shared seed; the kept shards alone, rescaled, are the oracle's
    # statistics of the kept rows divided by the kept fraction (local_MapReduce.py:119-129, 263-264)
    kept, frac = draw_kept_mask(world, 0.4, np.random.RandomState(5))
    assert 0 < sum(kept) < world
    out = ev.evaluate(False, kept_mask=kept, kept_fraction=frac)
    rows = np.concatenate([np.arange(cut[r], cut[r + 1]) for r in range(world) if kept[r]])
    st = Fz.phase1(d['Z'], d['sf2'], d['alpha'], d['Y'][rows], d['X_mu'][rows], d['X_S'][rows])
    for k in ('sum_exp_K_mi_K_im', 'exp_K_miY', 'sum_YYT', 'sum_exp_K_ii', 'KL'):
        st[k] = st[k] / frac
    gs = Fz.global_step(d['Z'], d['sf2'], d['alpha'], d['beta'], st, N, D)
    assert abs(out['F'] - gs['F']) <= 1e-6 * abs(gs['F']), ('drop-out', out['F'], gs['F'])
    eng.close()
# the jitter branch taken by all eight ranks together: statistics that make Kmm + beta Psi2 barely indefinite on every rank
d = Fz.synthetic_shard(N, D, M, Q, regime='A', seed=43, zseed=44, alpha_value=0.5)
from oracle import literal as L
Kmm = L.rbf_gram(d['Z'], d['sf2'], d['alpha'])
lam = np.linalg.eigvalsh(Kmm)[0]
class Fixed(ShardEngine):
    def phase1(self):
        ShardEngine.phase1(self)
        # every rank contributes 1/world of a Psi2 that shifts the smallest eigenvalue of Kmm + beta Psi2 to -5e-8
        self.set_local_statistics(1.0, -(lam + 5e-8) / d['beta'] * np.eye(M) / world, np.zeros((M, D)), d['sf2'] * (b - a), 0.0)
eng = Fixed(b - a, D, M, Q)
eng.upload_shard(d['Y'][a:b], d['X_mu'][a:b], d['X_S'][a:b])
eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'], N_global=N)
out = DistributedEvaluator(eng, device=torch.device('cuda', 0)).evaluate(False)
assert np.isfinite(out['F']) and np.all(np.isfinite(out['grad_Z']))
Fs = [None] * world
dist.all_gather_object(Fs, out['F'])
assert len(set(Fs)) == 1, Fs                      # the replicated global step (with the retry) gives the same bits on every rank
eng.close()
# resident SCG helpers at world_size 8: the reductions over eight shards equal the single-process values
d = Fz.synthetic_shard(N, D, M, Q, regime='B', seed=45, zseed=46, alpha_value=0.5)
raw = np.log(np.exp(d['X_S']) - 1.0)
model = ResidentModel([(d['Y'][a:b], d['X_mu'][a:b], raw[a:b])], M, Q, D, fixed_embeddings=False)
assert model.N == N
x = np.concatenate([d['Z'].ravel(), [0.5], np.full(Q, 0.3), [1.2]])
f, g = model.likelihood_and_gradient(x, 0)
cg = ResidentCG(model)
cg.embeddings_set_grads(None)
vals = np.array([cg.embeddings_get_grads_mu(), cg.embeddings_get_grads_kappa(), cg.embeddings_get_grads_current_grad()])
alls = [None] * world
dist.all_gather_object(alls, (f, vals))
assert all(abs(o[0] - alls[0][0]) == 0 and np.array_equal(o[1], alls[0][1]) for o in alls)
model.close()
"""


def test_eight_ranks_on_one_device(tmp_path):
    """world_size 8 (BASELINE configs[3] / [4] run at 8 ranks) on a 1-GPU box: eight processes share device 0 over gloo, one launch.  Covers
    what depends on the rank count: ragged eight-way sharding against the oracle (both regimes), the shared-seed drop-out mask with several
    dropped ranks, the jitter retry taken by all ranks together, the resident optimiser reductions, and bench.py --gpus 8 end to end
    with its all-reduce timings."""
    text = EIGHT_RANK_SCRIPT % {'root': ROOT} + BENCH_TAIL % {'bench_shape': ['--N', '20000', '--D', '12', '--M', '96', '--Q', '5']}
    _launch(tmp_path, 'eight_rank_script.py', text, 8, 600)
