"""Regime-B phase 2 on tile pairs (gparml_amd/csrc/psi2_tile.hip; reference: partial_terms.py:190-205, 273-284, 388-394, 421-427).

The kernel is in charge from Q = 17 on (below that the VALU kernels of psi2.hip are faster), so the seeded shapes of test_gpu_parity.py
reach it only at its three widest instantiations.  Here: (1) every compiled width (4 ... 52), several point chunks per launch, diagonal-only
and many-tile layouts, with the kernel forced (GPARML_B_PHASE2=tiles, read once per process: a child process); (2) BASELINE configs[4]'s
per-GPU shape (D=1000, M=1024, Q=50) with MORE points than inducing points (N=2048, alpha = 1/Q), against the oracle evaluated in
sixteen shards on the host's threads."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_cache
from conftest import ROOT, assert_close

pytestmark = pytest.mark.gpu

CHILD = r'''
import sys
import numpy as np
sys.path.insert(0, %(root)r)
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
# N, D, M, Q, alpha: widths 4, 8, 12, 16, 24, 32, 40, 52, 64; one slab .. five slabs (15 tiles); 9000 and 20000 points = 2 and 3 launches of 8192
SHAPES = [(300, 5, 20, 3, 0.5), (9000, 3, 200, 6, 0.3), (1000, 7, 130, 10, 0.3), (20000, 2, 64, 11, 0.2), (640, 3, 33, 13, 0.2), (500, 4, 300, 15, 0.1),
          (400, 2, 70, 20, 0.1), (700, 2, 129, 23, 0.1), (300, 2, 40, 30, 0.08), (350, 2, 65, 31, 0.05), (200, 2, 24, 50, 0.05), (257, 2, 1, 1, 1.0),
          (333, 2, 100, 51, 0.03), (300, 2, 70, 52, 0.03), (280, 3, 150, 60, 0.03), (9000, 2, 64, 63, 0.03), (450, 2, 140, 36, 0.05), (300, 2, 64, 39, 0.05)]
for (N, D, M, Q, alpha) in SHAPES:
    d = Fz.synthetic_shard(N, D, M, Q, regime='B', seed=31, zseed=32, alpha_value=alpha)
    ref = Fz.evaluate_sharded(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], shards=8, pairs='gemm')
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    out = eng.evaluate(True)
    keys = ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu', 'grad_X_S')
    errs = {k: float(np.max(np.abs(np.asarray(out[k]) - np.asarray(ref[k]))) / np.max(np.abs(ref[k]))) for k in keys}
    jit = eng.last_jitter      # none of these shapes needs the reference's jitter: a retry here is a factorisation that failed where it should not (conftest._no_silent_jitter)
    if jit or abs(out['F'] - ref['F']) > 1e-6 * abs(ref['F']) or max(errs.values()) > 1e-5:
        # (round 6: the cause was a race in the blocked Cholesky's panel solve on the first evaluation of a fresh process -- profiles/r06_first_evaluation_race.txt,
        # tests/test_gpu_first_evaluation.py; the diagnostics stay.)  A failure here was seen ONCE in round 5 (grad_Z 1e-4 off at (9000, 3, 200, 6) inside a full-suite run; 0 of 40 repeats since, with the
        # round-4 library as well: tools/stress_tile.sh): say everything that helps to place it -- the jitter branch, a repeat on the same context
        again = eng.evaluate(True)
        errs2 = {k: float(np.max(np.abs(np.asarray(again[k]) - np.asarray(ref[k]))) / np.max(np.abs(ref[k]))) for k in keys}
        print('TILE_FAIL', (N, D, M, Q), 'F', out['F'], ref['F'], 'errors', errs, 'jitter mask', jit, '| repeated on the same context:', errs2,
              'jitter mask', eng.last_jitter, flush=True)
        raise SystemExit(1)
    eng.close()
    print('TILE_OK', N, D, M, Q)
'''


def test_every_compiled_width_with_the_kernel_forced(tmp_path):
    script = tmp_path / 'tile_child.py'
    script.write_text(CHILD % {'root': ROOT})
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=1800, cwd=ROOT, env=dict(os.environ, GPARML_B_PHASE2='tiles'))
    assert r.returncode == 0 and r.stdout.count('TILE_OK') == 18, r.stdout[-1500:] + r.stderr[-3000:]


def test_config4_shape_with_more_points_than_inducing_points():
    """D=1000, M=1024, Q=50, free embeddings, N=2048 > M, alpha = 1/Q (the regime BASELINE configs[4] runs in: A = K_mm + beta Psi2 is
    dominated by the statistics, not a low-rank update of K_mm as in test_gpu_parity.test_config4_shape): 136 tiles, two point streams,
    the column-of-ones and padding columns of the 52-wide instantiation."""
    from gparml_amd.engine import ShardEngine
    from oracle import factorised as Fz
    N, D, M, Q = 2048, 1000, 1024, 50
    rs = np.random.RandomState(7)
    d = Fz.synthetic_shard(N, D, 64, Q, regime='B', seed=6, zseed=7, alpha_value=1.0 / Q)
    d['Z'] = d['X_mu'][rs.permutation(N)[:M]] + 0.3 * rs.randn(M, Q)
    ref = oracle_cache.get('config4_shape_N2048', d, lambda: Fz.evaluate_sharded(
        d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], shards=32, workers=min(32, os.cpu_count() or 8), pairs='gemm'))
    oracle_cache.done()
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    out = eng.evaluate(True)
    eng.close()
    assert_close(out['F'], ref['F'], 1e-6, what='F')
    for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu', 'grad_X_S'):
        assert_close(out[k], ref[k], 1e-5, what=k)
