"""Parity at the benchmark's own workload and FULL size (BASELINE configs[2]: N=1e6, D=100, M=512, Q=10, alpha=0.1, beta=10,
bench.py's generator) against an extended-precision truth, on TEN data / inducing-point draws at N = 1e5 and N = 1e6.

tests/golden/make_hp_truth_large.py -> oracle/hp_truth.c evaluates the workload in x87 80-bit long double end to end (its own
uncertainty, measured by re-running the global step on the reversed order of the inducing points: 0.7e-8 .. 2.6e-8 on grad_Z) and stores
the truth with the errors of the two float64 CPU arrangements.  cond(Kmm + beta Psi2) is 1.4e10 .. 3.1e10 over the draws, and on EVERY draw
both float64 CPU arrangements are outside the 1e-5 contract on grad_Z: the reference's LU arrangement by 2.1e-5 .. 8.9e-5, the Cholesky
port by 1.4e-5 .. 4.3e-5.  The device's round-3 global step (float64 + one refinement step of E) was 5.7e-6 .. 2.1e-5: inside the contract on
the benchmark's own draw, OUTSIDE it on draw (102, 12) at both sizes (profiles/r04_seed_floor_N1e5.txt, _N1e6.txt).

Round 4 located the error: it is the float64 ACCUMULATION of one product of the global step, G = Kmm^-1 Psi2 (entries of both signs around
1e5 times entries around N, the result a small difference).  With that product accumulated in double-double (csrc/linalg.hip,
ddacc_gemm_kernel; inputs and output float64; +0.05 ms) the device is 1.2e-8 .. 5.7e-8 from the truth on grad_Z -- the truth's own uncertainty --
on every draw and at both sizes.  The device path is held to

    err(GPU, truth) <= 1e-5   on every gradient block, every draw, both sizes (BASELINE.json's contract; relative to the block's largest magnitude)
    err(GPU, truth) <= 5e-7   on grad_Z (regression guard: what the double-double product delivers, with a margin over the truth's uncertainty)
    |F_gpu - F_truth| <= 1e-9 |F_truth|
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

BLOCKS = ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta')


def _draws():
    """(data seed, inducing-point seed) of every committed truth: the benchmark's own draw first, then the other data / inducing-point draws of round 4
    (tests/golden/make_hp_truth_large.py N seed z_seed) that exist at BOTH sizes."""
    import glob
    import re
    found = {}
    for f in glob.glob(os.path.join(GOLDEN_DIR, 'hp_truth_large_N*_s*_z*.npz')):
        m = re.search(r'_N(\d+)_s(\d+)_z(\d+)\.npz$', f)
        found.setdefault((int(m.group(2)), int(m.group(3))), set()).add(int(m.group(1)))
    return [(100, None)] + sorted(k for k, v in found.items() if {100000, 1000000} <= v)


DRAWS = _draws()


def _fixture(N, seed=100, z_seed=None):
    tag = '' if (seed, z_seed) == (100, None) else '_s%d_z%d' % (seed, z_seed)
    return np.load(os.path.join(GOLDEN_DIR, 'hp_truth_large_N%d%s.npz' % (N, tag)))


def _inputs(z):
    import bench
    N, D, M, Q, seed = (int(z[k]) for k in ('N', 'D', 'M', 'Q', 'seed'))
    z_seed = int(z['z_seed']) if 'z_seed' in z.files and int(z['z_seed']) >= 0 else None
    d = bench.synthetic(N, D, M, Q, seed=seed, z_seed=z_seed)
    # the truth belongs to these inputs: the generator must reproduce them (sin() may differ in the last bit between hosts)
    np.testing.assert_allclose(bench.input_checksums(d), z['input_checksums'], rtol=1e-11, atol=1e-9)
    return d, (N, D, M, Q)


def _err(x, truth):
    return float(np.max(np.abs(np.asarray(x) - truth)) / np.max(np.abs(truth)))


def test_fixtures_describe_the_benchmark_workload():
    assert len(DRAWS) >= 5
    for N in (100000, 1000000):
        for seed, z_seed in DRAWS:
            z = _fixture(N, seed, z_seed)
            assert (int(z['D']), int(z['M']), int(z['Q']), int(z['seed'])) == (100, 512, 10, seed)
            assert float(z['cond_A']) > 1e10
            assert float(z['truth_uncertainty'][1]) < 1e-7          # the truth itself is good to 1e-7 on grad_Z
            assert z['truth_grad_Z'].shape == (512, 10)
            # no float64 CPU arrangement is inside the contract on any draw: the reference's LU arrangement is 3e-5 .. 1e-4 away on grad_Z
            assert float(z['err_lu_grad_Z']) > 1e-5


def test_float64_cpu_paths_against_the_truth_1e5():
    """Both float64 arrangements re-run here at N = 1e5: their distance from the truth is what the fixture recorded (to within the
    run-to-run spread of a threaded BLAS), F agrees to 1e-9, and grad_Z is beyond 1e-5 for both -- the bar the device test uses."""
    from oracle import factorised as Fz
    z = _fixture(100000)
    d, (N, D, M, Q) = _inputs(z)
    for name, linalg in (('chol', 'cholesky'), ('lu', 'lu')):
        o = Fz.evaluate_blas(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], linalg=linalg)
        assert abs(o['F'] - float(z['truth_F'])) <= 1e-9 * abs(float(z['truth_F']))
        for k in BLOCKS:
            e, rec = _err(o[k], z['truth_' + k]), float(z['err_%s_%s' % (name, k)])
            assert e <= 10 * rec + 1e-12, (name, k, e, rec)
        assert _err(o['grad_Z'], z['truth_grad_Z']) > 2e-6     # the conditioning floor is real, not an artefact of one run


@pytest.mark.gpu
@pytest.mark.parametrize('N', [100000, 1000000])
@pytest.mark.parametrize('seed,z_seed', DRAWS)
def test_gpu_against_the_long_double_truth(N, seed, z_seed):
    """The benchmark's kernel sequence (fixed embeddings: p1v2_kernel -> global step -> p2_fast8_kernel<3>) on ten draws of the benchmark's
    workload, at N = 1e5 and at the full N = 1e6.  No escape clause: 1e-5 on every block."""
    from gparml_amd.engine import ShardEngine
    z = _fixture(N, seed, z_seed)
    d, (N, D, M, Q) = _inputs(z)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    out = eng.evaluate(False)
    eng.close()
    assert abs(out['F'] - float(z['truth_F'])) <= 1e-9 * abs(float(z['truth_F']))
    rep = {k: (_err(out[k], z['truth_' + k]), float(z['err_lu_' + k]), float(z['err_chol_' + k])) for k in BLOCKS}
    print('N=%d draw (%s, %s) cond %.2e error vs truth (gpu, float64 LU, float64 Cholesky):' % (N, seed, z_seed, float(z['cond_A'])),
          {k: '%.2e %.2e %.2e' % v for k, v in rep.items()})
    for k in BLOCKS:
        assert rep[k][0] <= 1e-5, (k, rep[k])
    assert rep['grad_Z'][0] <= 5e-7, rep['grad_Z']


@pytest.mark.gpu
def test_the_double_double_product_is_what_closes_the_gap():
    """The same evaluation with the round-3 global step (gp_debug_set_option('dd_kipsi2', 0): K_mm^-1 Psi2 on the float64 matrix core) on the
    draw where it fails the contract: 1.4e-5 at N = 1e5 (2.1e-5 at N = 1e6), against < 5e-7 with the double-double accumulation."""
    from gparml_amd import _lib
    from gparml_amd.engine import ShardEngine
    z = _fixture(100000, 102, 12)
    d, (N, D, M, Q) = _inputs(z)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    lib = _lib.load()
    try:
        assert lib.gp_debug_set_option(b'dd_kipsi2', 0) == 0
        plain = _err(eng.evaluate(False)['grad_Z'], z['truth_grad_Z'])
    finally:
        assert lib.gp_debug_set_option(b'dd_kipsi2', 1) == 0
    dd = _err(eng.evaluate(False)['grad_Z'], z['truth_grad_Z'])
    eng.close()
    print('draw (102, 12), N = 1e5: grad_Z vs truth  float64 product %.2e   double-double product %.2e' % (plain, dd))
    assert plain > 5e-6 and dd <= 5e-7 and dd < plain / 10
    assert lib.gp_debug_set_option(b'no_such_option', 1) == _lib.GP_ERR_BAD_ARG


@pytest.mark.gpu
def test_global_step_products_on_the_int8_matrix_core_at_M_1024():
    """csrc/gsi8.hip (round 5): from M = 1024 on, the global step's two double-double products -- K_mm^-1 Psi2 and the residual of the refinement step --
    run as ten-digit exact products on the int8 matrix core.  The benchmark's generator at N = 5e4, D = 100, M = 1024, Q = 10: cond(K_mm + beta Psi2) =
    5.6e11, where the float64 Cholesky port is 7.5e-4 and the reference's LU arrangement 1.8e-3 from the 80-bit truth on grad_Z and the truth's own
    uncertainty is 8.5e-7 (tests/golden/make_hp_truth_large.py 50000 100 -1 1024).  Both device paths must stay within the contract of the truth, and the
    int8 path must be as close to it as the double-double path it replaces."""
    from gparml_amd import _lib
    from gparml_amd.engine import ShardEngine
    z = np.load(os.path.join(GOLDEN_DIR, 'hp_truth_M1024_N50000.npz'))
    d, (N, D, M, Q) = _inputs(z)
    assert M == 1024
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    lib = _lib.load()
    rep = {}
    try:
        for tag, on in (('int8', 1), ('double-double', 0)):
            assert lib.gp_debug_set_option(b'gs_i8', on) == 0
            out = eng.evaluate(False)
            rep[tag] = {k: _err(out[k], z['truth_' + k]) for k in BLOCKS}
            rep[tag]['F'] = abs(out['F'] - float(z['truth_F'])) / abs(float(z['truth_F']))
    finally:
        assert lib.gp_debug_set_option(b'gs_i8', 1) == 0
    eng.close()
    unc = float(z['truth_uncertainty'][1])
    print('M = 1024, N = 5e4, cond %.1e: error vs the 80-bit truth (its own uncertainty on grad_Z %.1e; float64 Cholesky %.1e, LU %.1e):' % (
        float(z['cond_A']), unc, float(z['err_chol_grad_Z']), float(z['err_lu_grad_Z'])), {t: {k: '%.1e' % v for k, v in r.items()} for t, r in rep.items()})
    for tag in rep:
        assert rep[tag]['F'] <= 1e-9
        for k in BLOCKS:
            assert rep[tag][k] <= 1e-5, (tag, k, rep[tag][k])
    assert rep['int8']['grad_Z'] <= 2.0 * rep['double-double']['grad_Z'] + 2.0 * unc
