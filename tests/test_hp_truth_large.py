"""Parity at the benchmark's own workload and FULL size (BASELINE configs[2]: N=1e6, D=100, M=512, Q=10, alpha=0.1, beta=10,
bench.py's generator, seed 100) against an extended-precision truth.

tests/golden/make_hp_truth_large.py -> oracle/hp_truth.c evaluates the workload in x87 80-bit long double end to end (its own
uncertainty, measured by re-running the global step on the reversed order of the inducing points: 1.4e-8 on grad_Z) and stores
the truth with the errors of the two float64 CPU arrangements:

    N = 1e5: cond(Kmm + beta Psi2) = 1.37e10   float64 LU (the reference's arrangement) 5.3e-5, float64 Cholesky port 1.6e-5 on grad_Z
    N = 1e6: cond(Kmm + beta Psi2) = 1.41e10   float64 LU 8.9e-5,                      float64 Cholesky port 2.3e-5 on grad_Z

i.e. NO float64 evaluation of this bound certifies grad_Z to 1e-5 at this conditioning: with the long-double partials rounded to
double the float64 phase 2 reproduces grad_Z to 1.9e-10, so the whole error is the M x M global step, and even a long-double
global step on double-rounded statistics is left with 6e-9 on dF/dPsi2, which grad_Z (a ~1000-fold amplification through the
K_mm part) turns into ~1e-6..1e-5.  The device's global step refines E = (K_mm + beta Psi2)^-1 Psi1^T Y once with a
double-double residual (csrc/linalg.hip, solve_residual_kernel), which takes its grad_Z from 3.7e-5 / 1.3e-5 to 1.07e-5 / 7.5e-6 (N = 1e5 / 1e6).
The device path is held to

    err(GPU, truth) <= max(1e-5, best float64 CPU arrangement's error)    per gradient block, relative to the block's largest magnitude
    err(GPU, truth) <= 1e-5 on every block at the full size N = 1e6 (the headline configuration)
    |F_gpu - F_truth| <= 1e-9 |F_truth|
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

BLOCKS = ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta')


# (data seed, inducing-point seed): the benchmark's own draw first, then four more data / inducing-point draws (round 4)
DRAWS = [(100, None), (101, 11), (102, 12), (103, 13), (104, 14)]


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
def test_gpu_against_the_long_double_truth(N):
    """The benchmark's kernel sequence (fixed embeddings: p1v2_kernel -> global step -> p2_fast8_kernel<3>) on the benchmark's own
    inputs, at N = 1e5 and at the full N = 1e6."""
    from gparml_amd.engine import ShardEngine
    z = _fixture(N)
    d, (N, D, M, Q) = _inputs(z)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    out = eng.evaluate(False)
    eng.close()
    assert abs(out['F'] - float(z['truth_F'])) <= 1e-9 * abs(float(z['truth_F']))
    rep = {k: (_err(out[k], z['truth_' + k]), float(z['err_lu_' + k]), float(z['err_chol_' + k])) for k in BLOCKS}
    print('N=%d error vs truth (gpu, float64 LU, float64 Cholesky):' % N, {k: '%.2e %.2e %.2e' % v for k, v in rep.items()})
    for k in BLOCKS:
        assert rep[k][0] <= max(1e-5, min(rep[k][1], rep[k][2])), (k, rep[k])
        if N == 1000000:
            assert rep[k][0] <= 1e-5, (k, rep[k])
