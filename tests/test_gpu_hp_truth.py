"""Parity at the benchmark's own hyper-parameters (BASELINE configs[2]: D=100, M=512, Q=10, alpha=0.1, beta=10, bench.py's Z)
against an extended-precision truth, through the kernel the benchmark runs (fixed embeddings -> the fast phase-2 kernel).

tests/golden/make_hp_golden.py evaluates the same inputs three ways: in 80-bit long double (the truth), with the imported
reference (float64, LU inv/slogdet) and with oracle/factorised.py (float64, Cholesky).  cond(Kmm + beta Psi2) = 5.9e9 here and
grad_Z is the small difference of two large parts, so the two float64 CPU paths differ from each other by 1.4e-5 -- but each is
within 1e-5 of the truth (reference 8.5e-6, Cholesky port 6.1e-6).  The device path is held to the same bar:

    error(GPU, truth) <= max(1e-5, error(reference, truth))     per gradient block, relative to the block's largest magnitude
    |F_gpu - F_truth| <= 1e-6 |F_truth|
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

FIXTURE = os.path.join(GOLDEN_DIR, 'hp_truth_config2_N4000.npz')
BLOCKS = ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta')
G_RTOL = 1e-5
F_RTOL = 1e-6


def _load():
    z = np.load(FIXTURE)
    inp = dict(Y=z['in_Y'], X_mu=z['in_X_mu'], Z=z['in_Z'], sf2=float(z['in_sf2']), alpha=z['in_alpha'], beta=float(z['in_beta']))
    return z, inp


def _err(x, truth):
    return float(np.max(np.abs(np.asarray(x) - truth)) / np.max(np.abs(truth)))


def test_fixture_is_the_benchmark_workload():
    """The stored inputs are bench.py's synthetic() at the benchmark's hyper-parameters (same generator, same seed)."""
    import bench
    z, inp = _load()
    N, D, M, Q = (int(z['in_' + k]) for k in 'NDMQ')
    assert (D, M, Q) == (100, 512, 10)
    d = bench.synthetic(N, D, M, Q, seed=100)
    np.testing.assert_array_equal(d['Z'], inp['Z'])
    np.testing.assert_array_equal(d['X_mu'], inp['X_mu'])
    np.testing.assert_allclose(d['Y'], inp['Y'], rtol=0, atol=1e-15)      # sin() may differ in the last bit between CPUs
    assert d['alpha'][0] == inp['alpha'][0] == 0.1 and d['beta'] == inp['beta'] == 10.0
    assert float(z['cond_A']) > 1e9                                        # the regime the verdict asked about


def test_cpu_paths_against_the_truth():
    """The recorded errors of the reference and the port, and the port re-run here, stay under 1e-5 of the truth."""
    from oracle import factorised as Fz
    z, inp = _load()
    X_S = np.zeros_like(inp['X_mu'])
    out = Fz.evaluate(inp['Z'], inp['sf2'], inp['alpha'], inp['beta'], inp['Y'], inp['X_mu'], X_S, want_embeddings=False)
    assert abs(out['F'] - float(z['truth_F'])) <= F_RTOL * abs(float(z['truth_F']))
    for k in BLOCKS:
        assert float(z['err_ref_' + k]) <= G_RTOL, (k, float(z['err_ref_' + k]))
        assert _err(out[k], z['truth_' + k]) <= G_RTOL, k
        assert _err(z['ref_' + k], z['truth_' + k]) <= G_RTOL, k


@pytest.mark.gpu
@pytest.mark.parametrize('emb', [False, True])
def test_gpu_against_the_truth(emb):
    """emb=False is the benchmark's kernel sequence (p1_kernel8 -> global step -> p2_fast8_kernel<3>); emb=True the general one."""
    from gparml_amd.engine import ShardEngine
    z, inp = _load()
    N, D = inp['Y'].shape
    M, Q = inp['Z'].shape
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(inp['Y'], inp['X_mu'], np.zeros((N, Q)))
    eng.set_globals(inp['Z'], inp['sf2'], inp['alpha'], inp['beta'])
    eng.phase1()
    eng.global_step()
    eng.phase2(emb)
    out = eng.finish()
    eng.close()
    assert abs(out['F'] - float(z['truth_F'])) <= F_RTOL * abs(float(z['truth_F']))
    report = {}
    for k in BLOCKS:
        e = _err(out[k], z['truth_' + k])
        report[k] = (e, float(z['err_ref_' + k]), float(z['err_oracle_' + k]))
    print('error vs truth (gpu, reference LU, port Cholesky):', {k: '%.2e %.2e %.2e' % v for k, v in report.items()})
    for k in BLOCKS:
        assert report[k][0] <= max(G_RTOL, report[k][1]), (k, report[k])


@pytest.mark.gpu
def test_device_agrees_with_a_60_digit_evaluation_where_the_float64_oracle_does_not():
    """N = 17 points under M = 129 inducing points spread at random in three dimensions (cond(K_mm) 2.7e7, cond(K_mm + beta Psi2) 1.8e9): the float64
    CPU oracle is 4.5e-3 away from a 60-digit mpmath evaluation on grad_Z (tests/golden/make_mp_truth_small.py -> mp_truth_N17_M129.npz) -- it accumulates
    K_mm^-1 Psi2 in float64 -- while the device, which carries that one product in double-double (csrc/linalg.hip ddacc_gemm_kernel), stays inside the 1e-5 contract.
    Found by tools/soak.sh's shape fuzz, which compares the device with the oracle and flagged the case."""
    import os
    from conftest import GOLDEN_DIR
    from gparml_amd.engine import ShardEngine
    from oracle import factorised as Fz
    z = np.load(os.path.join(GOLDEN_DIR, 'mp_truth_N17_M129.npz'))
    N, D = z['Y'].shape
    M, Q = z['Z'].shape
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(z['Y'], z['X_mu'], np.zeros((N, Q)))
    eng.set_globals(z['Z'], float(z['sf2']), z['alpha'], float(z['beta']))
    out = eng.evaluate(False)
    eng.close()
    t = z['truth_grad_Z']
    err_dev = float(np.max(np.abs(out['grad_Z'] - t)) / np.max(np.abs(t)))
    ref = Fz.evaluate(z['Z'], float(z['sf2']), z['alpha'], float(z['beta']), z['Y'], z['X_mu'], np.zeros((N, Q)), want_embeddings=False)
    err_cpu = float(np.max(np.abs(ref['grad_Z'] - t)) / np.max(np.abs(t)))
    print('N = 17, M = 129: grad_Z vs 60-digit truth: device %.2e, float64 oracle %.2e (fixture %.2e)' % (err_dev, err_cpu, float(z['oracle_err_grad_Z'])))
    assert float(z['oracle_err_grad_Z']) > 1e-3            # the float64 arrangement carries three digits here
    assert err_dev <= 1e-5
