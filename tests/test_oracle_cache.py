"""tests/oracle_cache.py (committed outputs of the CPU oracle for the GPU tests whose oracle evaluation is minutes of host time): the sampled
comparison does what it says, a committed file is only used for exactly its inputs, one committed case is recomputed live in this process and
two more (every one with GPARML_AUDIT_ORACLE_CACHE_ALL=1) through the tests that use them, in audit mode."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import oracle_cache
from conftest import ROOT, assert_close


def test_sampled_comparison_sees_a_single_wrong_row_and_a_systematic_error():
    rs = np.random.RandomState(0)
    a = rs.randn(50000, 10)
    s = oracle_cache.Sampled.of(a)
    assert s.values.size <= oracle_cache.MAX_ELEMS and s.rows[0] == 0 and s.rows[-1] == a.shape[0] - 1 and s.rowp.shape == (a.shape[0], 2)
    assert {127, 128, 129, a.shape[0] - 128, (a.shape[0] // 128) * 128}.issubset(set(s.rows.tolist()))        # tile-boundary rows are among the exact ones
    assert_close(a, s, 1e-12, what='identical')
    assert_close(a * (1 + 1e-7), s, 1e-5, what='within tolerance')
    b = a.copy()
    b[s.rows[7], 3] += 1e-3                                   # a stored row
    with pytest.raises(AssertionError):
        assert_close(b, s, 1e-5)
    b = a.copy()
    free = np.setdiff1d(np.arange(a.shape[0]), s.rows)[123]
    b[free] += 0.5                                            # a row that is NOT stored: its projections see it
    with pytest.raises(AssertionError):
        assert_close(b, s, 1e-5)
    b = a.copy()
    b[free, 4] += 2e-4 * s.scale                              # ONE element of an unsampled row off by 20 x the tolerance (round 5's sums: 1265 x)
    with pytest.raises(AssertionError, match='row %d' % free):
        assert_close(b, s, 1e-5)
    with pytest.raises(AssertionError):
        assert_close(a + 1e-4, s, 1e-5)                       # a small systematic shift: every row within 2.3e-5 of the max-norm, the sums are not


def test_a_committed_case_is_used_only_for_its_inputs_and_reproduces_live():
    from oracle import factorised as Fz
    N, D, M, Q = 800, 2, 700, 7
    d = Fz.synthetic_shard(N, D, M, Q, regime='B', seed=11, zseed=12, alpha_value=0.3)       # tests/test_gpu_parity.py SHAPES
    calls = []

    def live():
        calls.append(1)
        return Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'])

    ref = oracle_cache.get('seeded_800_2_700_7_B', d, live)
    assert not calls and isinstance(ref['stats']['sum_exp_K_mi_K_im'], oracle_cache.Sampled)        # served from the committed file
    t = time.time()
    now = live()
    print('live oracle: %.1f s' % (time.time() - t))
    assert_close(now['F'], ref['F'], 1e-12, what='F')
    for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu', 'grad_X_S'):
        assert_close(now[k], ref[k], 1e-9, what=k)
    assert_close(now['stats']['sum_exp_K_mi_K_im'], ref['stats']['sum_exp_K_mi_K_im'], 1e-12, what='Psi2')
    # other inputs under the same key: the file is ignored and the oracle runs
    d2 = dict(d, Y=d['Y'] * (1 + 1e-9))
    stub = []
    assert oracle_cache.get('seeded_800_2_700_7_B', d2, lambda: stub.append(1) or {'F': 0.0}) == {'F': 0.0} and stub == [1]


# committed file -> the GPU test that builds its inputs and calls oracle_cache.get() before it touches the device
CASES = {
    'seeded_800_2_700_7_B': 'tests/test_gpu_parity.py::test_against_oracle_on_seeded_inputs[800-2-700-7-B-0.3]',
    'seeded_1100_2_1024_6_B': 'tests/test_gpu_parity.py::test_against_oracle_on_seeded_inputs[1100-2-1024-6-B-0.8]',
    'config1_full_B': 'tests/test_gpu_parity.py::test_config1_full_evaluation[B-True]',
    'config4_shape_N320': 'tests/test_gpu_parity.py::test_config4_shape',
    'config4_shape_N2048': 'tests/test_gpu_tile_phase2.py::test_config4_shape_with_more_points_than_inducing_points',
    'config2_full_blas_port_alpha0.3': 'tests/test_gpu_fullsize.py::test_full_size_against_the_blas_port',
    'config2_full_blas_port_alpha0.1': 'tests/test_gpu_fullsize.py::test_full_size_against_the_blas_port',
    'config4_fullsize_slice_2e4': 'tests/test_gpu_config4_fullsize.py::test_oracle_on_a_2e4_point_slice_of_the_same_workload',
}
DEFAULT_AUDIT = ('config4_shape_N320', 'config2_full_blas_port_alpha0.1', 'config2_full_blas_port_alpha0.3')     # + seeded_800_2_700_7_B in the test above: four of eight on every CPU run (~90 s)


def test_every_committed_file_has_a_case_and_the_other_way_round():
    files = sorted(f[:-4] for f in os.listdir(oracle_cache.CACHE_DIR) if f.endswith('.npz'))
    assert files == sorted(CASES)


def test_committed_files_against_the_live_oracle():
    """Audit mode of tests/oracle_cache.py through the very tests that use the files: get() regenerates the inputs, runs the oracle live and compares
    every stored output at 1e-9; done() then skips the device part.  Three files by default (GPARML_AUDIT_ORACLE_CACHE_ALL=1: all eight -- minutes of
    host time and 16 GB for the full-size cases; GPARML_SKIP_ORACLE_AUDIT=1: none)."""
    if os.environ.get('GPARML_SKIP_ORACLE_AUDIT'):
        pytest.skip('GPARML_SKIP_ORACLE_AUDIT')
    keys = sorted(CASES) if os.environ.get('GPARML_AUDIT_ORACLE_CACHE_ALL') else list(DEFAULT_AUDIT)
    nodes = sorted({CASES[k] for k in keys})
    env = dict(os.environ, GPARML_AUDIT_ORACLE_CACHE='1', GPARML_AUDIT_KEYS=','.join(keys))
    env.pop('GPARML_WRITE_ORACLE_CACHE', None)
    t = time.time()
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-s', '-m', 'gpu', '-p', 'no:cacheprovider'] + nodes, capture_output=True, text=True, cwd=ROOT, env=env,
                       timeout=7200)
    import re
    done = re.findall(r'ORACLE_CACHE_AUDIT (\S+) ok: [^\n]*', r.stdout)          # (pytest -s puts its progress characters in front of the line)
    print('\n'.join(m.group(0) for m in re.finditer(r'ORACLE_CACHE_AUDIT [^\n]*', r.stdout)), '\n(%.0f s)' % (time.time() - t))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert sorted(done) == sorted(keys), (done, r.stdout[-2000:])
