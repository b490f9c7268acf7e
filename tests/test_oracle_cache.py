"""tests/oracle_cache.py (committed outputs of the CPU oracle for the GPU tests whose oracle evaluation is minutes of host time): the sampled
comparison does what it says, a committed file is only used for exactly its inputs, and one committed case is recomputed live."""
import os
import time

import numpy as np
import pytest

import oracle_cache
from conftest import assert_close


def test_sampled_comparison_sees_a_single_wrong_row_and_a_systematic_error():
    rs = np.random.RandomState(0)
    a = rs.randn(50000, 10)
    s = oracle_cache.Sampled.of(a)
    assert s.values.size <= oracle_cache.MAX_ELEMS and s.rows[0] == 0 and s.rows[-1] == a.shape[0] - 1
    assert_close(a, s, 1e-12, what='identical')
    assert_close(a * (1 + 1e-7), s, 1e-5, what='within tolerance')
    b = a.copy()
    b[s.rows[7], 3] += 1e-3                                   # a stored row
    with pytest.raises(AssertionError):
        assert_close(b, s, 1e-5)
    b = a.copy()
    free = np.setdiff1d(np.arange(a.shape[0]), s.rows)[123]
    b[free] += 0.5                                            # a row that is NOT stored: the weighted sums over all rows see it
    with pytest.raises(AssertionError):
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
