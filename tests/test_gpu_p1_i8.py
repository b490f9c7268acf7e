"""Phase 1 on the int8 matrix core (gparml_amd/csrc/p1i8.hip; opt-in -- the outcome of the Ozaki gate, DESIGN.md section 6;
reference: partial_terms.py:45-52, 79-80 -- Psi2 = sum_n psi2_n, Psi1^T Y -- kernel_exp.py:13-49): exact integer products of six 7-bit
digits per operand, the 21 digit products with a + b <= 7, Psi2's diagonal from float64 sums of squares, instead of float64 MFMAs.

Checked with the path switched on (gp_debug_set_option('p1_i8', 1)): (1) the statistics against the float64 path of the same library
(p1v2_kernel) -- they differ by the 2^-42 truncation of the operands and the dropped digit products, 3.5e-15 in the exact CPU emulation;
(2) bound and gradients against the oracle at ragged shapes (five Psi2 row tiles, two Y column blocks, M = 1024); (3) bit-identical repeats
and independence of the slicing (integer sums); (4) the 80-bit truth of the benchmark workload at N = 1e5: grad_Z as close as with float64
statistics."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, assert_close

pytestmark = pytest.mark.gpu


def _lib():
    from gparml_amd import _lib
    return _lib.load()


def _eval(d, N, D, M, Q, i8):
    from gparml_amd.engine import ShardEngine
    lib = _lib()
    assert lib.gp_debug_set_option(b'p1_i8', 1 if i8 else 0) == 0
    try:
        eng = ShardEngine(N, D, M, Q)
        eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
        eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
        if i8:
            # the first int8 evaluation after an upload is the guard's check (both phase-1 paths, float64 statistics used): accepted here
            eng.evaluate(False)
            st = eng.i8_status()
            assert st['state'] == 1 and st['checks'] == 1 and 0 < st['rel_psi2'] < 1e-9 and 0 < st['rel_c'] < 1e-9 and st['cond_lower_bound'] > 1, st
            assert st['cond_lower_bound'] * max(st['rel_psi2'], st['rel_c']) <= 1e-4, st
        out = eng.evaluate(False)
        out['Psi2'], out['C'] = eng.download('PSI2_SUM'), eng.download('PSI1TY')
        out['timings'] = eng.timings()
        again = eng.evaluate(False)
        assert again['F'] == out['F'] and np.array_equal(again['grad_Z'], out['grad_Z'])          # bit-identical repeat
        eng.close()
    finally:
        lib.gp_debug_set_option(b'p1_i8', 0)
    return out


@pytest.mark.parametrize('N,D,M,Q,sf2,al', [(70000, 100, 512, 10, 1.0, 0.4), (66003, 130, 600, 7, 2.5, 0.4), (90000, 3, 512, 4, 0.7, 3.0), (65536, 40, 1024, 16, 1.0, 0.6)])
def test_int8_statistics_against_the_float64_path_and_the_oracle(N, D, M, Q, sf2, al):
    from oracle import factorised as Fz
    d = Fz.synthetic_shard(N, D, M, Q, regime='A', seed=11, zseed=12, alpha_value=al)
    d['sf2'] = sf2
    d['Y'] = d['Y'] * np.linspace(0.01, 30.0, D)[None, :]            # columns of very different scale: per-column digit scales
    a, b = _eval(d, N, D, M, Q, True), _eval(d, N, D, M, Q, False)
    assert not np.array_equal(a['Psi2'], b['Psi2'])                  # two different kernels did run
    # operands to 42 bits below their scale, digit products to order 8: the float64 kernel's own accumulation error is of the same size
    assert_close(a['Psi2'], b['Psi2'], 1e-11, what='Psi2 int8 vs float64')
    for dcol in range(D):
        assert_close(a['C'][:, dcol], b['C'][:, dcol], 1e-9, what='C[:, %d] int8 vs float64' % dcol)
    print((N, D, M, Q), 'p1 kernel ms: int8 %.3f float64 %.3f; psi1 ms %.3f %.3f' % (a['timings']['p1_kernel_ms'], b['timings']['p1_kernel_ms'],
                                                                               a['timings']['psi1_ms'], b['timings']['psi1_ms']))
    assert np.array_equal(a['Psi2'], a['Psi2'].T)                      # integer sums: exactly symmetric
    ref = Fz.evaluate_blas(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'])
    assert_close(a['F'], ref['F'], 1e-6, what='F')
    assert_close(a['F'], b['F'], 1e-9, what='F int8 vs float64')
    for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta'):
        assert_close(a[k], ref[k], 1e-5, what=k)
        assert_close(a[k], b[k], 2e-6, what=k + ' int8 vs float64')


def test_int8_sums_do_not_depend_on_the_slicing():
    """Integer accumulation is exact, so the digits' products are the same whatever the split of the rows: one shard against two shards added
    through the packed buffers (statistics equal to the rounding of the final float64 additions)."""
    from gparml_amd.engine import ShardEngine
    from oracle import factorised as Fz
    N, D, M, Q = 150000, 20, 512, 5
    d = Fz.synthetic_shard(N, D, M, Q, regime='A', seed=3, zseed=4, alpha_value=0.5)
    lib = _lib()
    assert lib.gp_debug_set_option(b'p1_i8', 1) == 0
    engines = []
    try:
        one = ShardEngine(N, D, M, Q); engines.append(one)
        one.upload_shard(d['Y'], d['X_mu'], d['X_S']); one.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
        one.evaluate(False); assert one.i8_status()['state'] == 1          # the guard's check (float64 statistics); from here on int8
        one.phase1()
        P1, C1 = one.download('PSI2_SUM'), one.download('PSI1TY')
        cut = 70001
        parts = []
        for sl in (slice(0, cut), slice(cut, N)):
            e = ShardEngine(sl.stop - sl.start, D, M, Q); engines.append(e)
            e.upload_shard(d['Y'][sl], d['X_mu'][sl], d['X_S'][sl]); e.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'], N_global=N)
            e.evaluate(False); assert e.i8_status()['state'] == 1
            e.phase1()
            parts.append(e)
        parts[0].combine(parts[1], 'stats', 'add')
        P2, C2 = parts[0].download('PSI2_SUM'), parts[0].download('PSI1TY')
    finally:
        for e in engines:
            e.close()
        lib.gp_debug_set_option(b'p1_i8', 0)
    # Y's digit scale is per shard (its own column maxima), so C may differ by the truncation; Psi2's digits do not depend on the shard
    assert_close(P2, P1, 1e-14, what='Psi2, two shards vs one')       # off the diagonal exact integers; the diagonal is a float64 sum of squares
    assert_close(C2, C1, 1e-9, what='C, two shards vs one')


def test_int8_phase1_against_the_long_double_truth():
    import bench
    z = np.load(os.path.join(GOLDEN_DIR, 'hp_truth_large_N100000.npz'))
    N, D, M, Q = 100000, 100, 512, 10
    d = bench.synthetic(N, D, M, Q, seed=100)
    out = _eval(d, N, D, M, Q, True)
    ref = _eval(d, N, D, M, Q, False)
    err = lambda o, k: float(np.max(np.abs(np.asarray(o[k]) - z['truth_' + k])) / np.max(np.abs(z['truth_' + k])))
    rep = {k: (err(out, k), err(ref, k)) for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta')}
    print('N=1e5 vs truth (int8 phase 1, float64 phase 1):', {k: '%.2e %.2e' % v for k, v in rep.items()},
          ' p1 kernel ms: int8 %.3f float64 %.3f  psi1 ms: %.3f %.3f' % (out['timings']['p1_kernel_ms'], ref['timings']['p1_kernel_ms'],
                                                                          out['timings']['psi1_ms'], ref['timings']['psi1_ms']))
    assert abs(out['F'] - float(z['truth_F'])) <= 1e-9 * abs(float(z['truth_F']))
    for k, (e8, e64) in rep.items():
        assert e8 <= 1e-5, (k, e8)
    assert rep['grad_Z'][0] <= 1e-6


def test_the_guard_measures_accepts_rejects_and_resets():
    """csrc/p1i8.hip "guard": the first int8 evaluation after an upload runs BOTH phase-1 paths, compares the statistics on the device and uses the
    float64 ones -- so its results are, bit for bit, those of the float64 library; gp_i8_status reports the measured distances and the lower bound
    of cond(K_mm + beta Psi2); an accepted context then runs on the int8 statistics (different bits, same numbers to 2e-6); a rejected one (test
    hook: threshold 0) keeps the float64 kernels -- bit-identical to the float64 library -- until the next upload resets the state."""
    from gparml_amd.engine import ShardEngine
    from oracle import factorised as Fz
    N, D, M, Q = 70000, 100, 512, 10
    d = Fz.synthetic_shard(N, D, M, Q, regime='A', seed=11, zseed=12, alpha_value=0.4)
    lib = _lib()
    keys = ('F', 'grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta')
    same = lambda a, b: all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in keys)
    eng = ShardEngine(N, D, M, Q)
    try:
        eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
        eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
        ref = eng.evaluate(False)                                   # the float64 library
        assert eng.i8_status()['state'] == 0 and eng.i8_status()['checks'] == 0
        assert lib.gp_debug_set_option(b'p1_i8', 1) == 0
        chk = eng.evaluate(False)                                   # the check: both paths, float64 statistics
        st = eng.i8_status()
        assert same(chk, ref) and st['state'] == 1 and st['checks'] == 1, st
        assert st['cond_lower_bound'] * max(st['rel_psi2'], st['rel_c']) <= 1e-4
        i8 = eng.evaluate(False)                                    # accepted: int8 statistics
        assert not same(i8, ref) and eng.i8_status()['checks'] == 1
        for k in keys:
            assert_close(i8[k], ref[k], 2e-6, what=k + ' int8 vs float64')
        # new data: measured again; with the threshold at zero the check rejects
        assert lib.gp_debug_set_option(b'i8_guard_strict', 1) == 0
        eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
        assert eng.i8_status()['state'] == 0
        eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
        chk2 = eng.evaluate(False)
        st2 = eng.i8_status()
        assert same(chk2, ref) and st2['state'] == 2 and st2['checks'] == 2, st2
        for _ in range(2):
            assert same(eng.evaluate(False), ref)                   # rejected: the float64 kernels, nothing else
        assert eng.i8_status()['checks'] == 2
    finally:
        lib.gp_debug_set_option(b'i8_guard_strict', 0); lib.gp_debug_set_option(b'p1_i8', 0)
        eng.close()
    small = ShardEngine(500, 3, 20, 2)
    assert small.i8_status()['state'] == -1                         # the path does not apply to this shape
    small.close()
