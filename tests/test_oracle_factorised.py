"""oracle/factorised.py (two-phase formulation = kernel spec + CPU baseline) against the golden vectors
from the imported reference and against oracle/literal.py on fresh seeded inputs.  Tolerances: 1e-9
relative to the array's max magnitude (Cholesky vs LU inverse, different summation order)."""
import numpy as np
import pytest

from conftest import assert_close
from oracle import factorised as Fz
from oracle import literal as L

RTOL = 1e-9


def test_against_golden(golden):
    name, inp, out = golden
    ev = Fz.evaluate(inp['Z'], inp['sf2'], inp['alpha'], inp['beta'], inp['Y'], inp['X_mu'], inp['X_S'],
                     N_global=inp['N'], chunk=7)
    st = ev['stats']
    assert_close(st['sum_exp_K_mi_K_im'], out['sum_exp_K_mi_K_im'], 1e-12, what='Psi2')
    assert_close(st['exp_K_miY'], out['exp_K_miY'], 1e-12, what='C')
    assert_close(st['sum_YYT'], out['sum_YYT'], 1e-12, what='sum_YYT')
    assert_close(st['KL'], out['KL'], 1e-12, what='KL')
    assert_close(ev['F'], out['F'], 1e-11, what='F')
    g = ev['gstep']
    assert_close(g['Abar'], out['dF_dexp_K_miY'], RTOL, what='Abar')
    assert_close(g['Bbar'], out['dF_dexp_K_mi_K_im'], RTOL, what='Bbar')
    assert_close(g['dF_dKmm'], out['dF_dKmm'], RTOL, what='dF_dKmm')
    for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu'):
        assert_close(ev[k], out[k], RTOL, what=k)
    if 'grad_X_S' in out:
        assert_close(ev['grad_X_S'], out['grad_X_S'], RTOL, what='grad_X_S')


@pytest.mark.parametrize('regime,N,D,M,Q', [('A', 70, 6, 12, 5), ('B', 33, 4, 9, 3), ('B', 20, 2, 5, 1)])
def test_against_literal_synthetic(regime, N, D, M, Q):
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=3, zseed=4, alpha_value=0.4)
    ref = L.full_evaluation(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'])
    ev = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], chunk=16)
    assert_close(ev['F'], ref['F'], 1e-10, what='F')
    for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu'):
        assert_close(ev[k], ref[k], 1e-8, what=k)
    if regime == 'B':
        assert_close(ev['grad_X_S'], ref['grad_X_S'], 1e-8, what='grad_X_S')


def test_two_shards_sum_to_one():
    """The map/reduce identity the multi-GPU path relies on (local_MapReduce.py:250-277)."""
    d = Fz.synthetic_shard(48, 5, 8, 3, regime='B', seed=5, zseed=6, alpha_value=0.5)
    whole = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'])
    parts = [slice(0, 20), slice(20, 48)]
    st = None
    for sl in parts:
        s = Fz.phase1(d['Z'], d['sf2'], d['alpha'], d['Y'][sl], d['X_mu'][sl], d['X_S'][sl])
        st = s if st is None else {k: st[k] + s[k] for k in st}
    gs = Fz.global_step(d['Z'], d['sf2'], d['alpha'], d['beta'], st, 48, 5)
    acc = None
    gmu = []
    for sl in parts:
        p2 = Fz.phase2(d['Z'], d['sf2'], d['alpha'], d['Y'][sl], d['X_mu'][sl], d['X_S'][sl], gs['Abar'], gs['Bbar'])
        gmu.append(p2['grad_X_mu'])
        part = dict(grad_Z_data=p2['grad_Z_data'], grad_alpha_data=p2['grad_alpha_data'])
        acc = part if acc is None else {k: acc[k] + part[k] for k in acc}
    out = Fz.finish(d['Z'], d['sf2'], d['alpha'], gs, acc, False)
    assert_close(out['F'], whole['F'], 1e-12, what='F')
    assert_close(out['grad_Z'], whole['grad_Z'], 1e-10, what='grad_Z')
    assert_close(out['grad_alpha'], whole['grad_alpha'], 1e-10, what='grad_alpha')
    assert_close(np.concatenate(gmu), whole['grad_X_mu'], 1e-10, what='grad_X_mu')


def test_finite_difference_sanity():
    """Independent sanity layer in the spirit of test.py:62-93, 270-296 (forward FD, percent-level)."""
    d = Fz.synthetic_shard(12, 3, 5, 2, regime='B', seed=7, zseed=8, alpha_value=0.7)
    base = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'])
    h = 1e-6

    def F(**kw):
        a = dict(d)
        a.update(kw)
        return Fz.evaluate(a['Z'], a['sf2'], a['alpha'], a['beta'], a['Y'], a['X_mu'], a['X_S'])['F']

    Zp = d['Z'].copy(); Zp[1, 0] += h
    assert abs((F(Z=Zp) - base['F']) / h - base['grad_Z'][1, 0]) < 1e-3 * max(1.0, abs(base['grad_Z'][1, 0]))
    ap = d['alpha'].copy(); ap[1] += h
    assert abs((F(alpha=ap) - base['F']) / h - base['grad_alpha'][1]) < 1e-3 * max(1.0, abs(base['grad_alpha'][1]))
    assert abs((F(sf2=d['sf2'] + h) - base['F']) / h - base['grad_sf2']) < 1e-3 * max(1.0, abs(base['grad_sf2']))
    assert abs((F(beta=d['beta'] + h) - base['F']) / h - base['grad_beta']) < 1e-3 * max(1.0, abs(base['grad_beta']))
    mp = d['X_mu'].copy(); mp[3, 1] += h
    assert abs((F(X_mu=mp) - base['F']) / h - base['grad_X_mu'][3, 1]) < 1e-3 * max(1.0, abs(base['grad_X_mu'][3, 1]))
    Sp = d['X_S'].copy(); Sp[2, 0] += h
    assert abs((F(X_S=Sp) - base['F']) / h - base['grad_X_S'][2, 0]) < 1e-3 * max(1.0, abs(base['grad_X_S'][2, 0]))


def test_blas_bound_baseline_equals_the_two_phase_port():
    """bench.py's CPU baseline (evaluate_blas: K kept between the phases, buffers reused) is the same evaluation as evaluate()."""
    from oracle import factorised as Fz
    d = Fz.synthetic_shard(5000, 7, 60, 6, regime='A', seed=3, zseed=4, alpha_value=0.4)
    ref = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], want_embeddings=False)
    work = {}
    for chunk in (1024, 5000):
        out = Fz.evaluate_blas(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], chunk=chunk, work=work)
        assert abs(out['F'] - ref['F']) <= 1e-11 * abs(ref['F'])
        for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta'):
            assert np.max(np.abs(np.asarray(out[k]) - np.asarray(ref[k]))) <= 1e-8 * np.max(np.abs(ref[k])), k


def test_threaded_and_lu_variants_of_the_baseline():
    """evaluate_blas with its chunks in a thread pool adds the partial sums in chunk order: bit-identical to the serial call; with
    linalg='lu' (the reference's LU arrangement of the global step, oracle/literal.py) it is the same evaluation to rounding."""
    from oracle import factorised as Fz
    d = Fz.synthetic_shard(6000, 5, 48, 4, regime='A', seed=5, zseed=6, alpha_value=0.5)
    a = Fz.evaluate_blas(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], chunk=1000)
    b = Fz.evaluate_blas(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], chunk=1000, workers=3)
    c = Fz.evaluate_blas(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], chunk=1000, linalg='lu')
    assert a['F'] == b['F'] and np.array_equal(a['grad_Z'], b['grad_Z']) and np.array_equal(a['grad_alpha'], b['grad_alpha'])
    assert abs(c['F'] - a['F']) <= 1e-11 * abs(a['F'])
    for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta'):
        assert np.max(np.abs(np.asarray(c[k]) - np.asarray(a[k]))) <= 1e-8 * np.max(np.abs(a[k])), k


def test_gemm_pair_tensor_and_sharded_evaluation_equal_the_direct_form():
    """The two accelerations the large-shape GPU tests use -- the per-point psi2 tensor through a batched GEMM (pairs='gemm') and the
    shards-in-threads evaluation -- are the same evaluation as evaluate() (both regimes)."""
    from oracle import factorised as Fz
    for regime in ('A', 'B'):
        d = Fz.synthetic_shard(700, 5, 40, 7, regime=regime, seed=1, zseed=2, alpha_value=0.3)
        ref = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'])
        for out in (Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], pairs='gemm'),
                    Fz.evaluate_sharded(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], shards=5, pairs='gemm')):
            assert abs(out['F'] - ref['F']) <= 1e-12 * abs(ref['F'])
            for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu') + (('grad_X_S',) if regime == 'B' else ()):
                assert np.max(np.abs(np.asarray(out[k]) - np.asarray(ref[k]))) <= 1e-9 * np.max(np.abs(ref[k])), (regime, k)
