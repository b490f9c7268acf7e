"""GPU parity tests proper (run with -m gpu on the MI355X box): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs and against the golden vectors captured from the
imported reference.

Tolerances (BASELINE.json north_star): bound within 1e-6 relative, gradients within 1e-5 relative to the
largest magnitude of the gradient block.  The cancellation between the data part and the Kmm part of grad_Z
grows with cond(Kmm + beta*Psi2); cases are chosen with cond <~ 1e8 (the same limit applies to the reference:
Cholesky-vs-LU on the CPU differ by the same amount)."""
import numpy as np
import pytest

import oracle_cache
from conftest import assert_close, golden_names, load_golden

pytestmark = pytest.mark.gpu

F_RTOL = 1e-6
G_RTOL = 1e-5


def _engine(inp_or_shapes):
    from gparml_amd.engine import ShardEngine
    return ShardEngine(*inp_or_shapes)


def _run(Z, sf2, alpha, beta, Y, X_mu, X_S, N_global=None, emb=True):
    N, D = Y.shape
    M, Q = Z.shape
    eng = _engine((N, D, M, Q))
    eng.upload_shard(Y, X_mu, X_S)
    eng.set_globals(Z, sf2, alpha, beta, N_global=N_global)
    eng.phase1()
    eng.global_step()
    eng.phase2(emb)
    out = eng.finish()
    out['stats'] = dict(Psi2=eng.download('PSI2_SUM'), C=eng.download('PSI1TY'), scal=eng.scalars(), Psi1=eng.download('PSI1'))
    out['partials'] = dict(Abar=eng.download('DF_DPSI1TY'), Bbar=eng.download('DF_DPSI2'), dFdK=eng.download('DF_DKMM'),
                           Kmm=eng.download('KMM'), Kmm_inv=eng.download('KMM_INV'), P=eng.download('KMM_PLUS_OP_INV'))
    if emb:
        out['grad_X_mu'] = eng.download('GRAD_X_MU')
        if not np.all(X_S == 0):
            out['grad_X_S'] = eng.download('GRAD_X_S')
    eng.close()
    return out


@pytest.mark.parametrize('name', golden_names())
def test_against_reference_golden(name):
    """Every golden case captured from the imported reference (tests/golden/make_golden.py)."""
    inp, ref = load_golden(name)
    out = _run(inp['Z'], inp['sf2'], inp['alpha'], inp['beta'], inp['Y'], inp['X_mu'], inp['X_S'], N_global=inp['N'])
    assert_close(out['stats']['Psi1'], ref['exp_K_mi'], 1e-12, what='exp_K_mi')
    assert_close(out['stats']['Psi2'], ref['sum_exp_K_mi_K_im'], 1e-11, what='sum_exp_K_mi_K_im')
    assert_close(out['stats']['C'], ref['exp_K_miY'], 1e-11, what='exp_K_miY')
    assert_close(out['stats']['scal']['sum_YYT'], ref['sum_YYT'], 1e-12, what='sum_YYT')
    assert_close(out['stats']['scal']['KL'], ref['KL'], 1e-11, atol=1e-300, what='KL')
    assert_close(out['partials']['Kmm'], ref['Kmm'], 1e-12, what='Kmm')
    assert_close(out['partials']['Kmm_inv'], ref['Kmm_inv'], 1e-8, what='Kmm_inv')
    assert_close(out['partials']['P'], ref['Kmm_plus_op_inv'], 1e-8, what='Kmm_plus_op_inv')
    assert_close(out['F'], ref['F'], F_RTOL, what='F')
    assert_close(out['partials']['Abar'], ref['dF_dexp_K_miY'], G_RTOL, what='dF_dexp_K_miY')
    assert_close(out['partials']['Bbar'], ref['dF_dexp_K_mi_K_im'], G_RTOL, what='dF_dexp_K_mi_K_im')
    assert_close(out['partials']['dFdK'], ref['dF_dKmm'], G_RTOL, what='dF_dKmm')
    assert_close(out['grad_Z'], ref['grad_Z'], G_RTOL, what='grad_Z')
    assert_close(out['grad_alpha'], ref['grad_alpha'], G_RTOL, what='grad_alpha')
    assert_close(out['grad_sf2'], ref['grad_sf2'], G_RTOL, what='grad_sf2')
    assert_close(out['grad_beta'], ref['grad_beta'], G_RTOL, what='grad_beta')
    assert_close(out['grad_X_mu'], ref['grad_X_mu'], G_RTOL, what='grad_X_mu')
    if 'grad_X_S' in ref:
        assert_close(out['grad_X_S'], ref['grad_X_S'], G_RTOL, what='grad_X_S')


SHAPES = [
    # N, D, M, Q, regime, alpha   (ragged sizes: nothing is a multiple of the 128/16 tiles)
    (300, 5, 20, 3, 'A', 0.5), (1000, 7, 130, 10, 'A', 0.3), (777, 3, 5, 2, 'A', 0.7), (2000, 10, 128, 13, 'A', 0.2),
    (900, 4, 64, 20, 'A', 0.1), (129, 1, 1, 1, 'A', 1.0), (4096, 100, 512, 10, 'A', 0.3),
    (300, 5, 20, 3, 'B', 0.5), (500, 4, 2, 2, 'B', 0.7), (1000, 7, 130, 10, 'B', 0.3), (257, 2, 1, 1, 'B', 1.0),
    (640, 3, 33, 13, 'B', 0.2),
    # every compiled latent width (4/10/16/24/32/52/64; 6/8 below) and more than four 64-column slabs of inducing points
    (500, 3, 300, 5, 'B', 1.0), (400, 2, 70, 20, 'B', 0.1), (300, 2, 40, 30, 'B', 0.08), (200, 2, 24, 50, 'B', 0.05),
    (150, 2, 12, 60, 'B', 0.05),
    # the tile-pair phase 2 (psi2_sym_kernel): eight slabs (four waves, nine rounds), an odd slab count with a bye, both latent widths
    (600, 3, 512, 10, 'B', 0.3), (260, 2, 200, 4, 'B', 4.0), (800, 2, 700, 7, 'B', 0.3),
    # the latent widths 6 and 8 (r04) on all three phase-2 forms: tile pairs (three .. sixteen slabs), the column kernel (two slabs), a single slab
    (500, 3, 300, 6, 'B', 0.8), (450, 2, 256, 8, 'B', 0.5), (300, 4, 100, 5, 'B', 1.0), (300, 4, 100, 8, 'B', 0.6), (200, 2, 30, 6, 'B', 1.0),
    (1100, 2, 1024, 6, 'B', 0.8),
    # widths 12 and 14 (the column kernel), several slab groups and a single slab
    (500, 3, 300, 11, 'B', 0.4), (400, 2, 200, 12, 'B', 0.4), (350, 4, 130, 13, 'B', 0.3), (300, 2, 50, 14, 'B', 0.3),
    # r06: psi2_sym_kernel<12> (Q = 11, 12: compact rt rows, the constant ones quad, the two-pass finish) on five slabs with a bye and on eight, the latent
    # widths 14 and 16 on five to ten column slabs (the column kernel's several slab groups), Q = 16 on two slabs
    (420, 3, 300, 12, 'B', 0.3), (520, 2, 512, 11, 'B', 0.25), (530, 2, 512, 13, 'B', 0.25), (650, 2, 640, 14, 'B', 0.2), (460, 3, 450, 16, 'B', 0.15), (520, 2, 513, 15, 'B', 0.15),
    (350, 2, 128, 16, 'B', 0.15),
]


def _check(out, ref, regime, emb):
    assert_close(out['stats']['Psi2'], ref['stats']['sum_exp_K_mi_K_im'], 1e-11, what='Psi2')
    assert_close(out['stats']['C'], ref['stats']['exp_K_miY'], 1e-11, what='C')
    assert_close(out['F'], ref['F'], F_RTOL, what='F')
    for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta') + (('grad_X_mu',) if emb else ()):
        assert_close(out[k], ref[k], G_RTOL, what=k)
    if regime == 'B' and emb:
        assert_close(out['grad_X_S'], ref['grad_X_S'], G_RTOL, what='grad_X_S')


@pytest.mark.parametrize('N,D,M,Q,regime,alpha', SHAPES)
def test_against_oracle_on_seeded_inputs(N, D, M, Q, regime, alpha):
    from oracle import factorised as Fz
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=11, zseed=12, alpha_value=alpha)
    live = lambda: Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'])
    # the two cases whose oracle evaluation takes 8 and 29 s on the host come from tests/golden/oracle_cache (tests/oracle_cache.py)
    ref = oracle_cache.get('seeded_%d_%d_%d_%d_%s' % (N, D, M, Q, regime), d, live) if (M >= 700 and regime == 'B') else live()
    oracle_cache.done()
    out = _run(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'])
    _check(out, ref, regime, True)


# Fixed embeddings (want_embedding_grads = 0): the kernel sequence bench.py times -- p1_kernel8, global step, the fast phase-2
# kernels (p2_fast8_kernel for Q <= 11, p2_gen8_kernel<false> on the features [mu | 1 | mu^2] beyond).
# Phase 1 (p1v2.hip): MT = ceil(M/128) diagonal+C jobs and MT(MT-1)/2 off-diagonal jobs with their own slice counts; NBY = 8 (D <= 32) or 26
# (D <= 104) C column groups ride on the diagonal jobs; D > 104 or more than 11 row tiles fall back to the tile-per-workgroup kernel.
FIXED_SHAPES = [s for s in SHAPES if s[4] == 'A'] + [(1500, 6, 200, 11, 'A', 0.2), (700, 5, 140, 12, 'A', 0.2), (600, 4, 70, 23, 'A', 0.1),
                                                     (500, 3, 40, 24, 'A', 0.1), (3000, 40, 300, 6, 'A', 1.0), (2500, 33, 1100, 9, 'A', 2.0),
                                                     (900, 104, 129, 5, 'A', 0.6), (900, 105, 129, 5, 'A', 0.6), (2000, 8, 1500, 10, 'A', 2.0),
                                                     (1300, 32, 256, 8, 'A', 0.8),
                                                     # 13 and 16 panels of 128 (r05): the inverse factor by halves with pairs cut off at every level, the split-k plans of the
                                                     # 128-tile kernel, the global step's int8 products (csrc/gsi8.hip: 1024 <= M <= 2048) at their upper end
                                                     (1700, 5, 1537, 7, 'A', 2.0), (2100, 6, 2048, 8, 'A', 2.0),
                                                     # wide latent spaces: p2_gen8_kernel<false> on the features [mu | 1 | mu^2] (no m-contraction, no point kernel)
                                                     (700, 7, 150, 30, 'A', 0.05), (900, 104, 129, 50, 'A', 0.03), (400, 3, 300, 63, 'A', 0.03),
                                                     (1000, 12, 64, 25, 'A', 0.06)]


@pytest.mark.parametrize('N,D,M,Q,regime,alpha', FIXED_SHAPES)
def test_fixed_embeddings_against_oracle(N, D, M, Q, regime, alpha):
    from oracle import factorised as Fz
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=21, zseed=22, alpha_value=alpha)
    ref = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], want_embeddings=False)
    out = _run(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], emb=False)
    _check(out, ref, regime, False)


@pytest.mark.parametrize('regime,emb', [('A', False), ('A', True), ('B', True)])
def test_config1_full_evaluation(regime, emb):
    """BASELINE configs[1] at its full size (N=1e5, D=10, M=128, Q=10, alpha = 1/Q): bound and every gradient against the oracle."""
    from oracle import factorised as Fz
    N, D, M, Q = 100000, 10, 128, 10
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=0, zseed=1, alpha_value=0.1)
    live = lambda: Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], want_embeddings=emb)
    ref = oracle_cache.get('config1_full_B', d, live) if regime == 'B' else live()      # regime B: 34 s of host time for the pair loops
    oracle_cache.done()
    out = _run(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], emb=emb)
    _check(out, ref, regime, emb)


def test_config4_shape():
    """BASELINE configs[4]'s per-GPU shape (D=1000, M=1024, Q=50, free embeddings) at an N the oracle finishes in a minute:
    16 inducing slabs / 32 strips of the wide-latent MFMA pair kernels, eight 128-column Y tiles, two and a half row tiles."""
    from oracle import factorised as Fz
    N, D, M, Q = 320, 1000, 1024, 50
    rs = np.random.RandomState(5)
    d = Fz.synthetic_shard(N, D, 64, Q, regime='B', seed=4, zseed=5, alpha_value=0.02)
    d['Z'] = d['X_mu'][rs.randint(0, N, size=M)] + 0.3 * rs.randn(M, Q)      # M > N: inducing points around re-used rows
    ref = oracle_cache.get('config4_shape_N320', d, lambda: Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S']))
    oracle_cache.done()
    out = _run(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'])
    _check(out, ref, 'B', True)


def test_errors_map_to_reference_exceptions():
    """Non-PD Cholesky -> LinAlgError, negative variance -> AssertionError (scg_adapted.py:55 relies on these)."""
    from gparml_amd.engine import ShardEngine
    rs = np.random.RandomState(0)
    N, D, M, Q = 50, 2, 4, 2
    Y, X = rs.randn(N, D), rs.randn(N, Q)
    eng = ShardEngine(N, D, M, Q)
    with pytest.raises(AssertionError):
        eng.upload_shard(Y, X, -np.ones((N, Q)))
    eng.upload_shard(Y, X, np.zeros((N, Q)))
    Z = np.tile(rs.randn(1, Q), (M, 1))          # identical inducing points: Kmm singular
    eng.set_globals(Z, 1.0, np.ones(Q), 1.0)
    eng.phase1()
    with pytest.raises(np.linalg.LinAlgError):
        eng.global_step()
    with pytest.raises(AssertionError):
        eng.set_globals(rs.randn(M, Q), 1.0, -np.ones(Q), 1.0)
    eng.close()


def test_full_size_properties():
    """Size-independent properties at a BASELINE-like size (config 2: N=1e5, D=10, M=128, Q=10):
    shard additivity of the statistics (the map/reduce identity), symmetry of Psi2, and scaling of C in Y."""
    from gparml_amd.engine import ShardEngine
    from oracle import factorised as Fz
    N, D, M, Q = 100000, 10, 128, 10
    d = Fz.synthetic_shard(N, D, M, Q, regime='A', seed=0, zseed=1, alpha_value=0.3)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    eng.phase1()
    P2, C = eng.download('PSI2_SUM'), eng.download('PSI1TY')
    assert np.max(np.abs(P2 - P2.T)) == 0.0
    eng.close()
    h = N // 3
    acc2, accC = 0.0, 0.0
    for sl in (slice(0, h), slice(h, N)):
        n = sl.stop - sl.start
        e = ShardEngine(n, D, M, Q)
        e.upload_shard(d['Y'][sl] * 2.0, d['X_mu'][sl], d['X_S'][sl])
        e.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'], N_global=N)
        e.phase1()
        acc2 = acc2 + e.download('PSI2_SUM')
        accC = accC + e.download('PSI1TY')
        e.close()
    assert_close(acc2, P2, 1e-12, what='Psi2 additivity')
    assert_close(accC, 2.0 * C, 1e-12, what='C linearity in Y')


