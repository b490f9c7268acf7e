"""Global step semantics on the device (run with -m gpu): the reference's jitter fallback (partial_terms.py:452-461), the deferred
failure report (one host synchronisation per evaluation, in finish) and run-to-run bit-identical results (fixed-order reductions)."""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


def _setup(seed=0, N=300, D=3, M=40, Q=3):
    from oracle import factorised as Fz
    d = Fz.synthetic_shard(N, D, M, Q, regime='A', seed=seed, zseed=seed + 1, alpha_value=0.5)
    return d, (N, D, M, Q)


def _indefinite_stats(d, shift):
    """Reduced statistics that make Kmm + beta*Psi2 indefinite by ``shift``: Psi2 = -(lambda_min(Kmm) + shift)/beta * I."""
    from oracle import literal as L
    Kmm = L.rbf_gram(d['Z'], d['sf2'], d['alpha'])
    lam = np.linalg.eigvalsh(Kmm)[0]
    M, D = Kmm.shape[0], d['Y'].shape[1]
    Psi2 = -(lam + shift) / d['beta'] * np.eye(M)
    C = np.random.RandomState(3).randn(M, D)
    return dict(sum_YYT=float(np.sum(d['Y'] ** 2)), Psi2=Psi2, C=C, Psi0=d['sf2'] * d['Y'].shape[0], KL=0.0), Kmm


@pytest.mark.jitter_expected
def test_jitter_retry_matches_the_reference_branch():
    """A barely indefinite Kmm + beta*Psi2 (smallest eigenvalue -5e-8): the reference adds 1e-7*I and carries on
    (partial_terms.py:454-456); the device path reports GP_RETRY_JITTER once and repeats the global step with the jitter."""
    from gparml_amd.engine import ShardEngine
    from gparml_amd import _lib
    from oracle import literal as L
    d, (N, D, M, Q) = _setup()
    st, Kmm = _indefinite_stats(d, 5e-8)
    # the reference's own branch, restated in oracle/literal.py (logmarglik): jittered log-determinant and inverse in the trace term
    pt = L.PartialTermsOracle(d['Z'], d['sf2'], d['alpha'], d['beta'], M, Q, N, D)
    pt.set_local_statistics(st['sum_YYT'], st['Psi2'], st['C'], st['Psi0'], st['KL'])
    F_ref = pt.logmarglik()
    assert np.isfinite(F_ref)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    eng.set_local_statistics(st['sum_YYT'], st['Psi2'], st['C'], st['Psi0'], st['KL'])
    # asynchronous form: the failure surfaces at the first synchronisation point as a retry request carrying the mask
    eng.global_step(sync=False)
    with pytest.raises(_lib.JitterRetry) as ei:
        eng.global_status()
    assert ei.value.mask == 2                      # bit 1: Kmm + beta*Psi2
    eng.global_step(sync=False, jitter=ei.value.mask)
    eng.global_status()
    F1 = eng.scalars()['F']
    # synchronous form (what the partial_terms class and the MapReduce surface use) does the same internally
    eng.global_step()
    F2 = eng.scalars()['F']
    assert F1 == F2
    assert_close(F1, F_ref, 1e-6, what='F with jitter')
    P = eng.download('KMM_PLUS_OP_INV')
    A_j = Kmm + d['beta'] * st['Psi2'] + 1e-7 * np.eye(M)
    assert_close(P, np.linalg.inv(A_j), 1e-5, what='inverse of the jittered matrix')
    eng.close()


def _relmax(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(np.asarray(b))), 1e-300))


@pytest.mark.jitter_expected
def test_jitter_branch_gradients_against_the_restated_reference_branch():
    """What the jitter branch does to the GRADIENTS.  The reference adds 1e-7*I only inside logmarglik (log-determinant and the trace
    term, partial_terms.py:452-461); dF_dKmm / dF_dexp_K_miY / dF_dexp_K_mi_K_im / grad_beta keep Kmm_plus_op_inv = inv(Kmm + beta Psi2)
    of the UN-jittered, here indefinite, matrix (:95, 102-131, 340-360).  The device path factorises by Cholesky, so it uses the inverse
    of the jittered matrix everywhere (documented deviation, include/gparml_hip.h).  Two comparisons on the same inputs:
      (a) against the reference's formulas with Kmm_plus_op_inv := inv(Kmm + beta Psi2 + 1e-7 I): the documented semantics, asserted;
      (b) against the reference branch as it stands (oracle/literal.py, LU inverse of the indefinite matrix): the deviation itself,
          measured and printed -- with the smallest eigenvalue at -5e-8 the two inverses differ by a sign flip of a 2e7-sized
          eigen-direction, so the partials differ at O(1); the branch's gradients carry no information in the reference either."""
    from gparml_amd.engine import ShardEngine
    from oracle import literal as L
    d, (N, D, M, Q) = _setup()
    st, Kmm = _indefinite_stats(d, 5e-8)
    pt = L.PartialTermsOracle(d['Z'], d['sf2'], d['alpha'], d['beta'], M, Q, N, D)
    pt.set_local_statistics(st['sum_YYT'], st['Psi2'], st['C'], st['Psi0'], st['KL'])
    ref = dict(dF_dKmm=pt.dF_dKmm(), Abar=pt.dF_dexp_K_miY(), Bbar=pt.dF_dexp_K_mi_K_im(), grad_beta=pt.grad_beta())
    ptj = L.PartialTermsOracle(d['Z'], d['sf2'], d['alpha'], d['beta'], M, Q, N, D)
    ptj.set_local_statistics(st['sum_YYT'], st['Psi2'], st['C'], st['Psi0'], st['KL'])
    ptj.Kmm_plus_op_inv = np.linalg.inv(Kmm + d['beta'] * st['Psi2'] + 1e-7 * np.eye(M))
    # grad_beta's trace terms with the jittered inverse (partial_terms.py:340-360 restated, P := jittered inverse)
    refj = dict(dF_dKmm=ptj.dF_dKmm(), Abar=ptj.dF_dexp_K_miY(), Bbar=ptj.dF_dexp_K_mi_K_im(), grad_beta=ptj.grad_beta())
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    eng.set_local_statistics(st['sum_YYT'], st['Psi2'], st['C'], st['Psi0'], st['KL'])
    eng.global_step()                                   # takes the retry internally
    out = dict(dF_dKmm=eng.download('DF_DKMM'), Abar=eng.download('DF_DPSI1TY'), Bbar=eng.download('DF_DPSI2'), grad_beta=eng.scalars()['grad_beta'])
    eng.close()
    dev = {k: _relmax(out[k], ref[k]) for k in out}
    print('jitter branch, device vs reference branch as it stands (LU inverse of the indefinite matrix):', {k: '%.2e' % v for k, v in dev.items()})
    for k in out:
        assert _relmax(out[k], refj[k]) <= 1e-5, (k, _relmax(out[k], refj[k]))
        assert np.all(np.isfinite(out[k]))


def test_coincident_and_nearly_coincident_inducing_points():
    """The situations the reference's 1e-7 fallback was written for, with real statistics (no hand-made matrices).
    (1) Two COINCIDENT inducing points: Kmm has two identical rows, the reference's linalg.inv raises "Singular matrix" before any
        jitter applies (partial_terms.py:95); the device path meets an exactly zero Cholesky pivot and raises the same class.
    (2) Two NEARLY coincident points (distance 1e-2, cond(Kmm) 3e7): neither path jitters; F agrees to 1e-6 and the gradients to the
        1e-5 contract times the conditioning of this toy problem (the two float64 CPU paths, LU and Cholesky, differ by 1e-5 .. 4e-5
        from each other here; measured device-vs-LU differences are printed).  Closer than that (1e-3: cond 3e9) the two CPU paths
        already disagree by 5 % on grad_beta -- the reference's own numbers are noise there, with or without a jitter."""
    from gparml_amd.engine import ShardEngine
    from oracle import literal as L
    d, (N, D, M, Q) = _setup(seed=6, N=400, M=30)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    Z = d['Z'].copy()
    Z[1] = Z[0]
    with pytest.raises(np.linalg.LinAlgError):
        L.full_evaluation(Z, d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], with_embeddings=False)
    eng.set_globals(Z, d['sf2'], d['alpha'], d['beta'])
    with pytest.raises(np.linalg.LinAlgError):
        eng.evaluate(False)
    Z[1] = Z[0] + 1e-2 * np.random.RandomState(0).randn(Q)
    ref = L.full_evaluation(Z, d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], with_embeddings=False)
    eng.set_globals(Z, d['sf2'], d['alpha'], d['beta'])
    out = eng.evaluate(False)
    eng.close()
    diffs = {k: _relmax(out[k], ref[k]) for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta')}
    print('nearly coincident inducing points (cond(Kmm) %.1e): F %.2e, gradients vs the LU path:' % (np.linalg.cond(ref['Kmm']), abs(out['F'] - ref['F']) / abs(ref['F'])),
          {k: '%.2e' % v for k, v in diffs.items()})
    assert abs(out['F'] - ref['F']) <= 1e-6 * abs(ref['F'])
    for k, v in diffs.items():
        assert v <= 2e-4, (k, v)


@pytest.mark.jitter_expected
def test_jitter_that_does_not_help_is_a_linalg_error():
    """Smallest eigenvalue -1e-6: still indefinite with 1e-7*I -> the reference's assertion (partial_terms.py:459-461) -> LinAlgError."""
    from gparml_amd.engine import ShardEngine
    d, (N, D, M, Q) = _setup(seed=2)
    st, _ = _indefinite_stats(d, 1e-6)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    eng.set_local_statistics(st['sum_YYT'], st['Psi2'], st['C'], st['Psi0'], st['KL'])
    with pytest.raises(np.linalg.LinAlgError):
        eng.global_step()
    eng.global_step(sync=False)
    eng.phase2(False)
    with pytest.raises(Exception):                 # deferred: the evaluation's only synchronisation reports it
        eng.finish()
    eng.close()


@pytest.mark.jitter_expected
def test_evaluate_recovers_through_the_retry():
    """ShardEngine.evaluate / DistributedEvaluator.evaluate repeat global step + phase 2 when finish() asks for the jitter."""
    from gparml_amd.dist import DistributedEvaluator
    from gparml_amd.engine import ShardEngine
    d, (N, D, M, Q) = _setup(seed=4)
    st, _ = _indefinite_stats(d, 5e-8)

    class Fixed(ShardEngine):
        bad = True
        phase2_calls = 0

        def phase1(self):                          # keep the hand-made statistics instead of the shard's own
            ShardEngine.phase1(self)
            if self.bad:
                self.set_local_statistics(st['sum_YYT'], st['Psi2'], st['C'], st['Psi0'], st['KL'])

        def phase2(self, want=False):
            self.phase2_calls += 1
            ShardEngine.phase2(self, want)

    eng = Fixed(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    a = eng.evaluate(False)
    assert eng.phase2_calls == 2 and eng._jitter_hint != 0          # the failure was seen in finish(): phase 2 ran twice
    b = DistributedEvaluator(eng).evaluate(False)
    # the next evaluation checks the factorisation before phase 2 (one wait instead of a wasted phase 2) and gives the same bits
    assert eng.phase2_calls == 3 and eng._jitter_hint != 0
    assert np.isfinite(a['F']) and a['F'] == b['F']
    assert np.all(np.isfinite(a['grad_Z'])) and np.array_equal(a['grad_Z'], b['grad_Z'])
    eng.bad = False                                                   # the shard's own (well-conditioned) statistics: the hint is dropped again
    c1 = eng.evaluate(False)
    assert eng.phase2_calls == 4 and eng._jitter_hint == 0
    c2 = eng.evaluate(False)
    assert eng.phase2_calls == 5 and c1['F'] == c2['F'] and np.array_equal(c1['grad_Z'], c2['grad_Z'])
    eng.close()


@pytest.mark.parametrize('regime,emb', [('A', False), ('B', True)])
def test_repeated_evaluations_are_bit_identical(regime, emb):
    """Every reduction on the path has a fixed order (no floating-point atomics): the same inputs give the same bits, on the same
    context and on a fresh one."""
    from gparml_amd.engine import ShardEngine
    from oracle import factorised as Fz
    N, D, M, Q = 3000, 6, 150, 5
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=8, zseed=9, alpha_value=0.4)
    outs = []
    for fresh in range(2):
        eng = ShardEngine(N, D, M, Q)
        eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
        eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
        for rep in range(3):
            o = eng.evaluate(emb)
            outs.append((o['F'], o['grad_Z'].copy(), o['grad_alpha'].copy(), o['grad_sf2'], o['grad_beta']))
        eng.close()
    for o in outs[1:]:
        assert o[0] == outs[0][0] and o[3] == outs[0][3] and o[4] == outs[0][4]
        assert np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2])


def test_fixed_embedding_preparation_cache_follows_the_hyper_parameters():
    """With fixed embeddings the per-point preparation runs once per upload and Psi1 takes alpha / sf2 as arguments: a second
    evaluation at other hyper-parameters (and one after a new upload, and one after a switch to embedding gradients and back) must
    equal a fresh context's."""
    from gparml_amd.engine import ShardEngine
    from oracle import factorised as Fz
    N, D, M, Q = 2500, 7, 140, 6
    d = Fz.synthetic_shard(N, D, M, Q, regime='A', seed=21, zseed=22, alpha_value=0.5)
    rs = np.random.RandomState(3)
    hypers = [(d['Z'], d['sf2'], d['alpha'], d['beta']),
              (d['Z'] + 0.01 * rs.randn(M, Q), 1.7, np.asarray(d['alpha']) * rs.uniform(0.5, 2.0, size=np.asarray(d['alpha']).shape), 4.0)]

    def fresh(Y, X_mu, h, emb=False):
        e = ShardEngine(N, D, M, Q)
        e.upload_shard(Y, X_mu, d['X_S'])
        e.set_globals(*h)
        o = e.evaluate(emb)
        e.close()
        return o

    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    X2 = d['X_mu'] + 0.1 * rs.randn(N, Q)
    steps = [(d['X_mu'], hypers[0], False), (d['X_mu'], hypers[1], False), (d['X_mu'], hypers[1], True), (d['X_mu'], hypers[0], False),
             (X2, hypers[0], False), (X2, hypers[1], False)]
    cur = d['X_mu']
    for X_mu, h, emb in steps:
        if X_mu is not cur:
            eng.upload_shard(d['Y'], X_mu, d['X_S'])
            cur = X_mu
        eng.set_globals(*h)
        out = eng.evaluate(emb)
        ref = fresh(d['Y'], X_mu, h, emb)
        # rounding-level agreement only: which of the two Psi1 forms (fixed-embedding or general) phase 1 used depends on the mode of
        # the PREVIOUS evaluation; a stale cache would be wrong in the leading digits
        assert abs(out['F'] - ref['F']) <= 1e-10 * abs(ref['F'])
        for k in ('grad_Z', 'grad_alpha'):
            assert np.max(np.abs(out[k] - ref[k])) <= 1e-8 * np.max(np.abs(ref[k])), k
    eng.close()


@pytest.mark.parametrize('N,D,M,Q,regime,emb', [(100000, 10, 128, 10, 'A', False), (100000, 10, 128, 10, 'A', True), (20000, 10, 128, 10, 'B', True),
                                              (3000, 5, 70, 4, 'A', False), (900, 128, 33, 11, 'B', True), (2000, 1, 1, 1, 'A', False)])
def test_short_one_panel_tail_is_bit_identical_to_the_fifteen_launches(N, D, M, Q, regime, emb):
    """M, D <= 128 (BASELINE configs[1] is M = 128, D = 10): everything of the global step behind the panel factorisation runs as SEVEN launches
    of one kernel (tail_stage_kernel, csrc/linalg.hip: products that do not depend on each other share a launch, the assembly rides on the
    tiles it needs) instead of fifteen kernels.  The stages call the same
    device functions as the separate kernels, so the bound, every gradient, the partials (partial_terms.py:102-138) and the inverses
    (partial_terms.py:60, 95) must agree BIT FOR BIT with the library run with the fused kernel switched off (gp_debug_set_option('gs_tail', 0)),
    also without the two extended-precision pieces, and on a repeated evaluation."""
    from gparml_amd import _lib
    from gparml_amd.engine import ShardEngine
    from oracle import factorised as Fz
    lib = _lib.load()
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=60, zseed=61, alpha_value=min(0.5, 1.0 / Q) if M > 1 else 0.5)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])

    def run(tail, dd=1, refine=1):
        assert lib.gp_debug_set_option(b'gs_tail', tail) == 0 and lib.gp_debug_set_option(b'dd_kipsi2', dd) == 0 and lib.gp_debug_set_option(b'refine_E', refine) == 0
        eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
        o = eng.evaluate(emb)
        res = {k: np.array(o[k]) for k in ('F', 'grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta')}
        for name in ('KMM_INV', 'KMM_PLUS_OP_INV', 'DF_DKMM', 'DF_DPSI1TY', 'DF_DPSI2'):
            res[name] = eng.download(name)
        if emb:
            res['gmu'], res['gS'] = eng.download('GRAD_X_MU'), eng.download('GRAD_X_S')
        return res

    try:
        for dd, refine in ((1, 1), (0, 0)):
            a, b, again = run(1, dd, refine), run(0, dd, refine), run(1, dd, refine)
            for k in a:
                assert np.array_equal(a[k], b[k]), ('fused vs separate', k, dd, refine, float(np.max(np.abs(a[k] - b[k]))))
                assert np.array_equal(a[k], again[k]), ('fused, repeated', k)
    finally:
        lib.gp_debug_set_option(b'gs_tail', 1); lib.gp_debug_set_option(b'dd_kipsi2', 1); lib.gp_debug_set_option(b'refine_E', 1)
        eng.close()
