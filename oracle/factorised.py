"""Two-phase, factorised CPU formulation of one bound+gradient evaluation (numpy, float64).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  This is (i) the functional specification
of what the HIP kernels compute, phase by phase, and (ii) the fair multi-core CPU baseline
(``bench.py`` ``cpu_baseline`` leg, kind "port"): chunked, vectorised, BLAS-backed, never
materialising the reference's (N_s, M, M) tensor (partial_terms.py:45).

It is pinned by ``tests/test_oracle_factorised.py`` against ``oracle/literal.py`` and against the
golden vectors captured from the imported reference.

Formulation (reference lines in brackets):
  u_nq = a_q/(a_q S_nq+1)           Psi1_nm = s2 prod_q(a_q S_nq+1)^-1/2 exp(-1/2 sum_q u_nq (mu_nq-z_mq)^2)   [kernel_exp.py:80]
  w_nq = a_q/(2 a_q S_nq+1)         psi2_n[m,m'] = s2^2 prod_q(2 a_q S_nq+1)^-1/2
                                                  exp(lnE_nm + lnE_nm' - 1/4 sum_q (a_q-w_nq)(z_mq-z_m'q)^2),
                                    lnE_nm = -1/2 sum_q w_nq (mu_nq-z_mq)^2                                   [kernel_exp.py:143-146]
  S == 0 everywhere ("regime A"):   Psi1 = K_nm, Psi2 = K_nm^T K_nm, KL = 0                                   [partial_terms.py:83-87]
  phase 1  -> {Psi2, C = Psi1^T Y, sum_YYT, Psi0, KL}      (summed over shards)                               [partial_terms.py:74-87]
  global   -> F, Abar = dF/dC, Bbar = dF/dPsi2, dF/dKmm, grad_beta, Kmm-parts of grad_Z/alpha/sf2            [partial_terms.py:102-138, 340-360, 436-473]
  phase 2  -> Psi1/Psi2-parts of grad_Z, grad_alpha (summed over shards); grad_X_mu, grad_X_S (local)        [partial_terms.py:162-205, 256-284, 367-431]
"""
import numpy as np
import scipy.linalg as sla


def _as_params(Z, sf2, alpha, beta):
    Z = np.ascontiguousarray(Z, dtype=float)
    alpha = np.atleast_1d(np.asarray(alpha, dtype=float).squeeze()).astype(float)
    return Z, float(sf2), alpha, float(beta)


def is_regime_A(X_S):
    """Fixed embeddings: every variance exactly zero (partial_terms.py:83)."""
    return bool(np.all(X_S == 0))


def _psi1_chunk(Z, s2, a, mu, S):
    d1 = a[None, :] * S + 1.0
    u = a[None, :] / d1
    c1 = s2 / np.sqrt(np.prod(d1, axis=1))
    # -1/2 sum_q u (mu - z)^2 = -1/2 [ sum u mu^2 - 2 (u mu) . z + u . z^2 ]
    e = -0.5 * (np.sum(u * mu * mu, axis=1)[:, None] - 2.0 * (u * mu).dot(Z.T) + u.dot((Z * Z).T))
    return c1[:, None] * np.exp(e), u


def _psi2_terms_chunk(Z, s2, a, mu, S):
    """Returns psi2 for a chunk as (n, M, M) plus w (n, Q).  Only called with small chunks."""
    d2 = 2.0 * a[None, :] * S + 1.0
    w = a[None, :] / d2
    c2 = s2 * s2 / np.sqrt(np.prod(d2, axis=1))
    lnE = -0.5 * (np.sum(w * mu * mu, axis=1)[:, None] - 2.0 * (w * mu).dot(Z.T) + w.dot((Z * Z).T))   # (n, M)
    dz2 = (Z[:, None, :] - Z[None, :, :]) ** 2                                                         # (M, M, Q)
    v = a[None, :] - w                                                                                 # (n, Q) >= 0
    coup = -0.25 * np.tensordot(v, dz2, axes=([1], [2]))                                               # (n, M, M)
    psi2 = c2[:, None, None] * np.exp(lnE[:, :, None] + lnE[:, None, :] + coup)
    return psi2, w, d2


def _psi2_terms_chunk_gemm(Z, s2, a, mu, S):
    """The same (n, M, M) tensor as _psi2_terms_chunk with the coupling term as a batched GEMM instead of a contraction with the
    (M, M, Q) table of squared differences: -1/4 sum_q v_q (z_mq - z_m'q)^2 = -1/4 (v.z_m^2) - 1/4 (v.z_m'^2) + 1/2 sum_q v_q z_mq z_m'q.
    Ten times faster at M = 1024, Q = 50 (the table is 400 MB there); used by the large-shape GPU tests (``pairs='gemm'``), pinned
    against the direct form by tests/test_oracle_factorised.py."""
    d2 = 2.0 * a[None, :] * S + 1.0
    w = a[None, :] / d2
    c2 = s2 * s2 / np.sqrt(np.prod(d2, axis=1))
    lnE = -0.5 * (np.sum(w * mu * mu, axis=1)[:, None] - 2.0 * (w * mu).dot(Z.T) + w.dot((Z * Z).T))   # (n, M)
    v = a[None, :] - w                                                                                 # (n, Q) >= 0
    A = lnE - 0.25 * v.dot((Z * Z).T)                                                                  # (n, M)
    G = np.matmul(Z[None, :, :] * v[:, None, :], Z.T)                                                  # (n, M, M) = Z diag(v_n) Z^T
    G *= 0.5
    G += A[:, :, None]
    G += A[:, None, :]
    np.exp(G, out=G)
    G *= c2[:, None, None]
    return G, w, d2


# ----------------------------------------------------------------------------------------- phase 1
def phase1(Z, sf2, alpha, Y, X_mu, X_S, chunk=4096, pairs='direct'):
    """Per-shard sufficient statistics.  [partial_terms.py:38-52, 74-87; kernel_exp.py:13-148]"""
    Z, s2, a, _ = _as_params(Z, sf2, alpha, 1.0)
    N_s, D = Y.shape
    M, Q = Z.shape
    regA = is_regime_A(X_S)
    Psi2 = np.zeros((M, M))
    C = np.zeros((M, D))
    sum_YYT = 0.0
    KL = 0.0
    if not regA:
        chunk = max(1, min(chunk, int(2.5e7 // max(1, M * M))))
    for lo in range(0, N_s, chunk):
        hi = min(N_s, lo + chunk)
        mu, S, Yc = X_mu[lo:hi], X_S[lo:hi], Y[lo:hi]
        P1, _ = _psi1_chunk(Z, s2, a, mu, S)
        C += P1.T.dot(Yc)
        sum_YYT += float(np.sum(Yc * Yc))
        if regA:
            Psi2 += P1.T.dot(P1)
        else:
            psi2, _, _ = (_psi2_terms_chunk_gemm if pairs == 'gemm' else _psi2_terms_chunk)(Z, s2, a, mu, S)
            Psi2 += psi2.sum(0)
            KL += 0.5 * float(np.sum(np.sum(S - np.log(S), 1) + np.sum(mu * mu, 1) - Q))
    return dict(sum_exp_K_mi_K_im=Psi2, exp_K_miY=C, sum_YYT=sum_YYT, sum_exp_K_ii=s2 * N_s, KL=KL if not regA else 0.0)


# ----------------------------------------------------------------------------------------- global step
def _global_step_lu(Z, s2, a, b, stats, N_global, D):
    """The global step in the REFERENCE's own arrangement: LU ``inv`` of Kmm and of Kmm + beta Psi2 (partial_terms.py:60, 82, 95),
    ``slogdet`` for the bound (:449-450), the partials as the products the reference forms (:102-131, 340-360) -- taken from the literal
    restatement of the class (oracle/literal.py) fed with the summed statistics.  Used as the float64 "reference" column where the
    imported reference itself cannot run (it stores an (N, M, M) tensor)."""
    from . import literal
    M, Q = Z.shape
    pt = literal.PartialTermsOracle(Z, s2, a, b, M, Q, N_global, D)
    pt.set_local_statistics(stats['sum_YYT'], stats['sum_exp_K_mi_K_im'], stats['exp_K_miY'], stats['sum_exp_K_ii'], stats['KL'])
    return dict(F=pt.logmarglik(), Abar=pt.dF_dexp_K_miY(), Bbar=pt.dF_dexp_K_mi_K_im(), dF_dKmm=pt.dF_dKmm(), grad_beta=pt.grad_beta(),
                Kmm=pt.Kmm, Kmm_inv=pt.Kmm_inv, Kmm_plus_op_inv=pt.Kmm_plus_op_inv)


def global_step(Z, sf2, alpha, beta, stats, N_global, D, fixed_beta=False, linalg='cholesky'):
    """Replicated M x M algebra on the all-reduced statistics.
    [partial_terms.py:89-95 (Kmm), 54-61 (A^-1), 436-473 (F), 102-138 (partials), 340-360 (grad_beta),
     146-160 / 247-254 / 306-308 (Kmm derivative parts), 322-333 (sf2 contraction)]
    ``linalg='lu'``: inverses, log-determinants and partials in the reference's LU arrangement (_global_step_lu)."""
    Z, s2, a, b = _as_params(Z, sf2, alpha, beta)
    M, Q = Z.shape
    Psi2, C = stats['sum_exp_K_mi_K_im'], stats['exp_K_miY']
    sum_YYT, Psi0, KL = stats['sum_YYT'], stats['sum_exp_K_ii'], stats['KL']
    dz = Z[:, None, :] - Z[None, :, :]
    if linalg == 'lu':
        lu = _global_step_lu(Z, s2, a, b, stats, N_global, D)
        Kmm, Abar, Bbar, dF_dKmm = lu['Kmm'], lu['Abar'], lu['Bbar'], lu['dF_dKmm']
        dF_dPsi0 = -0.5 * b * D
        S = (dF_dKmm + dF_dKmm.T) * Kmm
        gZ_K = -a[None, :] * (Z * S.sum(1)[:, None] - S.dot(Z))
        V = dF_dKmm * Kmm
        ga_K = -0.5 * np.einsum('ab,abq->q', V, dz * dz)
        gs = (np.sum(V) + np.sum(Abar * C) + 2.0 * np.sum(Bbar * Psi2) + dF_dPsi0 * Psi0) / s2
        return dict(F=lu['F'], Abar=Abar, Bbar=Bbar, dF_dKmm=dF_dKmm, dF_dPsi0=dF_dPsi0, grad_beta=0.0 if fixed_beta else lu['grad_beta'],
                    grad_Z_K=gZ_K, grad_alpha_K=ga_K, grad_sf2=gs, Kmm=Kmm, Kmm_inv=lu['Kmm_inv'], Kmm_plus_op_inv=lu['Kmm_plus_op_inv'],
                    BbarPsi2=Bbar * Psi2)
    Kmm = s2 * np.exp(-0.5 * np.sum(a[None, None, :] * dz * dz, axis=2))
    A = Kmm + b * Psi2
    try:
        Lk = np.linalg.cholesky(Kmm)
        La = np.linalg.cholesky(A)
    except np.linalg.LinAlgError:
        raise
    ld_K = 2.0 * np.sum(np.log(np.diag(Lk)))
    ld_A = 2.0 * np.sum(np.log(np.diag(La)))
    Ki = sla.cho_solve((Lk, True), np.eye(M))
    P = sla.cho_solve((La, True), np.eye(M))
    E = P.dot(C)                                             # (M, D)
    tr_KiPsi2 = np.sum(Ki * Psi2)
    tr_PPsi2 = np.sum(P * Psi2)
    tr_CtE = np.sum(C * E)
    F = (-0.5 * N_global * D * np.log(2.0 * np.pi) + 0.5 * D * N_global * np.log(b) + 0.5 * D * ld_K - 0.5 * D * ld_A
         - 0.5 * b * sum_YYT - 0.5 * b * D * Psi0 + 0.5 * b * D * tr_KiPsi2 + 0.5 * b * b * tr_CtE - KL)
    EEt = E.dot(E.T)
    Abar = b * b * E
    Bbar = 0.5 * b * D * (Ki - P) - 0.5 * b ** 3 * EEt
    dF_dKmm = 0.5 * D * (Ki - P) - 0.5 * b * D * Ki.dot(Psi2).dot(Ki) - 0.5 * b * b * EEt
    dF_dPsi0 = -0.5 * b * D
    grad_beta = (0.5 * N_global * D / b - 0.5 * D * tr_PPsi2 - 0.5 * sum_YYT - 0.5 * D * Psi0 + 0.5 * D * tr_KiPsi2
                 + b * tr_CtE - 0.5 * b * b * np.sum(E * Psi2.dot(E)))
    # Kmm-dependent parts of the hyper-parameter gradients
    S = (dF_dKmm + dF_dKmm.T) * Kmm                           # symmetrised, partial_terms.py:227-231
    gZ_K = -a[None, :] * (Z * S.sum(1)[:, None] - S.dot(Z))
    V = dF_dKmm * Kmm
    ga_K = -0.5 * np.einsum('ab,abq->q', V, dz * dz)
    # sf2: Kmm, C and Psi2 terms need only the summed statistics; the Psi0 term uses d Psi0/d sf2 = local N summed = Psi0/s2
    gs = (np.sum(V) + np.sum(Abar * C) + 2.0 * np.sum(Bbar * Psi2) + dF_dPsi0 * Psi0) / s2
    return dict(F=F, Abar=Abar, Bbar=Bbar, dF_dKmm=dF_dKmm, dF_dPsi0=dF_dPsi0, grad_beta=0.0 if fixed_beta else grad_beta,
                grad_Z_K=gZ_K, grad_alpha_K=ga_K, grad_sf2=gs, Kmm=Kmm, Kmm_inv=Ki, Kmm_plus_op_inv=P,
                BbarPsi2=Bbar * Psi2)


# ----------------------------------------------------------------------------------------- phase 2
def phase2(Z, sf2, alpha, Y, X_mu, X_S, Abar, Bbar, chunk=4096, want_embeddings=True, pairs='direct'):
    """Data-dependent parts of grad_Z / grad_alpha (to be summed over shards) and the local
    grad_X_mu / grad_X_S.  The alpha term -1/4 sum (Bbar o Psi2)(z-z')^2 that needs only the reduced
    Psi2 is added by ``finish`` (it is not a per-shard sum).
    [partial_terms.py:162-205, 256-284 (sums), 207-240, 286-299 (contractions), 367-431 (per-point)]"""
    Z, s2, a, _ = _as_params(Z, sf2, alpha, 1.0)
    N_s, D = Y.shape
    M, Q = Z.shape
    regA = is_regime_A(X_S)
    Z2 = Z * Z
    gZ = np.zeros((M, Q))
    ga = np.zeros(Q)
    gmu = np.zeros((N_s, Q)) if want_embeddings else None
    gS = np.zeros((N_s, Q)) if (want_embeddings and not regA) else None
    if not regA:
        chunk = max(1, min(chunk, int(2.5e7 // max(1, M * M))))
    for lo in range(0, N_s, chunk):
        hi = min(N_s, lo + chunk)
        mu, S, Yc = X_mu[lo:hi], X_S[lo:hi], Y[lo:hi]
        P1, u = _psi1_chunk(Z, s2, a, mu, S)
        d1 = a[None, :] * S + 1.0
        G = Yc.dot(Abar.T)                                    # dF/dPsi1
        if regA:
            G += 2.0 * P1.dot(Bbar)                           # back-prop through Psi2 = K^T K
        H = G * P1
        h = H.sum(1)                                          # (n,)
        HZ = H.dot(Z)
        HZ2 = H.dot(Z2)
        gZ += H.T.dot(u * mu) - Z * H.T.dot(u)
        quad1 = mu * mu * h[:, None] - 2.0 * mu * HZ + HZ2    # sum_m H (mu - z)^2
        ga += -0.5 * np.sum(quad1 / (d1 * d1) + (S / d1) * h[:, None], axis=0)
        if want_embeddings:
            gmu[lo:hi] = -mu - u * (mu * h[:, None] - HZ)
            if not regA:
                gS[lo:hi] = -0.5 * (1.0 - 1.0 / S) + 0.5 * u * u * quad1 - 0.5 * u * h[:, None]
        if not regA:
            psi2, w, d2 = (_psi2_terms_chunk_gemm if pairs == 'gemm' else _psi2_terms_chunk)(Z, s2, a, mu, S)
            T = psi2 * Bbar[None, :, :]
            r = T.sum(2)                                      # (n, M)
            t = T.dot(Z)                                      # (n, M, Q): sum_m' T[m,m'] z_m'q
            sr = r.sum(1)
            zr = r.dot(Z)
            z2r = r.dot(Z2)
            zt = np.einsum('nmq,mq->nq', t, Z)
            tsum = t.sum(0)
            gZ += (-a[None, :] * (Z * r.sum(0)[:, None] - tsum)
                   + 2.0 * r.T.dot(w * mu) - Z * r.T.dot(w) - np.einsum('nmq,nq->mq', t, w))
            quad2 = 4.0 * mu * mu * sr[:, None] - 8.0 * mu * zr + 2.0 * z2r + 2.0 * zt
            ga += np.sum(-0.25 * quad2 / (d2 * d2) - (S / d2) * sr[:, None], axis=0)
            if want_embeddings:
                gmu[lo:hi] += -w * (2.0 * mu * sr[:, None] - 2.0 * zr)
                gS[lo:hi] += 0.5 * w * w * quad2 - w * sr[:, None]
    return dict(grad_Z_data=gZ, grad_alpha_data=ga, grad_X_mu=gmu, grad_X_S=gS)


def finish(Z, sf2, alpha, gstep, p2_sum, regime_A):
    """Combine the global-step parts with the all-reduced phase-2 sums."""
    Z, s2, a, _ = _as_params(Z, sf2, alpha, 1.0)
    gZ = gstep['grad_Z_K'] + p2_sum['grad_Z_data']
    ga = gstep['grad_alpha_K'] + p2_sum['grad_alpha_data']
    if not regime_A:
        dz = Z[:, None, :] - Z[None, :, :]
        ga = ga - 0.25 * np.einsum('ab,abq->q', gstep['BbarPsi2'], dz * dz)
    return dict(F=gstep['F'], grad_Z=gZ, grad_alpha=ga, grad_sf2=gstep['grad_sf2'], grad_beta=gstep['grad_beta'])


def evaluate(Z, sf2, alpha, beta, Y, X_mu, X_S, N_global=None, chunk=4096, want_embeddings=True, fixed_beta=False, pairs='direct'):
    """One evaluation on a single shard (the sequence phase1 -> global_step -> phase2 -> finish).  ``pairs='gemm'``: the per-point psi2
    tensor through a batched GEMM (_psi2_terms_chunk_gemm) -- for the shapes where the direct form takes minutes."""
    N_s, D = Y.shape
    Ng = N_s if N_global is None else N_global
    st = phase1(Z, sf2, alpha, Y, X_mu, X_S, chunk, pairs=pairs)
    gs = global_step(Z, sf2, alpha, beta, st, Ng, D, fixed_beta)
    p2 = phase2(Z, sf2, alpha, Y, X_mu, X_S, gs['Abar'], gs['Bbar'], chunk, want_embeddings, pairs=pairs)
    out = finish(Z, sf2, alpha, gs, p2, is_regime_A(X_S))
    out['grad_X_mu'], out['grad_X_S'] = p2['grad_X_mu'], p2['grad_X_S']
    out['stats'], out['gstep'] = st, gs
    return out


# ----------------------------------------------------------------------------------------- synthetic data
def synthetic_shard(N, D, M, Q, regime='A', seed=0, zseed=1, alpha_value=None):
    """Seeded synthetic inputs of SURVEY.md section 8(d) / BASELINE.md section 3."""
    rs = np.random.RandomState(seed)
    X = rs.randn(N, Q)
    W = rs.randn(Q, D)
    Y = np.sin(X.dot(W)) + 0.1 * rs.randn(N, D)
    X_mu = X + 0.05 * rs.randn(N, Q)
    X_S = np.zeros((N, Q)) if regime == 'A' else rs.uniform(0.05, 0.55, size=(N, Q))
    rz = np.random.RandomState(zseed)
    idx = rz.permutation(N)[:M]
    Z = X_mu[idx] + 0.05 * rz.randn(M, Q)
    if alpha_value is None:
        alpha_value = min(1.0, 1.0 / Q)
    return dict(Y=Y, X_mu=X_mu, X_S=X_S, Z=Z, sf2=1.0, alpha=np.full(Q, float(alpha_value)), beta=10.0)


# ----------------------------------------------------------------------------------------- shards in threads
def evaluate_sharded(Z, sf2, alpha, beta, Y, X_mu, X_S, shards=8, workers=None, want_embeddings=True, pairs='direct', chunk=4096):
    """The same evaluation as ``evaluate`` with the points cut into ``shards`` contiguous shards whose phase 1 / phase 2 run in a thread
    pool (numpy releases the GIL in its large kernels) and are summed in shard order -- exactly the map/reduce the path is built on
    (local_MapReduce.py:115-171, 250-277).  For the large-shape GPU tests, where one thread needs minutes."""
    from multiprocessing.pool import ThreadPool
    N_s, D = Y.shape
    cuts = [int(round(i * N_s / float(shards))) for i in range(shards + 1)]
    parts = [(cuts[i], cuts[i + 1]) for i in range(shards) if cuts[i + 1] > cuts[i]]
    pool = ThreadPool(workers or len(parts))
    try:
        sts = pool.map(lambda ab: phase1(Z, sf2, alpha, Y[ab[0]:ab[1]], X_mu[ab[0]:ab[1]], X_S[ab[0]:ab[1]], chunk, pairs=pairs), parts)
        st = dict(sts[0])
        for o in sts[1:]:
            for k in ('sum_exp_K_mi_K_im', 'exp_K_miY', 'sum_YYT', 'sum_exp_K_ii', 'KL'):
                st[k] = st[k] + o[k]
        gs = global_step(Z, sf2, alpha, beta, st, N_s, D)
        p2s = pool.map(lambda ab: phase2(Z, sf2, alpha, Y[ab[0]:ab[1]], X_mu[ab[0]:ab[1]], X_S[ab[0]:ab[1]], gs['Abar'], gs['Bbar'], chunk,
                                         want_embeddings, pairs=pairs), parts)
    finally:
        pool.close()
    p2 = dict(grad_Z_data=sum(o['grad_Z_data'] for o in p2s), grad_alpha_data=sum(o['grad_alpha_data'] for o in p2s))
    out = finish(Z, sf2, alpha, gs, p2, is_regime_A(X_S))
    out['grad_X_mu'] = np.concatenate([o['grad_X_mu'] for o in p2s]) if p2s[0]['grad_X_mu'] is not None else None
    out['grad_X_S'] = np.concatenate([o['grad_X_S'] for o in p2s]) if p2s[0]['grad_X_S'] is not None else None
    out['stats'], out['gstep'] = st, gs
    return out


# ----------------------------------------------------------------------------------------- BLAS-bound CPU baseline
def evaluate_blas(Z, sf2, alpha, beta, Y, X_mu, N_global=None, chunk=32768, fixed_beta=False, work=None, linalg='cholesky', workers=1):
    """Regime A with fixed embeddings (X_S == 0, no per-point gradients) arranged so that the time goes into DGEMM: the same
    formulation as phase1 / global_step / phase2 / finish above, but K_nm is generated once per chunk and KEPT for phase 2 (the
    two-phase protocol regenerates it), the two back-propagation products are GEMMs on K and Y into one buffer, element-wise work is
    done in place in buffers that are reused from chunk to chunk (``work``: pass the same dict again to reuse them across calls),
    and grad_alpha's mu^2 term uses the row sums of W.  This is what bench.py times as the CPU baseline (kind "port"): large chunks,
    all BLAS threads.  ``workers`` > 1 runs the chunks of each phase in a thread pool (the element-wise numpy work -- exp, products, row
    sums over 32768 x 512 arrays -- is single-threaded per call, numpy releases the GIL) and adds the partial sums in chunk order; the
    caller divides the BLAS threads among the workers.  Checked against evaluate() by tests/test_oracle_factorised.py."""
    Z, s2, a, b = _as_params(Z, sf2, alpha, beta)
    N_s, D = Y.shape
    M, Q = Z.shape
    Ng = N_s if N_global is None else N_global
    work = {} if work is None else work
    nch = (N_s + chunk - 1) // chunk
    workers = max(1, min(int(workers), nch))
    if work.get('shape') != (N_s, M, chunk, workers):
        work.clear()
        work['shape'] = (N_s, M, chunk, workers)
        work['K'] = [np.empty((min(chunk, N_s - i * chunk), M)) for i in range(nch)]
        work['W'] = [np.empty((min(chunk, N_s), M)) for _ in range(workers)]
    Z2a = (Z * Z).dot(a)                                        # sum_q a_q z_mq^2
    Za = (Z * a[None, :]).T.copy()                              # (Q, M)
    if workers > 1:
        import queue
        from multiprocessing.pool import ThreadPool
        pool = ThreadPool(workers)
        run = lambda f: pool.map(f, range(nch))
        free_w = queue.Queue()                                  # one W buffer per pool thread, handed out per chunk (reused across calls)
        for w in range(workers):
            free_w.put(w)
    else:
        pool = None
        run = lambda f: [f(i) for i in range(nch)]

    def p1(i):
        lo, hi = i * chunk, min(N_s, (i + 1) * chunk)
        mu, Yc, E = X_mu[lo:hi], Y[lo:hi], work['K'][i]
        np.dot(mu, Za, out=E)                                   # (n, M)
        E *= 2.0
        E -= (mu * mu).dot(a)[:, None]
        E -= Z2a[None, :]
        E *= 0.5
        np.exp(E, out=E)
        if s2 != 1.0:
            E *= s2
        return E.T.dot(E), E.T.dot(Yc), float(np.einsum('ij,ij->', Yc, Yc))

    try:
        parts = run(p1)
        Psi2 = np.zeros((M, M))
        C = np.zeros((M, D))
        sum_YYT = 0.0
        for P_, C_, s_ in parts:                                # chunk order: the result does not depend on the worker count
            Psi2 += P_
            C += C_
            sum_YYT += s_
        st = dict(sum_exp_K_mi_K_im=Psi2, exp_K_miY=C, sum_YYT=sum_YYT, sum_exp_K_ii=s2 * N_s, KL=0.0)
        gs = global_step(Z, sf2, alpha, beta, st, Ng, D, fixed_beta, linalg=linalg)
        B2 = np.ascontiguousarray(2.0 * gs['Bbar'].T)              # G[n, j] = sum_m' K[n, m'] 2 Bbar[j, m'] (partial_terms.py:238)
        At = np.ascontiguousarray(gs['Abar'].T)                     # G = K (2 Bbar) + Y Abar^T

        def p2(i):
            lo, hi = i * chunk, min(N_s, (i + 1) * chunk)
            K, mu, Yc = work['K'][i], X_mu[lo:hi], Y[lo:hi]
            wid = 0 if workers == 1 else free_w.get()
            try:
                W = work['W'][wid][:hi - lo]
                np.dot(K, B2, out=W)
                W += Yc.dot(At)
                W *= K                                              # W = G o K
                return W.T.dot(mu), W.sum(0), W.sum(1).dot(mu * mu)
            finally:
                if workers > 1:
                    free_w.put(wid)

        parts = run(p2)
    finally:
        if pool is not None:
            pool.close()
    R1 = np.zeros((M, Q))
    R0 = np.zeros(M)
    hmu2 = np.zeros(Q)
    for r1, r0, h2 in parts:
        R1 += r1
        R0 += r0
        hmu2 += h2
    gZ = a[None, :] * (R1 - Z * R0[:, None])
    ga = -0.5 * (hmu2 - 2.0 * np.sum(Z * R1, axis=0) + (Z * Z).T.dot(R0))
    out = finish(Z, sf2, alpha, gs, dict(grad_Z_data=gZ, grad_alpha_data=ga), True)
    out['stats'], out['gstep'] = st, gs
    return out
