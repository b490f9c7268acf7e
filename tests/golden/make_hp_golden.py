#!/usr/bin/env python3
"""Higher-precision truth at the benchmark's own hyper-parameters (BASELINE configs[2]: D=100, M=512, Q=10, alpha=0.1,
beta=10, the Z construction of bench.py) at an N every CPU path can run.

Why: at these hyper-parameters cond(Kmm + beta*Psi2) is ~1e9-1e10 and grad_Z is the small difference of a data part and a
Kmm part, so two float64 evaluations that only differ in the factorisation (LU inv/slogdet in the reference,
partial_terms.py:60,82,95,449-450; Cholesky here) already disagree at the 1e-5 level.  To say which float64 path is closer,
this script computes the SAME evaluation in extended precision (x86 80-bit long double, eps 1.08e-19: Psi-statistics, both
factorisations, every contraction) and stores

    inputs                                  Y, X_mu, Z, sf2, alpha, beta            (bench.py synthetic(), seed 100)
    truth_*                                 F and every gradient block, rounded to float64 from the long-double evaluation
    truth_uncertainty                       change of the truth under one Newton refinement step of both inverses
    ref_*                                   the imported reference's float64 outputs (its own LU path) on the same inputs
    err_ref_* / err_oracle_*                their errors against the truth, relative to the block's largest magnitude
    cond_Kmm, cond_A

The GPU test (tests/test_gpu_hp_truth.py) asserts  err_gpu <= max(1e-5, err_ref)  block by block, F at 1e-6.

Runs only in the build container (imports /root/reference through make_golden.load_reference; ~15 min, ~10 GB: the
reference stores the (N, M, M) tensor, partial_terms.py:45).  Usage:  python tests/golden/make_hp_golden.py [N]
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

LD = np.longdouble


# ------------------------------------------------------------------------------------------- long-double linear algebra
def chol_ld(A):
    """Lower Cholesky factor in long double (left-looking, one matvec per column)."""
    n = A.shape[0]
    L = np.zeros((n, n), dtype=LD)
    for j in range(n):
        v = A[j:, j] - L[j:, :j].dot(L[j, :j])
        assert v[0] > 0, 'not positive definite at column %d' % j
        L[j, j] = np.sqrt(v[0])
        L[j + 1:, j] = v[1:] / L[j, j]
    return L


def tri_inv_ld(L):
    """Inverse of a lower-triangular matrix in long double (forward substitution, row by row)."""
    n = L.shape[0]
    X = np.zeros((n, n), dtype=LD)
    for i in range(n):
        r = -L[i, :i].dot(X[:i, :])
        r[i] += 1.0
        X[i, :] = r / L[i, i]
    return X


def spd_inv_logdet_ld(A):
    L = chol_ld(A)
    X = tri_inv_ld(L)
    return X.T.dot(X), 2.0 * np.sum(np.log(np.diag(L)))


def evaluate_ld(Z, sf2, alpha, beta, Y, X_mu, refine=False):
    """Regime A (X_S == 0) bound and hyper-parameter gradients, every operation in long double.
    Formulation: SURVEY.md section 7 / oracle/factorised.py (Psi1 = K_nm, Psi2 = K^T K; partial_terms.py:102-138,
    207-240, 286-299, 322-360, 436-473)."""
    Z, Y, X = Z.astype(LD), Y.astype(LD), X_mu.astype(LD)
    a, s2, b = alpha.astype(LD), LD(sf2), LD(beta)
    N, D = Y.shape
    M, Q = Z.shape
    half = LD(1) / LD(2)
    e = np.zeros((N, M), dtype=LD)
    for q in range(Q):
        d = X[:, q][:, None] - Z[:, q][None, :]
        e += a[q] * d * d
    K = s2 * np.exp(-half * e)
    Psi2 = K.T.dot(K)
    C = K.T.dot(Y)
    sumYY = np.sum(Y * Y)
    Psi0 = s2 * N
    dz = Z[:, None, :] - Z[None, :, :]
    Kmm = s2 * np.exp(-half * np.sum(a[None, None, :] * dz * dz, axis=2))
    A = Kmm + b * Psi2
    Ki, ldK = spd_inv_logdet_ld(Kmm)
    P, ldA = spd_inv_logdet_ld(A)
    if refine:   # one Newton step: X <- X (2I - A X)
        I2 = 2 * np.eye(M, dtype=LD)
        Ki = Ki.dot(I2 - Kmm.dot(Ki))
        P = P.dot(I2 - A.dot(P))
    E = P.dot(C)
    two_pi = 2 * np.arccos(LD(-1))
    trKi, trP, trCE = np.sum(Ki * Psi2), np.sum(P * Psi2), np.sum(C * E)
    F = (-half * N * D * np.log(two_pi) + half * D * N * np.log(b) + half * D * ldK - half * D * ldA - half * b * sumYY
         - half * b * D * Psi0 + half * b * D * trKi + half * b * b * trCE)
    EEt = E.dot(E.T)
    Abar = b * b * E
    Bbar = half * b * D * (Ki - P) - half * b ** 3 * EEt
    dFdK = half * D * (Ki - P) - half * b * D * Ki.dot(Psi2).dot(Ki) - half * b * b * EEt
    grad_beta = (half * N * D / b - half * D * trP - half * sumYY - half * D * Psi0 + half * D * trKi + b * trCE
                 - half * b * b * np.sum(E * Psi2.dot(E)))
    S = (dFdK + dFdK.T) * Kmm
    gZ_K = -a[None, :] * (Z * S.sum(1)[:, None] - S.dot(Z))
    V = dFdK * Kmm
    ga_K = np.array([-half * np.sum(V * dz[:, :, q] * dz[:, :, q]) for q in range(Q)], dtype=LD)
    grad_sf2 = (np.sum(V) + np.sum(Abar * C) + 2 * np.sum(Bbar * Psi2) - half * b * D * Psi0) / s2
    G = Y.dot(Abar.T) + 2 * K.dot(Bbar)
    W = G * K
    W1 = W.sum(0)
    WX = W.T.dot(X)
    WX2 = W.T.dot(X * X)
    gZ = a[None, :] * (WX - Z * W1[:, None]) + gZ_K
    ga = -half * np.sum(WX2 - 2 * Z * WX + Z * Z * W1[:, None], axis=0) + ga_K
    out = dict(F=F, grad_Z=gZ, grad_alpha=ga, grad_sf2=grad_sf2, grad_beta=grad_beta, Abar=Abar, Bbar=Bbar, dF_dKmm=dFdK,
               Psi2=Psi2, C=C, Kmm_inv=Ki, Kmm_plus_op_inv=P)
    return out


BLOCKS = ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'Abar', 'Bbar', 'dF_dKmm')
STORED = ('F', 'grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta')   # the M x M partials are compared through their error only (2 MB each)


def rel_err(x, truth):
    x, truth = np.asarray(x, dtype=LD), np.asarray(truth, dtype=LD)
    return float(np.max(np.abs(x - truth)) / np.max(np.abs(truth)))


def main():
    import bench
    import make_golden as mg
    from oracle import factorised as Fz
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    D, M, Q = 100, 512, 10
    d = bench.synthetic(N, D, M, Q, seed=100)                  # the benchmark's generator and hyper-parameters
    assert np.all(d['X_S'] == 0)
    t0 = time.time()
    tru = evaluate_ld(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'])
    tru2 = evaluate_ld(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], refine=True)
    print('[hp] long-double truth: %.0f s' % (time.time() - t0))
    save = dict(in_Y=d['Y'], in_X_mu=d['X_mu'], in_Z=d['Z'], in_sf2=np.float64(d['sf2']), in_alpha=d['alpha'],
                in_beta=np.float64(d['beta']), in_N=np.int64(N), in_D=np.int64(D), in_M=np.int64(M), in_Q=np.int64(Q))
    unc = {}
    for k in ('F',) + BLOCKS:
        if k in STORED:
            save['truth_' + k] = np.asarray(tru[k], dtype=np.float64)
        unc[k] = rel_err(tru2[k], tru[k])
    save['truth_uncertainty'] = np.array([unc[k] for k in ('F',) + BLOCKS])
    print('[hp] truth uncertainty (Newton-refined vs plain):', {k: '%.1e' % v for k, v in unc.items()})
    A64 = np.asarray(tru['Psi2'], dtype=np.float64) * d['beta']
    dz = d['Z'][:, None, :] - d['Z'][None, :, :]
    Kmm64 = d['sf2'] * np.exp(-0.5 * np.sum(d['alpha'][None, None, :] * dz * dz, axis=2))
    save['cond_Kmm'] = np.float64(np.linalg.cond(Kmm64))
    save['cond_A'] = np.float64(np.linalg.cond(Kmm64 + A64))
    print('[hp] cond(Kmm) %.2e  cond(Kmm+beta Psi2) %.2e' % (save['cond_Kmm'], save['cond_A']))

    # ---- float64 CPU port (oracle/factorised.py, Cholesky)
    orc = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], want_embeddings=False)
    orc.update(Abar=orc['gstep']['Abar'], Bbar=orc['gstep']['Bbar'], dF_dKmm=orc['gstep']['dF_dKmm'])
    for k in ('F',) + BLOCKS:
        save['err_oracle_' + k] = np.float64(rel_err(orc[k], tru[k]))

    # ---- the imported reference itself (LU inv + slogdet), its own call sequence
    t0 = time.time()
    mods = mg.load_reference()
    pt = mods['partial_terms'].partial_terms(d['Z'].copy(), d['sf2'], d['alpha'].copy(), d['beta'], M, Q, N, D)
    pt.set_data(d['Y'], d['X_mu'], d['X_S'], True)
    ref = {}
    ref['F'] = pt.logmarglik()
    ref['dF_dKmm'], ref['Abar'], ref['Bbar'] = pt.dF_dKmm(), pt.dF_dexp_K_miY(), pt.dF_dexp_K_mi_K_im()
    dFii = pt.dF_dexp_K_ii()
    ref['grad_Z'] = pt.grad_Z(ref['dF_dKmm'], pt.dKmm_dZ(), ref['Abar'], pt.dexp_K_miY_dZ(), ref['Bbar'], pt.dexp_K_mi_K_im_dZ())
    ref['grad_alpha'] = pt.grad_alpha(ref['dF_dKmm'], pt.dKmm_dalpha(), ref['Abar'], pt.dexp_K_miY_dalpha(), ref['Bbar'],
                                      pt.dexp_K_mi_K_im_dalpha())
    ref['grad_sf2'] = pt.grad_sf2(ref['dF_dKmm'], pt.dKmm_dsf2(), dFii, pt.dexp_K_ii_dsf2(), ref['Abar'], pt.dexp_K_miY_dsf2(),
                                  ref['Bbar'], pt.dexp_K_mi_K_im_dsf2())
    ref['grad_beta'] = pt.grad_beta()
    print('[hp] reference run: %.0f s' % (time.time() - t0))
    for k in ('F',) + BLOCKS:
        if k in STORED:
            save['ref_' + k] = np.asarray(ref[k], dtype=np.float64)
        save['err_ref_' + k] = np.float64(rel_err(ref[k], tru[k]))
    for k in ('F',) + BLOCKS:
        print('[hp] %-10s  reference(LU) %.2e   oracle(Cholesky) %.2e   truth+-%.1e' % (k, save['err_ref_' + k], save['err_oracle_' + k], unc[k]))
    out = os.path.join(HERE, 'hp_truth_config2_N%d.npz' % N)
    np.savez_compressed(out, **save)
    print('[hp] wrote', out, os.path.getsize(out), 'bytes')


if __name__ == '__main__':
    main()
