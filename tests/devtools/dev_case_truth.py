"""Developer probe (GPU box): one small case of dev_fuzz_shapes.py (identified by its iteration number and shape) -- who is right, the library or the float64 oracle?
Truth = the bound F restated in x86 80-bit long double (direct formulas, no factorised shortcuts: kernel_exp.py:80, 143-146; partial_terms.py:464-472; Cholesky in
long double) and grad_Z by central differences of THAT (h = 1e-5: truncation 1e-10, rounding 1e-19 / 1e-5).  Prints the errors of the oracle and of the first and
second GPU evaluation against it, the jitter masks, and the conditioning.  usage: dev_case_truth.py IT N D M Q REGIME"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from oracle import factorised as Fz

LD = np.longdouble


def chol_ld(A):
    n = A.shape[0]
    L = np.zeros((n, n), dtype=LD)
    for j in range(n):
        v = A[j:, j] - L[j:, :j].dot(L[j, :j])
        assert v[0] > 0, 'not positive definite at column %d' % j
        L[j, j] = np.sqrt(v[0])
        L[j + 1:, j] = v[1:] / L[j, j]
    return L


def solve_ld(L, B):          # (L L^T)^-1 B, forward and back substitution
    n = L.shape[0]
    Yv = np.zeros_like(B)
    for i in range(n):
        Yv[i] = (B[i] - L[i, :i].dot(Yv[:i])) / L[i, i]
    X = np.zeros_like(B)
    for i in range(n - 1, -1, -1):
        X[i] = (Yv[i] - L[i + 1:, i].dot(X[i + 1:])) / L[i, i]
    return X


def F_ld(Z, s2, a, b, Y, mu, S):
    Z, a, Y, mu, S = (np.asarray(x, dtype=LD) for x in (Z, a, Y, mu, S))
    s2, b = LD(s2), LD(b)
    N, D = Y.shape
    M, Q = Z.shape
    dz = Z[:, None, :] - Z[None, :, :]
    Kmm = s2 * np.exp(-LD(0.5) * np.sum(a[None, None, :] * dz * dz, axis=2))
    d1 = a[None, :] * S + 1
    dm = mu[:, None, :] - Z[None, :, :]
    Psi1 = s2 / np.sqrt(np.prod(d1, axis=1))[:, None] * np.exp(-LD(0.5) * np.sum((a[None, :] / d1)[:, None, :] * dm * dm, axis=2))
    d2 = 2 * a[None, :] * S + 1
    Psi2 = np.zeros((M, M), dtype=LD)
    for n in range(N):
        zb = (Z[:, None, :] + Z[None, :, :]) / 2
        e = -LD(0.25) * np.sum(a[None, None, :] * dz * dz, axis=2) - np.sum((a / d2[n])[None, None, :] * (mu[n][None, None, :] - zb) ** 2, axis=2)
        Psi2 += s2 * s2 / np.sqrt(np.prod(d2[n])) * np.exp(e)
    C = Psi1.T.dot(Y)
    KL = LD(0.5) * np.sum(np.sum(S - np.log(S), 1) + np.sum(mu * mu, 1) - Q) if np.any(S != 0) else LD(0)
    Lk, La = chol_ld(Kmm), chol_ld(Kmm + b * Psi2)
    ldK, ldA = 2 * np.sum(np.log(np.diag(Lk))), 2 * np.sum(np.log(np.diag(La)))
    E = solve_ld(La, C)
    KiPsi2 = solve_ld(Lk, Psi2)
    two_pi = 2 * np.arctan(LD(1)) * 4
    return (-LD(0.5) * N * D * np.log(two_pi) + LD(0.5) * D * N * np.log(b) + LD(0.5) * D * ldK - LD(0.5) * D * ldA - LD(0.5) * b * np.sum(Y * Y)
            - LD(0.5) * b * D * s2 * N + LD(0.5) * b * D * np.trace(KiPsi2) + LD(0.5) * b * b * np.sum(C * E) - KL)


def grad_Z_truth(d):
    """(F, grad_Z) of the long-double bound by central differences, rounded to float64"""
    M, Q = np.asarray(d['Z']).shape
    h = LD(1e-5)
    g = np.zeros((M, Q), dtype=LD)
    for m in range(M):
        for q in range(Q):
            Zp = np.asarray(d['Z'], dtype=LD).copy(); Zp[m, q] += h
            Zm = np.asarray(d['Z'], dtype=LD).copy(); Zm[m, q] -= h
            g[m, q] = (F_ld(Zp, d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S']) - F_ld(Zm, d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'])) / (2 * h)
    return float(F_ld(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'])), np.asarray(g, dtype=np.float64)


def main():
    it, N, D, M, Q = (int(x) for x in sys.argv[1:6])
    regime = sys.argv[6]
    d = Fz.synthetic_shard(N, D, min(M, N), Q, regime=regime, seed=100 + it, zseed=200 + it, alpha_value=min(0.5, 2.0 / Q))
    if M > N:
        d['Z'] = 1.5 * np.random.RandomState(300 + it).randn(M, Q)
    ref = Fz.evaluate_sharded(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], shards=4, pairs='gemm')
    F0, g = grad_Z_truth(d)
    sc = np.max(np.abs(g))
    print('truth: F %.15g  max |grad_Z| %.6g' % (float(F0), sc))
    print('oracle (float64): F rel err %.2e   grad_Z err %.2e' % (abs(ref['F'] - float(F0)) / abs(float(F0)), np.max(np.abs(ref['grad_Z'] - g)) / sc))
    try:
        from gparml_amd.engine import ShardEngine
        eng = ShardEngine(N, D, M, Q)
        eng.upload_shard(d['Y'], d['X_mu'], d['X_S']); eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
        for rep in range(2):
            out = eng.evaluate(regime == 'B')
            print('library evaluation %d: jitter mask %d   F rel err %.2e   grad_Z err %.2e   (against the oracle: %.2e)' % (
                rep + 1, eng.last_jitter, abs(out['F'] - float(F0)) / abs(float(F0)), np.max(np.abs(out['grad_Z'] - g)) / sc, np.max(np.abs(out['grad_Z'] - ref['grad_Z'])) / sc))
        eng.close()
    except Exception as e:      # no GPU here: the oracle's line alone
        print('library not run:', type(e).__name__, str(e)[:100])


if __name__ == '__main__':
    main()
