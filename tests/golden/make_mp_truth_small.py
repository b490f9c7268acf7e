#!/usr/bin/env python3
"""A 60-digit (mpmath) evaluation of grad_Z on a small ill-conditioned sparse-GP case where the float64 oracle itself is wrong by 4.5e-3:
N = 17 points, M = 129 inducing points spread at random (1.5 randn) in Q = 3 dimensions, D = 4 -- more inducing points than data, cond(K_mm) = 2.7e7 (dev_fuzz_shapes.py's draw 27),
cond(K_mm + beta Psi2) = 1.8e9.  Round 4's tools/soak.sh flagged it (device vs float64 oracle: 4.5e-3); this script shows the ORACLE is the one that is off
(K_mm^-1 Psi2 accumulated in float64, the product the device now carries in double-double).  Formulas: partial_terms.py:102-131, 207-240
(regime A: Psi1 = K_nm, Psi2 = K^T K).  Stored: the inputs, the truth's grad_Z, the float64 oracle's distance from it.
Usage (build container, ~1 min): python tests/golden/make_mp_truth_small.py  ->  tests/golden/mp_truth_N17_M129.npz"""
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import factorised as Fz  # noqa: E402


def main():
    mp.mp.dps = 60
    N, D, M, Q = 17, 4, 129, 3
    d = Fz.synthetic_shard(N, D, N, Q, regime='A', seed=127, zseed=227, alpha_value=min(0.5, 2.0 / Q))
    d['Z'] = 1.5 * np.random.RandomState(327).randn(M, Q)
    ref = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], want_embeddings=False)
    Z = mp.matrix(d['Z'].tolist()); Y = mp.matrix(d['Y'].tolist()); X = mp.matrix(d['X_mu'].tolist())
    al = [mp.mpf(float(a)) for a in np.asarray(d['alpha']).ravel()]; s2 = mp.mpf(float(d['sf2'])); b = mp.mpf(float(d['beta']))

    def kern(x, z):
        return s2 * mp.exp(-mp.mpf(1) / 2 * sum(al[q] * (x[q] - z[q]) ** 2 for q in range(Q)))

    Kmm = mp.matrix(M, M)
    for i in range(M):
        for j in range(M):
            Kmm[i, j] = kern([Z[i, q] for q in range(Q)], [Z[j, q] for q in range(Q)])
    K = mp.matrix(N, M)
    for n in range(N):
        for m in range(M):
            K[n, m] = kern([X[n, q] for q in range(Q)], [Z[m, q] for q in range(Q)])
    Psi2 = K.T * K; C = K.T * Y
    A = Kmm + b * Psi2
    Ki = mp.inverse(Kmm); P = mp.inverse(A)
    E = P * C
    Dm = mp.mpf(D)
    Bbar = b * Dm / 2 * (Ki - P) - b ** 3 / 2 * (E * E.T)
    dFdK = Dm / 2 * (Ki - P) - b * Dm / 2 * (Ki * Psi2 * Ki) - b * b / 2 * (E * E.T)
    G = K * (2 * Bbar) + Y * (b * b * E).T
    gZ = np.zeros((M, Q))
    for j in range(M):
        for q in range(Q):
            s = mp.mpf(0)
            for m2 in range(M):
                s += (dFdK[j, m2] + dFdK[m2, j]) * Kmm[j, m2] * (-al[q]) * (Z[j, q] - Z[m2, q])
            for n in range(N):
                s += G[n, j] * K[n, j] * al[q] * (X[n, q] - Z[j, q])
            gZ[j, q] = float(s)
    rel = float(np.max(np.abs(ref['grad_Z'] - gZ)) / np.max(np.abs(gZ)))
    Kf = np.array(Kmm.tolist(), dtype=float)
    out = os.path.join(HERE, 'mp_truth_N17_M129.npz')
    np.savez_compressed(out, Y=d['Y'], X_mu=d['X_mu'], Z=d['Z'], alpha=np.asarray(d['alpha'], dtype=float), sf2=np.float64(d['sf2']), beta=np.float64(d['beta']),
                        truth_grad_Z=gZ, oracle_err_grad_Z=np.float64(rel), cond_Kmm=np.float64(np.linalg.cond(Kf)),
                        cond_A=np.float64(np.linalg.cond(Kf + float(b) * np.array(Psi2.tolist(), dtype=float))))
    print('float64 oracle vs 60-digit truth: grad_Z %.2e; cond(Kmm) %.1e; wrote %s' % (rel, np.linalg.cond(Kf), out))


if __name__ == '__main__':
    main()
