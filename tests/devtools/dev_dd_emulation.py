"""Which part of the global step carries grad_Z's float64 error?  (round 4; CPU only, ~1 min at N = 1e5 on 8 cores)

The global step on float64 statistics with selected products in TRUE double-double (error-free transformations in numpy: two_prod by Dekker splitting,
two_sum; the arithmetic of ddacc_gemm_kernel / solve_residual_kernel in csrc/linalg.hip), float64 phase 2, against the 80-bit truth of the benchmark workload
(tests/golden/hp_truth_large_N<N>*.npz).  Result at N = 1e5, draw (100, -):
    current round-3 product (float64, E refined)                    1.14e-05
    + G = Kmm^-1 Psi2 accumulated in double-double (round 4)        1.59e-08      <- the one product that matters
    everything in double-double (refined inverses, all products)    1.49e-08
    G in double-double but E NOT refined                            1.71e-05
Usage: python tests/devtools/dev_dd_emulation.py [N [seed [z_seed]]]"""
import os, sys, time
import numpy as np
import scipy.linalg as sla
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from oracle import factorised as Fz
LD = np.longdouble
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
zseed = int(sys.argv[3]) if len(sys.argv) > 3 else None
D, M, Q = 100, 512, 10
d = bench.synthetic(N, D, M, Q, seed=seed, z_seed=zseed)
fx = np.load(ROOT + '/tests/golden/hp_truth_large_N%d%s.npz' % (N, '' if zseed is None else '_s%d_z%d' % (seed, zseed)))
truZ = fx['truth_grad_Z']
Z, s2, a, b = Fz._as_params(d['Z'], d['sf2'], d['alpha'], d['beta'])
Y, X_mu = d['Y'], d['X_mu']
chunk = 25000
Ks = []
Psi2 = np.zeros((M, M)); C = np.zeros((M, D))
Z2a = (Z * Z).dot(a); Za = (Z * a[None, :]).T.copy()
for lo in range(0, N, chunk):
    mu = X_mu[lo:lo + chunk]
    E = 2.0 * mu.dot(Za); E -= (mu * mu).dot(a)[:, None]; E -= Z2a[None, :]; E *= 0.5; np.exp(E, out=E); E *= s2
    Ks.append(E); Psi2 += E.T.dot(E); C += E.T.dot(Y[lo:lo + chunk])
dz = Z[:, None, :] - Z[None, :, :]
Kmm = s2 * np.exp(-0.5 * np.sum(a[None, None, :] * dz * dz, axis=2))
def phase2(Abar, Bbar):
    B2 = np.ascontiguousarray(2.0 * Bbar.T); At = np.ascontiguousarray(Abar.T)
    R1 = np.zeros((M, Q)); R0 = np.zeros(M)
    for i, lo in enumerate(range(0, N, chunk)):
        K = Ks[i]; W = K.dot(B2); W += Y[lo:lo + chunk].dot(At); W *= K
        R1 += W.T.dot(X_mu[lo:lo + chunk]); R0 += W.sum(0)
    return a[None, :] * (R1 - Z * R0[:, None])
def gradZ(Abar, Bbar, dFdK):
    S = (dFdK + dFdK.T) * Kmm
    return -a[None, :] * (Z * S.sum(1)[:, None] - S.dot(Z)) + phase2(Abar, Bbar)
err = lambda g: float(np.max(np.abs(g - truZ)) / np.max(np.abs(truZ)))
def chol_inv(X):
    L = np.linalg.cholesky(X); return sla.cho_solve((L, True), np.eye(X.shape[0]))

# ---- double-double emulation
def split(x):
    t = 134217729.0 * x; h = t - (t - x); return h, x - h
def two_prod(x, y):
    p = x * y
    xh, xl = split(x); yh, yl = split(y)
    e = ((xh * yh - p) + xh * yl + xl * yh) + xl * yl
    return p, e
def two_sum(x, y):
    s = x + y; bb = s - x
    return s, (x - (s - bb)) + (y - bb)
def ddgemm(Ah, Al, Bh, Bl):
    """(Ah + Al) (Bh + Bl) with dd accumulation; Al / Bl may be None.  Returns (hi, lo) unnormalised like the device kernel."""
    m, K = Ah.shape; n = Bh.shape[1]
    hi = np.zeros((m, n)); lo = np.zeros((m, n))
    for k in range(K):
        ah = Ah[:, k:k + 1]; bh = Bh[k:k + 1, :]
        p, e = two_prod(ah, bh)
        if Al is not None: e = e + Al[:, k:k + 1] * bh
        if Bl is not None: e = e + ah * Bl[k:k + 1, :]
        hi, er = two_sum(hi, p)
        lo = lo + (er + e)
    return hi, lo

t0 = time.time()
A64 = Kmm + b * Psi2
Ki0 = chol_inv(Kmm); P0 = chol_inv(A64)
E0 = P0.dot(C); h, l = ddgemm(A64, None, E0, None); R = (C - h) - l; E1 = E0 + P0.dot(R)
EEt = E1.dot(E1.T)
def fin(name, W, careful):
    if careful:
        dh, dl = two_sum(Ki0, -P0); yh, yl = two_sum(dh, -b * W); y = yh + (yl + dl); dKP = dh + dl
    else:
        dKP = Ki0 - P0; y = dKP - b * W
    print('%-60s grad_Z err %.2e   (%.0f s)' % (name, err(gradZ(b * b * E1, 0.5 * b * D * dKP - 0.5 * b ** 3 * EEt, 0.5 * D * y - 0.5 * b * b * EEt)), time.time() - t0), flush=True)
W0 = Ki0.dot(Psi2).dot(Ki0)
fin('current product (all f64, E refined)', W0, False)
fin('current + careful assembly', W0, True)
Gh, Gl = ddgemm(Ki0, None, Psi2, None); G = Gh + Gl
fin('G dd-accumulated (rounded), G Ki0 f64, plain assembly', G.dot(Ki0), False)
fin('G dd-accumulated (rounded), G Ki0 f64, careful assembly', G.dot(Ki0), True)
W0s = 0.5 * (W0 + W0.T)
fin('current, KPK symmetrised', W0s, False)
# which matrix product order: Ki0 (Psi2 Ki0)
fin('f64 Ki0.(Psi2.Ki0)', Ki0.dot(Psi2.dot(Ki0)), False)
# plain E (no refinement) with dd G
E = E0; EEt = E.dot(E.T); E1 = E0
fin('G dd, E NOT refined', G.dot(Ki0), False)
