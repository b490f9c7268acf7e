"""VERDICT r03 item 6 -- the Ozaki gate, step 1 (CPU only): grad_Z's distance from the 80-bit truth when the phase-1 statistics Psi2 = K^T K and
C = K^T Y come from signed 7-bit digit expansions (S digits per operand, the digit products with a + b <= L kept), emulated EXACTLY (integer-valued
float64 products), with the device's global step (double-double K_mm^-1 Psi2 product, numpy error-free transformations) and a float64 phase 2.
The int8 kernel (gparml_amd/csrc/p1i8.hip) reproduces these numbers to the digit (S = 6 / L = 7: 3.65e-6 here and on the MI355X).
Usage: python tests/devtools/dev_ozaki_gate.py [N [seed [z_seed]]]      (N = 1e5: ~6 min on 8 cores)   -> profiles/r04_ozaki_gate.txt"""
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

def gstep(Psi2, C):
    A64 = Kmm + b * Psi2
    Ki0 = chol_inv(Kmm); P0 = chol_inv(A64)
    E0 = P0.dot(C); h, l = ddgemm(A64, None, E0, None); R = (C - h) - l; E1 = E0 + P0.dot(R)
    Gh, Gl = ddgemm(Ki0, None, Psi2, None); G = Gh + Gl
    EEt = E1.dot(E1.T); dKP = Ki0 - P0
    return b * b * E1, 0.5 * b * D * dKP - 0.5 * b ** 3 * EEt, 0.5 * D * (dKP - b * G.dot(Ki0)) - 0.5 * b * b * EEt
Kfull = np.concatenate(Ks)
def digits(X, S, scale):
    r = X / scale; out = []
    for j in range(S):
        r = r * 128.0; dg = np.rint(r); r = r - dg; out.append(dg)
    return out
ymax = np.max(np.abs(Y), axis=0); ysc = 2.0 ** (np.ceil(np.log2(ymax)) + 1)
for S, L in ((5, 6), (5, 7), (5, 10), (6, 7), (6, 8), (6, 12), (7, 8)):
    dk = digits(Kfull, S, 2.0 * s2); dy = digits(Y, S, ysc[None, :])
    pairs = [(i, j) for i in range(S) for j in range(S) if i + j + 2 <= L]
    P2 = np.zeros((M, M)); Cq = np.zeros((M, D))
    for (i, j) in pairs:
        w = 128.0 ** -(i + j + 2)
        P2 += dk[i].T.dot(dk[j]) * w; Cq += dk[i].T.dot(dy[j]) * w
    P2 *= 4 * s2 * s2; Cq *= 2 * s2 * ysc[None, :]
    e = err(gradZ(*gstep(P2, Cq)))
    print('S = %d digits, pairs a+b <= %2d: %2d products   Psi2 max rel err %.1e  diag bias %.1e   grad_Z err vs truth %.2e' % (
        S, L, len(pairs), np.max(np.abs(P2 - Psi2)) / np.max(np.abs(Psi2)), np.mean(np.diag(P2 - Psi2) / np.diag(Psi2)), e), flush=True)
