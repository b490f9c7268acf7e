"""Ozaki gate, phase 2 (CPU only): G = [K | Y] [2 Bbar ; Abar^T] from S signed 7-bit digits per operand, digit products with a + b <= L, exact integer
emulation (same preamble as dev_ozaki_gate.py); grad_Z against the 80-bit truth at N = 1e5.  Result (profiles/r04_ozaki_gate.txt): K . Bbar cancels ten
digits, so phase 2 needs 49-bit operands -- S = 7 / L = 8, 28 products, seven int32 accumulator sets: 4.7e-8; S = 6: 5.2e-6 whatever is kept.
Usage: python tests/devtools/dev_ozaki_gate_phase2.py [N [seed [z_seed]]]"""
import os, sys
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dev_ozaki_gate.py')).read().split("for S, L in")[0])
Abar, Bbar, dFdK = gstep(Psi2, C)          # float64 statistics, double-double G in the global step
Bm = np.concatenate([2.0 * Bbar.T, Abar.T], axis=0)            # (M + D) x M: G = [K | Y] Bm
KY = np.concatenate([Kfull, Y], axis=1)
S_ = (dFdK + dFdK.T) * Kmm
gK = -a[None, :] * (Z * S_.sum(1)[:, None] - S_.dot(Z))
def finish(G):
    W = G * Kfull
    R1 = W.T.dot(X_mu); R0 = W.sum(0)
    return err(gK + a[None, :] * (R1 - Z * R0[:, None]))
print('float64 G: %.2e' % finish(KY.dot(Bm)), flush=True)
bmax = np.max(np.abs(Bm), axis=0); bsc = 2.0 ** (np.ceil(np.log2(bmax)) + 1)
ksc = np.concatenate([np.full(M, 2.0 * s2), ysc])
for S, L in ((6, 7), (6, 8), (7, 8), (7, 9), (8, 9)):
    dA = digits(KY, S, ksc[None, :]); dB = digits(Bm, S, bsc[None, :])
    # the row scale of [K | Y] differs per COLUMN k of the contraction: fold it into the B operand's rows instead (B'[k][m] = ksc[k] Bm[k][m])
    Bs = Bm * ksc[:, None]; b2 = np.max(np.abs(Bs), axis=0); b2 = 2.0 ** (np.ceil(np.log2(b2)) + 1)
    dB = digits(Bs, S, b2[None, :])
    pairs = [(i, j) for i in range(S) for j in range(S) if i + j + 2 <= L]
    G = np.zeros((N, M))
    for (i, j) in pairs:
        G += dA[i].dot(dB[j]) * 128.0 ** -(i + j + 2)
    G *= b2[None, :]
    print('S = %d, a+b <= %d: %2d products   G max rel err %.1e   grad_Z err %.2e' % (S, L, len(pairs), np.max(np.abs(G - KY.dot(Bm))) / np.max(np.abs(KY.dot(Bm))), finish(G)), flush=True)
