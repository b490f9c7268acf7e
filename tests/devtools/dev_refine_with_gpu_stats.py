"""Dev experiment (GPU box): the device's own statistics (Psi2, Psi1^T Y as the MFMA kernels accumulate them) through global steps of
increasing precision on the host -- where is the floor of grad_Z at the benchmark's conditioning, and how far is the device's float64 global
step from it?  Usage: python tests/devtools/dev_refine_with_gpu_stats.py [N [seed [z_seed]]]"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
from oracle import factorised as Fz
from make_hp_truth_large import unpack, rel
from gparml_amd.engine import ShardEngine

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
D, M, Q = 100, 512, 10
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ZSEED = int(sys.argv[3]) if len(sys.argv) > 3 else None
print('N %d seed %d z_seed %s' % (N, SEED, ZSEED), flush=True)
d = bench.synthetic(N, D, M, Q, seed=SEED, z_seed=ZSEED)
exe = os.path.join(ROOT, 'oracle', '_build', 'hp_truth')
os.makedirs(os.path.dirname(exe), exist_ok=True)
subprocess.check_call(['gcc', '-O2', '-fopenmp', '-o', exe, os.path.join(ROOT, 'oracle', 'hp_truth.c'), '-lm'])
work = tempfile.mkdtemp(prefix='hp_refine_')
d['Y'].tofile(work + '/Y.bin'); d['X_mu'].tofile(work + '/X.bin'); d['Z'].tofile(work + '/Z.bin'); d['alpha'].tofile(work + '/alpha.bin')
np.array([d['sf2'], d['beta']], dtype=np.float64).tofile(work + '/params.bin')
t0 = time.time()
subprocess.check_call([exe, work, str(N), str(D), str(M), str(Q)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                      env=dict(os.environ, OMP_NUM_THREADS='16'))   # 256 threads on 32-row blocks: 20x slower than 16
print('truth %.0f s' % (time.time() - t0), flush=True)
tru, _ = unpack(np.fromfile(work + '/truth_plain.bin'), M, Q)
part = {tag: np.fromfile(work + '/%s_plain.bin' % tag).reshape(M, -1) for tag in ('Abar', 'Bbar', 'dFdK')}
Psi2_t = np.fromfile(work + '/Psi2.bin').reshape(M, M); C_t = np.fromfile(work + '/C.bin').reshape(M, D)

eng = ShardEngine(N, D, M, Q)
eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
out = eng.evaluate(False)
g_Psi2, g_C = eng.download('PSI2_SUM'), eng.download('PSI1TY')
g_Abar, g_Bbar, g_dFdK = eng.download('DF_DPSI1TY'), eng.download('DF_DPSI2'), eng.download('DF_DKMM')
eng.close()
print('device        : grad_Z %.2e  grad_alpha %.2e  Abar %.2e  Bbar %.2e  dFdK %.2e   | statistics vs truth: Psi2 %.2e  C %.2e' % (
    rel(out['grad_Z'], tru['grad_Z']), rel(out['grad_alpha'], tru['grad_alpha']), rel(g_Abar, part['Abar']), rel(g_Bbar, part['Bbar']),
    rel(g_dFdK, part['dFdK']), rel(g_Psi2, Psi2_t), rel(g_C, C_t)), flush=True)

LD = np.longdouble
orig = Fz.global_step
MODE = {'m': 'plain', 'stats': 'cpu'}

def refine_inverse(A, X, iters):
    A_, X_ = A.astype(LD), X.astype(LD)
    I = np.eye(A.shape[0], dtype=LD)
    for _ in range(iters):
        R = I - A_.dot(X_)
        X_ = X_ + X.dot(R.astype(np.float64)).astype(LD)
        X = X_.astype(np.float64)
    return X_

def gstep(Z, sf2, alpha, beta, stats, N_global, D_, fixed_beta=False, linalg='cholesky'):
    if MODE['stats'] == 'gpu':
        stats = dict(stats); stats['sum_exp_K_mi_K_im'] = g_Psi2; stats['exp_K_miY'] = g_C
    elif MODE['stats'] == 'truth':
        stats = dict(stats); stats['sum_exp_K_mi_K_im'] = Psi2_t; stats['exp_K_miY'] = C_t
    g = orig(Z, sf2, alpha, beta, stats, N_global, D_, fixed_beta, linalg)
    mode = MODE['m']
    if mode == 'plain':
        return g
    Z, s2, a, b = Fz._as_params(Z, sf2, alpha, beta)
    Psi2, C = stats['sum_exp_K_mi_K_im'], stats['exp_K_miY']
    Kmm = g['Kmm']
    A = Kmm + b * Psi2
    P_ = refine_inverse(A, g['Kmm_plus_op_inv'], 2)
    Ki_ = refine_inverse(Kmm, g['Kmm_inv'], 2)
    if mode == 'refineP_E':
        P, Ki = P_.astype(np.float64), Ki_.astype(np.float64)
        E = P.dot(C)
        for _ in range(2):
            R = C.astype(LD) - A.astype(LD).dot(E.astype(LD))
            E = (E.astype(LD) + P.dot(R.astype(np.float64)).astype(LD)).astype(np.float64)
        EEt = E.dot(E.T); dKP = Ki - P; KPK = Ki.dot(Psi2).dot(Ki)
    else:
        C_ = C.astype(LD); Psi2_ = Psi2.astype(LD)
        E = P_.dot(C_); EEt = E.dot(E.T); dKP = Ki_ - P_; KPK = Ki_.dot(Psi2_).dot(Ki_)
    g = dict(g)
    g['Abar'] = np.asarray(b * b * E, dtype=np.float64)
    g['Bbar'] = np.asarray(0.5 * b * D_ * dKP - 0.5 * b ** 3 * EEt, dtype=np.float64)
    dF_dKmm = np.asarray(0.5 * D_ * dKP - 0.5 * b * D_ * KPK - 0.5 * b * b * EEt, dtype=np.float64)
    g['dF_dKmm'] = dF_dKmm
    dz = Z[:, None, :] - Z[None, :, :]
    S = (dF_dKmm + dF_dKmm.T) * Kmm
    g['grad_Z_K'] = -a[None, :] * (Z * S.sum(1)[:, None] - S.dot(Z))
    V = dF_dKmm * Kmm
    g['grad_alpha_K'] = -0.5 * np.einsum('ab,abq->q', V, dz * dz)
    g['BbarPsi2'] = g['Bbar'] * Psi2
    return g

Fz.global_step = gstep
wk = {}
for stats in ('cpu', 'gpu', 'truth'):
    for mode in ('plain', 'refineP_E', 'all_ld'):
        MODE['m'] = mode; MODE['stats'] = stats
        t0 = time.time()
        o = Fz.evaluate_blas(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], workers=8, work=wk)
        msg = '  '.join('%s %.2e' % (k, rel(o[k], tru[k])) for k in ('grad_Z', 'grad_alpha'))
        msg += '  ' + '  '.join('%s %.2e' % (tag, rel(o['gstep'][key], part[tag])) for tag, key in (('Abar', 'Abar'), ('Bbar', 'Bbar'), ('dFdK', 'dF_dKmm')))
        print('stats=%-5s %-10s (%.0f s): %s' % (stats, mode, time.time() - t0, msg), flush=True)
