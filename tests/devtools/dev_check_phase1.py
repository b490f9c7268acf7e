"""Developer check (GPU box): potrf/inverse hook, phase 1 and global step against the CPU oracle."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from gparml_amd import _lib
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz

lib = _lib.load()
print(lib.gp_version().decode())
rs = np.random.RandomState(0)
ok = True

def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)

for n in (5, 128, 200, 512):
    B = rs.randn(n, n); A = B.dot(B.T) + n * np.eye(n)
    L = np.empty((n, n)); Ai = np.empty((n, n)); ld = ctypes.c_double()
    rc = lib.gp_debug_potrf_inverse(0, n, A.ctypes.data_as(_lib._dp), L.ctypes.data_as(_lib._dp), Ai.ctypes.data_as(_lib._dp), ctypes.byref(ld))
    e1 = rel(L, np.linalg.cholesky(A)); e2 = rel(Ai, np.linalg.inv(A)); e3 = abs(ld.value - np.linalg.slogdet(A)[1])
    good = rc == 0 and e1 < 1e-12 and e2 < 1e-11 and e3 < 1e-9
    ok &= good
    print('potrf n=%d rc=%d L %.2e inv %.2e logdet %.2e %s' % (n, rc, e1, e2, e3, 'OK' if good else 'FAIL'))
A = -np.eye(4)
rc = lib.gp_debug_potrf_inverse(0, 4, A.ctypes.data_as(_lib._dp), None, None, None)
print('potrf non-PD rc=%d (expect 2)' % rc); ok &= rc == 2

for (N, D, M, Q, regime) in [(300, 5, 20, 3, 'A'), (1000, 7, 130, 10, 'A'), (5000, 100, 512, 10, 'A')]:
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=1, zseed=2)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    eng.phase1()
    st = Fz.phase1(d['Z'], d['sf2'], d['alpha'], d['Y'], d['X_mu'], d['X_S'])
    e = dict(psi1=rel(eng.download('PSI1'), Fz._psi1_chunk(d['Z'], d['sf2'], d['alpha'], d['X_mu'], d['X_S'])[0]),
             psi2=rel(eng.download('PSI2_SUM'), st['sum_exp_K_mi_K_im']), C=rel(eng.download('PSI1TY'), st['exp_K_miY']))
    sc = eng.scalars()
    e['yy'] = abs(sc['sum_YYT'] - st['sum_YYT']) / st['sum_YYT']
    eng.global_step()
    gs = Fz.global_step(d['Z'], d['sf2'], d['alpha'], d['beta'], st, N, D)
    sc = eng.scalars()
    e['F'] = abs(sc['F'] - gs['F']) / abs(gs['F'])
    e['gbeta'] = abs(sc['grad_beta'] - gs['grad_beta']) / abs(gs['grad_beta'])
    e['gsf2'] = abs(sc['grad_sf2'] - gs['grad_sf2']) / abs(gs['grad_sf2'])
    e['Ki'] = rel(eng.download('KMM_INV'), gs['Kmm_inv']); e['P'] = rel(eng.download('KMM_PLUS_OP_INV'), gs['Kmm_plus_op_inv'])
    e['Abar'] = rel(eng.download('DF_DPSI1TY'), gs['Abar']); e['Bbar'] = rel(eng.download('DF_DPSI2'), gs['Bbar'])
    e['dFdK'] = rel(eng.download('DF_DKMM'), gs['dF_dKmm'])
    good = all(v < 1e-8 for v in e.values())
    ok &= good
    print('N=%d D=%d M=%d Q=%d %s: %s  %s  t=%s' % (N, D, M, Q, regime, ' '.join('%s=%.1e' % kv for kv in e.items()), 'OK' if good else 'FAIL', eng.timings()))
    eng.close()
sys.exit(0 if ok else 1)
