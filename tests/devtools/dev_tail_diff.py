"""Developer probe (GPU box): fused one-panel tail vs separate launches, buffer by buffer (gp_debug_peek)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from gparml_amd import _lib
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
lib = _lib.load()
lib.gp_debug_peek.restype = ctypes.c_int
lib.gp_debug_peek.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.c_long]
NAMES = ['Linv', 'Inv', 'E', 'T1', 'PsiE', 'dFdK', 'Bbar', 'Abar', 'Bm', 'gK', 'gs', 'T2']


def peek(eng, name):
    buf = np.zeros(2 * 128 * 256 + 1024)
    rc = lib.gp_debug_peek(eng.h, name.encode(), buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), buf.size)
    assert rc == 0, (name, rc)
    return buf


for (N, D, M, Q, regime) in [(40, 4, 2, 2, 'A'), (300, 5, 16, 3, 'A'), (3000, 5, 70, 4, 'A')]:
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=60, zseed=61, alpha_value=0.3)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    res = {}
    for tail in (1, 0, 1):
        lib.gp_debug_set_option(b'gs_tail', tail)
        eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
        eng.phase1(); eng.global_step(sync=False)
        res.setdefault(tail, []).append({k: peek(eng, k) for k in NAMES})
        eng.phase2(False); o = eng.finish()
        print('  tail', tail, 'F', o['F'])
    a, b, a2 = res[1][0], res[0][0], res[1][1]
    print((N, D, M, Q, regime))
    for k in NAMES:
        dab, daa = np.abs(a[k] - b[k]), np.abs(a[k] - a2[k])
        print('   %-5s fused-separate max %.2e at %d (n wrong %d)   fused-fused %.2e' % (k, dab.max(), int(dab.argmax()), int((dab > 0).sum()), daa.max()), flush=True)
    eng.close()
