"""Developer probe (GPU box): scope of the spurious first-evaluation jitter.  argv: N D M Q regime [warm]   (warm: another context evaluated first)"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from gparml_amd import _lib
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
N, D, M, Q = [int(x) for x in sys.argv[1:5]]
regime = sys.argv[5]
lib = _lib.load()
if len(sys.argv) > 6:
    dw = Fz.synthetic_shard(600, 3, 300, 5, regime='B', seed=1, zseed=2, alpha_value=0.8)
    e = ShardEngine(600, 3, 300, 5); e.upload_shard(dw['Y'], dw['X_mu'], dw['X_S']); e.set_globals(dw['Z'], dw['sf2'], dw['alpha'], dw['beta']); e.evaluate(True); print('warm-up jitter', e.last_jitter); e.close()
d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=11, zseed=12, alpha_value=0.8)
eng = ShardEngine(N, D, M, Q)
eng.upload_shard(d['Y'], d['X_mu'], d['X_S']); eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
def peek(name, cnt):
    buf = np.empty(cnt); rc = lib.gp_debug_peek(eng.h, name.encode(), buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), buf.size); return buf if rc == 0 else None
Mp = (M + 127) // 128 * 128
for rep in range(2):
    eng.phase1()
    lib.gp_global_step_jitter(eng.h, 0)
    gs = peek('gs', 24 + 8 * 64)
    st = peek('stats', Mp * Mp + Mp * ((D + 127) // 128 * 128) + 8)
    P2 = st[:Mp * Mp].reshape(Mp, Mp)
    Km = peek('KmmKeep', Mp * Mp).reshape(Mp, Mp)
    print('rep %d: flags %s logdets %s | Psi2 finite %s sym err %.1e diag min %.3e | Kmm finite %s diag min %.3e' % (
        rep, gs[16:18], gs[0:2], np.all(np.isfinite(P2)), np.max(np.abs(P2 - P2.T)), np.min(np.diag(P2)[:M]), np.all(np.isfinite(Km)), np.min(np.diag(Km)[:M])), flush=True)
eng.close()
