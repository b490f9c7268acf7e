"""Developer probe (GPU box): where the NaNs of the spurious first-evaluation Cholesky failure sit (M = 1024, free embeddings, fresh process)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from gparml_amd import _lib
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
N, D, M, Q = 1100, 2, 1024, 8
lib = _lib.load()
d = Fz.synthetic_shard(N, D, M, Q, regime='B', seed=11, zseed=12, alpha_value=0.8)
eng = ShardEngine(N, D, M, Q)
eng.upload_shard(d['Y'], d['X_mu'], d['X_S']); eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
def peek(name, cnt):
    buf = np.empty(cnt); rc = lib.gp_debug_peek(eng.h, name.encode(), buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), buf.size); return buf if rc == 0 else None
Mp = 1024
eng.phase1()
lib.gp_global_step_jitter(eng.h, 0)
gs = peek('gs', 24 + 8 * 64)
L = peek('Kmm', 2 * Mp * Mp).reshape(2, Mp, Mp)
X = peek('Linv', 2 * Mp * Mp).reshape(2, Mp, Mp)
def blockmap(A):
    bad = ~np.isfinite(A)
    return [''.join('#' if bad[128 * i:128 * i + 128, 128 * j:128 * j + 128].any() else '.' for j in range(8)) for i in range(8)]
print('flags', gs[16:18], 'logdets', gs[0:2])
if gs[16] != 0 or gs[17] != 0:
    for b in range(2):
        print('matrix %d: L non-finite blocks (rows = block row):' % b, ' '.join(blockmap(L[b])), '| Linv:', ' '.join(blockmap(X[b])))
        bad = np.argwhere(~np.isfinite(L[b]))
        if len(bad):
            print('   first non-finite entries of L:', bad[:6].tolist(), ' count', len(bad), ' columns touched', sorted(set((bad[:, 1] // 128).tolist())), ' min row', bad[:, 0].min())
eng.close()
