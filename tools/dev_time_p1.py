"""Developer timing of phase 1 only (ablations via GP_P1_DBG)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gparml_amd.engine import ShardEngine
N, D, M, Q = 1000000, 100, 512, 10
rs = np.random.RandomState(0)
Y = rs.randn(N, D); X = rs.randn(N, Q); Z = X[:M] + 0.05 * rs.randn(M, Q)
eng = ShardEngine(N, D, M, Q)
eng.upload_shard(Y, X, np.zeros((N, Q)))
eng.set_globals(Z, 1.0, np.full(Q, 0.1), 10.0)
ts = []
for it in range(6):
    eng.phase1()
    ts.append(eng.timings()['p1_kernel_ms'])
print('GP_P1_DBG=%s p1_kernel_ms: %s' % (os.environ.get('GP_P1_DBG', '0'), ' '.join('%.2f' % t for t in ts)))
