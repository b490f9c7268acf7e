"""Developer timing of phase 2 only (ablations via GP_P2_DBG)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
N, D, M, Q = 1000000, 100, 512, 10
d = Fz.synthetic_shard(N, D, M, Q, regime='A', seed=0, zseed=1)
eng = ShardEngine(N, D, M, Q)
eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
eng.phase1(); eng.global_step()
ts = []
for it in range(5):
    eng.phase2(False)
    ts.append(eng.timings()['p2_kernel_ms'])
print('GP_P2_DBG=%s p2_kernel_ms: %s' % (os.environ.get('GP_P2_DBG', '0'), ' '.join('%.2f' % t for t in ts)))
