"""Developer timing (GPU box): the free-embedding phase 1 alone (gp_phase1; no global step, so timing builds with wrong statistics can be measured).
usage: [GPARML_LIB=...] dev_time_phase1_b.py N D M Q"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
N, D, M, Q = (int(x) for x in sys.argv[1:5])
d = Fz.synthetic_shard(N, D, M, Q, regime='B', seed=3, zseed=4, alpha_value=min(0.5, 3.0 / Q))
eng = ShardEngine(N, D, M, Q)
eng.set_timing(2)
eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
best = 1e9
for i in range(5):
    eng.set_globals(d['Z'] + 1e-4 * i, d['sf2'], d['alpha'], d['beta'])
    eng.phase1()
    best = min(best, eng.timings()['p1_kernel_ms'])
print('phase-1 pair kernel (N %d, M %d, Q %d): %.3f ms' % (N, M, Q, best), flush=True)
eng.close()
