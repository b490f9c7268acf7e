import os, sys
sys.path.insert(0, ".")
import numpy as np
from gparml_amd.engine import ShardEngine
N, D, M, Q = 1000000, 100, 512, 10
rs = np.random.RandomState(0)
eng = ShardEngine(N, D, M, Q)
eng.upload_shard(rs.randn(N, D), rs.randn(N, Q), np.zeros((N, Q)))
eng.set_globals(rs.randn(M, Q), 1.0, np.full(Q, 0.1), 10.0)
ts = []
for i in range(4):
    eng.phase1(); ts.append(round(float(eng.timings()["psi1_ms"]), 3))
print("GP_PSI1_DBG", os.environ.get("GP_PSI1_DBG"), "psi1_ms", ts)
