import sys, os
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from gparml_amd.engine import ShardEngine
N, D, M, Q = 1000000, 100, 512, 10
d = bench.synthetic(N, D, M, Q, seed=100)
eng = ShardEngine(N, D, M, Q)
eng.upload_shard(d['Y'], d['X_mu'], d['X_S']); eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
for emb in (False, True):
    for _ in range(3):
        eng.phase1(); eng.global_step(sync=False); eng.phase2(emb); out = eng.finish()
    print('want_emb', emb, {k: round(v, 3) for k, v in eng.timings().items()}, flush=True)
eng.close()
