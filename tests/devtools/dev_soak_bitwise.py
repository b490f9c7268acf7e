"""Dev soak (GPU box): many repeated evaluations of one shape must be bit-identical (a race in an LDS exchange, a missing barrier or wait
shows up as a differing bit long before it shows up as a wrong digit).  Usage: dev_soak_bitwise.py N D M Q regime reps"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
N, D, M, Q = (int(a) for a in sys.argv[1:5]); regime = sys.argv[5]; reps = int(sys.argv[6])
d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=5, zseed=6, alpha_value=min(0.3, 3.0 / Q))
eng = ShardEngine(N, D, M, Q)
eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
eng.evaluate(True)
ref = eng.evaluate(True)
bad = 0
t0 = time.time()
for r in range(reps):
    out = eng.evaluate(True)
    for k, v in ref.items():
        if isinstance(v, (float, np.ndarray)) and not np.array_equal(np.asarray(v), np.asarray(out[k])):
            bad += 1
            print('rep', r, 'differs in', k, float(np.max(np.abs(np.asarray(v) - np.asarray(out[k])))), flush=True)
print('SOAK', (N, D, M, Q, regime), 'reps', reps, 'differing blocks', bad, '%.1f ms/eval' % (1e3 * (time.time() - t0) / reps), flush=True)
eng.close()
