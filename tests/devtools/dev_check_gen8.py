"""Dev check of p2_gen8_kernel (regime A with embedding gradients; regime B) against the oracle."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz

def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(np.asarray(b))), 1e-300))

shapes = [(300, 5, 20, 3, 'A'), (1000, 7, 130, 10, 'A'), (3000, 20, 200, 6, 'A'), (40000, 12, 300, 5, 'A'), (40000, 12, 300, 5, 'B'), (3000, 20, 600, 12, 'A')]
for (N, D, M, Q, regime) in shapes:
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=11, zseed=12, alpha_value=0.4)
    ref = Fz.evaluate_sharded(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], shards=8, pairs='gemm')
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    print((N, D, M, Q, regime), 'start', flush=True)
    out = eng.evaluate(True)
    eng.close()
    print((N, D, M, Q, regime), 'F %.1e' % (abs(out['F'] - ref['F']) / abs(ref['F'])),
          {k: '%.1e' % rel(out[k], ref[k]) for k in ('grad_Z', 'grad_alpha', 'grad_X_mu')}, flush=True)
