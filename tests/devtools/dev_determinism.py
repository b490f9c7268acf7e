"""Dev check: are two consecutive evaluations bit-identical?  Prints the keys that differ, per shape and per evaluate(want_embedding_grads)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
shapes = [(300, 5, 20, 3, 'A', 0.4), (1000, 7, 130, 10, 'A', 0.4), (3000, 20, 200, 6, 'A', 0.4), (5000, 1, 128, 2, 'B', 0.5), (20000, 12, 140, 5, 'A', 1.0)]
for (N, D, M, Q, regime, alpha) in shapes:
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=11, zseed=12, alpha_value=alpha)
    for emb in (True, False):
        eng = ShardEngine(N, D, M, Q)
        eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
        eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
        outs = [eng.evaluate(emb) for _ in range(4)]
        eng.close()
        diff = {}
        for k, v in outs[0].items():
            if isinstance(v, (float, np.ndarray)):
                for i in range(1, 4):
                    if not np.array_equal(np.asarray(v), np.asarray(outs[i][k])):
                        diff.setdefault(k, []).append((i, float(np.max(np.abs(np.asarray(v) - np.asarray(outs[i][k]))) / (np.max(np.abs(np.asarray(v))) + 1e-300))))
        print((N, D, M, Q, regime), 'emb' if emb else 'noemb', diff, flush=True)
