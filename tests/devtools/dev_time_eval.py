"""Developer timing (GPU box): per-phase device times of one evaluation at a given size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
N, D, M, Q = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (1000000, 100, 512, 10))]
regime = sys.argv[5] if len(sys.argv) > 5 else 'A'
emb = regime == 'B'
t = time.time(); d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=0, zseed=1); print('synth %.1fs' % (time.time() - t))
eng = ShardEngine(N, D, M, Q)
t = time.time(); eng.upload_shard(d['Y'], d['X_mu'], d['X_S']); print('upload %.2fs' % (time.time() - t))
eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
for it in range(4):
    t = time.time()
    eng.phase1(); eng.global_step(); eng.phase2(emb); out = eng.finish()
    wall = (time.time() - t) * 1e3
    tm = eng.timings()
    print('iter %d wall %.2f ms | %s | F=%.6e' % (it, wall, ' '.join('%s=%.3f' % kv for kv in tm.items()), out['F']))
W = N * M * (3.0 * M + 4.0 * D + 12.0 * Q)
print('algorithmic flop (regime A formula) %.3e -> %.1f TFLOP/s at device total' % (W, W / (tm['total_ms'] * 1e-3) / 1e12))
