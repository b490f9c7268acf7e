"""Developer check (GPU box): create / evaluate / destroy contexts repeatedly in both regimes and watch free device memory."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
free0 = None
for it in range(6):
    for regime, Q in (('A', 10), ('B', 10), ('B', 30)):
        N, D, M = 20000, 10, 200
        d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=it, zseed=1, alpha_value=0.5)
        e = ShardEngine(N, D, M, Q)
        e.upload_shard(d['Y'], d['X_mu'], d['X_S'])
        e.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
        for k in range(3):
            e.phase1(); e.global_step(); e.phase2(regime == 'B'); out = e.finish()
        e.close()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if free0 is None:
        free0 = free
    print('iter %d free %.1f MB (delta %.1f MB) F=%.6e' % (it, free / 2**20, (free - free0) / 2**20, out['F']))
assert abs(free - free0) < 64 * 2**20, 'device memory is leaking'
print('LEAK_OK')
