"""Developer probe (GPU box): where does the free-embedding evaluation at M = 1024 stop being positive definite -- conditioning or a wrapped index?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from test_gpu_index_range import _generate
from gparml_amd.engine import ShardEngine
for (N, Q, alpha) in [(int(a) for a in x.split(',')[:2]) + (float(x.split(',')[2]),) if False else (int(x.split(',')[0]), int(x.split(',')[1]), float(x.split(',')[2])) for x in sys.argv[1:]]:
    D, M = 8, 1024
    d = _generate(N, D, M, Q, 'B', seed=50 + Q)
    d['alpha'] = np.full(Q, alpha)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    eng.phase1()
    P2 = eng.download('PSI2_SUM')
    ev = np.linalg.eigvalsh(P2)
    print('N=%d Q=%d alpha=%.2f: Psi2 finite %s, symmetric %.1e, eig min %.3e max %.3e, diag min %.3e' % (N, Q, alpha, np.all(np.isfinite(P2)),
          np.max(np.abs(P2 - P2.T)), ev[0], ev[-1], P2.diagonal().min()), flush=True)
    try:
        eng.global_step(); eng.phase2(True); out = eng.finish(); print('  F', out['F'])
    except Exception as e:
        print('  FAILED:', str(e)[:150])
    eng.close()
