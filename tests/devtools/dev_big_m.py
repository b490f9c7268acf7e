import sys, os
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
for (N, D, M, Q, regime, al) in ((3000, 20, 1500, 8, 'A', 1.0), (2500, 100, 2048, 10, 'A', 1.0), (2000, 10, 1537, 6, 'B', 1.5), (2000, 7, 1100, 20, 'B', 0.4)):
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=3, zseed=4, alpha_value=al)
    try:
        ref = Fz.evaluate_sharded(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], shards=32, workers=32, pairs='gemm')
    except np.linalg.LinAlgError:
        print((N, D, M, Q, regime), 'oracle not PD'); continue
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S']); eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    out = eng.evaluate(True); eng.close()
    errs = {k: float(np.max(np.abs(np.asarray(out[k]) - np.asarray(ref[k]))) / np.max(np.abs(ref[k]))) for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu')}
    errs['F'] = abs(out['F'] - ref['F']) / abs(ref['F'])
    print((N, D, M, Q, regime), {k: '%.1e' % v for k, v in errs.items()}, flush=True)
