"""Developer probe (GPU box): the forced tile-pair phase 2 over a sequence of shapes in ONE process (engines created and closed in turn), errors of
every output against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
shapes = [tuple(float(v) if '.' in v else int(v) for v in s.split(',')) for s in sys.argv[1:]] or [(300, 5, 20, 3, 0.5), (9000, 3, 200, 6, 0.3)]
for (N, D, M, Q, alpha) in shapes:
    d = Fz.synthetic_shard(N, D, M, Q, regime='B', seed=31, zseed=32, alpha_value=alpha)
    ref = Fz.evaluate_sharded(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], shards=8, pairs='gemm')
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    out = eng.evaluate(True)
    Zdev = eng.download('KMM')
    out2 = eng.evaluate(True)
    eng.close()
    errs = {k: float(np.max(np.abs(np.asarray(out[k]) - np.asarray(ref[k]))) / np.max(np.abs(ref[k]))) for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu', 'grad_X_S')}
    print((N, D, M, Q), 'F rel %.1e' % (abs(out['F'] - ref['F']) / abs(ref['F'])), {k: '%.1e' % v for k, v in errs.items()},
          'repeat identical', out['F'] == out2['F'] and np.array_equal(out['grad_Z'], out2['grad_Z']), flush=True)
