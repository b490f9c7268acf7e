"""Developer probe (GPU box): what does a SPURIOUS jitter retry do to round 5's one failing shape (tests/test_gpu_tile_phase2.py, (9000, 3, 200, 6), seed 31 / 32:
grad_Z 9.97e-5 off the oracle, once)?  Evaluates the shape normally and with the reference's 1e-7 jitter forced on K_mm, on K_mm + beta Psi2 and on both (the masks
gp_global_status reports), and prints every block's deviation from the (jitter-free) oracle."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
N, D, M, Q, alpha = 9000, 3, 200, 6, 0.3
d = Fz.synthetic_shard(N, D, M, Q, regime='B', seed=31, zseed=32, alpha_value=alpha)
ref = Fz.evaluate_sharded(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], shards=8, pairs='gemm')
eng = ShardEngine(N, D, M, Q)
eng.upload_shard(d['Y'], d['X_mu'], d['X_S']); eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
keys = ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta')
for mask in (0, 1, 2, 3):
    eng.phase1()
    eng.global_step(sync=False, jitter=mask)
    eng.phase2(True)
    out = eng.finish()
    errs = {k: float(np.max(np.abs(np.asarray(out[k]) - np.asarray(ref[k]))) / np.max(np.abs(ref[k]))) for k in keys}
    print('jitter mask %d: F %.2e  ' % (mask, abs(out['F'] - ref['F']) / abs(ref['F'])) + '  '.join('%s %.3e' % (k, errs[k]) for k in keys), flush=True)
eng.close()
