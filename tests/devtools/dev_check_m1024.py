"""Developer probe (GPU box): psi2_sym_kernel<8> at M = 1024 (eight waves per workgroup, new in round 6 through the compact rt rows) against the oracle on a
well-conditioned case; run once as is and once with GPARML_B_SYM_MAXQ=0 (column kernel)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
for (N, D, M, Q, alpha) in [(1100, 2, 1024, 8, 0.8), (1100, 2, 1024, 7, 0.8), (1100, 2, 960, 8, 0.8), (1100, 2, 1024, 6, 0.8), (1030, 16, 1024, 7, 0.29)]:
    d = Fz.synthetic_shard(N, D, M, Q, regime='B', seed=11, zseed=12, alpha_value=alpha)
    ref = Fz.evaluate_sharded(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], shards=8, pairs='gemm')
    dz = d['Z'][:, None, :] - d['Z'][None, :, :]
    Kmm = d['sf2'] * np.exp(-0.5 * np.sum(np.asarray(d['alpha']) * dz * dz, axis=2))
    cond = np.linalg.cond(Kmm + d['beta'] * np.asarray(ref['stats']['sum_exp_K_mi_K_im']))
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S']); eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    out = eng.evaluate(True)
    eng.close()
    keys = ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu', 'grad_X_S')
    errs = {k: float(np.max(np.abs(np.asarray(out[k]) - np.asarray(ref[k]))) / np.max(np.abs(ref[k]))) for k in keys}
    print((N, D, M, Q, alpha), 'cond %.1e' % cond, 'F %.1e' % (abs(out['F'] - ref['F']) / abs(ref['F'])), {k: '%.1e' % v for k, v in errs.items()}, flush=True)
