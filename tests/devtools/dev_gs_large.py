"""The replicated global step at M >= 1024 (round-4 review item 6): its time (library HIP events) and its results with each of round 5's changes switched
off and on -- xtx_tri (A^-1 = X^T X from the lower tiles only, k from the first non-zero row, mirrored), residual_dd (refinement residual through
ddacc_block), gemm_big (128 x 128-tile kernel for the M x M x {M, D} products) -- on one box, plus parity of the default against the oracle.
usage (through gpurun): python tests/devtools/dev_gs_large.py [M ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from gparml_amd import _lib
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz

OPTS = ('xtx_tri', 'residual_dd', 'gemm_big', 'trtri_rec', 'gs_i8')


def setopt(**kw):
    lib = _lib.load()
    for k in OPTS:
        rc = lib.gp_debug_set_option(k.encode(), int(kw.get(k, 1)))
        assert rc == 0, k


def run(eng, reps=6):
    best, out = 1e9, None
    for _ in range(reps):
        out = eng.evaluate(False)
        best = min(best, eng.timings()['global_ms'])
    return best, out


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b)))


for M in [int(x) for x in sys.argv[1:]] or [512, 1024, 2048]:
    N, D, Q = (20000, 1000, 50) if M >= 1024 else (20000, 100, 10)
    d = Fz.synthetic_shard(N, D, M, Q, regime='A', seed=3, zseed=4, alpha_value=3.0 / Q if M >= 1024 else 0.1)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    setopt()
    t_all, ref_dev = run(eng)
    print('M = %d, D = %d, Q = %d, N = %d' % (M, D, Q, N), flush=True)
    print('  all on (default)            global step %.3f ms' % t_all, flush=True)
    rows = [('all off', {o: 0 for o in OPTS})]
    rows += [('only %s' % k, dict({o: 0 for o in OPTS}, **{k: 1})) for k in OPTS]
    for name, kw in rows:
        setopt(**kw)
        t, out = run(eng)
        print('  %-26s  global step %.3f ms   F rel %.1e  grad_Z rel %.1e  grad_alpha rel %.1e (against the default)' % (
            name, t, abs(out['F'] - ref_dev['F']) / abs(ref_dev['F']), rel(out['grad_Z'], ref_dev['grad_Z']), rel(out['grad_alpha'], ref_dev['grad_alpha'])), flush=True)
    setopt()
    if M <= 1024:
        t0 = time.time()
        ref = Fz.evaluate_sharded(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], shards=16, workers=16, pairs='gemm')
        print('  default against the oracle (%.0f s): F %.1e  grad_Z %.1e  grad_alpha %.1e  grad_sf2 %.1e  grad_beta %.1e' % (
            time.time() - t0, abs(ref_dev['F'] - ref['F']) / abs(ref['F']), rel(ref_dev['grad_Z'], ref['grad_Z']), rel(ref_dev['grad_alpha'], ref['grad_alpha']),
            rel(ref_dev['grad_sf2'], ref['grad_sf2']), rel(ref_dev['grad_beta'], ref['grad_beta'])), flush=True)
    eng.close()
