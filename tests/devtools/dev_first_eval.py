"""Developer probe (GPU box): the FIRST evaluation of a fresh process at M = 1024 (round 6: one such evaluation came back with F 2.9e-8 and grad_Z 4.1e-6 off --
the signature of the 1e-7 jitter -- where the same inputs give 5e-15 / 5e-11).  Prints F and grad_Z errors against a stored reference, the jitter mask and a repeat."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
N, D, M, Q, alpha = 1100, 2, 1024, 8, 0.8
d = Fz.synthetic_shard(N, D, M, Q, regime='B', seed=11, zseed=12, alpha_value=alpha)
path = '/tmp/first_eval_ref.npz'
if not os.path.exists(path):
    ref = Fz.evaluate_sharded(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], shards=8, pairs='gemm')
    np.savez(path, F=ref['F'], grad_Z=ref['grad_Z'])
z = np.load(path)
eng = ShardEngine(N, D, M, Q)
eng.upload_shard(d['Y'], d['X_mu'], d['X_S']); eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
res = []
for rep in range(3):
    out = eng.evaluate(True)
    res.append((abs(out['F'] - float(z['F'])) / abs(float(z['F'])), float(np.max(np.abs(out['grad_Z'] - z['grad_Z'])) / np.max(np.abs(z['grad_Z']))), eng.last_jitter))
eng.close()
print('evaluations 1..3: ' + ' | '.join('F %.1e gZ %.1e jitter %d' % r for r in res), flush=True)
