"""Dev check of the regime-B tile-pair phase 2 against the oracle on a few shapes (run on the GPU box)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz

def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(np.asarray(b))), 1e-300))

shapes = [(300, 5, 20, 3, 0.5), (500, 4, 2, 2, 0.7), (1000, 7, 130, 10, 0.3), (640, 3, 33, 13, 0.2), (400, 2, 70, 20, 0.1), (300, 2, 40, 30, 0.08),
          (200, 2, 24, 50, 0.05), (600, 3, 512, 10, 0.3), (257, 2, 1, 1, 1.0), (9000, 3, 200, 6, 0.3)]
if len(sys.argv) > 1:
    shapes = [tuple(float(x) if '.' in x else int(x) for x in sys.argv[1].split(','))]
for (N, D, M, Q, alpha) in shapes:
    d = Fz.synthetic_shard(N, D, M, Q, regime='B', seed=11, zseed=12, alpha_value=alpha)
    ref = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'])
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    out = eng.evaluate(True)
    tm = eng.timings()
    eng.close()
    print((N, D, M, Q), 'F %.1e' % (abs(out['F'] - ref['F']) / abs(ref['F'])), {k: '%.1e' % rel(out[k], ref[k]) for k in ('grad_Z', 'grad_alpha', 'grad_X_mu', 'grad_X_S')},
          'p2 kernel %.3f ms' % tm['p2_kernel_ms'])
    if rel(out['grad_Z'], ref['grad_Z']) > 1e-5:
        e = np.abs(out['grad_Z'] - ref['grad_Z']) / np.max(np.abs(ref['grad_Z']))
        bad = np.argwhere(e > 1e-5)
        print('   bad grad_Z entries: %d of %d; rows %s cols %s' % (len(bad), e.size, sorted(set(bad[:, 0]))[:20], sorted(set(bad[:, 1]))[:20]))
        print('   sample out/ref:', out['grad_Z'][bad[0][0], bad[0][1]], ref['grad_Z'][bad[0][0], bad[0][1]])
    if rel(out['grad_X_mu'], ref['grad_X_mu']) > 1e-5:
        e = np.abs(out['grad_X_mu'] - ref['grad_X_mu']) / np.max(np.abs(ref['grad_X_mu']))
        bad = np.argwhere(e > 1e-5)
        print('   bad grad_X_mu entries: %d of %d; points %s cols %s' % (len(bad), e.size, sorted(set(bad[:, 0]))[:12], sorted(set(bad[:, 1]))[:12]))
