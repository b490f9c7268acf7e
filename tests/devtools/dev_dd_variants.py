"""Dev: time the global step (HIP events) for the variants of the double-double G = Ki Psi2 product and report grad_Z against the truth.
Usage: python tests/devtools/dev_dd_variants.py   (runs itself once per variant in a child process)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import numpy as np
    import bench
    from gparml_amd.engine import ShardEngine
    for (N, D, M, Q) in ((100000, 100, 512, 10), (20000, 100, 1024, 10), (100000, 10, 128, 10)):
        d = bench.synthetic(N, D, M, Q, seed=100)
        eng = ShardEngine(N, D, M, Q); eng.upload_shard(d['Y'], d['X_mu'], d['X_S']); eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
        g = []
        for _ in range(6):
            out = eng.evaluate(False); g.append(eng.timings()['global_ms'])
        msg = ''
        f = os.path.join(ROOT, 'tests', 'golden', 'hp_truth_large_N%d.npz' % N)
        if (D, M, Q) == (100, 512, 10) and os.path.exists(f):
            z = np.load(f); t = z['truth_grad_Z']
            msg = 'grad_Z err vs truth %.2e' % (np.max(np.abs(out['grad_Z'] - t)) / np.max(np.abs(t)))
        print('  %s M=%d: global step %.4f ms (min of 5)  %s' % (sys.argv[1], M, min(g[1:]), msg), flush=True)
        eng.close()
else:
    for env in ({'GPARML_DD_KIPSI2': '0'}, {'GPARML_DD_VARIANT': '0'}, {'GPARML_DD_VARIANT': '1'}, {'GPARML_DD_VARIANT': '2'}, {'GPARML_DD_VARIANT': '3'},
                {'GPARML_DD_VARIANT': '4'}, {'GPARML_DD_VARIANT': '5'}):
        subprocess.call([sys.executable, os.path.abspath(__file__), str(env)], env=dict(os.environ, **env))
