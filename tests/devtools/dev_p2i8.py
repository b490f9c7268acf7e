"""Developer probe (GPU box): int8 phase 2 (csrc/p2i8.hip) against the float64 library and the 80-bit truth; timings of the three configurations."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from gparml_amd import _lib
from gparml_amd.engine import ShardEngine
lib = _lib.load()
sizes = [int(x) for x in sys.argv[1:]] or [100000]
keys = ('F', 'grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta')
for N in sizes:
    D, M, Q = 100, 512, 10
    d = bench.synthetic(N, D, M, Q, seed=100)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    res, tms = {}, {}
    for name, p1, p2 in (('float64', 0, 0), ('int8 phase 1', 1, 0), ('int8 phase 1 + 2', 1, 1)):
        lib.gp_debug_set_option(b'p1_i8', p1); lib.gp_debug_set_option(b'p2_i8', p2)
        eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
        for rep in range(4):
            out = eng.evaluate(False)
        res[name] = out; tms[name] = eng.timings()
        te = bench.truth_errors(d, out, N, D, M, Q, 100)
        print('N=%d %-18s F %.12e  vs float64: %s  vs truth grad_Z %s  status %s' % (
            N, name, out['F'], {k: '%.1e' % (np.max(np.abs(np.asarray(out[k]) - np.asarray(res['float64'][k]))) / np.max(np.abs(np.asarray(res['float64'][k])))) for k in keys},
            ('%.2e' % te['grad_Z_err_vs_truth']) if te else 'n/a', eng.i8_status()['state']), flush=True)
        print('      device ms:', {k: round(v, 3) for k, v in tms[name].items()}, flush=True)
    eng.close()
lib.gp_debug_set_option(b'p1_i8', 0); lib.gp_debug_set_option(b'p2_i8', 0)
