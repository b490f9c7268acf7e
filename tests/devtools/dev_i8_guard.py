"""Developer probe (GPU box): the int8 guard's numbers (rel_psi2, rel_c, cond_lower_bound) next to grad_Z's distance from the 80-bit truth, at
the benchmark workload (N = 1e5, 1e6) and on the int8 test shapes; run once with the product library and once with a GP_I8_DIGITS=5 build
(GPARML_LIB) to see both sides of the threshold."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from gparml_amd import _lib
from gparml_amd.engine import ShardEngine
lib = _lib.load()
lib.gp_debug_set_option(b'p1_i8', 1)
for N in (100000, 1000000):
    D, M, Q = 100, 512, 10
    d = bench.synthetic(N, D, M, Q, seed=100)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    o_check = eng.evaluate(False)              # the checked evaluation: float64 statistics
    st = eng.i8_status()
    lib.gp_debug_set_option(b'p1_i8', 1)
    o_i8 = eng.evaluate(False)                 # int8 statistics if accepted
    st2 = eng.i8_status()
    te = bench.truth_errors(d, o_i8, N, D, M, Q, 100)
    tc = bench.truth_errors(d, o_check, N, D, M, Q, 100)
    print('N=%d: status %s -> score %.3e; grad_Z vs truth: checked evaluation (float64 stats) %.2e, next evaluation %.2e; F differs %s' % (
        N, st, st['cond_lower_bound'] * max(st['rel_psi2'], st['rel_c']), tc['grad_Z_err_vs_truth'] if tc else -1,
        te['grad_Z_err_vs_truth'] if te else -1, o_i8['F'] != o_check['F']), st2['state'], flush=True)
    eng.close()
lib.gp_debug_set_option(b'p1_i8', 0)
