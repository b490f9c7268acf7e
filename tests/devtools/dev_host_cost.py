"""Host cost of the calls of one evaluation at configs[1]'s size (N = 1e5, D = 10, M = 128, Q = 10): each call timed inside a back-to-back loop of evaluations
(steady state: with idle time in front of a call the device's wake-up dominates -- 180 us for a finish with nothing to wait for); the asynchronous calls show Python + ctypes +
the library's host code + the enqueue of their launches, finish shows the wait for whatever of the 0.32 ms of kernels is left.  usage (through gpurun): python tests/devtools/dev_host_cost.py"""
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np

import bench
from gparml_amd.engine import ShardEngine

N, D, M, Q = 100000, 10, 128, 10
d = bench.synthetic(N, D, M, Q, seed=11)
eng = ShardEngine(N, D, M, Q)
eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
eng.set_timing(0)
for _ in range(200):
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    eng.evaluate(False)
acc = {}
R = 300
for _ in range(R):
    for name, fn in (('set_globals', lambda: eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])), ('phase1', eng.phase1), ('global_step', lambda: eng.global_step(sync=False)),
                     ('phase2', lambda: eng.phase2(False)), ('finish (incl. waiting for 0.32 ms of kernels)', eng.finish)):
        t0 = time.perf_counter()
        fn()
        acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) / R
for k, v in acc.items():
    print('%-50s %7.1f us' % (k, v * 1e6))
t0 = time.perf_counter()
for _ in range(R):
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    eng.evaluate(False)
print('%-50s %7.1f us' % ('set_globals + evaluate, back to back', (time.perf_counter() - t0) / R * 1e6))
# finish with the device already idle: the cost of the synchronisation call and the read-back themselves
eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta']); eng.phase1(); eng.global_step(); eng.phase2(False)
time.sleep(0.01)
t0 = time.perf_counter(); eng.finish(); print('%-50s %7.1f us' % ('finish with everything already complete', (time.perf_counter() - t0) * 1e6))
eng.close()
