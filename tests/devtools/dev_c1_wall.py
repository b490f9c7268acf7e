"""Why does configs[1]'s wall clock per evaluation read 1-2 ms inside bench.py and 0.37 ms in a fresh process?  Runs bench.config1_extra between the stages of
bench.main() and prints the wall time after each.  usage (through gpurun): python tests/devtools/dev_c1_wall.py"""
import os
import sys
import threading
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np

import bench
from gparml_amd.engine import ShardEngine


def c1(tag):
    r = bench.config1_extra(0, steps=50)
    print('%-60s wall %.3f ms  device %.3f ms  (threads in the process: %d)' % (tag, r['ms_per_eval_wall'], r['device_ms'], threading.active_count()), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == 'torch':          # as bench.main does: torch first (the library then shares torch's HIP runtime)
    import torch
    torch.cuda.set_device(0)
    x = torch.zeros(8, device='cuda:0')
    torch.cuda.synchronize()
c1('first thing in the process' + (' after torch initialised the device' if len(sys.argv) > 1 else ''))
if len(sys.argv) > 2 and sys.argv[2] == 'pg':
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
    dist.init_process_group('nccl', rank=0, world_size=1)
    c1('after init_process_group(nccl, world 1)')
    t = torch.zeros(4, device='cuda:0'); dist.all_reduce(t); torch.cuda.synchronize()
    c1('after one all_reduce')
N, D, M, Q = 1000000, 100, 512, 10
d = bench.synthetic(N, D, M, Q, seed=100)
c1('after synthetic(1e6 x 100)')
c1('again')
time.sleep(3.0)
c1('after 3 s of sleep')
import ctypes, gc
gc.collect()
c1('after gc.collect')
d2 = bench.synthetic(100000, 10, 128, 10, seed=11)
c1('after a small synthetic')
eng = ShardEngine(N, D, M, Q, device=0)
eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'], N_global=N)
for _ in range(5):
    out = eng.evaluate(False)
c1('after five headline evaluations (engine still open)')
res = bench.extended_precision_cost(eng, d, N, D, M, Q, 100)
c1('after extended_precision_cost')
te = bench.truth_errors(d, out, N, D, M, Q, 100)
c1('after truth_errors')
iv = bench.int8_variants(eng, d, N, D, M, Q, 100)
c1('after int8_variants')
eng.close()
c1('after closing the headline engine')
