#!/usr/bin/env python3
"""Wall time of one Driver.likelihood_and_gradient call through the drop-in backend (gpu_MapReduce: files in, files out) against the device time of
the evaluation it wraps -- what a parallel_GPLVM.py run pays per optimiser step around the kernels.
usage (GPU box): python tests/devtools/dev_driver_timing.py [N [fixed(1/0) [fast(1/0)]]]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from gparml_amd import gpu_MapReduce  # noqa: E402
from gparml_amd.driver import Driver, transform_back  # noqa: E402,F401


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    fixed = (sys.argv[2] != '0') if len(sys.argv) > 2 else True
    fast = (sys.argv[3] != '0') if len(sys.argv) > 3 else True
    D, M, Q = 100, 512, 10
    d = bench.synthetic(N, D, M, Q, seed=100)
    work = tempfile.mkdtemp(prefix='drv_', dir='/tmp')
    dirs = {k: os.path.join(work, k) for k in ('input', 'embeddings', 'statistics', 'tmp')}
    for v in dirs.values():
        os.makedirs(v)
    t = time.time()
    np.savetxt(os.path.join(dirs['input'], 'shard_0'), d['Y'], delimiter=',', fmt='%.17g')
    print('wrote the CSV shard in %.1f s' % (time.time() - t))
    np.save(os.path.join(dirs['embeddings'], 'shard_0.embedding.npy'), d['X_mu'])
    np.save(os.path.join(dirs['embeddings'], 'shard_0.variance.npy'), np.zeros((N, Q)) if fixed else np.full((N, Q), -2.0))
    options = dict(input=dirs['input'], embeddings=dirs['embeddings'], statistics=dirs['statistics'], tmp=dirs['tmp'], parallel='local', keep=True, load=False,
                   M=M, Q=Q, D=D, N=N, fixed_embeddings=fixed, fixed_beta=False, drop_out_fraction=0)
    drv = Driver(options, gpu_MapReduce, fast=fast)
    gs = {'Z': d['Z'], 'sf2': np.array([[d['sf2']]]), 'alpha': np.asarray(d['alpha']).reshape(1, -1), 'beta': np.array([[d['beta']]])}
    x_t = drv.flatten_global_statistics(gs)
    # the optimiser's vector is the inverse softplus of the positive entries
    from gparml_amd.driver import transform_back as tb
    x = np.array([tb(b, v) for b, v in zip(options['flat_global_statistics_bounds'], x_t)])
    # where the host time goes: accumulate the backend's file traffic
    acc = {'save': 0.0, 'load': 0.0}
    _save, _load = gpu_MapReduce.save, gpu_MapReduce.load

    def save(name, obj):
        t0 = time.time(); _save(name, obj); acc['save'] += time.time() - t0

    def load(name):
        t0 = time.time(); r = _load(name); acc['load'] += time.time() - t0
        return r
    gpu_MapReduce.save, gpu_MapReduce.load = save, load
    times = []
    for it in range(6):
        t = time.time()
        f, g = drv.likelihood_and_gradient(x + 1e-4 * it, it)
        times.append(time.time() - t)
    if os.environ.get('DRV_PROFILE'):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for it in range(6, 11):
            drv.likelihood_and_gradient(x + 1e-4 * it, it)
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
    eng = list(gpu_MapReduce._shards.values())[0]['engine']
    print('N = %d fixed = %s fast = %s: first call %.1f ms (CSV parse + upload), then %s ms per call; device time of the evaluation %.2f ms; F = %.6e'
          % (N, fixed, fast, times[0] * 1e3, ['%.1f' % (t * 1e3) for t in times[1:]], eng.timings()['total_ms'], -f))
    print('   file traffic per call: np.save %.1f ms, np.load %.1f ms' % (1e3 * acc['save'] / 6, 1e3 * acc['load'] / 6))
    for k, v in drv.time_acc.items():
        if v:
            print('   %s: %.1f ms' % (k, 1e3 * float(np.mean(v[1:]))))
    shutil.rmtree(work)


if __name__ == '__main__':
    main()
