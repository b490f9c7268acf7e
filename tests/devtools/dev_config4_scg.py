"""Dev soak (GPU box): the resident SCG loop at BASELINE configs[4]'s FULL per-GPU size -- N = 1e6, D = 1000, M = 1024, Q = 50, free embeddings, two
5e5-point shards on one device (scg_adapted.py; scg_adapted_local_MapReduce.py:29-243; parallel_GPLVM.py:222-369).  tests/test_gpu_config4_fullsize.py runs
one iteration; this runs several and prints the objective after every accepted step and the time per evaluation (~5 s each).
Usage: python tests/devtools/dev_config4_scg.py [iterations=4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from gparml_amd.driver import transform_back
from gparml_amd.resident import ResidentCG, ResidentModel
from gparml_amd.scg_adapted import SCG_adapted

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N, D, M, Q = 1000000, 1000, 1024, 50
t = time.time()
d = bench.synthetic_threaded(N, D, M, Q, seed=40, regime='B')
print('generated in %.1f s' % (time.time() - t), flush=True)
S_raw = np.log(np.expm1(d['X_S']))
h = N // 2
model = ResidentModel([(d['Y'][:h], d['X_mu'][:h], S_raw[:h]), (d['Y'][h:], d['X_mu'][h:], S_raw[h:])], M, Q, D, fixed_embeddings=False)
x0 = np.concatenate([d['Z'].ravel(), [float(d['sf2'])], np.asarray(d['alpha'], dtype=float), [float(d['beta'])]])
x0 = np.array([transform_back(b, v) for b, v in zip(model.bounds, x0)])
calls = []
def f_and_g(x, it, step=0):
    t = time.time(); f, g = model.likelihood_and_gradient(x, it, step); calls.append((float(f), time.time() - t))
    print('  evaluation %d (iteration %s): objective %.9e  %.2f s' % (len(calls), it, f, calls[-1][1]), flush=True)
    return f, g
t0 = time.time()
x, flog, nfe, status = SCG_adapted(f_and_g, x0, ResidentCG(model), fixed_embeddings=False, maxiters=iters, xtol=0, ftol=0, gtol=0)
fl = [float(f) for f in flog]
free, total = model.engines[0].memory_info()
print('SCG at configs[4] per-GPU size: %d iterations, %d evaluations in %.1f s (%.2f s each); objective %s; monotone=%s finite=%s; device memory in use %.1f GiB' % (
    iters, len(calls), time.time() - t0, (time.time() - t0) / len(calls), ' -> '.join('%.6e' % f for f in fl),
    all(b <= a + 1e-9 * abs(a) for a, b in zip(fl, fl[1:])), bool(np.all(np.isfinite(fl))), (total - free) / 2.0 ** 30))
model.close()
