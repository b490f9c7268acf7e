#!/usr/bin/env python3
"""Golden vectors of the reference's test-time inference (predict.py:19-144), captured by running the reference in memory.
Build container only (needs /root/reference).

A small Bayesian GPLVM is trained by the reference's own pipeline (parallel_GPLVM.main, two shards, keep=True), then
predict.test(options, Y_test, is_random_init=True) optimises the latent mean and variance of three new points against the stored
accumulated statistics.  Every predict.likelihood_and_gradient call of that run is recorded (flat vector in, objective and gradient out)
together with what the callback reads: the trained global statistics, the five accumulated base statistics of iteration 'f', N, D and
Y_test.  Only numbers are stored (tests/golden/predict_gplvm_2shards.npz); tests/test_gpu_predict.py replays the calls on
gparml_amd/predict.py.
"""
import contextlib
import io
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_pipeline_golden as mpg  # noqa: E402


def main():
    mpg.MODS.append('predict')
    mods = mpg.load_reference()
    pg, lmr, pred = mods['parallel_GPLVM'], mods['local_MapReduce'], mods['predict']
    seed, shards, D, M, Q = 31, (34, 29), 3, 5, 2
    work = tempfile.mkdtemp(prefix='gparml_predgold_')
    dirs = {k: os.path.join(work, k) for k in ('input', 'embeddings', 'statistics', 'tmp')}
    for d in dirs.values():
        os.makedirs(d)
    rs = np.random.RandomState(seed)
    W = rs.randn(Q, D)
    for i, n in enumerate(shards):
        X = rs.randn(n, Q)
        np.savetxt(os.path.join(dirs['input'], 'shard_%d' % i), np.sin(X.dot(W)) + 0.1 * rs.randn(n, D), delimiter=',', fmt='%.17g')
    options = dict(input=dirs['input'], embeddings=dirs['embeddings'], statistics=dirs['statistics'], tmp=dirs['tmp'], parallel='local',
                   iterations=3, keep=True, load=False, init='PCA', optimiser='SCG_adapted', drop_out_fraction=0,
                   local_no_pool=False, M=M, Q=Q, D=D, fixed_embeddings=False, fixed_beta=False)
    np.random.seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        pg.main(options)
    trained = dict(pg.options)
    rec = {'D': np.int64(D), 'M': np.int64(M), 'Q': np.int64(Q), 'N': np.int64(trained['N'])}
    orig = pred.likelihood_and_gradient
    # run A: three new points, initial means from the nearest training outputs (predict.py:45-66); run B: one new point started from a random
    # inducing point with one restart (predict.py:38-41, 93-108 -- that branch only works for a single test point: X_mu is one row of Z)
    # round 5: what predict.test's INITIALISATION reads (predict.py:45-66: every shard's outputs and trained embeddings) is stored too, and a
    # run C with a ``mask`` (nearest training output over a subset of the output columns) -- appended AFTER runs A and B so that their random
    # streams, and therefore their recorded numbers, are what round 4 committed
    for i in range(len(shards)):
        name = 'shard_%d' % i
        rec['train_Y_%d' % i] = np.atleast_2d(np.genfromtxt(os.path.join(dirs['input'], name), delimiter=','))
        rec['train_X_%d' % i] = np.load(os.path.join(dirs['embeddings'], name + '.embedding.npy'))
    for tag, n_test, random_init, mask in (('A', 3, False, None), ('B', 1, True, None), ('C', 4, False, [0, 2])):
        Xt = rs.randn(n_test, Q)
        Y_test = np.sin(Xt.dot(W)) + 0.1 * rs.randn(n_test, D)
        rec[tag + '_Y_test'] = Y_test
        calls = []

        def wrapped(flat_array, iteration=0, step_size=0):
            f, g = orig(flat_array, iteration, step_size)
            k = len(calls)
            rec['%s_call%d_x' % (tag, k)] = np.array(flat_array, dtype=float)
            rec['%s_call%d_f' % (tag, k)] = np.array(f, dtype=float)
            rec['%s_call%d_g' % (tag, k)] = np.array(g, dtype=float)
            calls.append(k)
            return f, g

        pred.likelihood_and_gradient = wrapped
        np.random.seed(seed + 1)
        with contextlib.redirect_stdout(io.StringIO()):
            best = pred.test(trained, Y_test, mask=mask, is_random_init=random_init, random_iterations=3, random_restarts=1)
        if mask is not None:
            rec[tag + '_mask'] = np.asarray(mask, dtype=np.int64)
        rec[tag + '_n_calls'] = np.int64(len(calls))
        rec[tag + '_best_X_mu'], rec[tag + '_best_X_S'], rec[tag + '_best_likelihood'] = np.asarray(best[0]), np.asarray(best[1]), np.float64(best[2])
        print('predict run %s: %d calls, objective %s -> best likelihood %s' % (tag, len(calls), rec[tag + '_call0_f'], best[2]))
    for key in ('Z', 'sf2', 'alpha', 'beta'):
        rec['global_' + key] = np.asarray(pred.global_statistics[key], dtype=float)
    for key in ('sum_YYT', 'sum_exp_K_mi_K_im', 'sum_exp_K_miY', 'sum_exp_K_ii', 'sum_KL'):
        rec['acc_' + key] = np.asarray(pred.accumulated_statistics[key], dtype=float)
    path = os.path.join(HERE, 'predict_gplvm_2shards.npz')
    np.savez_compressed(path, **rec)
    shutil.rmtree(work)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    import multiprocessing
    multiprocessing.set_start_method('fork')
    main()
