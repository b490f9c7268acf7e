"""Drop-in test of the MapReduce backend + driver on the GPU: every likelihood_and_gradient call of three reference
runs (2-shard GPLVM, 2-shard sparse GP with fixed embeddings, 1-shard config-1 sizes) is replayed from the captured
file state, in compat mode (the reference's own call sequence through statistics_MR / partial_terms / embeddings_MR,
12 statistics incl. the derivative 3-tensors) and in fast mode (two-phase device protocol)."""
import tempfile

import numpy as np
import pytest

from conftest import assert_close
from pipeline_util import call_args, load_pipeline, pipeline_names, write_call_state

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('fast', [False, True])
@pytest.mark.parametrize('name', pipeline_names())
def test_replay_reference_pipeline(name, fast):
    from gparml_amd import gpu_MapReduce
    from gparml_amd.driver import Driver
    g = load_pipeline(name)
    gpu_MapReduce._reset()
    with tempfile.TemporaryDirectory() as work:
        for k in range(int(g['n_calls'])):
            options = write_call_state(g, k, work)
            drv = Driver(options, gpu_MapReduce, fast=fast)
            x, it, step = call_args(g, k)
            f, grad = drv.likelihood_and_gradient(x, it, step)
            assert_close(f, g['call%d_f' % k], 1e-6, what='%s call %d f' % (name, k))
            assert_close(grad, g['call%d_g' % k], 1e-5, what='%s call %d grad' % (name, k))
            its = 'f' if it == 'f' else str(it)
            keys = options['accumulated_statistics_names'] if not fast else ['sum_YYT', 'sum_exp_K_mi_K_im', 'sum_exp_K_miY', 'sum_exp_K_ii', 'sum_KL']
            for key in keys:
                got = np.load(options['statistics'] + '/accumulated_statistics_%s_%s.npy' % (key, its))
                assert_close(got, g['call%d_acc_%s' % (k, key)], 1e-9, atol=1e-300, what='%s call %d %s' % (name, k, key))
            for key in options['partial_derivatives_names']:
                got = np.load(options['statistics'] + '/partial_derivatives_%s_%s.npy' % (key, its))
                assert_close(got, g['call%d_pd_%s' % (k, key)], 1e-5, what='%s call %d %s' % (name, k, key))
            if not options['fixed_embeddings']:
                for i in range(int(g['n_shards'])):
                    got = np.load(options['embeddings'] + '/shard_%d.grad_latest.npy' % i)
                    assert_close(got, g['call%d_out_shard%d_grad_latest' % (k, i)], 1e-5, what='grad_latest shard %d' % i)
    gpu_MapReduce._reset()


def test_fixed_embeddings_are_read_once_and_again_when_the_file_changes():
    """--fixed_embeddings: a resident shard keeps its device copy of the embeddings while the two files are the ones it was loaded from (size and
    modification time); a rewritten file is read again.  (The reference re-reads them in every map call, local_MapReduce.py:200-203; at N = 1e6 that
    was 25 of the 55 ms of a likelihood_and_gradient call.)"""
    import os
    from gparml_amd import gpu_MapReduce
    from gparml_amd.driver import Driver, transform_back
    from oracle import factorised as Fz
    N, D, M, Q = 700, 3, 20, 4
    d = Fz.synthetic_shard(N, D, M, Q, regime='A', seed=5, zseed=6, alpha_value=0.5)
    gpu_MapReduce._reset()
    with tempfile.TemporaryDirectory() as work:
        dirs = {k: os.path.join(work, k) for k in ('input', 'embeddings', 'statistics', 'tmp')}
        for v in dirs.values():
            os.makedirs(v)
        np.savetxt(os.path.join(dirs['input'], 'shard_0'), d['Y'], delimiter=',', fmt='%.17g')
        emb = os.path.join(dirs['embeddings'], 'shard_0.embedding.npy')
        np.save(emb, d['X_mu'])
        np.save(os.path.join(dirs['embeddings'], 'shard_0.variance.npy'), np.zeros((N, Q)))
        options = dict(input=dirs['input'], embeddings=dirs['embeddings'], statistics=dirs['statistics'], tmp=dirs['tmp'], parallel='local', keep=True,
                       load=False, M=M, Q=Q, D=D, N=N, fixed_embeddings=True, fixed_beta=False, drop_out_fraction=0)
        drv = Driver(options, gpu_MapReduce, fast=True)
        gs = {'Z': d['Z'], 'sf2': np.array([[d['sf2']]]), 'alpha': np.asarray(d['alpha']).reshape(1, -1), 'beta': np.array([[d['beta']]])}
        x = np.array([transform_back(b, v) for b, v in zip(options['flat_global_statistics_bounds'], drv.flatten_global_statistics(gs))])
        loads = []
        real_load = gpu_MapReduce.load
        gpu_MapReduce.load = lambda name: (loads.append(os.path.basename(name)), real_load(name))[1]
        try:
            f0, _ = drv.likelihood_and_gradient(x, 0)
            assert sorted(loads) == ['shard_0.embedding.npy', 'shard_0.variance.npy']
            f1, _ = drv.likelihood_and_gradient(x, 1)
            assert len(loads) == 2 and f1 == f0                       # nothing re-read, same resident shard, same bound
            X2 = d['X_mu'] + 0.05
            np.save(emb, X2)
            st = os.stat(emb)
            os.utime(emb, ns=(st.st_atime_ns, st.st_mtime_ns + 10_000_000))      # a later modification time whatever the file system's granularity
            f2, g2 = drv.likelihood_and_gradient(x, 2)
            assert len(loads) == 4
        finally:
            gpu_MapReduce.load = real_load
        ref = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], X2, np.zeros((N, Q)), want_embeddings=False)
        assert_close(-f2, ref['F'], 1e-6, what='bound at the rewritten embeddings')
        assert abs(f2 - f0) > 1e-6 * abs(f0)
    gpu_MapReduce._reset()
