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
