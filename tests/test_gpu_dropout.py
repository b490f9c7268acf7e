"""Node drop-out (local_MapReduce.py:119-129, 263-264) and multi-shard reduction paths of the GPU backend (run with -m gpu):
the reference's branch in statistics_MR / statistics_reducer (compat mode), the same semantics in the two-phase fast mode, the
'every node dropped' fallback with its 1/(n+1) divisor, and the cross-device reduce path of gp_buffer_combine."""
import tempfile

import numpy as np
import pytest

from conftest import assert_close
from pipeline_util import call_args, load_pipeline, write_call_state

pytestmark = pytest.mark.gpu


def _seed_with_draw(want, fraction):
    """A numpy seed whose first uniform draw of len(want) gives exactly the drop pattern ``want`` (True = dropped)."""
    for seed in range(1000):
        if list(np.random.RandomState(seed).uniform(size=len(want)) < fraction) == list(want):
            return seed
    raise AssertionError('no seed found')


def _expected(g, kept, frac):
    """The reference's semantics with the oracle: every statistic summed over the kept shards, divided by frac; gradient sums likewise."""
    from oracle import factorised as Fz
    x, it, step = call_args(g, 0)
    M, Q, D, N = int(g['M']), int(g['Q']), int(g['D']), int(g['N'])
    xt = x.copy()
    xt[M * Q:] = np.log(1 + np.exp(x[M * Q:]))
    Z, sf2, alpha, beta = xt[:M * Q].reshape(M, Q), xt[M * Q], xt[M * Q + 1:M * Q + 1 + Q], xt[-1]
    st = None
    shards = [(g['Y_%d' % i], g['call0_in_shard%d_embedding' % i], g['call0_in_shard%d_variance' % i]) for i in kept]
    for (Y, mu, S) in shards:
        s = Fz.phase1(Z, sf2, alpha, Y, mu, S)
        st = s if st is None else {k: st[k] + s[k] for k in s}
    st = {k: v / frac for k, v in st.items()}
    gs = Fz.global_step(Z, sf2, alpha, beta, st, N, D)
    acc = None
    for (Y, mu, S) in shards:
        p2 = Fz.phase2(Z, sf2, alpha, Y, mu, S, gs['Abar'], gs['Bbar'], want_embeddings=False)
        p = dict(grad_Z_data=p2['grad_Z_data'], grad_alpha_data=p2['grad_alpha_data'])
        acc = p if acc is None else {k: acc[k] + p[k] for k in p}
    acc = {k: v / frac for k, v in acc.items()}
    out = Fz.finish(Z, sf2, alpha, gs, acc, True)
    grad = np.concatenate([out['grad_Z'].ravel(), [out['grad_sf2']], out['grad_alpha'], [out['grad_beta']]])
    grad[M * Q:] *= 1 / (np.exp(-x[M * Q:]) + 1)
    return -out['F'], -grad


@pytest.mark.parametrize('fast', [False, True])
@pytest.mark.parametrize('case', ['one_dropped', 'all_dropped'])
def test_drop_out_branch(case, fast):
    from gparml_amd import gpu_MapReduce
    from gparml_amd.driver import Driver
    g = load_pipeline('sparsegp_2shards')                      # fixed embeddings: statistics_MR is the whole evaluation
    gpu_MapReduce._reset()
    with tempfile.TemporaryDirectory() as work:
        options = write_call_state(g, 0, work)
        if case == 'one_dropped':
            options['drop_out_fraction'] = 0.5
            seed = _seed_with_draw([False, True], 0.5)
            kept, frac = [0], 0.5
        else:
            options['drop_out_fraction'] = 1.0                  # everything dropped: one random node kept, divisor 1/(n+1) = 1/3
            seed = 5
            rs = np.random.RandomState(seed)
            rs.uniform(size=2)
            kept, frac = [int(rs.randint(0, 2))], 1.0 / 3.0
        x, it, step = call_args(g, 0)
        np.random.seed(seed)
        f, grad = Driver(options, gpu_MapReduce, fast=fast).likelihood_and_gradient(x, it, step)
        assert list(gpu_MapReduce.non_dropped_out_nodes) == kept
        f_ref, grad_ref = _expected(g, kept, frac)
        assert_close(f, f_ref, 1e-6, what='f with drop-out')
        assert_close(grad, grad_ref, 1e-5, what='grad with drop-out')
    gpu_MapReduce._reset()


def test_cross_device_reduce_path():
    """gp_buffer_combine's peer-copy path (shards on different GPUs of one process, options['devices']) forced on one device: the
    fast driver gives the same numbers as with the same-device reduce; threads drive the shards as with several devices."""
    from gparml_amd import _lib, gpu_MapReduce
    from gparml_amd.driver import Driver
    g = load_pipeline('gplvm_2shards')
    lib = _lib.load()
    outs = []
    for staged in (0, 1):
        gpu_MapReduce._reset()
        lib.gp_debug_force_staging(staged)
        try:
            with tempfile.TemporaryDirectory() as work:
                options = write_call_state(g, 0, work)
                options['devices'] = [0, 0]
                x, it, step = call_args(g, 0)
                outs.append(Driver(options, gpu_MapReduce, fast=True).likelihood_and_gradient(x, it, step))
        finally:
            lib.gp_debug_force_staging(0)
    gpu_MapReduce._reset()
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])
    assert_close(outs[1][0], g['call0_f'], 1e-6, what='f')
    assert_close(outs[1][1], g['call0_g'], 1e-5, what='grad')
