"""The reference's second optimiser (gd.py, `--optimiser GD`) on the device-resident model: a 9-iteration run of
parallel_GPLVM.main with GD captured from the imported reference (tests/golden/gdpipe_*.npz: 15 evaluations, accepted and
rejected steps, step doubling / halving) is replayed with gparml_amd.gd.GD and the resident vector algebra
(gparml_amd.resident.ResidentGD = gd_local_MapReduce.py:14-105 through gp_cg_update / gp_cg_abs)."""
import os

import numpy as np
import pytest

from conftest import assert_close
from pipeline_util import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def test_resident_gd_reproduces_reference_run():
    from gparml_amd.gd import GD
    from gparml_amd.resident import ResidentGD, ResidentModel
    z = np.load(os.path.join(GOLDEN_DIR, 'gdpipe_gplvm_2shards.npz'))
    g = {k: z[k] for k in z.files}
    M, Q, D, N = int(g['M']), int(g['Q']), int(g['D']), int(g['N'])
    shards = [(g['Y_%d' % i], g['call0_in_shard%d_embedding' % i], g['call0_in_shard%d_variance' % i]) for i in range(int(g['n_shards']))]
    model = ResidentModel(shards, M, Q, D, fixed_embeddings=False)
    assert model.N == N
    calls = []

    def f_and_g(x, iteration, step_size=0):
        f, grad = model.likelihood_and_gradient(x, iteration, step_size)
        calls.append((np.array(x), f, grad, step_size))
        return f, grad

    ops = ResidentGD(model)
    x_opt, flog, _, status = GD(f_and_g, g['call0_x'].copy(), ops, fixed_embeddings=False, maxiters=9)
    f_and_g(x_opt, 'f')                                    # parallel_GPLVM.py:120
    ncalls = int(g['n_calls'])
    assert len(calls) == ncalls, (len(calls), ncalls)
    for k, (x, f, grad, step) in enumerate(calls):
        assert step == float(g['call%d_step' % k])
        assert_close(x, g['call%d_x' % k], 1e-7, atol=1e-12, what='call %d x' % k)
        assert_close(f, g['call%d_f' % k], 1e-6, what='call %d f' % k)
        assert_close(grad, g['call%d_g' % k], 2e-5, what='call %d grad' % k)
    # the reductions the optimiser prints / tests against gtol, against the reference's files of the last call
    last = ncalls - 1
    gl = np.concatenate([g['call%d_out_shard%d_grad_latest' % (last, i)].ravel() for i in range(int(g['n_shards']))])
    ops.embeddings_set_grads_update_grad_now(None)
    assert_close(ops.embeddings_get_grads_current_grad(None), np.sum(np.abs(gl)), 2e-5, what='sum |grad_now|')
    assert_close(ops.embeddings_get_grads_max_gradnow(None), np.max(np.abs(gl)), 2e-5, what='max |grad_now|')
    model.close()
