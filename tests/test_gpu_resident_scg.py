"""End-to-end drop-in evidence on the GPU: the reference's own optimisation runs (parallel_GPLVM.main with SCG_adapted,
captured in tests/golden/pipe_*.npz) are re-run with every shard, embedding, search direction and gradient vector
resident in HBM (gparml_amd.resident) and the optimiser loop of gparml_amd.scg_adapted.  Every likelihood_and_gradient
call must see the same x and return the same f and gradient as the reference did -- which also checks the trial-point
protocol (X + step*d), the resident CG dot products / axpy updates against the reference's file-based helpers
(scg_adapted_local_MapReduce.py), and the final 'f' evaluation of parallel_GPLVM.py:115-123."""
import numpy as np
import pytest

from conftest import assert_close
from pipeline_util import load_pipeline, pipeline_names

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', pipeline_names())
def test_resident_scg_reproduces_reference_run(name):
    from gparml_amd.resident import ResidentCG, ResidentModel
    from gparml_amd.scg_adapted import SCG_adapted
    g = load_pipeline(name)
    M, Q, D, N = int(g['M']), int(g['Q']), int(g['D']), int(g['N'])
    fixed = bool(g['fixed'])
    shards = [(g['Y_%d' % i], g['call0_in_shard%d_embedding' % i], g['call0_in_shard%d_variance' % i]) for i in range(int(g['n_shards']))]
    model = ResidentModel(shards, M, Q, D, fixed_embeddings=fixed)
    assert model.N == N
    calls = []

    def f_and_g(x, iteration, step_size=0):
        f, grad = model.likelihood_and_gradient(x, iteration, step_size)
        calls.append((np.array(x), f, grad))
        return f, grad

    ncalls = int(g['n_calls'])
    iters = 2
    x_opt, flog, nfe, status = SCG_adapted(f_and_g, g['call0_x'].copy(), ResidentCG(model), fixed_embeddings=fixed, maxiters=iters,
                                           xtol=0, ftol=0, gtol=0)
    f_and_g(x_opt, 'f')                                    # parallel_GPLVM.py:120
    assert len(calls) == ncalls, (len(calls), ncalls)
    for k, (x, f, grad) in enumerate(calls):
        assert_close(x, g['call%d_x' % k], 1e-7, atol=1e-12, what='%s call %d x' % (name, k))
        assert_close(f, g['call%d_f' % k], 1e-6, what='%s call %d f' % (name, k))
        assert_close(grad, g['call%d_g' % k], 2e-5, what='%s call %d grad' % (name, k))
    model.close()
