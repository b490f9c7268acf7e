"""The resident SCG loop at BASELINE configs[4]'s per-GPU SHAPE (D=1000, M=1024, Q=50, free embeddings; reference: scg_adapted.py,
scg_adapted_local_MapReduce.py:29-243, parallel_GPLVM.py:226-369).

The reference's own optimisation runs replayed in test_gpu_resident_scg.py are small (the reference stores an (N, M, M) tensor); the
evaluation at this shape is checked against the oracle in test_gpu_tile_phase2.py.  This test closes the gap between the two: a few SCG
iterations with everything resident (two shards on one device, 2 x 1024 points x 50 latent dimensions of embeddings, variances, search
directions and gradient vectors in HBM, the matrix-core tile kernels of psi2_tile.hip in charge of both pairwise phases), then the state the
optimiser LEFT on the device -- embeddings and variances after the resident axpy updates, hyper-parameters through the softplus transforms
-- is downloaded and evaluated by the oracle: bound and gradient must agree with what the device reports for that state."""
import os

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


def test_scg_iterations_at_config4_shape_leave_a_state_the_oracle_agrees_with():
    from gparml_amd.driver import transform_back, transform_grad_vec, transform_vec
    from gparml_amd.resident import ResidentCG, ResidentModel
    from gparml_amd.scg_adapted import SCG_adapted
    from oracle import factorised as Fz
    N, D, M, Q = 2048, 1000, 1024, 50
    rs = np.random.RandomState(3)
    d = Fz.synthetic_shard(N, D, 64, Q, regime='B', seed=16, zseed=17, alpha_value=1.0 / Q)
    Z0 = d['X_mu'][rs.permutation(N)[:M]] + 0.3 * rs.randn(M, Q)
    S_raw = np.log(np.expm1(d['X_S']))                                    # the optimiser's unconstrained variances (softplus inverse)
    h = N // 2
    shards = [(d['Y'][:h], d['X_mu'][:h], S_raw[:h]), (d['Y'][h:], d['X_mu'][h:], S_raw[h:])]
    model = ResidentModel(shards, M, Q, D, fixed_embeddings=False)
    try:
        x0 = np.concatenate([Z0.ravel(), [float(d['sf2'])], np.asarray(d['alpha'], dtype=float), [float(d['beta'])]])
        x0 = np.array([transform_back(b, v) for b, v in zip(model.bounds, x0)])
        flog_calls = []

        def f_and_g(x, iteration, step_size=0):
            f, g = model.likelihood_and_gradient(x, iteration, step_size)
            flog_calls.append(float(f))
            return f, g

        x, flog, nfe, status = SCG_adapted(f_and_g, x0, ResidentCG(model), fixed_embeddings=False, maxiters=3, xtol=0, ftol=0, gtol=0)
        fl = [float(v) for v in flog]
        assert np.all(np.isfinite(fl)) and np.all(np.isfinite(flog_calls))
        assert all(b <= a + 1e-9 * abs(a) for a, b in zip(fl, fl[1:])), fl   # accepted steps never increase the objective (-F)
        assert fl[-1] < fl[0]                                                  # and the embeddings / hyper-parameters did move
        f_dev, g_dev = model.likelihood_and_gradient(x, 'f', 0)               # parallel_GPLVM.py:120: the final evaluation at step 0
        mu = np.concatenate([e.download('X_MU_TRIAL') for e in model.engines])
        S = np.concatenate([e.download('X_S_TRIAL') for e in model.engines])
        gmu_dev = np.concatenate([e.download('GRAD_X_MU') for e in model.engines])
        gS_dev = np.concatenate([e.download('GRAD_X_S') for e in model.engines])
    finally:
        model.close()
    assert np.max(np.abs(mu - d['X_mu'])) > 1e-6 and np.all(S > 0)            # the resident updates reached the embeddings
    xt = transform_vec(model._pos, x)
    Z, sf2, alpha, beta = xt[:M * Q].reshape(M, Q), xt[M * Q], xt[M * Q + 1:M * Q + 1 + Q], xt[M * Q + 1 + Q]
    ref = Fz.evaluate_sharded(Z, sf2, alpha, beta, d['Y'], mu, S, shards=32, workers=min(32, os.cpu_count() or 8), pairs='gemm')
    assert_close(-f_dev, ref['F'], 1e-6, what='F after the optimiser steps')
    g_ref = np.concatenate([ref['grad_Z'].ravel(), [ref['grad_sf2']], ref['grad_alpha'], [ref['grad_beta']]]) * transform_grad_vec(model._pos, x)
    assert_close(-g_dev, g_ref, 1e-5, what='transformed hyper-parameter gradient')
    assert_close(gmu_dev, ref['grad_X_mu'], 1e-5, what='grad_X_mu')
    assert_close(gS_dev, ref['grad_X_S'], 1e-5, what='grad_X_S')
