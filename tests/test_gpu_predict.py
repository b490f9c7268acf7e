"""predict.py's callback (predict.py:116-144) on the GPU against the oracle restatement of the same call sequence."""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


def test_predict_callback_matches_oracle():
    from gparml_amd.predict import Predictor
    from oracle import factorised as Fz
    from oracle import literal as L
    d = Fz.synthetic_shard(80, 4, 9, 3, regime='B', seed=31, zseed=32, alpha_value=0.6)
    M, Q, D, N = 9, 3, 4, 80
    train = L.PartialTermsOracle(d['Z'], d['sf2'], d['alpha'], d['beta'], M, Q, N, D)
    train.set_data(d['Y'], d['X_mu'], d['X_S'], True)
    st = train.get_local_statistics()
    acc = dict(sum_YYT=st['sum_YYT'], sum_exp_K_mi_K_im=st['sum_exp_K_mi_K_im'], sum_exp_K_miY=st['exp_K_miY'],
               sum_exp_K_ii=st['sum_exp_K_ii'], sum_KL=st['KL'])
    rs = np.random.RandomState(5)
    Yt = rs.randn(3, D)
    Xm, Xs = rs.randn(3, Q), rs.uniform(0.2, 0.8, size=(3, Q))
    gs = dict(Z=d['Z'], sf2=d['sf2'], alpha=d['alpha'], beta=d['beta'])
    p = Predictor(gs, acc, N, D)
    p.Y_test, p.shape = Yt, Xm.shape
    p.bounds = [(None, None)] * Xm.size + [(0, None)] * Xm.size
    x = np.concatenate((Xm.flatten(), np.log(np.exp(Xs.flatten()) - 1.0)))
    f, g = p.likelihood_and_gradient(x)
    # oracle: the same sequence with the literal restatement of partial_terms
    o = L.PartialTermsOracle(d['Z'], d['sf2'], d['alpha'], d['beta'], M, Q, N, D)
    o.set_data(Yt, Xm, Xs, True)
    new = o.get_local_statistics()
    o.set_local_statistics(acc['sum_YYT'] + new['sum_YYT'], acc['sum_exp_K_mi_K_im'] + new['sum_exp_K_mi_K_im'],
                           acc['sum_exp_K_miY'] + new['exp_K_miY'], acc['sum_exp_K_ii'] + new['sum_exp_K_ii'], acc['sum_KL'] + new['KL'])
    fo = -o.logmarglik()
    go = -np.concatenate((o.grad_X_mu().flatten(), o.grad_X_S().flatten() * (1.0 / (1.0 + np.exp(-x[Xm.size:])))))
    assert_close(f, fo, 1e-6, what='predict f')
    assert_close(g, go, 1e-5, what='predict grad')
    # and a short optimisation decreases the objective
    res = p.test(Yt, Xm, Xs, iterations=3)
    assert -res[2] <= f + 1e-9


def _replay_reference_fixture(make_callback):
    """The reference's own predict.test runs (tests/golden/make_predict_golden.py: a model trained by parallel_GPLVM.main, then
    predict.likelihood_and_gradient called by its SCG; run A = three new points from nearest-neighbour means, run B = one point from a random
    inducing point with a restart): every recorded call is replayed -- same flat vector in, objective 1e-6 and gradient 1e-5 out."""
    import os
    from conftest import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, 'predict_gplvm_2shards.npz'))
    gs = dict(Z=z['global_Z'], sf2=z['global_sf2'], alpha=z['global_alpha'], beta=z['global_beta'])
    acc = {k: z['acc_' + k] for k in ('sum_YYT', 'sum_exp_K_mi_K_im', 'sum_exp_K_miY', 'sum_exp_K_ii', 'sum_KL')}
    Q = int(z['Q'])
    n = 0
    for tag in ('A', 'B'):
        Yt = z[tag + '_Y_test']
        cb = make_callback(gs, acc, int(z['N']), int(z['D']), Yt, (Yt.shape[0], Q))
        for k in range(int(z[tag + '_n_calls'])):
            f, g = cb(z['%s_call%d_x' % (tag, k)])
            assert_close(f, z['%s_call%d_f' % (tag, k)], 1e-6, what='run %s call %d: objective' % (tag, k))
            assert_close(g, z['%s_call%d_g' % (tag, k)], 1e-5, what='run %s call %d: gradient' % (tag, k))
            n += 1
    assert n == 21


def test_predict_callback_replays_the_reference_runs():
    from gparml_amd.predict import Predictor

    def make(gs, acc, N, D, Yt, shape):
        p = Predictor(gs, acc, N, D)
        p.Y_test, p.shape = Yt, shape
        p.bounds = [(None, None)] * (shape[0] * shape[1]) + [(0, None)] * (shape[0] * shape[1])
        return p.likelihood_and_gradient

    _replay_reference_fixture(make)


def run_reference_protocol(make_predictor, xtol, ftol, gtol):
    """predict.test end to end as the reference ran it (tests/golden/make_predict_golden.py): the same global numpy seed, the training shards and
    trained embeddings the reference's initialisation read, then -- run A: nearest-training-output start (predict.py:44-66), run B: a random
    inducing point and one restart keeping the best likelihood (predict.py:38-41, 93-108), run C: nearest output over a column ``mask``.
    Every evaluation the restated driver makes must be the one the reference made (same flat vector, objective, gradient), the number of
    evaluations must agree, and so must the returned [X_mu, X_S, likelihood]."""
    import os
    from conftest import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, 'predict_gplvm_2shards.npz'))
    gs = dict(Z=z['global_Z'], sf2=z['global_sf2'], alpha=z['global_alpha'], beta=z['global_beta'])
    acc = {k: z['acc_' + k] for k in ('sum_YYT', 'sum_exp_K_mi_K_im', 'sum_exp_K_miY', 'sum_exp_K_ii', 'sum_KL')}
    training = [(z['train_Y_%d' % i], z['train_X_%d' % i]) for i in range(2)]
    for tag, kw in (('A', dict(training=training)), ('B', dict(is_random_init=True, random_restarts=1)),
                    ('C', dict(training=training, mask=[int(v) for v in z['C_mask']]))):
        p = make_predictor(gs, acc, int(z['N']), int(z['D']))
        calls = []
        inner = p.likelihood_and_gradient

        def recorded(x, iteration=0, step_size=0, inner=inner, calls=calls):
            f, g = inner(x, iteration, step_size)
            calls.append((np.array(x, dtype=float), float(f), np.array(g, dtype=float)))
            return f, g

        p.likelihood_and_gradient = recorded
        np.random.seed(32)                                             # make_predict_golden.py: numpy.random.seed(seed + 1) before predict.test
        best = p.test(z[tag + '_Y_test'], iterations=3, **kw)
        assert len(calls) == int(z[tag + '_n_calls']), (tag, len(calls))
        # the starting point is host arithmetic only (k-d tree / random inducing point, clipped variances, softplus inverse): to rounding
        assert_close(calls[0][0], z[tag + '_call0_x'], 1e-13, what='run %s: x0' % tag)
        for k, (x, f, g) in enumerate(calls):
            assert_close(x, z['%s_call%d_x' % (tag, k)], xtol, what='run %s call %d: x' % (tag, k))
            assert_close(f, z['%s_call%d_f' % (tag, k)], ftol, what='run %s call %d: objective' % (tag, k))
            assert_close(g, z['%s_call%d_g' % (tag, k)], gtol, what='run %s call %d: gradient' % (tag, k))
        assert_close(best[0], z[tag + '_best_X_mu'], xtol, what='run %s: best X_mu' % tag)
        assert_close(best[1], z[tag + '_best_X_S'], xtol, what='run %s: best X_S' % tag)
        assert_close(best[2], z[tag + '_best_likelihood'], ftol, what='run %s: best likelihood' % tag)


def test_predict_test_initialisation_and_restarts_reproduce_the_reference_runs():
    from gparml_amd.predict import Predictor
    run_reference_protocol(lambda gs, acc, N, D: Predictor(gs, acc, N, D), 1e-6, 1e-6, 1e-5)
