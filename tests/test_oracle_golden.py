"""Pins oracle/literal.py against golden vectors captured from the imported reference
(tests/golden/make_golden.py).  Tolerance 1e-12 relative to the array's max magnitude: both sides
are float64 numpy doing the same arithmetic in a different association order."""
import numpy as np

from conftest import assert_close
from oracle import literal as L

RTOL = 1e-12


def _build(inp):
    pt = L.PartialTermsOracle(inp['Z'], inp['sf2'], inp['alpha'], inp['beta'], inp['M'], inp['Q'], inp['N'], inp['D'])
    pt.set_data(inp['Y'], inp['X_mu'], inp['X_S'], True)
    return pt


def test_kernel_pieces(golden):
    name, inp, out = golden
    assert_close(L.rbf_gram(inp['Z'], inp['sf2'], inp['alpha']), out['Kmm'], RTOL, what='Kmm')
    assert_close(L.psi1(inp['Z'], inp['sf2'], inp['alpha'], inp['X_mu'], inp['X_S']), out['exp_K_mi'], RTOL, what='psi1')
    for n in range(min(3, inp['X_mu'].shape[0])):
        assert_close(L.psi2_point(inp['Z'], inp['sf2'], inp['alpha'], inp['X_mu'][n], inp['X_S'][n]),
                     out['exp_K_mi_K_im'][n], RTOL, what='psi2[%d]' % n)
    assert_close(L.psi2_point_scalar(inp['Z'], inp['sf2'], inp['alpha'], inp['X_mu'][0], inp['X_S'][0]),
                 out['psi2_scalar_point0'], RTOL, what='psi2 scalar twin')
    assert_close(L.psi1_T_Y(inp['Z'], inp['sf2'], inp['alpha'], inp['X_mu'], inp['X_S'], inp['Y']),
                 out['exp_K_miY'], RTOL, what='psi1^T Y')


def test_statistics_and_bound(golden):
    name, inp, out = golden
    pt = _build(inp)
    st = pt.get_local_statistics()
    assert_close(st['sum_YYT'], out['sum_YYT'], RTOL, what='sum_YYT')
    assert_close(st['sum_exp_K_mi_K_im'], out['sum_exp_K_mi_K_im'], RTOL, what='Psi2')
    assert_close(st['exp_K_miY'], out['exp_K_miY'], RTOL, what='C')
    assert_close(st['sum_exp_K_ii'], out['sum_exp_K_ii'], RTOL, what='Psi0')
    assert_close(st['KL'], out['KL'], RTOL, what='KL')
    assert_close(pt.Kmm_inv, out['Kmm_inv'], 1e-10, what='Kmm_inv')
    assert_close(pt.Kmm_plus_op_inv, out['Kmm_plus_op_inv'], 1e-10, what='(Kmm+beta Psi2)^-1')
    assert_close(pt.logmarglik(), out['F'], RTOL, what='F')


def test_partials_and_gradients(golden):
    name, inp, out = golden
    pt = _build(inp)
    for meth in ('dF_dKmm', 'dF_dexp_K_miY', 'dF_dexp_K_mi_K_im', 'dF_dexp_K_ii',
                 'dKmm_dZ', 'dexp_K_miY_dZ', 'dexp_K_mi_K_im_dZ',
                 'dKmm_dalpha', 'dexp_K_miY_dalpha', 'dexp_K_mi_K_im_dalpha',
                 'dKmm_dsf2', 'dexp_K_miY_dsf2', 'dexp_K_mi_K_im_dsf2', 'dexp_K_ii_dsf2',
                 'grad_beta', 'grad_X_mu'):
        assert_close(getattr(pt, meth)(), out[meth], 1e-10, what=meth)
    if 'grad_X_S' in out:
        assert_close(pt.grad_X_S(), out['grad_X_S'], 1e-10, what='grad_X_S')
    gZ = pt.grad_Z(out['dF_dKmm'], out['dKmm_dZ'], out['dF_dexp_K_miY'], out['dexp_K_miY_dZ'],
                   out['dF_dexp_K_mi_K_im'], out['dexp_K_mi_K_im_dZ'])
    assert_close(gZ, out['grad_Z'], 1e-11, what='grad_Z')
    ga = pt.grad_alpha(out['dF_dKmm'], out['dKmm_dalpha'], out['dF_dexp_K_miY'], out['dexp_K_miY_dalpha'],
                       out['dF_dexp_K_mi_K_im'], out['dexp_K_mi_K_im_dalpha'])
    assert_close(ga, out['grad_alpha'], 1e-11, what='grad_alpha')
    gs = pt.grad_sf2(out['dF_dKmm'], out['dKmm_dsf2'], out['dF_dexp_K_ii'], out['dexp_K_ii_dsf2'],
                     out['dF_dexp_K_miY'], out['dexp_K_miY_dsf2'], out['dF_dexp_K_mi_K_im'], out['dexp_K_mi_K_im_dsf2'])
    assert_close(gs, out['grad_sf2'], 1e-11, what='grad_sf2')


def test_full_evaluation_helper(golden):
    name, inp, out = golden
    ev = L.full_evaluation(inp['Z'], inp['sf2'], inp['alpha'], inp['beta'], inp['Y'], inp['X_mu'], inp['X_S'], N_global=inp['N'])
    assert_close(ev['F'], out['F'], RTOL, what='F')
    for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu'):
        assert_close(ev[k], out[k], 1e-10, what=k)


def test_softplus_transforms():
    x = np.linspace(-10, 30, 13)
    y = L.transformVar(x)
    assert np.allclose(L.transformVar_back(y), x, rtol=1e-9, atol=1e-9)
    h = 1e-6
    assert np.allclose((L.transformVar(x + h) - L.transformVar(x - h)) / (2 * h), L.transformVar_grad(x), atol=1e-8)


def test_literal_oracle_replays_the_reference_predict_runs():
    """predict.likelihood_and_gradient (predict.py:116-144) as the reference ran it (tests/golden/make_predict_golden.py) on the literal
    restatement of partial_terms: the sequence tests/test_gpu_predict.py holds the GPU class to."""
    import os
    from conftest import GOLDEN_DIR
    from oracle import literal as L
    z = np.load(os.path.join(GOLDEN_DIR, 'predict_gplvm_2shards.npz'))
    M, Q, N, D = int(z['M']), int(z['Q']), int(z['N']), int(z['D'])
    acc = {k: z['acc_' + k] for k in ('sum_YYT', 'sum_exp_K_mi_K_im', 'sum_exp_K_miY', 'sum_exp_K_ii', 'sum_KL')}
    for tag in ('A', 'B'):
        Yt = z[tag + '_Y_test']
        n = Yt.shape[0] * Q
        for k in range(int(z[tag + '_n_calls'])):
            x = z['%s_call%d_x' % (tag, k)]
            Xm, Xs = x[:n].reshape(-1, Q), np.log1p(np.exp(x[n:])).reshape(-1, Q)
            o = L.PartialTermsOracle(z['global_Z'], float(z['global_sf2'].reshape(-1)[0]), z['global_alpha'].reshape(-1),
                                     float(z['global_beta'].reshape(-1)[0]), M, Q, N, D)
            o.set_data(Yt, Xm, Xs, True)
            new = o.get_local_statistics()
            o.set_local_statistics(acc['sum_YYT'] + new['sum_YYT'], acc['sum_exp_K_mi_K_im'] + new['sum_exp_K_mi_K_im'],
                                   acc['sum_exp_K_miY'] + new['exp_K_miY'], acc['sum_exp_K_ii'] + new['sum_exp_K_ii'], acc['sum_KL'] + new['KL'])
            f = -o.logmarglik()
            g = -np.concatenate((o.grad_X_mu().flatten(), o.grad_X_S().flatten() / (1.0 + np.exp(-x[n:]))))
            assert abs(f - float(z['%s_call%d_f' % (tag, k)])) <= 1e-9 * abs(f)
            gr = z['%s_call%d_g' % (tag, k)]
            assert np.max(np.abs(g - gr)) <= 1e-8 * np.max(np.abs(gr))


def test_mp_truth_fixture_documents_the_float64_oracles_limit():
    """tests/golden/mp_truth_N17_M129.npz (60-digit mpmath, make_mp_truth_small.py): on N = 17 points under M = 129 random inducing points the float64
    oracle is ~5e-3 from the truth on grad_Z (cond(K_mm) 2.8e7, cond(K_mm + beta Psi2) 1.8e9: K_mm^-1 Psi2 accumulated in float64).  The device test
    (tests/test_gpu_hp_truth.py) holds the library to 1e-5 there; this one pins the oracle's own number so that nobody trusts it on such a case."""
    import os
    from conftest import GOLDEN_DIR
    from oracle import factorised as Fz
    z = np.load(os.path.join(GOLDEN_DIR, 'mp_truth_N17_M129.npz'))
    N, Q = z['X_mu'].shape
    ref = Fz.evaluate(z['Z'], float(z['sf2']), z['alpha'], float(z['beta']), z['Y'], z['X_mu'], np.zeros((N, Q)), want_embeddings=False)
    t = z['truth_grad_Z']
    err = float(np.max(np.abs(ref['grad_Z'] - t)) / np.max(np.abs(t)))
    assert 1e-4 < err < 1e-1 and float(z['cond_A']) > 1e9


def test_predict_test_host_logic_reproduces_the_reference_runs_on_the_literal_oracle():
    """gparml_amd/predict.py's restatement of predict.test (predict.py:19-111: nearest-training-output start with ``mask``, random inducing point,
    restarts that keep the best likelihood) with the GPU class swapped for the literal CPU restatement: the host logic alone against the
    reference's recorded runs (the GPU twin is tests/test_gpu_predict.py)."""
    from gparml_amd.predict import Predictor
    from oracle import literal as L
    from test_gpu_predict import run_reference_protocol
    run_reference_protocol(lambda gs, acc, N, D: Predictor(gs, acc, N, D, partial_terms_class=L.PartialTermsOracle), 1e-8, 1e-9, 1e-8)
