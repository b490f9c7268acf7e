"""The reference's own tests (test.py), restated against gparml_amd.partial_terms.partial_terms on the GPU, plus the
method-by-method golden comparison.  test.py's GPy halves are unavailable here; the finite-difference
tests test_dF_dZ (:62-93), test_dF_dbeta first half (:186-201), test_mu (:270-282), test_S (:284-296) and the GPy-free
parts of test_dF_dalpha / test_dF_dsf2 are restated with the same step sizes and thresholds."""
import numpy as np
import pytest

from conftest import assert_close, golden_names, load_golden

pytestmark = pytest.mark.gpu


def _pt(inp):
    from gparml_amd.partial_terms import partial_terms
    pt = partial_terms(inp['Z'], inp['sf2'], inp['alpha'], inp['beta'], inp['M'], inp['Q'], inp['N'], inp['D'])
    pt.set_data(inp['Y'], inp['X_mu'], inp['X_S'], is_set_statistics=True)
    return pt


@pytest.mark.parametrize('name', golden_names())
def test_every_method_against_golden(name):
    inp, ref = load_golden(name)
    pt = _pt(inp)
    assert_close(pt.Kmm, ref['Kmm'], 1e-12, what='Kmm')
    assert_close(pt.Kmm_inv, ref['Kmm_inv'], 1e-8, what='Kmm_inv')
    assert_close(pt.exp_K_mi, ref['exp_K_mi'], 1e-12, what='exp_K_mi')
    assert_close(pt.exp_K_mi_K_im, ref['exp_K_mi_K_im'], 1e-12, what='exp_K_mi_K_im')
    st = pt.get_local_statistics()
    assert_close(st['sum_YYT'], ref['sum_YYT'], 1e-12, what='sum_YYT')
    assert_close(st['sum_exp_K_mi_K_im'], ref['sum_exp_K_mi_K_im'], 1e-11, what='sum_exp_K_mi_K_im')
    assert_close(st['exp_K_miY'], ref['exp_K_miY'], 1e-11, what='exp_K_miY')
    assert_close(st['sum_exp_K_ii'], ref['sum_exp_K_ii'], 1e-12, what='sum_exp_K_ii')
    assert_close(st['KL'], ref['KL'], 1e-11, atol=1e-300, what='KL')
    assert_close(pt.Kmm_plus_op_inv, ref['Kmm_plus_op_inv'], 1e-8, what='Kmm_plus_op_inv')
    assert_close(pt.logmarglik(), ref['F'], 1e-6, what='logmarglik')
    for meth in ('dF_dKmm', 'dF_dexp_K_miY', 'dF_dexp_K_mi_K_im', 'dF_dexp_K_ii'):
        assert_close(getattr(pt, meth)(), ref[meth], 1e-5, what=meth)
    for meth in ('dKmm_dZ', 'dexp_K_miY_dZ', 'dexp_K_mi_K_im_dZ', 'dKmm_dalpha', 'dexp_K_miY_dalpha', 'dexp_K_mi_K_im_dalpha',
                 'dKmm_dsf2', 'dexp_K_miY_dsf2', 'dexp_K_mi_K_im_dsf2'):
        assert_close(getattr(pt, meth)(), ref[meth], 1e-10, what=meth)
    assert pt.dexp_K_ii_dsf2() == int(ref['dexp_K_ii_dsf2'])
    gZ = pt.grad_Z(ref['dF_dKmm'], ref['dKmm_dZ'], ref['dF_dexp_K_miY'], ref['dexp_K_miY_dZ'], ref['dF_dexp_K_mi_K_im'], ref['dexp_K_mi_K_im_dZ'])
    assert_close(gZ, ref['grad_Z'], 1e-10, what='grad_Z(parts)')
    ga = pt.grad_alpha(ref['dF_dKmm'], ref['dKmm_dalpha'], ref['dF_dexp_K_miY'], ref['dexp_K_miY_dalpha'], ref['dF_dexp_K_mi_K_im'],
                       ref['dexp_K_mi_K_im_dalpha'])
    assert_close(ga, ref['grad_alpha'], 1e-10, what='grad_alpha(parts)')
    gs = pt.grad_sf2(ref['dF_dKmm'], ref['dKmm_dsf2'], ref['dF_dexp_K_ii'], ref['dexp_K_ii_dsf2'], ref['dF_dexp_K_miY'],
                     ref['dexp_K_miY_dsf2'], ref['dF_dexp_K_mi_K_im'], ref['dexp_K_mi_K_im_dsf2'])
    assert_close(gs, ref['grad_sf2'], 1e-10, what='grad_sf2(parts)')
    assert_close(pt.grad_beta(), ref['grad_beta'], 1e-5, what='grad_beta')
    assert_close(pt.grad_X_mu(), ref['grad_X_mu'], 1e-5, what='grad_X_mu')
    if 'grad_X_S' in ref:
        assert_close(pt.grad_X_S(), ref['grad_X_S'], 1e-5, what='grad_X_S')
    # the sequence parallel_GPLVM.calculate_global_derivatives runs (:340-359), all from this object
    gZ2 = pt.grad_Z(pt.dF_dKmm(), pt.dKmm_dZ(), pt.dF_dexp_K_miY(), pt.dexp_K_miY_dZ(), pt.dF_dexp_K_mi_K_im(), pt.dexp_K_mi_K_im_dZ())
    assert_close(gZ2, ref['grad_Z'], 1e-5, what='grad_Z(own parts)')
    fast = pt.gradients(want_embeddings=False)
    assert_close(fast['grad_Z'], ref['grad_Z'], 1e-5, what='grad_Z(fast path)')
    assert_close(fast['grad_alpha'], ref['grad_alpha'], 1e-5, what='grad_alpha(fast path)')
    assert_close(fast['grad_sf2'], ref['grad_sf2'], 1e-5, what='grad_sf2(fast path)')


def _check_grad(func, grad, x0, h):
    """nputil.check_grad (nputil.py:11-54): max percentage difference between gradient and finite differences.  The
    reference uses forward differences, which makes its own tests flaky at the 1 % threshold (3.96 %
    for grad_X_mu on one draw); central differences keep the check independent and make the threshold meaningful."""
    g = np.atleast_1d(grad(x0)).reshape(-1)
    fd = np.zeros_like(g)
    for d in range(x0.size):
        off = np.zeros(x0.size)
        off[d] = h
        fd[d] = (func(x0 + off) - func(x0 - off)) / (2 * h)
    return np.max(np.abs((g - fd) / fd * 100.0))


@pytest.fixture
def fixture_testpy():
    """test.py:24-60 with a fixed seed: D=7, Q=2, N=5, M=10, X_S = 0.2, data from the GP prior."""
    rs = np.random.RandomState(4)
    D, Q, N, M = 7, 2, 5, 10
    sf = 0.5 + np.exp(0.3 * rs.randn())
    ard = np.exp(0.3 * rs.randn(Q))
    sn = rs.uniform(0.05, 0.1)
    X = rs.randn(N, Q)
    from oracle.literal import rbf_gram
    KXX = rbf_gram(X, sf * sf, ard ** -2.0)
    Y = np.linalg.cholesky(KXX + 1e-10 * np.eye(N)).dot(rs.randn(N, D)) + rs.randn(N, D) * sn
    Z = rs.randn(M, Q)
    X_mu = X + 0.05 * rs.randn(N, Q)
    X_S = 0.2 * np.ones((N, Q))
    from gparml_amd.partial_terms import partial_terms
    pt = partial_terms(Z, sf ** 2, ard ** -2.0, sn ** -2, M, Q, N, D)
    pt.set_data(Y, X_mu, X_S, is_set_statistics=True)
    return pt, dict(Y=Y, X_mu=X_mu, X_S=X_S, Z=Z, M=M, Q=Q)


def test_dF_dZ(fixture_testpy):
    pt, d = fixture_testpy                                             # test.py:62-93

    def f(z):
        pt.Z = z.reshape(d['M'], d['Q'])
        pt.set_data(d['Y'], d['X_mu'], d['X_S'], is_set_statistics=True)
        pt.update_global_statistics()
        return pt.logmarglik()

    def g(z):
        f(z)
        return pt.grad_Z(pt.dF_dKmm(), pt.dKmm_dZ(), pt.dF_dexp_K_miY(), pt.dexp_K_miY_dZ(), pt.dF_dexp_K_mi_K_im(),
                         pt.dexp_K_mi_K_im_dZ()).flatten()

    assert _check_grad(f, g, d['Z'].flatten(), 1e-4) < 1.0


def test_dF_dbeta(fixture_testpy):
    pt, d = fixture_testpy                                             # test.py:186-201
    g = pt.grad_beta()
    F1 = pt.logmarglik()
    pt.beta += 1e-4
    pt.set_data(d['Y'], d['X_mu'], d['X_S'], is_set_statistics=True)
    pt.update_global_statistics()
    fd = (pt.logmarglik() - F1) / 1e-4
    assert abs((fd - g) / fd * 100) < 1.0


def test_dF_dsf2_and_dalpha(fixture_testpy):
    pt, d = fixture_testpy                                             # test.py:96-184, the GPy-free parts
    res = pt.gradients()
    F1 = pt.logmarglik()
    h = 1e-4
    sf0 = pt.hyp.sf
    g_sf = res['grad_sf2'] * 2 * sf0
    pt.hyp.sf = sf0 + h
    pt.set_data(d['Y'], d['X_mu'], d['X_S'], is_set_statistics=True)
    pt.update_global_statistics()
    assert abs((g_sf - (pt.logmarglik() - F1) / h) / g_sf) < 0.01
    pt.hyp.sf = sf0
    ard0 = pt.hyp.ard.copy()
    for q in range(d['Q']):
        ard = ard0.copy()
        ard[q] += 1e-5
        pt.hyp.ard = ard
        pt.set_data(d['Y'], d['X_mu'], d['X_S'], is_set_statistics=True)
        pt.update_global_statistics()
        fd = (pt.logmarglik() - F1) / 1e-5
        g = res['grad_alpha'][q] * -2 * ard0[q] ** -3                  # test.py:123
        assert abs((fd - g) / fd * 100) < 5.0
    pt.hyp.ard = ard0


def test_mu_and_S(fixture_testpy):
    pt, d = fixture_testpy                                             # test.py:270-296

    def f_mu(x):
        pt.set_data(d['Y'], x.reshape(d['X_mu'].shape), d['X_S'], is_set_statistics=True)
        return pt.logmarglik()

    def f_S(x):
        pt.set_data(d['Y'], d['X_mu'], x.reshape(d['X_S'].shape), is_set_statistics=True)
        return pt.logmarglik()

    def g_mu(x):
        f_mu(x)
        return pt.grad_X_mu().flatten()

    def g_S(x):
        f_S(x)
        return pt.grad_X_S().flatten()

    assert _check_grad(f_mu, g_mu, d['X_mu'].flatten(), 1e-5) < 1.0
    assert _check_grad(f_S, g_S, d['X_S'].flatten(), 1e-5) < 1.0


def test_global_only_object_like_the_master(fixture_testpy):
    """parallel_GPLVM.calculate_global_statistics (:302-334): an object without data, fed the reduced sums."""
    pt, d = fixture_testpy
    from gparml_amd.partial_terms import partial_terms
    st = pt.get_local_statistics()
    master = partial_terms(pt.Z, pt.hyp.sf ** 2, pt.hyp.ard ** -2.0, pt.beta, pt.M, pt.Q, pt.N, pt.D, update_global_statistics=False)
    master.set_global_statistics(pt.Kmm, pt.Kmm_inv)
    master.set_local_statistics(st['sum_YYT'], st['sum_exp_K_mi_K_im'], st['exp_K_miY'], st['sum_exp_K_ii'], st['KL'])
    assert_close(master.logmarglik(), pt.logmarglik(), 1e-12, what='F')
    assert_close(master.dF_dKmm(), pt.dF_dKmm(), 1e-10, what='dF_dKmm')
    assert_close(master.grad_beta(), pt.grad_beta(), 1e-10, what='grad_beta')
    assert_close(master.dKmm_dZ(), pt.dKmm_dZ(), 1e-12, what='dKmm_dZ')
