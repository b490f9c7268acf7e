"""Pins the oracle on the WHOLE evaluation (trial point X + step*d, softplus of the raw variances, reduce over shards,
global step, chain rule through the softplus of the globals, the (2,N_s,Q) negative-gradient layout) against the
pipeline goldens captured from the reference (parallel_GPLVM.py:222-279, local_MapReduce.py:183-363)."""
import numpy as np
import pytest

from conftest import assert_close
from oracle import factorised as Fz
from oracle import literal as L
from pipeline_util import call_args, load_pipeline, pipeline_names


def oracle_call(g, k):
    M, Q, D, N = int(g['M']), int(g['Q']), int(g['D']), int(g['N'])
    fixed = bool(g['fixed'])
    x, it, step = call_args(g, k)
    nz = M * Q
    Z = x[:nz].reshape(M, Q)
    sp = lambda v: np.log(1 + np.exp(v))
    sf2, alpha, beta = sp(x[nz]), sp(x[nz + 1:nz + 1 + Q]), sp(x[nz + 1 + Q])
    shards = []
    for i in range(int(g['n_shards'])):
        mu = g['call%d_in_shard%d_embedding' % (k, i)].copy()
        S = g['call%d_in_shard%d_variance' % (k, i)].copy()
        if not fixed:
            dk = 'call%d_in_shard%d_grad_d' % (k, i)
            if dk in g and step != 0:
                mu += g[dk][0] * step
                S += g[dk][1] * step
            Sraw = S
            S = L.transformVar(S)
        else:
            Sraw = None
        shards.append((g['Y_%d' % i], mu, S, Sraw))
    st = None
    for Y, mu, S, _ in shards:
        s = Fz.phase1(Z, sf2, alpha, Y, mu, S)
        st = s if st is None else {kk: st[kk] + s[kk] for kk in st}
    gs = Fz.global_step(Z, sf2, alpha, beta, st, N, D)
    acc, latest = None, []
    for Y, mu, S, Sraw in shards:
        p2 = Fz.phase2(Z, sf2, alpha, Y, mu, S, gs['Abar'], gs['Bbar'], want_embeddings=not fixed)
        part = dict(grad_Z_data=p2['grad_Z_data'], grad_alpha_data=p2['grad_alpha_data'])
        acc = part if acc is None else {kk: acc[kk] + part[kk] for kk in acc}
        if not fixed:
            latest.append(-np.array([p2['grad_X_mu'], p2['grad_X_S'] * L.transformVar_grad(Sraw)]))
    out = Fz.finish(Z, sf2, alpha, gs, acc, fixed)
    grad = np.concatenate([out['grad_Z'].flatten(), [out['grad_sf2']], out['grad_alpha'], [out['grad_beta']]])
    chain = np.concatenate([np.ones(nz), 1.0 / (np.exp(-x[nz:]) + 1.0)])
    return -out['F'], -grad * chain, st, latest


@pytest.mark.parametrize('name', pipeline_names())
def test_oracle_replays_reference_pipeline(name):
    g = load_pipeline(name)
    for k in range(int(g['n_calls'])):
        f, grad, st, latest = oracle_call(g, k)
        assert_close(f, g['call%d_f' % k], 1e-10, what='%s call %d f' % (name, k))
        assert_close(grad, g['call%d_g' % k], 1e-8, what='%s call %d grad' % (name, k))
        assert_close(st['sum_exp_K_mi_K_im'], g['call%d_acc_sum_exp_K_mi_K_im' % k], 1e-11, what='Psi2')
        assert_close(st['exp_K_miY'], g['call%d_acc_sum_exp_K_miY' % k], 1e-11, what='C')
        assert_close(st['KL'], g['call%d_acc_sum_KL' % k], 1e-11, atol=1e-300, what='KL')
        for i, lat in enumerate(latest):
            assert_close(lat, g['call%d_out_shard%d_grad_latest' % (k, i)], 1e-8, what='grad_latest shard %d' % i)
