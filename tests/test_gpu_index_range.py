"""Index ranges beyond 2**31 ELEMENTS (round-4 review, missing 2): every array of the path that scales with N x M is addressed with 64-bit
offsets, and these cases are the first that actually cross the line -- BASELINE configs[4]'s per-GPU Kaug is
1 000 064 x 2048 = 2.048e9 doubles, 4.6 % UNDER 2**31.  The array this replaces in the reference is the (N, M, M) tensor of
partial_terms.py:45 and the (N, M) / (N, D) operands of partial_terms.py:47-52.

No oracle runs at these sizes; the checks are the size-independent ones of test_gpu_fullsize.py: a repeated evaluation is bit-identical,
one shard equals two ragged shards reduced through the packed buffers (a wrapped 32-bit offset would read or write a different row in
the one-shard context -- whose rows sit ABOVE the line -- than in the two half-size contexts, whose rows sit below it), and a
directional central finite difference of the bound against the analytic gradient.

  * fixed embeddings, N = 1.2e6, D = 1000, M = 1024, Q = 50: Kaug = 1 200 128 x 2048 = 2.46e9 doubles (19.7 GB);
  * free embeddings, N = 2.2e6, D = 8, M = 1024: LE / LEA = 2 200 064 x 1024 = 2.25e9 doubles each, Kaug 2.53e9 -- at Q = 6 (phase 2 on
    psi2_sym_kernel, phase 1 on psi2_pairs_kernel) and at Q = 20 (psi2_tile_kernel / psi2_pairs_mfma_kernel); the per-point partials
    HZp [8][Np][CZp] and pp follow the same row index.
"""
import os
from multiprocessing.pool import ThreadPool

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


def _generate(N, D, M, Q, regime, seed):
    """SURVEY.md 8(d)'s synthetic shard in row chunks on the host's threads (one random stream per chunk)."""
    Wmap = np.random.RandomState(1234).randn(Q, D)
    Y, X_mu, X_S = np.empty((N, D)), np.empty((N, Q)), np.zeros((N, Q))
    step = 50000

    def chunk(i):
        rs = np.random.RandomState(seed * 1000 + i)
        a, b = i * step, min(N, (i + 1) * step)
        X = rs.randn(b - a, Q)
        Y[a:b] = np.sin(X.dot(Wmap))
        Y[a:b] += 0.1 * rs.randn(b - a, D)
        X_mu[a:b] = X + 0.05 * rs.randn(b - a, Q)
        if regime == 'B':
            X_S[a:b] = rs.uniform(0.05, 0.55, size=(b - a, Q))

    with ThreadPool(min(32, os.cpu_count() or 8)) as pool:
        pool.map(chunk, range((N + step - 1) // step))
    rs = np.random.RandomState(seed + 1)
    Z = X_mu[rs.permutation(N)[:M]] + 0.3 * rs.randn(M, Q)
    return dict(Y=Y, X_mu=X_mu, X_S=X_S, Z=Z, sf2=1.0, alpha=np.full(Q, min(0.3, 3.0 / Q)), beta=10.0)


def _engines(d, cuts, N, D, M, Q):
    from gparml_amd.engine import ShardEngine
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        e = ShardEngine(b - a, D, M, Q)
        e.upload_shard(d['Y'][a:b], d['X_mu'][a:b], d['X_S'][a:b])
        e.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'], N_global=N)
        out.append(e)
    return out


def _run(engines, emb):
    for e in engines:
        e.phase1()
    root = engines[0]
    for e in engines[1:]:
        root.combine(e, 'stats', 'add')
    for e in engines[1:]:
        e.combine(root, 'stats', 'copy')
    for e in engines:
        e.global_step()
        e.phase2(emb)
    for e in engines[1:]:
        root.combine(e, 'grads', 'add')
    out = root.finish()
    if emb:
        out['grad_X_mu'] = np.concatenate([e.download('GRAD_X_MU') for e in engines])
        out['grad_X_S'] = np.concatenate([e.download('GRAD_X_S') for e in engines])
    return out


@pytest.mark.parametrize('N,D,M,Q,regime,alpha', [(1200000, 1000, 1024, 50, 'A', None), (2200000, 8, 1024, 6, 'B', 0.8), (2200000, 8, 1024, 20, 'B', None)])
def test_arrays_beyond_two_to_the_31_elements(N, D, M, Q, regime, alpha):
    emb = regime == 'B'
    Np, Mp, Dp = -(-N // 128) * 128, -(-M // 128) * 128, -(-D // 128) * 128
    assert Np * (Mp + Dp) > 2 ** 31 and (regime == 'A' or Np * Mp > 2 ** 31)          # the case is what its name says
    d = _generate(N, D, M, Q, regime, seed=50 + Q)
    if alpha is not None:
        d['alpha'] = np.full(Q, alpha)          # 1024 inducing points in a six-dimensional latent space: a shorter length scale keeps K_mm factorisable
    blocks = ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta') + (('grad_X_mu', 'grad_X_S') if emb else ())
    one = _engines(d, [0, N], N, D, M, Q)
    ref = _run(one, emb)
    assert np.isfinite(ref['F']) and all(np.all(np.isfinite(ref[k])) for k in blocks)
    if emb:
        # the LAST rows' per-point gradients are real numbers (their LE / LEA / HZp rows are the ones past the 2**31st element)
        assert np.max(np.abs(ref['grad_X_mu'][-1000:])) > 0 and np.max(np.abs(ref['grad_X_S'][-1000:])) > 0
    # (1) bit-identical repeat
    again = _run(one, emb)
    assert again['F'] == ref['F'] and all(np.array_equal(again[k], ref[k]) for k in blocks)
    del again
    # (2) two ragged shards -- each below the line on its own -- against the one shard that crosses it
    two = _engines(d, [0, N // 2 + 4711, N], N, D, M, Q)
    out = _run(two, emb)
    for e in two:
        e.close()
    assert_close(out['F'], ref['F'], 1e-11, what='F (2 shards vs 1)')
    for k in blocks:
        assert_close(out[k], ref[k], 1e-8, what=k + ' (2 shards vs 1)')
    del out
    # (3) directional derivative along a random direction of (Z, sf2, alpha, beta), central difference
    rs = np.random.RandomState(5)
    dZ, ds, da, db = rs.randn(M, Q), rs.randn(), rs.randn(Q), rs.randn()
    scale = 1e-6
    ana = float(np.sum(ref['grad_Z'] * dZ) + ref['grad_sf2'] * ds * d['sf2'] + np.sum(ref['grad_alpha'] * da * d['alpha'])
                + ref['grad_beta'] * db * d['beta'])
    Fs = []
    eng = one[0]
    for sgn in (+1.0, -1.0):
        h = sgn * scale
        eng.set_globals(d['Z'] + h * dZ, d['sf2'] * (1 + h * ds), d['alpha'] * (1 + h * da), d['beta'] * (1 + h * db), N_global=N)
        Fs.append(_run([eng], False)['F'])
    eng.close()
    fd = (Fs[0] - Fs[1]) / (2 * scale)
    print('N=%d D=%d M=%d Q=%d regime %s: Kaug %.3e doubles; directional derivative fd %.10e analytic %.10e (rel %.1e)'
          % (N, D, M, Q, regime, Np * (Mp + Dp), fd, ana, abs(fd - ana) / abs(ana)))
    assert abs(fd - ana) <= 2e-5 * abs(ana) + 1e-9 * abs(ref['F']), 'directional derivative: fd %.10e vs analytic %.10e' % (fd, ana)
