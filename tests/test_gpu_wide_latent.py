"""Free embeddings with a latent space wider than the compiled tables (Q >= 64; gparml_amd/csrc/psi2_generic.hip): the reference has no limit on Q
(kernel_exp.py:126-148, partial_terms.py:367-431), so neither has the library -- rounds 1-5 refused Q > 64 with GP_ERR_UNSUPPORTED.
Bound and every gradient, embedding gradients included, against the oracle at Q = 64 (the first width without a spare column for the tile
kernel), 65, 100 and 130; several point chunks (M = 300: 93 points per chunk), M = 1, one shard = two ragged shards."""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu

# N, D, M, Q, alpha
SHAPES = [(300, 3, 40, 64, 0.03), (257, 2, 33, 65, 0.03), (400, 4, 300, 100, 0.02), (130, 2, 1, 130, 0.02), (1000, 2, 20, 70, 0.03)]
KEYS = ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu', 'grad_X_S')


def _eval(d, rows=None):
    from gparml_amd.engine import ShardEngine
    sl = slice(None) if rows is None else rows
    Y, mu, S = d['Y'][sl], d['X_mu'][sl], d['X_S'][sl]
    eng = ShardEngine(Y.shape[0], Y.shape[1], d['Z'].shape[0], d['Z'].shape[1])
    eng.upload_shard(Y, mu, S)
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'], N_global=d['Y'].shape[0])
    return eng


@pytest.mark.parametrize('N,D,M,Q,alpha', SHAPES)
def test_wide_latent_space_against_the_oracle(N, D, M, Q, alpha):
    from oracle import factorised as Fz
    d = Fz.synthetic_shard(N, D, M, Q, regime='B', seed=41, zseed=42, alpha_value=alpha)
    ref = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'])
    eng = _eval(d)
    out = eng.evaluate(True)
    again = eng.evaluate(True)
    stats = {k: eng.download(k) for k in ('PSI2_SUM', 'PSI1TY')}
    eng.close()
    assert_close(stats['PSI2_SUM'], ref['stats']['sum_exp_K_mi_K_im'], 1e-10, what='Psi2')
    assert_close(out['F'], ref['F'], 1e-6, what='F')
    for k in KEYS:
        assert_close(out[k], ref[k], 1e-5, what=k)
        assert np.array_equal(np.asarray(out[k]), np.asarray(again[k])), k + ': repeat differs'


def test_one_shard_equals_two_ragged_shards_at_q_70():
    """the statistics and gradient sums of two shards (431 + 569 rows) add up to the single shard's: the generic kernels' chunking does not depend on
    where a shard starts."""
    from oracle import factorised as Fz
    from test_gpu_fullsize import _eval_sharded, _run
    N, D, M, Q, alpha = SHAPES[-1]
    d = Fz.synthetic_shard(N, D, M, Q, regime='B', seed=41, zseed=42, alpha_value=alpha)
    one = _eval_sharded(d, N, D, M, Q, [0, N], True)
    ref = _run(one, True)
    one[0].close()
    two = _eval_sharded(d, N, D, M, Q, [0, 431, N], True)
    out = _run(two, True)
    for e in two:
        e.close()
    assert_close(out['F'], ref['F'], 1e-11, what='F (2 shards vs 1)')
    for k in KEYS:
        assert_close(out[k], ref[k], 1e-8, what=k + ' (2 shards vs 1)')
