"""General phase 2 with embedding gradients (p2_gen8_kernel + point_kernel, gparml_amd/csrc/psi.hip; reference: partial_terms.py:162-205,
256-284 for the inducing-point sums, :367-431 for grad_X_mu / grad_X_S).

The kernel contracts W = G o Psi1 twice -- over the points straight from the accumulator registers (four-block MFMA partials added through
LDS) and over the inducing points through an LDS slab pair with a 16-column MFMA, the four 32-column partials of a row exchanged through
LDS -- so the shapes below vary what those paths depend on: the number of feature quads (Q = 1 ... 50: one to 26 quads, one to seven
16-column groups, a last group that is full, half full or a single quad), the number of 128-column tiles (M = 1 ... 600, tiles that are
almost all padding), the number of row tiles per slice and a ragged last tile, both regimes (regime B leaves the K_mm part of the k-loop
out), and D not a multiple of the k-chunk (the k-steps that only multiply zero padding are skipped)."""
import numpy as np
import pytest

from conftest import assert_close  # noqa: F401  (conftest puts the repo root on sys.path)

pytestmark = pytest.mark.gpu

# N, D, M, Q, regime, alpha
SHAPES = [
    (300, 5, 20, 3, 'A', 0.4), (257, 2, 1, 1, 'A', 1.0), (1000, 7, 130, 10, 'A', 0.4), (3000, 20, 200, 6, 'A', 0.4), (3000, 20, 600, 12, 'A', 0.3),
    (20000, 12, 140, 5, 'A', 1.0), (20000, 12, 140, 5, 'B', 1.0), (2000, 16, 257, 13, 'B', 0.2), (900, 33, 129, 7, 'B', 0.3), (640, 3, 70, 15, 'A', 0.1),
    (700, 100, 512, 10, 'B', 0.1), (500, 4, 300, 24, 'A', 0.1), (400, 2, 70, 31, 'A', 0.05), (333, 3, 100, 50, 'B', 0.03), (5000, 1, 128, 4, 'B', 2.0),
]


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'N%d-D%d-M%d-Q%d-%s' % s[:5])
def test_embedding_gradients_against_the_oracle(shape):
    from gparml_amd.engine import ShardEngine
    from oracle import factorised as Fz
    N, D, M, Q, regime, alpha = shape
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=11, zseed=12, alpha_value=alpha)
    ref = Fz.evaluate_sharded(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], shards=8, pairs='gemm')
    eng = ShardEngine(N, D, M, Q)
    try:
        eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
        eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
        eng.evaluate(True)          # (the first call after set_globals may still run phase 1 in its fixed-embedding form)
        out = eng.evaluate(True)
        again = eng.evaluate(True)
    finally:
        eng.close()
    assert abs(out['F'] - ref['F']) <= 1e-6 * abs(ref['F'])
    # gradients: 1e-5 of the block's largest entry (north_star); grad_Z carries the conditioning of K_mm + beta Psi2, which these shapes keep mild
    keys = ['grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu'] + (['grad_X_S'] if regime == 'B' else [])
    for k in keys:
        err = np.max(np.abs(np.asarray(out[k]) - np.asarray(ref[k]))) / np.max(np.abs(ref[k]))
        assert err <= 1e-5, (k, err)
    # fixed summation order everywhere (LDS exchanges, no atomics): a second evaluation is bit-identical
    for k in keys:
        assert np.array_equal(np.asarray(out[k]), np.asarray(again[k])), k
