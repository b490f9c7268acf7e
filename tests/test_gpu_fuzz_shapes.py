"""Odd shapes: seeded random (N, D, M, Q, regime) with N below a row tile, M = 1, M > N, Q up to 63, D around the phase-1 kernels' column
limits -- every gradient block against the oracle (1e-5 of the block's largest entry, F 1e-6).  Shapes whose K_mm + beta Psi2 has a
condition number beyond 1e11 carry no float64 digits on either side and are skipped (tests/devtools/dev_fuzz_shapes.py is the same loop
with more shapes and a multi-tile mode)."""
import numpy as np
import pytest

from conftest import assert_close  # noqa: F401

pytestmark = pytest.mark.gpu


def _shapes(count, seed):
    rs = np.random.RandomState(seed)
    out = []
    for it in range(count):
        regime = 'AB'[rs.randint(2)]
        Q = int(rs.choice([1, 2, 3, 5, 8, 10, 11, 12, 16, 17, 23, 24, 25, 31, 40, 50, 51, 52, 60, 63]))
        M = int(rs.choice([1, 2, 7, 16, 33, 64, 65, 100, 128, 129, 200, 257, 300]))
        N = int(rs.choice([1, 2, 17, 63, 64, 127, 128, 129, 300, 777, 1500]))
        D = int(rs.choice([1, 2, 3, 4, 5, 15, 16, 17, 33, 100, 104, 105, 130, 300]))
        if regime == 'B' and Q > 24 and M > 130:
            M = 64
        out.append((it, N, D, M, Q, regime))
    return out


@pytest.mark.parametrize('shape', _shapes(40, 1), ids=lambda s: 'N%d-D%d-M%d-Q%d-%s' % s[1:])
def test_random_shape_against_the_oracle(shape):
    from gparml_amd.engine import ShardEngine
    from oracle import factorised as Fz
    it, N, D, M, Q, regime = shape
    d = Fz.synthetic_shard(N, D, min(M, N), Q, regime=regime, seed=100 + it, zseed=200 + it, alpha_value=min(0.5, 2.0 / Q))
    if M > N:
        d['Z'] = 1.5 * np.random.RandomState(300 + it).randn(M, Q)          # more inducing points than data points
    try:
        ref = Fz.evaluate_sharded(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], shards=4, pairs='gemm')
    except np.linalg.LinAlgError:
        pytest.skip('the float64 oracle cannot factorise this K_mm')
    dz = d['Z'][:, None, :] - d['Z'][None, :, :]
    Kmm = d['sf2'] * np.exp(-0.5 * np.sum(np.asarray(d['alpha'])[None, None, :] * dz * dz, axis=2))
    if np.linalg.cond(Kmm + d['beta'] * ref['stats']['sum_exp_K_mi_K_im']) > 1e11:
        pytest.skip('cond(K_mm + beta Psi2) > 1e11: no float64 digits to compare')
    eng = ShardEngine(N, D, M, Q)
    try:
        eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
        eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
        out = eng.evaluate(True)
    finally:
        eng.close()
    assert abs(out['F'] - ref['F']) <= 1e-6 * abs(ref['F'])
    for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu') + (('grad_X_S',) if regime == 'B' else ()):
        err = np.max(np.abs(np.asarray(out[k]) - np.asarray(ref[k]))) / max(np.max(np.abs(ref[k])), 1e-300)
        assert err <= 1e-5, (k, err)
