"""Size-independent properties at BASELINE.json's headline size (configs[2]: N=1e6, D=100, M=512, Q=10), where no CPU
oracle finishes in test time: (1) the map/reduce identity -- one 1e6-point shard equals two shards reduced through the
packed device buffers, for the bound and every gradient; (2) a directional central finite difference of the bound
against the analytic gradient (the check test.py:36-94 makes per coordinate).  Regime B at a reduced N (the pair kernels
cost 0.5 s per 1e6 points), and at BASELINE configs[4]'s per-GPU shape (D=1000, M=1024, Q=50) on 2e4 points: three launches of the
tile-pair kernel per evaluation, ragged shards."""
import numpy as np
import pytest

import oracle_cache
from conftest import assert_close

pytestmark = pytest.mark.gpu


def _synthetic(N, D, M, Q, regime):
    rs = np.random.RandomState(0)
    X = rs.randn(N, Q)
    W = np.random.RandomState(1234).randn(Q, D)
    Y = np.sin(X.dot(W)) + 0.1 * rs.randn(N, D)
    X_mu = X + 0.05 * rs.randn(N, Q)
    X_S = np.zeros((N, Q)) if regime == 'A' else rs.uniform(0.05, 0.55, size=(N, Q))
    rz = np.random.RandomState(1)
    Z = np.random.RandomState(2).randn(4 * M, Q)[rz.permutation(4 * M)[:M]] + 0.05 * rz.randn(M, Q)
    return dict(Y=Y, X_mu=X_mu, X_S=X_S, Z=Z, sf2=1.0, alpha=np.full(Q, min(0.3, 3.0 / Q)), beta=10.0)


def _eval_sharded(d, N, D, M, Q, cuts, emb, Z=None, sf2=None, alpha=None, beta=None):
    from gparml_amd.engine import ShardEngine
    Z = d['Z'] if Z is None else Z
    sf2 = d['sf2'] if sf2 is None else sf2
    alpha = d['alpha'] if alpha is None else alpha
    beta = d['beta'] if beta is None else beta
    engines = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        e = ShardEngine(b - a, D, M, Q)
        e.upload_shard(d['Y'][a:b], d['X_mu'][a:b], d['X_S'][a:b])
        e.set_globals(Z, sf2, alpha, beta, N_global=N)
        engines.append(e)
    return engines


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


@pytest.mark.parametrize('N,D,M,Q,regime', [(1000000, 100, 512, 10, 'A'), (60000, 20, 512, 10, 'B'), (20000, 1000, 1024, 50, 'B')])
def test_map_reduce_identity_and_directional_derivative(N, D, M, Q, regime):
    emb = regime == 'B'
    d = _synthetic(N, D, M, Q, regime)
    one = _eval_sharded(d, N, D, M, Q, [0, N], emb)
    ref = _run(one, emb)
    two = _eval_sharded(d, N, D, M, Q, [0, N // 3 + 17, N], emb)
    out = _run(two, emb)
    for e in two:
        e.close()
    assert_close(out['F'], ref['F'], 1e-11, what='F (2 shards vs 1)')
    for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta') + (('grad_X_mu', 'grad_X_S') if emb else ()):
        assert_close(out[k], ref[k], 1e-8, what=k + ' (2 shards vs 1)')
    # directional derivative along a random direction of (Z, sf2, alpha, beta), central difference
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
    assert abs(fd - ana) <= 2e-5 * abs(ana) + 1e-6 * abs(ref['F']) * 1e-3, 'directional derivative: fd %.10e vs analytic %.10e' % (fd, ana)


def test_full_size_against_the_blas_port():
    """BASELINE configs[2] at its FULL size (N=1e6, D=100, M=512, Q=10, fixed embeddings -- the kernel sequence bench.py times)
    against the CPU port arranged for BLAS (oracle/factorised.evaluate_blas, ~13 s on the GPU box's host; checked against the
    kernel-spec port by tests/test_oracle_factorised.py).  Two hyper-parameter sets on the same data:
      alpha = 0.3: well conditioned -- measured F 3e-14, grad_Z 7e-10, the other gradients <= 1e-12 (relative to the block's
                   largest magnitude); asserted at 1e-9 / 1e-7;
      alpha = 0.1: the benchmark's value -- cond(K_mm + beta Psi2) = 1.4e10 and grad_Z amplifies the global step's rounding ~1000-fold.
                   Two float64 evaluations cannot referee each other there, so the bound comes from the extended-precision truth of the
                   benchmark workload (tests/test_hp_truth_large.py, tests/golden/hp_truth_large_N1000000.npz): the float64 Cholesky port is
                   2.3e-5 from the truth on grad_Z and the reference's LU arrangement 8.9e-5, so two float64 paths may differ by their
                   sum; asserted: grad_Z within err_lu + err_chol = 1.1e-4 of the port, F 1e-8, the other gradients 1e-6 (measured F 1e-11,
                   grad_Z 2.8e-5, grad_alpha 6e-9, grad_sf2 4e-10, grad_beta 5e-13).  The device's own distance from the truth is asserted
                   in tests/test_hp_truth_large.py."""
    import os
    from conftest import GOLDEN_DIR
    from gparml_amd.engine import ShardEngine
    from oracle import factorised as Fz
    zt = np.load(os.path.join(GOLDEN_DIR, 'hp_truth_large_N1000000.npz'))
    gz_bound = float(zt['err_lu_grad_Z']) + float(zt['err_chol_grad_Z'])
    N, D, M, Q = 1000000, 100, 512, 10
    d = _synthetic(N, D, M, Q, 'A')
    work, refs = {}, {}
    cases = ((0.3, 1e-9, 1e-7, 1e-7), (0.1, 1e-8, gz_bound, 1e-6))
    for alpha, _, _, _ in cases:
        al = np.full(Q, alpha)
        # ~13 s of host time per hyper-parameter set when computed live: committed oracle outputs (tests/oracle_cache.py)
        refs[alpha] = oracle_cache.get('config2_full_blas_port_alpha%.1f' % alpha, dict(d, alpha=al),
                                       lambda: Fz.evaluate_blas(d['Z'], d['sf2'], al, d['beta'], d['Y'], d['X_mu'], work=work))
    oracle_cache.done()
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    for alpha, f_tol, gz_tol, g_tol in cases:
        al, ref = np.full(Q, alpha), refs[alpha]
        eng.set_globals(d['Z'], d['sf2'], al, d['beta'])
        out = eng.evaluate(False)
        assert_close(out['F'], ref['F'], f_tol, what='F (alpha %.1f)' % alpha)
        assert_close(out['grad_Z'], ref['grad_Z'], gz_tol, what='grad_Z (alpha %.1f)' % alpha)
        for k in ('grad_alpha', 'grad_sf2', 'grad_beta'):
            assert_close(out[k], ref[k], g_tol, what='%s (alpha %.1f)' % (k, alpha))
    eng.close()
