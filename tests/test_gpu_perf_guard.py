"""A coarse guard on the headline kernels' durations (round 5: a run-time branch around two non-temporal stores stopped hipcc merging them and psi1_kernel went from 0.96 to
2.28 ms at N = 1e6 -- every parity test still passed; only bench.py showed it).  One fifth of the benchmark's shard (N = 2e5, D = 100, M = 512, Q = 10, fixed embeddings),
the library's own HIP events, best of eight evaluations.  Limits (r06): 1.15 x the largest value measured on the pool's boxes in rounds 5 and 6 (they differ by
2-5 %); the test runs LAST in the suite (tests/conftest.py) so that a contended box cannot stop the parity files under -x, and skips when the timing
events are switched off."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# ms at N = 2e5 on boxes of the pool: psi1_kernel 0.25 / 0.25, p1v2_kernel 1.16 / 1.11, p2_fast8_kernel 2.12 / 2.07, global step 0.48 / 0.49 (r05 / r06)
LIMITS = {'psi1_ms': 0.29, 'p1_kernel_ms': 1.34, 'p2_kernel_ms': 2.44, 'global_ms': 0.57}


def test_headline_kernels_are_not_grossly_slower_than_measured():
    import bench
    from gparml_amd.engine import ShardEngine
    N, D, M, Q = 200000, 100, 512, 10
    d = bench.synthetic(N, D, M, Q, seed=100)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    best = {}
    for i in range(8):
        eng.set_globals(d['Z'] + 1e-4 * i, d['sf2'], d['alpha'], d['beta'])
        out = eng.evaluate(False)
        for k, v in eng.timings().items():
            best[k] = min(best.get(k, 1e9), v)
    eng.close()
    assert np.isfinite(out['F'])
    if max(best.get(k, 0.0) for k in LIMITS) <= 0.0:
        pytest.skip('no per-phase timings (GPARML_TIMING=0/1 in the environment): nothing to guard')
    print('kernel ms at N = 2e5:', {k: round(best[k], 4) for k in LIMITS})
    for k, lim in LIMITS.items():
        assert best[k] <= lim, '%s = %.3f ms at N = 2e5 (limit %.2f): a headline kernel is grossly slower than measured -- run bench.py' % (k, best[k], lim)
