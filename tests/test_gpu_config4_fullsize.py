"""BASELINE configs[4] at its FULL per-GPU size: N_s = 1e6, D = 1000, M = 1024, Q = 50, free embeddings (Bayesian GPLVM, regime B) --
the largest configuration BASELINE.json names; reference path partial_terms.py:367-431 (grad_X_mu / grad_X_S), kernel_exp.py:126-148
(psi2_n), local_MapReduce.py:183-363 (the two mappers), scg_adapted_local_MapReduce.py:29-243 (the optimiser's resident vectors).

No CPU oracle finishes there (W_B = 2.2e14 flop per evaluation), so at the full size the checks are the size-independent ones of
test_gpu_fullsize.py -- (1) a repeated evaluation is bit-identical, (2) one 1e6-point shard equals two ragged shards reduced through the
packed device buffers, for the bound and every gradient including the per-point ones, (3) a directional central finite difference of
the bound against the analytic gradient (test.py:36-94 per coordinate) -- plus (4) one resident SCG iteration on two 5e5-point shards
(finite, decreasing objective), and (5) the oracle itself on a 2e4-point slice of the SAME data and inducing points, evaluated in shards on
all host threads (bound 1e-6, every gradient 1e-5).  The device memory actually taken (hipMemGetInfo before / after, gp_memory_info) is
printed and, when gpurun_out/ exists, written to gpurun_out/config4_fullsize.json: DESIGN.md section 4 quotes it."""
import json
import os
import time
from multiprocessing.pool import ThreadPool

import numpy as np
import pytest

import oracle_cache
from conftest import ROOT, assert_close

pytestmark = pytest.mark.gpu

N, D, M, Q = 1000000, 1000, 1024, 50
REPORT = {}


def _report(**kw):
    REPORT.update(kw)
    out = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'config4_fullsize.json'), 'w') as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)


def _generate(N, D, M, Q, seed=40):
    """SURVEY.md 8(d)'s synthetic shard, generated in row chunks on the host's threads (8 GB of Y): every chunk has its own stream."""
    Wmap = np.random.RandomState(1234).randn(Q, D)
    Y = np.empty((N, D))
    X_mu = np.empty((N, Q))
    X_S = np.empty((N, Q))
    step = 50000

    def chunk(i):
        rs = np.random.RandomState(seed * 1000 + i)
        a, b = i * step, min(N, (i + 1) * step)
        X = rs.randn(b - a, Q)
        Y[a:b] = np.sin(X.dot(Wmap))
        Y[a:b] += 0.1 * rs.randn(b - a, D)
        X_mu[a:b] = X + 0.05 * rs.randn(b - a, Q)
        X_S[a:b] = rs.uniform(0.05, 0.55, size=(b - a, Q))

    with ThreadPool(min(32, os.cpu_count() or 8)) as pool:
        pool.map(chunk, range((N + step - 1) // step))
    rs = np.random.RandomState(seed + 1)
    Z = X_mu[rs.permutation(N)[:M]] + 0.3 * rs.randn(M, Q)
    return dict(Y=Y, X_mu=X_mu, X_S=X_S, Z=Z, sf2=1.0, alpha=np.full(Q, 1.0 / Q), beta=10.0)


@pytest.fixture(scope='module')
def data():
    t = time.time()
    d = _generate(N, D, M, Q)
    _report(host_generate_s=round(time.time() - t, 1))
    return d


def _engines(d, cuts, N_global):
    from gparml_amd.engine import ShardEngine
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        e = ShardEngine(b - a, D, M, Q)
        e.upload_shard(d['Y'][a:b], d['X_mu'][a:b], d['X_S'][a:b])
        e.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'], N_global=N_global)
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


GB = 1.0 / (1 << 30)


def test_full_size_evaluation_identity_and_directional_derivative(data):
    from gparml_amd.engine import ShardEngine
    d = data
    probe = ShardEngine(128, 1, 1, 1)
    free0, total = probe.memory_info()
    t = time.time()
    eng = ShardEngine(N, D, M, Q)
    free1, _ = eng.memory_info()
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'], N_global=N)
    t_up = time.time() - t
    t = time.time()
    ref = _run([eng], True)
    t_first = time.time() - t
    free2, _ = eng.memory_info()
    tm = eng.timings()
    print('configs[4] per-GPU size: device memory total %.1f GB, taken by gp_create %.2f GB, after the first evaluation (regime-B buffers) %.2f GB'
          % (total * GB, (free0 - free1) * GB, (free0 - free2) * GB))
    print('first evaluation %.2f s wall (upload %.1f s); device ms: %s' % (t_first, t_up, {k: round(v, 1) for k, v in tm.items()}))
    _report(device_total_GB=total * GB, taken_by_gp_create_GB=(free0 - free1) * GB, taken_after_first_evaluation_GB=(free0 - free2) * GB,
            upload_s=round(t_up, 1), first_evaluation_wall_s=round(t_first, 2), device_ms={k: float(v) for k, v in tm.items()}, F=ref['F'])
    assert np.isfinite(ref['F'])
    for k in ('grad_Z', 'grad_alpha', 'grad_X_mu', 'grad_X_S'):
        assert np.all(np.isfinite(ref[k])), k
    assert np.max(np.abs(ref['grad_X_mu'])) > 0 and np.max(np.abs(ref['grad_X_S'])) > 0
    # (1) bit-identical repeat
    again = _run([eng], True)
    assert again['F'] == ref['F']
    for k in ('grad_Z', 'grad_alpha', 'grad_X_mu', 'grad_X_S'):
        assert np.array_equal(again[k], ref[k]), k
    assert again['grad_sf2'] == ref['grad_sf2'] and again['grad_beta'] == ref['grad_beta']
    del again
    # (2) two ragged shards, reduced through the packed buffers, next to the resident one-shard context
    two = _engines(d, [0, N // 3 + 17, N], N)
    free3, _ = eng.memory_info()
    out = _run(two, True)
    for e in two:
        e.close()
    _report(taken_with_three_contexts_GB=(free0 - free3) * GB)
    assert_close(out['F'], ref['F'], 1e-11, what='F (2 shards vs 1)')
    for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu', 'grad_X_S'):
        assert_close(out[k], ref[k], 1e-8, what=k + ' (2 shards vs 1)')
    del out
    # (3) directional derivative along a random direction of (Z, sf2, alpha, beta), central difference
    rs = np.random.RandomState(5)
    dZ, ds, da, db = rs.randn(M, Q), rs.randn(), rs.randn(Q), rs.randn()
    scale = 1e-6
    ana = float(np.sum(ref['grad_Z'] * dZ) + ref['grad_sf2'] * ds * d['sf2'] + np.sum(ref['grad_alpha'] * da * d['alpha'])
                + ref['grad_beta'] * db * d['beta'])
    Fs = []
    for sgn in (+1.0, -1.0):
        h = sgn * scale
        eng.set_globals(d['Z'] + h * dZ, d['sf2'] * (1 + h * ds), d['alpha'] * (1 + h * da), d['beta'] * (1 + h * db), N_global=N)
        Fs.append(_run([eng], False)['F'])
    eng.close()
    probe.close()
    fd = (Fs[0] - Fs[1]) / (2 * scale)
    _report(directional_derivative={'fd': fd, 'analytic': ana, 'rel': abs(fd - ana) / abs(ana)})
    assert abs(fd - ana) <= 2e-5 * abs(ana) + 1e-9 * abs(ref['F']), 'directional derivative: fd %.10e vs analytic %.10e' % (fd, ana)


def test_one_resident_scg_iteration_at_full_size(data):
    """scg_adapted.py's loop with everything resident: two 5e5-point shards on one device, 2 x 5e5 x 50 embeddings, variances, search directions
    and gradient vectors in HBM; one iteration = the initial evaluation, the sigma-probe and the trial step."""
    from gparml_amd.driver import transform_back
    from gparml_amd.resident import ResidentCG, ResidentModel
    from gparml_amd.scg_adapted import SCG_adapted
    d = data
    S_raw = np.log(np.expm1(d['X_S']))
    h = N // 2
    shards = [(d['Y'][:h], d['X_mu'][:h], S_raw[:h]), (d['Y'][h:], d['X_mu'][h:], S_raw[h:])]
    model = ResidentModel(shards, M, Q, D, fixed_embeddings=False)
    calls = []
    try:
        x0 = np.concatenate([d['Z'].ravel(), [float(d['sf2'])], np.asarray(d['alpha'], dtype=float), [float(d['beta'])]])
        x0 = np.array([transform_back(b, v) for b, v in zip(model.bounds, x0)])

        def f_and_g(x, iteration, step_size=0):
            t = time.time()
            f, g = model.likelihood_and_gradient(x, iteration, step_size)
            calls.append((float(f), time.time() - t))
            return f, g

        x, flog, nfe, status = SCG_adapted(f_and_g, x0, ResidentCG(model), fixed_embeddings=False, maxiters=1, xtol=0, ftol=0, gtol=0)
        mu = model.engines[0].download('X_MU_TRIAL')
    finally:
        model.close()
    fl = [float(v) for v in flog]
    print('SCG at configs[4] per-GPU size: objective', fl, 'evaluations (f, s):', [(f, round(s, 2)) for f, s in calls])
    _report(scg_objective=fl, scg_evaluations=[{'f': f, 'wall_s': round(s, 2)} for f, s in calls])
    assert np.all(np.isfinite(fl)) and np.all(np.isfinite([c[0] for c in calls]))
    assert fl[-1] < fl[0], fl                                  # the accepted step decreased the objective (-F)
    assert np.max(np.abs(mu - d['X_mu'][:h])) > 0              # and the resident update reached the embeddings


def test_oracle_on_a_2e4_point_slice_of_the_same_workload(data):
    """The oracle (oracle/factorised.py, sharded over the host's threads) on the first 2e4 points of the full-size data with the same 1024
    inducing points: three launches of psi2_tile_kernel<52> per evaluation, N >> M."""
    from gparml_amd.engine import ShardEngine
    from oracle import factorised as Fz
    d = data
    n = 20000
    Y, mu, S = d['Y'][:n], d['X_mu'][:n], d['X_S'][:n]
    from threadpoolctl import threadpool_limits
    ncpu = os.cpu_count() or 8
    workers = min(32, ncpu)            # OpenBLAS is built for 64 caller threads: more concurrent callers corrupt its buffer table
    t = time.time()

    def live():
        with threadpool_limits(limits=max(1, ncpu // workers)):
            return Fz.evaluate_sharded(d['Z'], d['sf2'], d['alpha'], d['beta'], Y, mu, S, shards=64, workers=workers, pairs='gemm')

    # 95 s on 256 host threads when computed live: the oracle's outputs for this seeded slice are committed (tests/oracle_cache.py)
    ref = oracle_cache.get('config4_fullsize_slice_2e4', dict(Y=Y, X_mu=mu, X_S=S, Z=d['Z'], alpha=d['alpha'], beta=d['beta'], sf2=d['sf2']), live)
    t_ref = time.time() - t
    oracle_cache.done()
    eng = ShardEngine(n, D, M, Q)
    eng.upload_shard(Y, mu, S)
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    out = eng.evaluate(True)
    ms = eng.timings()['total_ms']
    eng.close()
    print('oracle on 2e4 points: %.1f s (live on %d host threads, or read from tests/golden/oracle_cache); device %.1f ms' % (t_ref, ncpu, ms))
    errs = {k: float(np.max(np.abs(np.asarray(out[k]) - np.asarray(ref[k]))) / np.max(np.abs(ref[k])))
            for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu', 'grad_X_S') if not hasattr(ref[k], 'rows')}
    _report(oracle_2e4={'oracle_s': round(t_ref, 1), 'host_threads': ncpu, 'device_ms': ms, 'F_rel': abs(out['F'] - ref['F']) / abs(ref['F']), 'errors': errs})
    assert_close(out['F'], ref['F'], 1e-6, what='F')
    for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta', 'grad_X_mu', 'grad_X_S'):
        assert_close(out[k], ref[k], 1e-5, what=k)
