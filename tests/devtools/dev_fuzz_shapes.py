"""Dev fuzz (GPU box): random small shapes in both regimes against the oracle -- looks for crashes / wrong answers at odd sizes
(N below a tile, M = 1, M > N, Q up to 63, D up to 300).  r03: 70 shapes, 68 within 1e-5 (most 1e-12), two flagged with cond 2e14 / 9e11.  Usage: dev_fuzz_shapes.py [count] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz
count = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(count):
    regime = 'AB'[rs.randint(2)]
    Q = int(rs.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 20, 23, 24, 25, 30, 31, 32, 36, 39, 40, 45, 50, 51, 52, 55, 60, 63]))
    M = int(rs.choice([1, 2, 7, 16, 33, 64, 65, 100, 128, 129, 200, 257, 300]))
    N = int(rs.choice([1, 2, 17, 63, 64, 127, 128, 129, 300, 777, 1500]))
    D = int(rs.choice([1, 2, 3, 4, 5, 15, 16, 17, 33, 100, 104, 105, 130, 300]))
    big = len(sys.argv) > 3 and sys.argv[3] == 'big'
    if big:                                         # several row tiles / inducing tiles / slices
        N = int(rs.choice([2000, 5000, 12000, 33000])); M = int(rs.choice([130, 257, 513, 700, 1025, 1100])); D = int(rs.choice([10, 100, 333, 1000]))
        Q = int(rs.choice([2, 5, 6, 8, 10, 12, 14, 16, 20, 24, 30, 36, 50, 60]))
        if regime == 'B': N = min(N, 5000); M = min(M, 513 if Q <= 24 else 257)
    elif regime == 'B' and Q > 24 and M > 130:      # keep the oracle's pairwise tensor small
        M = 64
    d = Fz.synthetic_shard(N, D, min(M, N), Q, regime=regime, seed=100 + it, zseed=200 + it, alpha_value=(min(1.0, 6.0 / Q) if big else min(0.5, 2.0 / Q)))
    if M > N:
        d['Z'] = 1.5 * np.random.RandomState(300 + it).randn(M, Q)      # more inducing points than data points
    try:
        ref = Fz.evaluate_sharded(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], shards=16 if big else 4, pairs='gemm')
    except np.linalg.LinAlgError:
        print('skip (oracle: not PD)', (N, D, M, Q, regime), flush=True)
        continue
    eng = ShardEngine(N, D, M, Q)
    try:
        eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
        eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
        emb = regime == 'B' or bool(rs.randint(2))      # regime A: half the shapes with fixed embeddings (p2_fast8_kernel / p2_gen8_kernel<false>)
        out = eng.evaluate(emb)
        errs = {k: float(np.max(np.abs(np.asarray(out[k]) - np.asarray(ref[k]))) / max(np.max(np.abs(ref[k])), 1e-300))
                for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta') + (('grad_X_mu',) if emb else ()) + (('grad_X_S',) if regime == 'B' else ())}
        errs['F'] = abs(out['F'] - ref['F']) / abs(ref['F'])
        worst = max(errs.values())
        flag = ''
        if worst >= 1e-5:
            dz = d['Z'][:, None, :] - d['Z'][None, :, :]
            Kmm = d['sf2'] * np.exp(-0.5 * np.sum(np.asarray(d['alpha'])[None, None, :] * dz * dz, axis=2))
            cond = np.linalg.cond(Kmm + d['beta'] * ref['stats']['sum_exp_K_mi_K_im'])
            flag = '   <<<<<< %s, cond(Kmm + beta Psi2) = %.1e%s' % (max(errs, key=errs.get), cond, ' (float64 carries no digits there)' if cond > 1e11 else '')
            if cond <= 1e11:
                # (r06) the float64 oracle is not the truth at such conditioning: arbitrate with the long-double bound (dev_case_truth.py) where the case is small.
                # Two cases of seeds 606 / 607 "failed" with the ORACLE 0.63 / 2.6e-3 off the truth and the library 1.2e-4 / 1.6e-5 (profiles/r06_fuzz_shapes.txt).
                if max(errs, key=errs.get) == 'grad_Z' and N * M * M <= 2e6 and M * Q <= 500:
                    from dev_case_truth import grad_Z_truth
                    _, g = grad_Z_truth(d)
                    e_lib, e_ora = (float(np.max(np.abs(np.asarray(x) - g)) / np.max(np.abs(g))) for x in (out['grad_Z'], ref['grad_Z']))
                    flag += '  | against the long-double truth: library %.1e, oracle %.1e' % (e_lib, e_ora)
                    if e_lib > max(1e-5, e_ora): bad += 1
                else:
                    bad += 1
        print((N, D, M, Q, regime), 'worst %.1e%s' % (worst, flag), flush=True)
    except Exception as e:
        bad += 1
        print((N, D, M, Q, regime), 'EXCEPTION', type(e).__name__, str(e)[:200], flush=True)
    finally:
        eng.close()
print('FUZZ done, flagged', bad)
