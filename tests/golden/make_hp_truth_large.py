#!/usr/bin/env python3
"""Extended-precision truth at the benchmark's own configuration and FULL size (BASELINE configs[2]: N=1e6, D=100, M=512,
Q=10, alpha=0.1, beta=10, bench.py's generator), and the errors of the float64 CPU paths against it.

tests/golden/make_hp_golden.py does this at N = 4000 with numpy long double and the imported reference.  Beyond that the
reference cannot run (it stores an (N, M, M) tensor, partial_terms.py:45) and numpy's long-double matmul is too slow, so:

  truth        oracle/hp_truth.c -- the whole evaluation in x87 80-bit long double, streamed over N with OpenMP; run twice from
               one pass over the statistics (inducing points in their given and in reversed order): the difference is the
               truth's own uncertainty.  Checked against the N = 4000 fixture (agrees to 6e-9 on grad_Z, the fixture's own
               uncertainty being 8e-8).
  "reference"  float64, the reference's LU arrangement of the global step (inv / slogdet and its products,
               partial_terms.py:60,82,95,102-131,340-360,449-450 via oracle/literal.py) on BLAS-accumulated statistics
               (oracle/factorised.evaluate_blas(linalg='lu'))
  "port"       float64, Cholesky (oracle/factorised.evaluate_blas)

Stored (tests/golden/hp_truth_large_N<N>.npz, a few hundred KB): the generator's arguments and checksums of the inputs it
must reproduce, truth_{F, grad_Z, grad_alpha, grad_sf2, grad_beta}, truth_uncertainty, err_lu_*, err_chol_* (relative to each
block's largest magnitude), cond(Kmm), cond(Kmm + beta Psi2).  Inputs are NOT stored: tests regenerate them with
bench.synthetic(N, D, M, Q, seed=100) and check the checksums.

Usage (build container, 8 cores: ~10 min at N = 1e6):  python tests/golden/make_hp_truth_large.py [N [seed [z_seed]]]
  python tests/golden/make_hp_truth_large.py 50000 100 -1 1024     (M = 1024: hp_truth_M1024_N50000.npz)
(seed 100 with the benchmark's own inducing points is the default and keeps the file name hp_truth_large_N<N>.npz; other draws are
stored as hp_truth_large_N<N>_s<seed>_z<z_seed>.npz -- round 4: four more data / inducing-point draws at both sizes)
"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

BLOCKS = ('F', 'grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta')


def checksums(d):
    """Order-independent and order-dependent sums of the generated inputs (float64, exact to reproduce with the same numpy)."""
    w = np.cos(np.arange(d['Y'].shape[0], dtype=np.float64))
    return np.array([d['Y'].sum(), np.abs(d['Y']).sum(), w.dot(d['Y']).sum(), d['X_mu'].sum(), w.dot(d['X_mu']).sum(),
                     d['Z'].sum(), np.abs(d['Z']).sum()])


def rel(x, t):
    x, t = np.asarray(x, dtype=np.float64), np.asarray(t, dtype=np.float64)
    return float(np.max(np.abs(x - t)) / np.max(np.abs(t)))


def unpack(o, M, Q):
    n = 1 + M * Q + Q + 2
    hi, lo = o[:n], o[n:]
    def blk(v):
        return dict(F=v[0], grad_Z=v[1:1 + M * Q].reshape(M, Q), grad_alpha=v[1 + M * Q:1 + M * Q + Q], grad_sf2=v[-2], grad_beta=v[-1])
    return blk(hi), blk(lo)


def main():
    import bench
    from oracle import factorised as Fz
    N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
    D, M, Q = 100, 512, 10
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    z_seed = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) >= 0 else None
    if len(sys.argv) > 4:
        M = int(sys.argv[4])          # round 5: M = 1024 (the global step's int8 products, csrc/gsi8.hip) -> hp_truth_M1024_N<N>.npz
    d = bench.synthetic(N, D, M, Q, seed=seed, z_seed=z_seed)
    assert np.all(d['X_S'] == 0)
    exe = os.path.join(ROOT, 'oracle', '_build', 'hp_truth')
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(['gcc', '-O2', '-fopenmp', '-o', exe, os.path.join(ROOT, 'oracle', 'hp_truth.c'), '-lm'])
    work = tempfile.mkdtemp(prefix='hp_truth_')
    d['Y'].tofile(work + '/Y.bin'); d['X_mu'].tofile(work + '/X.bin'); d['Z'].tofile(work + '/Z.bin'); d['alpha'].tofile(work + '/alpha.bin')
    np.array([d['sf2'], d['beta']], dtype=np.float64).tofile(work + '/params.bin')
    t0 = time.time()
    subprocess.check_call([exe, work, str(N), str(D), str(M), str(Q)])
    print('[hp-large] long-double truth: %.0f s' % (time.time() - t0))
    tru, tru_lo = unpack(np.fromfile(work + '/truth_plain.bin'), M, Q)
    rev, rev_lo = unpack(np.fromfile(work + '/truth_reversed.bin'), M, Q)
    save = dict(N=np.int64(N), D=np.int64(D), M=np.int64(M), Q=np.int64(Q), seed=np.int64(seed), z_seed=np.int64(-1 if z_seed is None else z_seed),
                input_checksums=checksums(d))
    unc = {}
    for k in BLOCKS:
        save['truth_' + k] = np.asarray(tru[k], dtype=np.float64)
        diff = (np.asarray(rev[k]) - np.asarray(tru[k])) + (np.asarray(rev_lo[k]) - np.asarray(tru_lo[k]))
        unc[k] = float(np.max(np.abs(diff)) / np.max(np.abs(tru[k])))
    save['truth_uncertainty'] = np.array([unc[k] for k in BLOCKS])
    print('[hp-large] truth uncertainty (reversed vs given order):', {k: '%.1e' % v for k, v in unc.items()})
    Psi2 = np.fromfile(work + '/Psi2.bin').reshape(M, M)
    dz = d['Z'][:, None, :] - d['Z'][None, :, :]
    Kmm = d['sf2'] * np.exp(-0.5 * np.sum(d['alpha'][None, None, :] * dz * dz, axis=2))
    save['cond_Kmm'] = np.float64(np.linalg.cond(Kmm))
    save['cond_A'] = np.float64(np.linalg.cond(Kmm + d['beta'] * Psi2))
    print('[hp-large] cond(Kmm) %.2e  cond(Kmm + beta Psi2) %.2e' % (save['cond_Kmm'], save['cond_A']))
    part = {}
    for tag in ('Abar', 'Bbar', 'dFdK'):
        part[tag] = np.fromfile(work + '/%s_plain.bin' % tag).reshape(M, -1)
    for name, linalg in (('chol', 'cholesky'), ('lu', 'lu')):
        t0 = time.time()
        o = Fz.evaluate_blas(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], linalg=linalg)
        for k in BLOCKS:
            save['err_%s_%s' % (name, k)] = np.float64(rel(o[k], tru[k]))
        for tag, key in (('Abar', 'Abar'), ('Bbar', 'Bbar'), ('dFdK', 'dF_dKmm')):
            save['err_%s_%s' % (name, tag)] = np.float64(rel(o['gstep'][key], part[tag]))
        print('[hp-large] float64 %-8s (%.0f s): ' % (linalg, time.time() - t0) +
              '  '.join('%s %.2e' % (k, save['err_%s_%s' % (name, k)]) for k in BLOCKS + ('Abar', 'Bbar', 'dFdK')))
    out = os.path.join(HERE, 'hp_truth_%s_N%d%s.npz' % ('large' if M == 512 else 'M%d' % M, N,
                                                       '' if (seed, z_seed) == (100, None) else '_s%d_z%d' % (seed, -1 if z_seed is None else z_seed)))
    np.savez_compressed(out, **save)
    print('[hp-large] wrote', out, os.path.getsize(out), 'bytes;  scratch in', work)


if __name__ == '__main__':
    main()
