"""Contexts driven CONCURRENTLY from several host threads on one GPU (what gpu_MapReduce._for_each does with one thread per shard; the ctypes calls release
the GIL), round 6: every thread owns one context of its own shape, all start together -- in a FRESH process, so the first launch of every kernel happens under
contention -- and evaluate REPS times; every result must be bit-identical to the same shape evaluated alone in another fresh process.
usage (GPU box): python3 tools/stress_threads.py [REPS]        (the solo runs are child processes of this script: --solo INDEX)"""
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (N, D, M, Q, regime, alpha): the headline kernels, the generic fixed-embedding kernel, tile pairs, columns, tiles, eight panels
SHAPES = [(20000, 100, 512, 10, 'A', 0.3), (5000, 10, 128, 13, 'A', 0.2), (3000, 3, 512, 10, 'B', 0.3), (3000, 3, 512, 14, 'B', 0.3), (1500, 3, 256, 33, 'B', 0.05),
          (1100, 2, 1024, 8, 'B', 0.8), (4000, 30, 300, 5, 'A', 0.5), (2000, 3, 200, 6, 'B', 0.3)]
KEYS = ('grad_Z', 'grad_alpha')


def make(i):
    from gparml_amd.engine import ShardEngine
    from oracle import factorised as Fz
    N, D, M, Q, regime, alpha = SHAPES[i]
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=41 + i, zseed=51 + i, alpha_value=alpha)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    return eng, regime == 'B'


def digest(out):
    return [float(out['F'])] + [np.asarray(out[k], dtype=float).tobytes().hex() for k in KEYS]


def solo(i):
    eng, emb = make(i)
    out = eng.evaluate(emb)
    np.save('/tmp/stress_threads_solo_%d.npy' % i, np.array(digest(out), dtype=object), allow_pickle=True)
    eng.close()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == '--solo':
        return solo(int(sys.argv[2]))
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    t0 = time.time()
    for i in range(len(SHAPES)):
        subprocess.run([sys.executable, os.path.abspath(__file__), '--solo', str(i)], check=True, cwd=ROOT)
    want = [list(np.load('/tmp/stress_threads_solo_%d.npy' % i, allow_pickle=True)) for i in range(len(SHAPES))]
    engines = [make(i) for i in range(len(SHAPES))]          # contexts created, nothing evaluated yet: every kernel's first launch happens in the threads
    start = threading.Barrier(len(SHAPES))
    bad, jit = [], []

    def work(i):
        eng, emb = engines[i]
        start.wait()
        for rep in range(reps):
            out = eng.evaluate(emb)
            if digest(out) != want[i]:
                bad.append((SHAPES[i], rep, float(out['F']), want[i][0]))
            if eng.last_jitter:
                jit.append((SHAPES[i], rep, eng.last_jitter))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(SHAPES))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for eng, _ in engines:
        eng.close()
    for b in bad[:10]:
        print('NOT IDENTICAL', b)
    print('STRESS_THREADS %d threads x %d evaluations, first evaluations concurrent in a fresh process: %d not bit-identical to the solo run, %d jitter evaluations, %.0f s'
          % (len(SHAPES), reps, len(bad), len(jit), time.time() - t0), flush=True)
    return 1 if bad or jit else 0


if __name__ == '__main__':
    sys.exit(main())
