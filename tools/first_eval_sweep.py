"""Cold-start sweep (round 6, GPU box): every kernel family's FIRST evaluation in a fresh process against the second and third of the same process.
The race of profiles/r06_first_evaluation_race.txt was invisible to every warm-process check (stress, shuffled suites, poison); this runs one fresh child per shape
and repetition: the first evaluation must be bit-identical to the next two and take the same branch (the jitter mask the same in all three: some of these
shapes -- 100 inducing points in two dimensions -- need the reference's jitter every time).  usage: python3 tools/first_eval_sweep.py [repetitions]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_first_evaluation import CHILD   # noqa: E402

# (N, D, M, Q, regime, alpha), environment -- fixed embeddings: p2_fast8 (Q <= 11), p2_gen8 (Q > 11), M from one panel to sixteen, D below / above 128;
# free embeddings: tile pairs (Q <= 12), columns (13..16 and M = 1024 at Q = 10), tiles (17..63), generic (>= 64), forced modes
SHAPES = [((3000, 3, 100, 2, 'A', 0.5), {}), ((5000, 10, 128, 3, 'A', 0.5), {}), ((4096, 100, 256, 10, 'A', 0.3), {}), ((4096, 100, 384, 10, 'A', 0.3), {}),
          ((4096, 100, 512, 10, 'A', 0.3), {}), ((4096, 200, 640, 11, 'A', 0.3), {}), ((3000, 30, 896, 5, 'A', 0.5), {}), ((3000, 1000, 1024, 12, 'A', 0.3), {}),
          ((3000, 5, 1536, 4, 'A', 0.8), {}), ((3000, 5, 2048, 4, 'A', 0.8), {}), ((3000, 16, 512, 20, 'A', 0.2), {}),
          ((3000, 3, 100, 2, 'B', 0.5), {}), ((3000, 3, 128, 4, 'B', 0.5), {}), ((3000, 3, 256, 6, 'B', 0.5), {}), ((3000, 3, 384, 8, 'B', 0.5), {}),
          ((3000, 3, 512, 10, 'B', 0.3), {}), ((3000, 3, 512, 12, 'B', 0.3), {}), ((3000, 3, 512, 14, 'B', 0.3), {}), ((3000, 3, 512, 16, 'B', 0.2), {}),
          ((3000, 3, 640, 9, 'B', 0.3), {}), ((2000, 3, 896, 5, 'B', 0.5), {}), ((1100, 2, 1024, 8, 'B', 0.8), {}), ((1100, 2, 1024, 10, 'B', 0.8), {}),
          ((1700, 2, 1536, 4, 'B', 0.8), {}), ((2100, 2, 2048, 3, 'B', 0.8), {}),
          ((2000, 3, 256, 20, 'B', 0.1), {}), ((2000, 3, 512, 33, 'B', 0.05), {}), ((1500, 3, 256, 52, 'B', 0.05), {}), ((1000, 3, 128, 70, 'B', 0.03), {}),
          ((9000, 3, 200, 6, 'B', 0.3), {'GPARML_B_PHASE2': 'tiles'}), ((3000, 3, 512, 10, 'B', 0.3), {'GPARML_B_PHASE2': 'cols'}),
          ((4096, 100, 512, 10, 'A', 0.3), {'GPARML_P1_I8': '1'}), ((3000, 100, 1024, 10, 'A', 0.3), {'GPARML_GS_I8': '0'})]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    t0, bad, runs = time.time(), [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for i, (shape, env) in enumerate(SHAPES):
            script = os.path.join(tmp, 'first_%d.py' % i)
            open(script, 'w').write(CHILD % {'root': ROOT, 'shape': shape, 'jitter_ok': True})   # a shape that needs the jitter needs it every time
            res = []
            for _ in range(reps):
                r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, **env))
                line = [ln for ln in r.stdout.splitlines() if ln.startswith('FIRST_EVAL')]
                runs += 1
                res.append('ok' if r.returncode == 0 else 'BAD')
                if r.returncode != 0:
                    bad.append((shape, env, line[0] if line else r.stderr[-300:]))
            print(shape, env, ' '.join(res), flush=True)
    for b in bad:
        print('BAD', b)
    print('FIRST_EVAL_SWEEP %d shapes x %d fresh processes: %d bad, %d s' % (len(SHAPES), reps, len(bad), time.time() - t0))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
