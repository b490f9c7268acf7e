import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from test_gpu_p1_i8 import _eval
from oracle import factorised as Fz
for (N, D, M, Q, sf2, al) in [(5003, 130, 600, 7, 2.5, 0.4), (5003, 130, 600, 7, 2.5, 1.0), (70000, 3, 512, 4, 0.7, 1.0), (1300, 40, 1024, 16, 1.0, 0.4), (9000, 40, 1024, 16, 1.0, 0.6)]:
  try:
    d = Fz.synthetic_shard(N, D, M, Q, regime='A', seed=11, zseed=12, alpha_value=al)
    d['sf2'] = sf2
    d['Y'] = d['Y'] * np.linspace(0.01, 30.0, D)[None, :]
    a, b = _eval(d, N, D, M, Q, True), _eval(d, N, D, M, Q, False)
    ref = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], want_embeddings=False)
    A = ref['gstep']['Kmm'] + d['beta'] * ref['stats']['sum_exp_K_mi_K_im']
    rel = lambda x, y: np.max(np.abs(np.asarray(x) - np.asarray(y))) / np.max(np.abs(y))
    print((N, D, M, Q), 'cond(A) %.1e cond(Kmm) %.1e' % (np.linalg.cond(A), np.linalg.cond(ref['gstep']['Kmm'])),
          'Psi2 i8 vs f64 %.1e' % rel(a['Psi2'], b['Psi2']), 'C %.1e' % rel(a['C'], b['C']),
          ' grad_Z: i8 vs oracle %.1e  f64 vs oracle %.1e  i8 vs f64 %.1e' % (rel(a['grad_Z'], ref['grad_Z']), rel(b['grad_Z'], ref['grad_Z']), rel(a['grad_Z'], b['grad_Z'])),
          ' F %.1e %.1e' % (abs(a['F'] - ref['F']) / abs(ref['F']), abs(b['F'] - ref['F']) / abs(ref['F'])), flush=True)

  except Exception as e:
    print((N, D, M, Q, al), 'FAILED', repr(e)[:200], flush=True)
