"""Developer probe (GPU box): where does the first NaN appear when allocations and per-evaluation buffers are poisoned (gp_debug_set_option
"poison_alloc")?  One evaluation stage by stage; after each stage the padded device images (gp_debug_peek) are scanned: non-finite entries in the
REAL region are bugs of the stage that wrote it (or of what it read), non-finite entries in the padding are harmless unless a later stage reads them.
usage: python3 tests/devtools/dev_poison_probe.py [A|Ae|B] N D M Q"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gparml_amd import _lib                     # noqa: E402
from gparml_amd.engine import ShardEngine       # noqa: E402
from oracle import factorised as Fz             # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'A'
N, D, M, Q = [int(x) for x in sys.argv[2:6]] if len(sys.argv) > 5 else (1000, 7, 130, 10)
lib = _lib.load()
assert lib.gp_debug_set_option(b'poison_alloc', 1) == 0
regime = 'B' if mode == 'B' else 'A'
emb = mode in ('Ae', 'B')
d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=11, zseed=12, alpha_value=0.3)
eng = ShardEngine(N, D, M, Q)
Np, Mp, Dp = (N + 127) // 128 * 128, (M + 127) // 128 * 128, (D + 127) // 128 * 128
CXp = (2 * Q + 1 + 3) // 4 * 4


def peek(name, cnt):
    buf = np.empty(cnt, dtype=np.float64)
    rc = lib.gp_debug_peek(eng.h, name.encode(), buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), buf.size)
    if rc != 0:
        return None
    return buf


def report(stage, items):
    print('--- after', stage)
    for name, cnt, shape, real in items:
        b = peek(name, cnt)
        if b is None:
            print('  %-8s (not available)' % name)
            continue
        a = b[:int(np.prod(shape))].reshape(shape)
        bad = ~np.isfinite(a)
        mask = np.zeros(shape, dtype=bool)
        mask[real] = True
        print('  %-8s non-finite: real region %d of %d, padding %d of %d' % (name, int(bad[mask].sum()), int(mask.sum()), int(bad[~mask].sum()), int((~mask).sum())))


mm, md = Mp * Mp, Mp * Dp
eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
report('set_globals', [('Kaug', Np * (Mp + Dp), (Np, Mp + Dp), (slice(0, N), slice(Mp, Mp + D))), ('Z', Mp * Q, (Mp, Q), (slice(0, M), slice(None))),
                       ('Zaug', Mp * CXp, (Mp, CXp), (slice(0, M), slice(0, 2 * Q + 1)))])
eng.phase1()
st = peek('stats', mm + md + 8)
report('phase1', [('Kaug', Np * (Mp + Dp), (Np, Mp + Dp), (slice(0, N), slice(0, M))), ('mu', Np * Q, (Np, Q), (slice(0, N), slice(None))),
                  ('S', Np * Q, (Np, Q), (slice(0, N), slice(None))), ('Xa', Np * CXp, (Np, CXp), (slice(0, N), slice(0, 2 * Q + 1))),
                  ('LE', Np * Mp, (Np, Mp), (slice(0, N), slice(0, M))), ('LEA', Np * Mp, (Np, Mp), (slice(0, N), slice(0, M)))])
if st is not None:
    P2, C, sc = st[:mm].reshape(Mp, Mp), st[mm:mm + md].reshape(Mp, Dp), st[mm + md:]
    print('  stats: Psi2 real %d pad %d | C real %d pad %d | scalars %s' % ((~np.isfinite(P2[:M, :M])).sum(), (~np.isfinite(P2)).sum() - (~np.isfinite(P2[:M, :M])).sum(),
                                                                          (~np.isfinite(C[:M, :D])).sum(), (~np.isfinite(C)).sum() - (~np.isfinite(C[:M, :D])).sum(), sc[:4]))
try:
    eng.global_step()
except Exception as e:      # noqa: BLE001
    print('global_step raised', type(e).__name__, e)
items = [(n, 2 * mm, (2, Mp, Mp), (slice(None), slice(0, M), slice(0, M))) for n in ('Kmm', 'Linv', 'Inv')]
items += [(n, mm, (Mp, Mp), (slice(0, M), slice(0, M))) for n in ('KmmKeep', 'T1', 'T2', 'dFdK', 'Bbar')]
items += [(n, md, (Mp, Dp), (slice(0, M), slice(0, D))) for n in ('E', 'PsiE', 'Abar')]
items += [('Bm', (Mp + Dp) * Mp, (Mp + Dp, Mp), (slice(0, Mp + D), slice(0, M))), ('gK', M * Q + Q, (M * Q + Q,), (slice(None),)), ('gs', 24, (24,), (slice(0, 13),))]
report('global_step', items)
try:
    eng.phase2(emb)
    report('phase2', [('grads', M * Q + Q, (M * Q + Q,), (slice(None),))])
    out = eng.finish()
    print('F', out['F'], 'grad_Z finite', np.all(np.isfinite(out['grad_Z'])), 'grad_alpha', out['grad_alpha'])
    if emb:
        print('grad_X_mu non-finite', (~np.isfinite(eng.download('GRAD_X_MU'))).sum(), 'grad_X_S', (~np.isfinite(eng.download('GRAD_X_S'))).sum())
except Exception as e:      # noqa: BLE001
    print('phase2 / finish raised', type(e).__name__, e)
eng.close()
