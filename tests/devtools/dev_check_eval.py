"""Developer check (GPU box): full single-shard evaluation against the factorised CPU oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from gparml_amd.engine import ShardEngine
from oracle import factorised as Fz

def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)

ok = True
cases = [(300, 5, 20, 3, 'A', False), (300, 5, 20, 3, 'A', True), (1000, 7, 130, 10, 'A', False), (1000, 7, 130, 10, 'A', True),
         (5000, 100, 512, 10, 'A', False), (777, 3, 5, 1, 'A', True), (2000, 10, 128, 13, 'A', False), (900, 4, 64, 20, 'A', True)]
if len(sys.argv) > 1 and sys.argv[1] == 'B':
    cases = [(300, 5, 20, 3, 'B', True), (1000, 7, 130, 10, 'B', True), (500, 4, 2, 2, 'B', True), (400, 3, 20, 20, 'B', True), (300, 2, 12, 40, 'B', True)]
for (N, D, M, Q, regime, emb) in cases:
    d = Fz.synthetic_shard(N, D, M, Q, regime=regime, seed=1, zseed=2)
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    eng.phase1(); eng.global_step(); eng.phase2(emb)
    out = eng.finish()
    ref = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], want_embeddings=emb)
    e = dict(F=abs(out['F'] - ref['F']) / abs(ref['F']), gZ=rel(out['grad_Z'], ref['grad_Z']), ga=rel(out['grad_alpha'], ref['grad_alpha']),
             gsf2=abs(out['grad_sf2'] - ref['grad_sf2']) / abs(ref['grad_sf2']), gbeta=abs(out['grad_beta'] - ref['grad_beta']) / abs(ref['grad_beta']))
    if emb:
        e['gmu'] = rel(eng.download('GRAD_X_MU'), ref['grad_X_mu'])
        if regime == 'B':
            e['gS'] = rel(eng.download('GRAD_X_S'), ref['grad_X_S'])
    good = all(v < 1e-7 for v in e.values())
    ok &= good
    print('N=%d D=%d M=%d Q=%d %s emb=%d: %s %s' % (N, D, M, Q, regime, emb, ' '.join('%s=%.1e' % kv for kv in e.items()), 'OK' if good else 'FAIL'))
    eng.close()
sys.exit(0 if ok else 1)
