"""Shared helpers to replay the pipeline goldens (tests/golden/pipe_*.npz, captured by make_pipeline_golden.py from the
reference's parallel_GPLVM.likelihood_and_gradient + local_MapReduce)."""
import os

import numpy as np

from conftest import GOLDEN_DIR

BOUND_POS = (0, None)


def pipeline_names():
    return sorted(f[5:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith('pipe_') and f.endswith('.npz'))


def load_pipeline(name):
    z = np.load(os.path.join(GOLDEN_DIR, 'pipe_%s.npz' % name))
    return {k: z[k] for k in z.files}


def write_call_state(g, k, work):
    """Materialise the directories the mappers read for call k; returns the options dict (parallel_GPLVM.py:414-460)."""
    dirs = {d: os.path.join(work, d) for d in ('input', 'embeddings', 'statistics', 'tmp')}
    for d in dirs.values():
        os.makedirs(d, exist_ok=True)
    ns = int(g['n_shards'])
    for i in range(ns):
        np.savetxt(os.path.join(dirs['input'], 'shard_%d' % i), g['Y_%d' % i], delimiter=',', fmt='%.17g')
        base = os.path.join(dirs['embeddings'], 'shard_%d' % i)
        for ext in ('embedding', 'variance', 'grad_d'):
            key = 'call%d_in_shard%d_%s' % (k, i, ext)
            f = base + '.' + ext + '.npy'
            if key in g:
                np.save(f, g[key])
            elif os.path.exists(f):
                os.remove(f)
    return dict(input=dirs['input'], embeddings=dirs['embeddings'], statistics=dirs['statistics'], tmp=dirs['tmp'], parallel='local',
                keep=True, load=False, M=int(g['M']), Q=int(g['Q']), D=int(g['D']), N=int(g['N']), fixed_embeddings=bool(g['fixed']),
                fixed_beta=False, drop_out_fraction=0)


def call_args(g, k):
    it = int(g['call%d_iter' % k])
    return g['call%d_x' % k], ('f' if it == -2 else it), float(g['call%d_step' % k])
