#!/usr/bin/env python3
"""Capture golden vectors of the reference's WHOLE evaluation pipeline (parallel_GPLVM.likelihood_and_gradient through
local_MapReduce) by running the reference in memory.  Build container only (needs /root/reference).

Every module is passed through lib2to3 in memory (print statements, tuple parameters, dict.items() lists, integer
division) and exec'd into sys.modules; scipy.save/load/ones/randn are re-pointed to numpy.
For each likelihood_and_gradient call we record: the flat parameter vector, iteration, step size, the per-shard
files the mappers read (embedding, raw variance, search direction) and everything they produce (f, gradient, the 12
accumulated statistics, the per-shard .grad_latest).  Only numbers are stored (tests/golden/pipe_*.npz).
"""
import builtins
import os
import shutil
import sys
import tempfile
import types
import warnings

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
MODS = ['kernels', 'kernel_exp', 'partial_terms', 'nputil', 'supporting_functions', 'local_MapReduce',
        'scg_adapted_local_MapReduce', 'scg_adapted', 'gd_local_MapReduce', 'gd', 'parallel_GPLVM']


# in-memory source substitutions on top of lib2to3 (python-2 semantics it cannot see): numpy '== None' tests, integer '/' in slice bounds
PATCHES = {'kernels': [('if ard==None:', 'if ard is None:'), ('if X2==None:', 'if X2 is None:')],
           'predict': [('len(flat_array_transformed)/2', 'len(flat_array_transformed)//2'), ('if mask == None:', 'if mask is None:')]}


def _patched(name, src):
    for old, new in PATCHES.get(name, ()):
        src = src.replace(old, new)
    return src


def load_reference():
    warnings.filterwarnings('ignore')
    from lib2to3.refactor import RefactoringTool, get_fixers_from_package
    import scipy
    sys.dont_write_bytecode = True
    builtins.xrange = range
    for name, fn in (('save', np.save), ('load', np.load), ('ones', np.ones), ('zeros', np.zeros), ('randn', np.random.randn)):
        if not hasattr(scipy, name):
            setattr(scipy, name, fn)
    tool = RefactoringTool(get_fixers_from_package('lib2to3.fixes'))
    mods = {}
    for name in MODS:
        src = _patched(name, open(os.path.join(REF, name + '.py')).read())
        src = str(tool.refactor_string(src + '\n', name))
        mod = types.ModuleType(name)
        mod.__file__ = os.path.join(REF, name + '.py')
        sys.modules[name] = mod
        mods[name] = mod
    for name in MODS:                       # exec after all are registered so intra-reference imports resolve
        src = _patched(name, open(os.path.join(REF, name + '.py')).read())
        src = str(tool.refactor_string(src + '\n', name))
        if name == 'nputil':
            src = src.replace("np.seterr(all='raise')", "pass")
        exec(compile(src, mods[name].__file__, 'exec'), mods[name].__dict__)
    return mods


def run(name, seed, shards, D, M, Q, fixed, iterations, optimiser='SCG_adapted', prefix='pipe'):
    mods = load_reference()
    pg, lmr = mods['parallel_GPLVM'], mods['local_MapReduce']
    work = tempfile.mkdtemp(prefix='gparml_gold_')
    dirs = {k: os.path.join(work, k) for k in ('input', 'embeddings', 'statistics', 'tmp')}
    for d in dirs.values():
        os.makedirs(d)
    rs = np.random.RandomState(seed)
    Ys, Xs = [], []
    W = rs.randn(Q, D)
    for i, n in enumerate(shards):
        X = rs.randn(n, Q)
        Y = np.sin(X.dot(W)) + 0.1 * rs.randn(n, D)
        np.savetxt(os.path.join(dirs['input'], 'shard_%d' % i), Y, delimiter=',', fmt='%.17g')
        Ys.append(Y)
        Xs.append(X)
        if fixed:
            np.save(os.path.join(dirs['embeddings'], 'shard_%d.embedding.npy' % i), X + 0.05 * rs.randn(n, Q))
    options = dict(input=dirs['input'], embeddings=dirs['embeddings'], statistics=dirs['statistics'], tmp=dirs['tmp'], parallel='local',
                   iterations=iterations, keep=True, load=False, init='PCA', optimiser=optimiser, drop_out_fraction=0,
                   local_no_pool=False, M=M, Q=Q, D=D, fixed_embeddings=fixed, fixed_beta=False)
    rec = {'n_shards': np.int64(len(shards)), 'D': np.int64(D), 'M': np.int64(M), 'Q': np.int64(Q), 'fixed': np.int64(fixed)}
    for i, Y in enumerate(Ys):
        rec['Y_%d' % i] = Y
    calls = []
    orig = pg.likelihood_and_gradient

    def snapshot(prefix, k):
        for i in range(len(shards)):
            base = os.path.join(dirs['embeddings'], 'shard_%d' % i)
            for ext in ('embedding', 'variance', 'grad_d', 'grad_latest'):
                f = base + '.' + ext + '.npy'
                if os.path.exists(f):
                    rec['call%d_%s_shard%d_%s' % (k, prefix, i, ext)] = np.load(f)

    def wrapped(flat_array, iteration, step_size=0):
        k = len(calls)
        snapshot('in', k)
        f, g = orig(flat_array, iteration, step_size)
        rec['call%d_x' % k] = np.array(flat_array, dtype=float)
        rec['call%d_iter' % k] = np.array(-2 if iteration == 'f' else iteration, dtype=np.int64)
        rec['call%d_step' % k] = np.array(step_size, dtype=float)
        rec['call%d_f' % k] = np.array(f, dtype=float)
        rec['call%d_g' % k] = np.array(g, dtype=float)
        snapshot('out', k)
        it = 'f' if iteration == 'f' else str(iteration)
        for key in pg.options['accumulated_statistics_names']:
            rec['call%d_acc_%s' % (k, key)] = np.load(os.path.join(dirs['statistics'], 'accumulated_statistics_%s_%s.npy' % (key, it)))
        for key in pg.options['partial_derivatives_names']:
            rec['call%d_pd_%s' % (k, key)] = np.load(os.path.join(dirs['statistics'], 'partial_derivatives_%s_%s.npy' % (key, it)))
        calls.append(k)
        return f, g

    pg.likelihood_and_gradient = wrapped
    np.random.seed(seed)
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        pg.main(options)
    rec['n_calls'] = np.int64(len(calls))
    rec['N'] = np.int64(sum(shards))
    path = os.path.join(HERE, '%s_%s.npz' % (prefix, name))
    np.savez_compressed(path, **rec)
    shutil.rmtree(work)
    print('%-12s %d calls, f: %s -> %s (%d bytes)' % (name, len(calls), rec['call0_f'], rec['call%d_f' % (len(calls) - 1)], os.path.getsize(path)))


if __name__ == '__main__':
    import multiprocessing
    multiprocessing.set_start_method('fork')
    run('gplvm_2shards', 21, (30, 26), 3, 4, 2, False, 2)
    run('sparsegp_2shards', 22, (40, 33), 4, 6, 3, True, 2)
    run('config1_1shard', 23, (120,), 4, 2, 2, False, 2)
    # the gradient-descent optimiser (parallel_GPLVM.py:104-105, gd.py): accepted and rejected steps, free embeddings
    run('gplvm_2shards', 24, (28, 31), 3, 4, 2, False, 9, optimiser='GD', prefix='gdpipe')
