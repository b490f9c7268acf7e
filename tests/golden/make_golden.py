#!/usr/bin/env python3
"""Generate golden vectors for the partial_terms hot path by IMPORTING the reference.

Runs only in the build container (needs /root/reference); the GPU box never sees the reference.
Nothing of the reference (source, bytecode, converted text) is written into the repo: only the
seeded inputs and the numeric outputs of its functions, as small .npz files next to this script.

Loader recipe: the reference is Python 2; partial_terms.py and
kernel_exp.py import unmodified once ``builtins.xrange = range``; kernels.py needs its two
``== None`` tests (kernels.py:20, :89) turned into ``is None`` because numpy compares elementwise.
Sources are patched IN MEMORY and exec'd into fresh modules registered in sys.modules.

Usage:  python tests/golden/make_golden.py            (rewrites tests/golden/pt_*.npz)
"""
import builtins
import os
import sys
import types

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sys.dont_write_bytecode = True
    builtins.xrange = range
    mods = {}
    for name in ('kernels', 'kernel_exp', 'partial_terms'):
        with open(os.path.join(REF, name + '.py')) as f:
            src = f.read()
        if name == 'kernels':
            src = src.replace('if ard==None:', 'if ard is None:').replace('if X2==None:', 'if X2 is None:')
        mod = types.ModuleType(name)
        mod.__file__ = os.path.join(REF, name + '.py')
        sys.modules[name] = mod
        exec(compile(src, mod.__file__, 'exec'), mod.__dict__)
        mods[name] = mod
    return mods


def make_case(rs, N, D, M, Q, regime, N_global=None, alpha_scale=1.0):
    """Seeded inputs in the spirit of test.py:24-60 (random hypers, Z ~ N(0,1))."""
    X = rs.randn(N, Q)
    W = rs.randn(Q, D)
    Y = np.sin(X.dot(W)) + 0.1 * rs.randn(N, D)
    X_mu = X + 0.05 * rs.randn(N, Q)
    if regime == 'A':
        X_S = np.zeros((N, Q))
    elif regime == 'const':
        X_S = 0.2 * np.ones((N, Q))              # test.py:60
    else:
        X_S = rs.uniform(0.05, 0.55, size=(N, Q))
    Z = rs.randn(M, Q)
    sf2 = float((0.5 + np.exp(0.3 * rs.randn())) ** 2)
    alpha = alpha_scale * np.exp(0.5 * rs.randn(Q))
    beta = float(rs.uniform(2.0, 20.0))
    return dict(Y=Y, X_mu=X_mu, X_S=X_S, Z=Z, sf2=sf2, alpha=alpha, beta=beta,
                N=np.int64(N if N_global is None else N_global), D=np.int64(D), M=np.int64(M), Q=np.int64(Q))


def run_reference(mods, c):
    pt_mod = mods['partial_terms']
    kexp = mods['kernel_exp']
    M, Q, N, D = int(c['M']), int(c['Q']), int(c['N']), int(c['D'])
    pt = pt_mod.partial_terms(c['Z'].copy(), c['sf2'], c['alpha'].copy(), c['beta'], M, Q, N, D)
    pt.set_data(c['Y'], c['X_mu'], c['X_S'], True)
    o = {}
    o['Kmm'] = pt.Kmm
    o['Kmm_inv'] = pt.Kmm_inv
    o['exp_K_mi'] = pt.exp_K_mi
    o['exp_K_mi_K_im'] = pt.exp_K_mi_K_im
    o['psi2_scalar_point0'] = kexp.calc_expect_K_mi_K_im_old(c['Z'], pt.hyp, c['X_mu'][:1], c['X_S'][:1])
    st = pt.get_local_statistics()
    o['sum_YYT'] = np.float64(st['sum_YYT'])
    o['sum_exp_K_mi_K_im'] = st['sum_exp_K_mi_K_im']
    o['exp_K_miY'] = st['exp_K_miY']
    o['sum_exp_K_ii'] = np.float64(st['sum_exp_K_ii'])
    o['KL'] = np.float64(st['KL'])
    o['Kmm_plus_op_inv'] = pt.Kmm_plus_op_inv
    o['F'] = np.float64(pt.logmarglik())
    o['dF_dKmm'] = pt.dF_dKmm()
    o['dF_dexp_K_miY'] = pt.dF_dexp_K_miY()
    o['dF_dexp_K_mi_K_im'] = pt.dF_dexp_K_mi_K_im()
    o['dF_dexp_K_ii'] = np.float64(pt.dF_dexp_K_ii())
    o['dKmm_dZ'] = pt.dKmm_dZ()
    o['dexp_K_miY_dZ'] = pt.dexp_K_miY_dZ()
    o['dexp_K_mi_K_im_dZ'] = pt.dexp_K_mi_K_im_dZ()
    o['grad_Z'] = pt.grad_Z(o['dF_dKmm'], o['dKmm_dZ'], o['dF_dexp_K_miY'], o['dexp_K_miY_dZ'],
                            o['dF_dexp_K_mi_K_im'], o['dexp_K_mi_K_im_dZ'])
    o['dKmm_dalpha'] = pt.dKmm_dalpha()
    o['dexp_K_miY_dalpha'] = pt.dexp_K_miY_dalpha()
    o['dexp_K_mi_K_im_dalpha'] = pt.dexp_K_mi_K_im_dalpha()
    o['grad_alpha'] = pt.grad_alpha(o['dF_dKmm'], o['dKmm_dalpha'], o['dF_dexp_K_miY'], o['dexp_K_miY_dalpha'],
                                    o['dF_dexp_K_mi_K_im'], o['dexp_K_mi_K_im_dalpha'])
    o['dKmm_dsf2'] = pt.dKmm_dsf2()
    o['dexp_K_miY_dsf2'] = pt.dexp_K_miY_dsf2()
    o['dexp_K_mi_K_im_dsf2'] = pt.dexp_K_mi_K_im_dsf2()
    o['dexp_K_ii_dsf2'] = np.int64(pt.dexp_K_ii_dsf2())
    o['grad_sf2'] = np.float64(pt.grad_sf2(o['dF_dKmm'], o['dKmm_dsf2'], o['dF_dexp_K_ii'], o['dexp_K_ii_dsf2'],
                                           o['dF_dexp_K_miY'], o['dexp_K_miY_dsf2'],
                                           o['dF_dexp_K_mi_K_im'], o['dexp_K_mi_K_im_dsf2']))
    o['grad_beta'] = np.float64(pt.grad_beta())
    o['grad_X_mu'] = pt.grad_X_mu()
    if not np.all(c['X_S'] == 0):
        o['grad_X_S'] = pt.grad_X_S()
    return o


CASES = [
    # name,            seed, N,  D, M,  Q, regime, N_global, alpha_scale
    ('testpy_fixture', 11,   5,  7, 10, 2, 'const', None, 1.0),     # sizes of test.py:24-45
    ('config1_small',  12,   40, 4, 2,  2, 'B',     None, 1.0),     # README.txt:25 sizes, N shrunk
    ('regimeA_small',  13,   12, 4, 6,  3, 'A',     None, 1.0),     # fixed embeddings, X_S == 0
    ('q1_m1',          14,   6,  3, 1,  1, 'B',     None, 1.0),     # degenerate M=1, Q=1
    ('mid_q10',        15,   30, 5, 16, 10, 'B',    None, 0.1),     # Q=10 like configs 2-4
    ('regimeA_q10',    16,   50, 9, 20, 10, 'A',    None, 0.1),
    ('shard_of_global', 17,  16, 3, 7,  4, 'B',     64,   0.5),     # N_global != local N (grad_beta, F)
]


def main():
    mods = load_reference()
    for name, seed, N, D, M, Q, regime, Ng, ascale in CASES:
        rs = np.random.RandomState(seed)
        c = make_case(rs, N, D, M, Q, regime, Ng, ascale)
        with np.errstate(all='ignore'):
            o = run_reference(mods, c)
        payload = {('in_' + k): v for k, v in c.items()}
        payload.update({('out_' + k): v for k, v in o.items()})
        path = os.path.join(HERE, 'pt_%s.npz' % name)
        np.savez_compressed(path, **payload)
        print('%-16s F=%.10g  cond(Kmm)=%.3g  -> %s (%d bytes)' % (
            name, o['F'], np.linalg.cond(o['Kmm']), os.path.basename(path), os.path.getsize(path)))


if __name__ == '__main__':
    main()
