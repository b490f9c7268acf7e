"""Host orchestration of one evaluation: Python-3 restatement of ``parallel_GPLVM.likelihood_and_gradient``
(parallel_GPLVM.py:222-279) and its helpers ``calculate_global_statistics`` (:302-334), ``calculate_global_derivatives``
(:336-369), ``flatten/rebuild_global_statistics`` (:286-299), ``clean`` (:373-404), against a MapReduce backend module
with the ``local_MapReduce`` function set (``gparml_amd.gpu_MapReduce``).

Two modes produce the same numbers and the same files:
  * ``fast=False`` (compat): exactly the reference's call sequence -- cache, statistics_MR (12 statistics incl. the four
    derivative 3-tensors), partial_terms.grad_Z/grad_alpha/grad_sf2/grad_beta on the master, embeddings_MR.
  * ``fast=True``: the two-phase device protocol: phase 1 on every shard, device-side reduce of the
    packed statistics, replicated global step, phase 2 with the Z/alpha gradients contracted on the device, device-side
    reduce of the packed gradient sums.  The 3-tensors are never formed.

The optimiser keeps its contract: ``f, g = likelihood_and_gradient(x, iteration, step_size)`` returns ``(-F, -grad)`` in
the softplus-inverse parametrisation, flat order Z (row-major), sf2, alpha, beta (parallel_GPLVM.py:139-141, 203-212).
"""
import os
import time

import numpy

from ._lib import JitterRetry


def _f(x):
    return float(numpy.asarray(x).reshape(-1)[0])


LIM_VAL = -numpy.log(numpy.finfo(float).eps)          # supporting_functions.py:125


def transform(b, x):
    if b == (0, None):                                # supporting_functions.py:127-132
        assert -LIM_VAL < x < LIM_VAL
        return numpy.log(1 + numpy.exp(x))
    return x


def transform_back(b, x):
    if b == (0, None):                                # supporting_functions.py:135-140
        assert numpy.finfo(float).eps < x < LIM_VAL
        return numpy.log(-1 + numpy.exp(x))
    return x


def transform_grad(b, x):
    if b == (0, None):                                # supporting_functions.py:143-148
        assert -LIM_VAL < x < LIM_VAL
        return 1 / (numpy.exp(-x) + 1)
    return 1


def positive_mask(bounds):
    """Boolean mask of the (0, None)-bounded entries of a flat parameter vector."""
    return numpy.array([b == (0, None) for b in bounds], dtype=bool)


def transform_vec(mask, x):
    """``[transform(b, x_i)]`` over a whole flat vector at once (supporting_functions.py:127-132): softplus on the positive entries,
    identity elsewhere, the same range assertion.  The per-element Python loop costs ~50 k calls per evaluation at M*Q = 51 200."""
    x = numpy.asarray(x, dtype=float)
    out = x.copy()
    xp = x[mask]
    assert numpy.all((-LIM_VAL < xp) & (xp < LIM_VAL))
    out[mask] = numpy.log(1 + numpy.exp(xp))
    return out


def transform_grad_vec(mask, x):
    """``[transform_grad(b, x_i)]`` as a vector (supporting_functions.py:143-148)."""
    x = numpy.asarray(x, dtype=float)
    out = numpy.ones_like(x)
    xp = x[mask]
    assert numpy.all((-LIM_VAL < xp) & (xp < LIM_VAL))
    out[mask] = 1 / (numpy.exp(-xp) + 1)
    return out


def init_statistics(map_reduce, options):
    """parallel_GPLVM.init_statistics (:134-214): the names the backends pass around, the initial global statistics -- inducing points
    Z by k-means over the first shards' embeddings (scipy.cluster.vq.kmeans, topped up with the first embeddings when k-means returns
    fewer than M centres) plus 0.05 * randn, sf2 = alpha = beta = 1 (:179-194), or the ``*_f.npy`` files of a previous run with
    ``options['load']`` (:195-200) -- and the optimisation bounds.  Returns (options, global_statistics).  One-off host work."""
    M, Q = options['M'], options['Q']
    Driver(options, map_reduce)          # fills the *_names entries and the flat bounds exactly as the evaluations expect them
    if not options.get('load'):
        names = sorted(os.listdir(options['input'] + '/'))
        idx = 0
        embeddings = map_reduce.load(options['embeddings'] + '/' + names[idx] + '.embedding.npy')
        while embeddings.shape[0] < M:                                                  # :172-176
            idx += 1
            embeddings = numpy.concatenate((embeddings, map_reduce.load(options['embeddings'] + '/' + names[idx] + '.embedding.npy')))
        if embeddings.shape[1] != Q:
            raise Exception('Given Q does not equal existing embedding data dimensions!')
        import scipy.cluster.vq as cl
        Z = cl.kmeans(embeddings, M)[0]                                                 # :180-181
        missing = M - Z.shape[0]
        if missing > 0:
            Z = numpy.concatenate((Z, embeddings[:missing]))                            # :183-185
        Z = Z + numpy.random.randn(M, Q) * 0.05                                         # :187
        gs = {'Z': Z, 'sf2': numpy.array([[1.0]]), 'alpha': numpy.ones((1, Q)), 'beta': numpy.array([[1.0]])}
    else:
        gs = {key: map_reduce.load(options['statistics'] + '/global_statistics_' + key + '_f.npy')
              for key in options['global_statistics_names']}
    return options, gs


def initial_flat_vector(options, global_statistics):
    """The optimiser's starting point (parallel_GPLVM.py:98-100): flat order Z, sf2, alpha, beta with the positive entries mapped
    through transform_back (softplus inverse)."""
    flat = numpy.concatenate([numpy.asarray(global_statistics[k], dtype=float).flatten() for k in ('Z', 'sf2', 'alpha', 'beta')])
    return numpy.array([transform_back(b, v) for b, v in zip(options['flat_global_statistics_bounds'], flat)])


class Driver(object):
    def __init__(self, options, map_reduce=None, fast=True):
        if map_reduce is None:
            from . import gpu_MapReduce as map_reduce
        self.map_reduce = map_reduce
        self.options = options
        self.fast = fast
        self.time_acc = {'time_acc_statistics_map_reduce': [], 'time_acc_statistics_mapper': [], 'time_acc_statistics_reducer': [],
                         'time_acc_calculate_global_statistics': [], 'time_acc_embeddings_MR': [], 'time_acc_embeddings_MR_mapper': []}
        o = options
        # parallel_GPLVM.init_statistics (:139-157, 203-212)
        o['global_statistics_names'] = {'Z': (o['M'], o['Q']), 'sf2': (1, 1), 'alpha': (1, o['Q']), 'beta': (1, 1)}
        o['accumulated_statistics_names'] = ['sum_YYT', 'sum_exp_K_mi_K_im', 'sum_exp_K_miY', 'sum_exp_K_ii', 'sum_KL',
                                             'sum_d_exp_K_miY_d_Z', 'sum_d_exp_K_mi_K_im_d_Z', 'sum_d_exp_K_miY_d_alpha',
                                             'sum_d_exp_K_mi_K_im_d_alpha', 'sum_d_exp_K_ii_d_sf2', 'sum_d_exp_K_miY_d_sf2',
                                             'sum_d_exp_K_mi_K_im_d_sf2']
        o['partial_derivatives_names'] = ['F', 'dF_dsum_exp_K_ii', 'dF_dKmm', 'dF_dsum_exp_K_miY', 'dF_dsum_exp_K_mi_K_im']
        o['cache_names'] = ['Kmm', 'Kmm_inv']
        o['flat_global_statistics_bounds'] = ([(None, None)] * (o['M'] * o['Q']) + [(0, None)] + [(0, None)] * o['Q'] + [(0, None)])
        self._pos = positive_mask(o['flat_global_statistics_bounds'])
        o.setdefault('keep', True)
        o.setdefault('fixed_beta', False)
        o.setdefault('drop_out_fraction', 0)

    # ---- parallel_GPLVM.py:286-299
    def flatten_global_statistics(self, gs):
        return numpy.concatenate([numpy.asarray(gs[k], dtype=float).flatten() for k in ('Z', 'sf2', 'alpha', 'beta')])

    def rebuild_global_statistics(self, flat):
        gs, start = {}, 0
        for key, shape in self.options['global_statistics_names'].items():
            size = shape[0] * shape[1]
            gs[key] = numpy.asarray(flat[start:start + size], dtype=float).reshape(shape)
            start += size
        return gs

    # ---- parallel_GPLVM.py:373-404
    def clean(self):
        o = self.options
        if not o['keep'] and o['i'] != 'f':
            for group, prefix in (('global_statistics_names', 'global_statistics_'), ('accumulated_statistics_names', 'accumulated_statistics_'),
                                  ('partial_derivatives_names', 'partial_derivatives_'), ('cache_names', 'cache_')):
                for key in o[group]:
                    for it in (-1, o['i'] - 1, o['i']):
                        self.map_reduce.remove(o['statistics'] + '/' + prefix + key + '_' + str(it) + '.npy')

    # ---- parallel_GPLVM.py:222-279
    def likelihood_and_gradient(self, flat_array, iteration, step_size=0):
        o, mr = self.options, self.map_reduce
        flat_t = transform_vec(self._pos, flat_array)
        gs = self.rebuild_global_statistics(flat_t)
        o['i'] = iteration
        o['step_size'] = step_size
        self.clean()
        for key in gs:
            mr.save(o['statistics'] + '/global_statistics_' + key + '_' + str(o['i']) + '.npy', gs[key])
        if self.fast:
            F, gradient = self._evaluate_fast(gs)
        else:
            F, gradient = self._evaluate_compat(gs)
        grad = self.flatten_global_statistics(gradient)
        grad = grad * transform_grad_vec(self._pos, flat_array)
        return -1 * F, -1 * grad

    # ---- the reference's sequence through the backend surface (parallel_GPLVM.py:243-265, 302-369)
    def _evaluate_compat(self, gs):
        o, mr = self.options, self.map_reduce
        t0 = time.time()
        mr.cache(o, gs)
        files, mt, rt = mr.statistics_MR(o)
        t1 = time.time()
        acc = {k: mr.load(f) for k, f in files}
        pt = mr.load_partial_terms(o, gs)
        mr.load_cache(o, pt)
        pt.set_local_statistics(acc['sum_YYT'], acc['sum_exp_K_mi_K_im'], acc['sum_exp_K_miY'], acc['sum_exp_K_ii'], acc['sum_KL'])
        pd = {'F': pt.logmarglik(), 'dF_dsum_exp_K_ii': pt.dF_dexp_K_ii(), 'dF_dsum_exp_K_miY': pt.dF_dexp_K_miY(),
              'dF_dsum_exp_K_mi_K_im': pt.dF_dexp_K_mi_K_im(), 'dF_dKmm': pt.dF_dKmm()}
        for key in pd:
            mr.save(o['statistics'] + '/partial_derivatives_' + key + '_' + str(o['i']) + '.npy', pd[key])
        grad_Z = pt.grad_Z(pd['dF_dKmm'], pt.dKmm_dZ(), pd['dF_dsum_exp_K_miY'], acc['sum_d_exp_K_miY_d_Z'],
                           pd['dF_dsum_exp_K_mi_K_im'], acc['sum_d_exp_K_mi_K_im_d_Z'])
        grad_alpha = pt.grad_alpha(pd['dF_dKmm'], pt.dKmm_dalpha(), pd['dF_dsum_exp_K_miY'], acc['sum_d_exp_K_miY_d_alpha'],
                                   pd['dF_dsum_exp_K_mi_K_im'], acc['sum_d_exp_K_mi_K_im_d_alpha'])
        grad_sf2 = pt.grad_sf2(pd['dF_dKmm'], pt.dKmm_dsf2(), pd['dF_dsum_exp_K_ii'], acc['sum_d_exp_K_ii_d_sf2'],
                               pd['dF_dsum_exp_K_miY'], acc['sum_d_exp_K_miY_d_sf2'], pd['dF_dsum_exp_K_mi_K_im'],
                               acc['sum_d_exp_K_mi_K_im_d_sf2'])
        gradient = {'Z': grad_Z, 'sf2': grad_sf2, 'alpha': grad_alpha,
                    'beta': numpy.zeros((1, 1)) if o['fixed_beta'] else pt.grad_beta()}
        t2 = time.time()
        self.time_acc['time_acc_statistics_map_reduce'].append(t1 - t0)
        self.time_acc['time_acc_statistics_mapper'].append(mt)
        self.time_acc['time_acc_statistics_reducer'].append(rt)
        self.time_acc['time_acc_calculate_global_statistics'].append(t2 - t1)
        if not o['fixed_embeddings']:
            t3 = time.time()
            et = mr.embeddings_MR(o)
            self.time_acc['time_acc_embeddings_MR'].append(time.time() - t3)
            self.time_acc['time_acc_embeddings_MR_mapper'].append(et)
        return pd['F'], gradient

    # ---- two-phase device protocol
    def _evaluate_fast(self, gs):
        o, mr = self.options, self.map_reduce
        t0 = time.time()
        files = mr._input_files(o)
        engines = mr._prepare_shards(o, files, gs)
        # node drop-out (local_MapReduce.py:119-129, 263-264): the same draw as statistics_MR; dropped shards contribute to neither
        # reduction and both reduced buffers are divided by kept/(kept+dropped)
        kept, frac = list(range(len(files))), None
        if o.get('drop_out_fraction', 0) > 0:
            kept, frac = mr._draw_drop_out(len(files), o['drop_out_fraction'])
        mr._for_each(engines, lambda e: e.phase1())
        root = engines[kept[0]]
        for i in kept[1:]:
            root.combine(engines[i], 'stats', 'add')            # statistics_reducer on the device(s)
        if frac is not None:
            root.scale_buffer('stats', 1.0 / frac)
        for e in engines:
            if e is not root:
                e.combine(root, 'stats', 'copy')       # every shard needs the global sums (local_MapReduce.py:318-320)
        t1 = time.time()
        want_emb = not o['fixed_embeddings']
        jitter = 0
        while True:
            def second(e):
                e.global_step(sync=False, jitter=jitter)   # replicated M x M algebra
                e.phase2(want_emb)
            mr._for_each(engines, second)
            for i in kept[1:]:
                root.combine(engines[i], 'grads', 'add')
            if frac is not None:
                root.scale_buffer('grads', 1.0 / frac)
            try:
                res = root.finish()
                break
            except JitterRetry as r:
                jitter = r.mask
        sc = root.scalars()
        # the artefacts other tools read (--load, predict.py): the five base sums and the partial derivatives
        it = str(o['i'])
        base = {'sum_YYT': sc['sum_YYT'], 'sum_exp_K_ii': sc['sum_exp_K_ii'], 'sum_KL': sc['KL'],
                'sum_exp_K_mi_K_im': root.download('PSI2_SUM'), 'sum_exp_K_miY': root.download('PSI1TY')}
        for key, val in base.items():
            mr.save(o['statistics'] + '/accumulated_statistics_' + key + '_' + it + '.npy', val)
        pd = {'F': res['F'], 'dF_dsum_exp_K_ii': -0.5 * _f(gs['beta']) * o['D'], 'dF_dKmm': root.download('DF_DKMM'),
              'dF_dsum_exp_K_miY': root.download('DF_DPSI1TY'), 'dF_dsum_exp_K_mi_K_im': root.download('DF_DPSI2')}
        for key, val in pd.items():
            mr.save(o['statistics'] + '/partial_derivatives_' + key + '_' + it + '.npy', val)
        if want_emb:
            for f, e in zip(files, engines):
                mr.save(o['embeddings'] + '/' + os.path.basename(f) + '.grad_latest.npy', e.download('GRAD_LATEST'))
        t2 = time.time()
        self.time_acc['time_acc_statistics_map_reduce'].append(t1 - t0)
        self.time_acc['time_acc_calculate_global_statistics'].append(t2 - t1)
        gradient = {'Z': res['grad_Z'], 'sf2': numpy.array([[res['grad_sf2']]]), 'alpha': res['grad_alpha'].reshape(1, -1),
                    'beta': numpy.zeros((1, 1)) if o['fixed_beta'] else numpy.array([[res['grad_beta']]])}
        return res['F'], gradient
