"""A MapReduce backend with the function set of the reference's ``local_MapReduce`` module, running on the GPU.

``parallel_GPLVM.py`` selects its backend with ``import local_MapReduce as map_reduce`` (parallel_GPLVM.py:87-95) and
then calls exactly: ``init, cache, statistics_MR, embeddings_MR, load, save, remove, exists, load_partial_terms,
load_cache`` (call sites parallel_GPLVM.py:95,167,238,243-244,264,309-314,332,379-404; predict.py:30,35,122-124).
This module provides the same names, arguments, return values and on-disk artefacts:

    {statistics}/global_statistics_{Z|sf2|alpha|beta}_{i}.npy      read   (written by the caller, :236-238)
    {statistics}/cache_{Kmm|Kmm_inv}_{i}.npy                       written by cache()
    {statistics}/accumulated_statistics_{12 keys}_{i}.npy          written by statistics_MR()
    {embeddings}/{shard}.embedding.npy / .variance.npy / .grad_d.npy   read
    {embeddings}/{shard}.grad_latest.npy  (2,N_s,Q)                written by embeddings_MR()

Differences by design: each shard's Y is parsed from CSV once and then stays resident in HBM
(the reference re-parses it in every mapper call, local_MapReduce.py:197,325); mappers run in this process, one
ShardEngine per shard, instead of a multiprocessing.Pool; Kmm is rebuilt on the device rather than read from the
cache files.  ``options['gpu_compat_tensors'] = False`` skips the four derivative 3-tensors (the fast driver in
``gparml_amd.driver`` never needs them).
"""
import glob
import os
import time
from os.path import basename

import numpy

from .engine import ShardEngine
from .partial_terms import partial_terms as _partial_terms

def _f(x):
    return float(numpy.asarray(x).reshape(-1)[0])


# ------------------------------------------------------------------------------------------------- module state
dropped_out_nodes = []
non_dropped_out_nodes = []
_shards = {}          # input file -> dict(engine, N_s)


def _reset():
    for s in _shards.values():
        s['engine'].close()
    _shards.clear()


# ------------------------------------------------------------------------------------------------- file helpers
def save(file_name, obj):
    numpy.save(file_name, obj)                                      # local_MapReduce.py:370-371


def load(file_name):
    return numpy.load(file_name)                                    # local_MapReduce.py:373-374


def exists(file_name):
    return os.path.exists(file_name)                                # local_MapReduce.py:376-377


def remove(file_name):
    if exists(file_name):                                           # local_MapReduce.py:379-381
        os.remove(file_name)


def _input_files(options):
    return sorted(glob.glob(options['input'] + '/*'))


def _read_csv(path):
    """numpy.genfromtxt(path, delimiter=',') (local_MapReduce.py:197-199) through the library's parallel parser
    (gp_csv_shape / gp_csv_read): a 1e6 x 100 shard takes seconds instead of ~300 s, and is parsed once per run."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    rows, cols = ctypes.c_int64(), ctypes.c_int64()
    _lib.raise_for(lib.gp_csv_shape(path.encode(), ctypes.byref(rows), ctypes.byref(cols)), lib, None, 'gp_csv_shape')
    Y = numpy.empty((rows.value, cols.value))
    _lib.raise_for(lib.gp_csv_read(path.encode(), Y.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), rows.value, cols.value, 0),
                   lib, None, 'gp_csv_read')
    return Y                                                        # always 2-D: a one-column file is (N, 1) as after :198-199


# ------------------------------------------------------------------------------------------------- init
def _streaming_pca(options, names):
    """supporting_functions.PCA (supporting_functions.py:102-121: left singular vectors of the centred data, each scaled to unit standard
    deviation) over ALL shards (local_MapReduce.py:54-65) without ever holding more than one shard: the reference concatenates every shard
    on one host and takes a thin SVD of the whole matrix (64 GB at BASELINE configs[4]).  The same principal axes are the eigenvectors of the
    D x D scatter matrix, which is a SUM over shards: pass 1 accumulates, per shard, the row count, the sum and the Gram matrix of the rows
    shifted by a provisional centre (the first shard's mean: keeps the later correction for the true mean small against the scatter);
    `eigh` of the D x D matrix gives the axes V and the variances lambda / N; pass 2 projects each shard, X_s = (Y_s - mean) V / sqrt(lambda / N).
    Identical to the SVD form up to the sign of a component (fixed here: the largest entry of every axis is positive) and rounding
    (tests/test_init_against_reference.py: 1e-9 against the reference's own files).  Returns the per-shard projection."""
    Q = options['Q']
    n_tot, shift, ssum, gram = 0, None, None, None
    for name in names:
        Y = _read_csv(options['input'] + '/' + name)
        if shift is None:
            shift = Y.mean(axis=0)
            ssum, gram = numpy.zeros(Y.shape[1]), numpy.zeros((Y.shape[1], Y.shape[1]))
        Yc = Y - shift
        n_tot += Y.shape[0]
        ssum += Yc.sum(axis=0)
        gram += Yc.T.dot(Yc)
    delta = ssum / n_tot                                            # true mean - provisional centre
    scatter = gram - n_tot * numpy.outer(delta, delta)
    lam, V = numpy.linalg.eigh(scatter)
    order = numpy.argsort(lam)[::-1][:Q]
    lam, V = lam[order], V[:, order]
    # rank-deficient data (Q beyond the rank of the centred data, or cancellation in the mean correction) leaves zero or slightly negative
    # trailing eigenvalues: dividing by their root would write inf / nan embeddings without a word (the reference's SVD form divides by a
    # tiny standard deviation in the same case and returns noise)
    floor = numpy.finfo(float).eps * max(float(lam[0]), 0.0) * scatter.shape[0]
    if not (lam[-1] > floor):
        raise numpy.linalg.LinAlgError('PCA initialisation: the data has fewer than Q = %d principal directions (eigenvalue %d of the scatter '
                                       'matrix is %.3e against a largest one of %.3e)' % (Q, int(numpy.sum(lam > floor)) + 1, lam[-1], lam[0]))
    V = V * numpy.sign(V[numpy.argmax(numpy.abs(V), axis=0), numpy.arange(V.shape[1])])[None, :]
    mean, std = shift + delta, numpy.sqrt(lam / n_tot)              # X.std(axis=0) of the projected data (ddof = 0)

    def project(name):
        return (_read_csv(options['input'] + '/' + name) - mean).dot(V) / std

    return project


def init(options):
    """local_MapReduce.init (local_MapReduce.py:27-104): count the points; create embeddings / variances unless
    loading or using fixed embeddings.  The PCA / random initialisation is one-off host preprocessing (out of the hot
    path); PCA streams over the shards (_streaming_pca: per-shard D x D scatter sums + eigh) instead of concatenating all data on one
    host; PPCA / FA initialisers of supporting_functions.py are not provided."""
    names = sorted(os.listdir(options['input'] + '/'))
    lengths = []
    for name in names:
        with open(options['input'] + '/' + name) as f:
            lengths.append(sum(1 for line in f if line.strip()))
    options['N'] = sum(lengths)
    if not options['fixed_embeddings'] and not options['load']:
        if options['init'] == 'PCA':
            X = None
            project = _streaming_pca(options, names)
        elif options['init'] == 'random':
            X = numpy.random.randn(options['N'], options['Q'])
            project = None
        else:
            raise Exception("init '%s' is not provided by the GPU backend (PCA or random)" % options['init'])
        start = 0
        for name, n in zip(names, lengths):
            base = options['embeddings'] + '/' + name
            save(base + '.embedding.npy', X[start:start + n] if project is None else project(name))
            v = numpy.clip(0.5 * numpy.ones((n, options['Q'])) + 0.01 * numpy.random.randn(n, options['Q']), 0.001, 1)
            save(base + '.variance.npy', numpy.log(numpy.exp(v) - 1.0))            # transformVar_back, :90-93
            start += n
    if options['fixed_embeddings']:
        for name, n in zip(names, lengths):
            base = options['embeddings'] + '/' + name
            if not exists(base + '.embedding.npy'):
                raise Exception('No embedding file ' + base + '.embedding.npy')
            save(base + '.variance.npy', numpy.zeros((n, options['Q'])))             # :94-103
    return options


# ------------------------------------------------------------------------------------------------- shards on the GPU
def _globals(options):
    gs = {}
    for key in options['global_statistics_names']:
        gs[key] = load(options['statistics'] + '/global_statistics_' + key + '_' + str(options['i']) + '.npy')
    return gs


def _devices(options):
    """GPUs the shards are spread over: options['devices'] (a list of device indices) or the single options['device'] (default 0).
    The reference runs one pool worker per shard (local_MapReduce.py:134,155,299); here shard i lives on devices[i % len(devices)]."""
    devs = options.get('devices')
    if devs:
        return [int(d) for d in devs]
    return [int(options.get('device', 0))]


def _device_of(options, input_file_name):
    files = _input_files(options)
    devs = _devices(options)
    try:
        return devs[files.index(input_file_name) % len(devs)]
    except ValueError:
        return devs[0]


def _for_each(items, fn):
    """Apply fn to every item -- from one thread per item when there are several (the ctypes calls release the GIL, so shards on
    different GPUs run concurrently: the Pool.map of local_MapReduce.py:134-137); results in order, the first exception re-raised."""
    items = list(items)
    if len(items) <= 1:
        return [fn(x) for x in items]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(items)) as ex:
        return list(ex.map(fn, items))


def _prepare_shards(options, files, global_statistics):
    return _for_each(files, lambda f: _prepare_shard(options, f, global_statistics))


def _draw_drop_out(n_nodes, fraction):
    """The keep/drop draw of statistics_MR (local_MapReduce.py:119-129) from numpy's global generator, like the reference.  Returns the
    kept indices and the divisor kept/(kept+dropped); sets the module globals the reference keeps between its two MapReduces.  When
    every node is dropped the reference keeps one random node but leaves ALL n nodes in dropped_out_nodes: divisor 1/(n+1)."""
    global non_dropped_out_nodes, dropped_out_nodes
    drop = numpy.random.uniform(size=n_nodes) < fraction
    dropped_out_nodes = numpy.arange(n_nodes)[drop]
    non_dropped_out_nodes = numpy.arange(n_nodes)[~drop]
    if len(non_dropped_out_nodes) == 0:
        non_dropped_out_nodes = [numpy.random.randint(0, n_nodes)]
    frac = float(len(non_dropped_out_nodes)) / (len(non_dropped_out_nodes) + len(dropped_out_nodes))
    return [int(i) for i in non_dropped_out_nodes], frac


def _prepare_shard(options, input_file_name, global_statistics):
    """Everything statistics_mapper / embeddings_mapper do before partial_terms.set_data
    (local_MapReduce.py:189-214, 315-341): resident Y, embeddings, trial point, globals."""
    key = os.path.abspath(input_file_name)
    sh = _shards.get(key)
    base = options['embeddings'] + '/' + basename(input_file_name)
    # With --fixed_embeddings nothing rewrites the embedding files during a run (the reference still re-reads them in every map call,
    # local_MapReduce.py:200-203): a resident shard whose two files are byte-for-byte the ones it was loaded from (size and modification time)
    # keeps its device copy -- at N = 1e6, Q = 10 the two numpy.load + upload were 25 of the 55 ms of a likelihood_and_gradient call.
    sig = None
    if options['fixed_embeddings']:
        st = [os.stat(base + ext) for ext in ('.embedding.npy', '.variance.npy')]
        sig = tuple((x.st_size, x.st_mtime_ns) for x in st)
    if sh is not None and sig is not None and sh.get('emb_sig') == sig and sh['shape'][1:] == (options['D'], options['M'], options['Q']):
        eng = sh['engine']
    else:
        X_mu = load(base + '.embedding.npy')
        X_S = load(base + '.variance.npy')
        if sh is None or sh['shape'] != (X_mu.shape[0], options['D'], options['M'], options['Q']):
            if sh is not None:
                sh['engine'].close()
            Y = _read_csv(input_file_name)
            eng = ShardEngine(Y.shape[0], options['D'], options['M'], options['Q'], device=_device_of(options, input_file_name))
            eng.set_timing(0)          # nobody reads per-kernel device timings here (time_acc is host time): no timing events on the stream
            eng.upload_shard(Y, X_mu, X_S, xs_is_raw=not options['fixed_embeddings'])
            sh = _shards[key] = dict(engine=eng, shape=(Y.shape[0], options['D'], options['M'], options['Q']))
        else:
            eng = sh['engine']
            eng.upload_embeddings(X_mu, X_S, xs_is_raw=not options['fixed_embeddings'])
        sh['emb_sig'] = sig
    d = None
    step = 0.0
    if not options['fixed_embeddings']:
        dname = base + '.grad_d.npy'
        if exists(dname) and options['step_size'] != 0:              # :204-211
            d = load(dname)
            step = options['step_size']
    eng.set_direction(d)
    eng.set_globals(global_statistics['Z'], _f(global_statistics['sf2']), numpy.squeeze(global_statistics['alpha']).reshape(-1),
                    _f(global_statistics['beta']), N_global=options['N'], step_size=step)
    return eng


# ------------------------------------------------------------------------------------------------- statistics MR
BASE_KEYS = ['sum_YYT', 'sum_exp_K_ii', 'sum_exp_K_mi_K_im', 'sum_exp_K_miY', 'sum_KL']


def statistics_mapper(arg):
    """local_MapReduce.statistics_mapper (:183-248) for one shard; returns the 12-key dictionary (arrays, not files)."""
    input_file_name, options = arg
    start = time.time()
    gs = _globals(options)
    eng = _prepare_shard(options, input_file_name, gs)
    eng.phase1()
    sc = eng.scalars()
    sf2 = _f(gs['sf2'])
    out = {
        'sum_YYT': numpy.float64(sc['sum_YYT']),
        'sum_exp_K_ii': numpy.float64(sc['sum_exp_K_ii']),
        'sum_exp_K_mi_K_im': eng.download('PSI2_SUM'),
        'sum_exp_K_miY': eng.download('PSI1TY'),
        'sum_KL': numpy.float64(sc['KL']),
    }
    out['sum_d_exp_K_ii_d_sf2'] = eng.N_s                                           # an int, partial_terms.py:318-320
    out['sum_d_exp_K_miY_d_sf2'] = out['sum_exp_K_miY'] / sf2
    out['sum_d_exp_K_mi_K_im_d_sf2'] = 2.0 * out['sum_exp_K_mi_K_im'] / sf2
    if options.get('gpu_compat_tensors', True):
        out['sum_d_exp_K_miY_d_Z'] = eng.download('DPSI1TY_DZ')
        out['sum_d_exp_K_mi_K_im_d_Z'] = eng.download('DPSI2_DZ')
        out['sum_d_exp_K_miY_d_alpha'] = eng.download('DPSI1TY_DALPHA')
        out['sum_d_exp_K_mi_K_im_d_alpha'] = eng.download('DPSI2_DALPHA')
    return out, time.time() - start


def statistics_MR(options):
    """local_MapReduce.statistics_MR (:115-171): map every shard, reduce by key, write accumulated_statistics files."""
    global non_dropped_out_nodes, dropped_out_nodes
    input_files = _input_files(options)
    frac = None
    if options.get('drop_out_fraction', 0) > 0:                                      # :119-129
        kept, frac = _draw_drop_out(len(input_files), options['drop_out_fraction'])
        input_files = [input_files[i] for i in kept]
    responses = _for_each(input_files, lambda f: statistics_mapper((f, options)))   # one worker per shard, :131-137
    mapped = [r[0] for r in responses]
    mapper_times = [r[1] for r in responses]
    files, reducer_times = [], []
    for key in mapped[0].keys():                                                     # statistics_reducer, :250-277
        start = time.time()
        acc = mapped[0][key]
        for m in mapped[1:]:
            acc = acc + m[key]
        if frac is not None:                                                          # :263-264, 272-273
            acc = acc / frac
        name = options['statistics'] + '/accumulated_statistics_' + key + '_' + str(options['i']) + '.npy'
        save(name, acc)
        files.append((key, name))
        reducer_times.append(time.time() - start)
    return files, mapper_times, reducer_times


# ------------------------------------------------------------------------------------------------- embeddings MR
def embeddings_mapper(arg):
    """local_MapReduce.embeddings_mapper (:310-363): per-point gradients at the trial point from the GLOBAL sums;
    writes -[grad_X_mu, grad_X_S * softplus'(raw)] as (2,N_s,Q) to .grad_latest.npy."""
    input_file_name, options = arg
    start = time.time()
    gs = _globals(options)
    acc = {}
    for key in BASE_KEYS:
        acc[key] = load(options['statistics'] + '/accumulated_statistics_' + key + '_' + str(options['i']) + '.npy')
    eng = _prepare_shard(options, input_file_name, gs)
    eng.phase1()                                    # Psi1 and the psi2 tables at the trial point (set_data(..., False), :348)
    eng.set_local_statistics(_f(acc['sum_YYT']), acc['sum_exp_K_mi_K_im'], acc['sum_exp_K_miY'], _f(acc['sum_exp_K_ii']),
                             _f(acc['sum_KL']))  # :350-354
    eng.global_step()
    eng.phase2(True)
    g = eng.download('GRAD_LATEST')                 # :357-359
    save(options['embeddings'] + '/' + basename(input_file_name) + '.grad_latest.npy', g)
    return time.time() - start


def embeddings_MR(options):
    files = _input_files(options)                                                    # all nodes, dropped or not (:293-295)
    return _for_each(files, lambda f: embeddings_mapper((f, options)))                # :284-308


# ------------------------------------------------------------------------------------------------- cache / partial_terms
def load_partial_terms(options, global_statistics):
    # local_MapReduce.py:403-409
    return _partial_terms(global_statistics['Z'], _f(global_statistics['sf2']), numpy.squeeze(global_statistics['alpha']).reshape(-1),
                          _f(global_statistics['beta']), options['M'], options['Q'], options['N'], options['D'],
                          update_global_statistics=False, device=_devices(options)[0])


def cache(options, global_statistics):
    # local_MapReduce.py:383-394: Kmm and Kmm_inv once per evaluation, for all nodes
    pt = load_partial_terms(options, global_statistics)
    pt.update_global_statistics()
    save(options['statistics'] + '/cache_Kmm_' + str(options['i']) + '.npy', pt.Kmm)
    save(options['statistics'] + '/cache_Kmm_inv_' + str(options['i']) + '.npy', pt.Kmm_inv)


def load_cache(options, partial_terms):
    # local_MapReduce.py:396-401
    Kmm = load(options['statistics'] + '/cache_Kmm_' + str(options['i']) + '.npy')
    Kmm_inv = load(options['statistics'] + '/cache_Kmm_inv_' + str(options['i']) + '.npy')
    partial_terms.set_global_statistics(Kmm, Kmm_inv)
