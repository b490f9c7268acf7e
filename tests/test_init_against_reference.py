"""the initialisation path against what the REFERENCE produced.

tests/golden/make_pipeline_golden.py runs the reference's parallel_GPLVM.main with init='PCA' after numpy.random.seed(seed); the files
the first likelihood_and_gradient call reads -- every shard's .embedding.npy (supporting_functions.PCA over all data,
local_MapReduce.py:54-65, supporting_functions.py:102-121) and .variance.npy (clip(0.5 + 0.01 randn, 0.001, 1) through transformVar_back,
local_MapReduce.py:88-93) -- and its flat parameter vector (k-means inducing points + 0.05 randn, unit hyper-parameters, through
transform_back: parallel_GPLVM.py:98-100, 160-194) are in the fixtures as call0_in_shard*_embedding / _variance and call0_x.  Here the
same seed and inputs go through gparml_amd.gpu_MapReduce.init and gparml_amd.driver.init_statistics / initial_flat_vector.

The reference walks os.listdir order when it draws the variances (an unsorted directory listing); the replay tries the shard orders and
requires one of them to reproduce every file.  PCA is sign- and order-independent up to rounding (1e-9 asserted)."""
import itertools
import os

import numpy as np
import pytest

from pipeline_util import load_pipeline

SEEDS = {'gplvm_2shards': 21, 'config1_1shard': 23}     # the seeds make_pipeline_golden.py passes (free-embedding runs only)


@pytest.mark.parametrize('name', sorted(SEEDS))
def test_pca_embeddings_variances_and_initial_vector(name, tmp_path, monkeypatch):
    from gparml_amd import gpu_MapReduce as mr
    from gparml_amd import driver
    g = load_pipeline(name)
    ns, D, M, Q = int(g['n_shards']), int(g['D']), int(g['M']), int(g['Q'])
    matched = None
    for order in itertools.permutations(range(ns)):
        work = tmp_path / ('o' + ''.join(map(str, order)))
        dirs = {d: str(work / d) for d in ('input', 'embeddings', 'statistics', 'tmp')}
        for d in dirs.values():
            os.makedirs(d)
        for i in range(ns):
            np.savetxt(os.path.join(dirs['input'], 'shard_%d' % i), g['Y_%d' % i], delimiter=',', fmt='%.17g')
        names = ['shard_%d' % i for i in order]
        monkeypatch.setattr(mr.os, 'listdir', lambda p, names=names: list(names))       # the directory order of this replay
        monkeypatch.setattr(mr, 'sorted', lambda x: list(x), raising=False)              # init walks the listing as it comes
        options = dict(input=dirs['input'], embeddings=dirs['embeddings'], statistics=dirs['statistics'], tmp=dirs['tmp'], parallel='local',
                       iterations=2, keep=True, load=False, init='PCA', optimiser='SCG_adapted', drop_out_fraction=0, local_no_pool=False,
                       M=M, Q=Q, D=D, fixed_embeddings=False, fixed_beta=False)
        np.random.seed(SEEDS[name])
        options = mr.init(options)
        assert options['N'] == int(g['N'])
        ok = True
        for i in range(ns):
            emb = np.load(os.path.join(dirs['embeddings'], 'shard_%d.embedding.npy' % i))
            var = np.load(os.path.join(dirs['embeddings'], 'shard_%d.variance.npy' % i))
            ref_e, ref_v = g['call0_in_shard%d_embedding' % i], g['call0_in_shard%d_variance' % i]
            assert emb.shape == ref_e.shape and var.shape == ref_v.shape
            # PCA: identical up to the sign of a component and rounding, whatever the row order
            for q in range(Q):
                s = np.sign(np.dot(emb[:, q], ref_e[:, q]))
                assert np.max(np.abs(s * emb[:, q] - ref_e[:, q])) <= 1e-9 * np.max(np.abs(ref_e[:, q])), (name, i, q)
            ok = ok and np.allclose(var, ref_v, rtol=0, atol=1e-12)
        if not ok:
            continue
        # same component signs as the reference from here on (k-means sees the embeddings)
        for i in range(ns):
            np.save(os.path.join(dirs['embeddings'], 'shard_%d.embedding.npy' % i), g['call0_in_shard%d_embedding' % i])
        monkeypatch.undo()
        options, gs = driver.init_statistics(mr, options)
        x0 = driver.initial_flat_vector(options, gs)
        np.testing.assert_allclose(x0, g['call0_x'], rtol=1e-9, atol=1e-12)
        matched = order
        break
    assert matched is not None, 'no shard order reproduces the reference variances'


def test_streaming_pca_equals_the_svd_of_all_data(tmp_path):
    """gpu_MapReduce._streaming_pca (per-shard scatter sums + eigh, one shard in memory at a time) against supporting_functions.PCA's
    arithmetic on the concatenated data (thin SVD of the centred matrix, left singular vectors scaled to unit standard deviation,
    supporting_functions.py:102-121): three ragged shards with a large common offset, D = 40, Q = 5."""
    from gparml_amd import gpu_MapReduce as mr
    rs = np.random.RandomState(8)
    D, Q = 40, 5
    W = rs.randn(7, D) * np.array([5, 4, 3, 2, 1.5, 0.3, 0.2])[:, None]
    shards = [rs.randn(n, 7).dot(W) + 0.05 * rs.randn(n, D) + 100.0 for n in (57, 130, 21)]
    os.makedirs(str(tmp_path / 'input'))
    names = ['s%d' % i for i in range(3)]
    for n, Y in zip(names, shards):
        np.savetxt(str(tmp_path / 'input' / n), Y, delimiter=',', fmt='%.17g')
    project = mr._streaming_pca({'input': str(tmp_path / 'input'), 'Q': Q}, names)
    X = np.concatenate([project(n) for n in names])
    Y = np.concatenate(shards)
    U = np.linalg.svd(Y - Y.mean(axis=0), full_matrices=False)[0][:, :Q]
    U = U / U.std(axis=0)
    for q in range(Q):
        s = np.sign(np.dot(X[:, q], U[:, q]))
        assert np.max(np.abs(s * X[:, q] - U[:, q])) <= 1e-9 * np.max(np.abs(U[:, q])), q
    np.testing.assert_allclose(X.std(axis=0), 1.0, rtol=1e-12)
