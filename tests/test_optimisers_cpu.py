"""The two optimisers (gparml_amd.scg_adapted, gparml_amd.gd) without a GPU: the split formulation (global part on the
host, per-point part behind an ``ops`` object) must follow exactly the same trajectory as the same problem with every
parameter in the host vector.  The numpy ``ops`` below implement the function names of scg_adapted_local_MapReduce.py:29-243
and gd_local_MapReduce.py:14-105 on in-memory vectors (what the reference does through .npy files)."""
import numpy as np
import pytest

from gparml_amd.gd import GD
from gparml_amd.scg_adapted import SCG_adapted


class NumpyOps(object):
    """Per-point part X (shape (n,)) with the reference's trial-point protocol: the objective is evaluated at X + step*d."""

    def __init__(self, X0):
        self.X = X0.copy()
        self.latest = np.zeros_like(X0)
        self.new = self.old = self.d = None

    # scg_adapted_local_MapReduce.py / gd_local_MapReduce.py names
    def embeddings_set_grads(self, folder=None):
        self.new, self.old, self.d = self.latest.copy(), self.latest.copy(), -self.latest

    def embeddings_get_grads_mu(self, folder=None):
        return float(np.dot(self.new, self.d))

    def embeddings_get_grads_kappa(self, folder=None):
        return float(np.dot(self.d, self.d))

    def embeddings_get_grads_theta(self, folder=None):
        return float(np.dot(self.d, self.latest - self.new))

    def embeddings_get_grads_current_grad(self, folder=None):
        return float(np.dot(self.new, self.new)) if not self.gd else float(np.sum(np.abs(self.new)))

    def embeddings_get_grads_gamma(self, folder=None):      # sum grad_new * grad_old (:128-140); |grad_new|^2 is in current_grad
        return float(np.dot(self.new, self.old))

    def embeddings_get_grads_max_d(self, folder, alpha):
        return float(np.max(np.abs(alpha * self.d)))

    def embeddings_get_grads_max_gradnow(self, folder=None):
        return float(np.max(np.abs(self.new)))

    def embeddings_set_grads_reset_d(self, folder=None):
        self.d = -self.new

    def embeddings_set_grads_update_d(self, folder, gamma):
        self.d = (gamma * self.d - self.new) if not self.gd else -(self.new + gamma * self.d)

    def embeddings_set_grads_update_X(self, folder, alpha):
        self.X = self.X + alpha * self.d

    def embeddings_set_grads_update_grad_old(self, folder=None):
        self.old = self.new.copy()

    def embeddings_set_grads_update_grad_new(self, folder=None):
        self.new = self.latest.copy()

    embeddings_set_grads_update_grad_now = embeddings_set_grads_update_grad_new
    gd = False


def _problem(n_host=40, n_local=7, seed=0):   # n_host > iterations: the restart after x.size successes (scg_adapted.py:250) counts the host part only
    rs = np.random.RandomState(seed)
    A = rs.randn(n_host + n_local, n_host + n_local)
    H = A.dot(A.T) + 0.5 * np.eye(n_host + n_local)
    b = rs.randn(n_host + n_local)

    def f_full(v):                                   # smooth, non-quadratic, bounded below
        q = 0.5 * v.dot(H).dot(v) - b.dot(v)
        return q + 0.1 * np.sum(v ** 4), H.dot(v) - b + 0.4 * v ** 3
    return f_full, rs.randn(n_host), rs.randn(n_local)


@pytest.mark.parametrize('optimiser', ['scg', 'gd'])
def test_split_vectors_follow_the_joint_trajectory(optimiser):
    f_full, x0, X0 = _problem()
    nh = x0.size

    def joint(v, iteration, step_size=0):
        return f_full(v)

    ops = NumpyOps(X0)
    ops.gd = optimiser == 'gd'

    def split(x, iteration, step_size=0):
        Xtrial = ops.X + (step_size * ops.d if ops.d is not None and step_size != 0 else 0.0)    # local_MapReduce.py:205-211
        f, g = f_full(np.concatenate((x, Xtrial)))
        ops.latest = g[nh:].copy()
        return f, g[:nh]

    if optimiser == 'scg':
        xj, flog_j, nfe_j, _ = SCG_adapted(joint, np.concatenate((x0, X0)), None, fixed_embeddings=True, maxiters=25, xtol=0, ftol=0, gtol=0)
        xs, flog_s, nfe_s, _ = SCG_adapted(split, x0.copy(), ops, fixed_embeddings=False, maxiters=25, xtol=0, ftol=0, gtol=0)
        assert nfe_j == nfe_s
    else:
        xj, flog_j, _, _ = GD(joint, np.concatenate((x0, X0)), None, fixed_embeddings=True, maxiters=25, ftol=0, gtol=0)
        xs, flog_s, _, _ = GD(split, x0.copy(), ops, fixed_embeddings=False, maxiters=25, ftol=0, gtol=0)
    assert len(flog_j) == len(flog_s) and len(flog_j) > 5
    np.testing.assert_allclose(flog_s, flog_j, rtol=1e-9)
    np.testing.assert_allclose(np.concatenate((xs, ops.X)), xj, rtol=1e-7, atol=1e-9)
    assert flog_j[-1] < flog_j[0]


def test_failure_cap_of_the_safe_wrapper():
    """scg_adapted.py:44-76: 100 consecutive failures are absorbed (f = inf, grad = ones), the 101st is re-raised; a success resets."""
    from gparml_amd import scg_adapted as S
    S._fail_count = 0
    calls = {'n': 0}

    def bad(x, it, step):
        calls['n'] += 1
        raise np.linalg.LinAlgError('singular')

    x = np.zeros(3)
    for _ in range(S._allowed_failures):
        f, g = S.safe_f_and_grad_f(bad, x)
        assert f == np.inf and np.array_equal(g, np.ones(3))
    with pytest.raises(np.linalg.LinAlgError):
        S.safe_f_and_grad_f(bad, x)
    assert calls['n'] == S._allowed_failures + 1
    S._fail_count = 50
    assert S.safe_f_and_grad_f(lambda x, it, step: (1.0, x), x)[0] == 1.0 and S._fail_count == 0


def test_init_statistics_follows_the_reference(tmp_path):
    """parallel_GPLVM.init_statistics (:134-214): k-means inducing points + 0.05 randn, unit hyper-parameters, bounds, --load branch."""
    from gparml_amd import driver as Dr

    class Files(object):
        load = staticmethod(np.load)
        save = staticmethod(np.save)

    rs = np.random.RandomState(0)
    for d in ('input', 'embeddings', 'statistics'):
        (tmp_path / d).mkdir()
    emb = []
    for i in range(2):
        (tmp_path / 'input' / ('shard_%d' % i)).write_text('0\n')
        e = rs.randn(30, 3) + 4.0 * (i == 1)
        np.save(str(tmp_path / 'embeddings' / ('shard_%d.embedding.npy' % i)), e)
        emb.append(e)
    opts = dict(input=str(tmp_path / 'input'), embeddings=str(tmp_path / 'embeddings'), statistics=str(tmp_path / 'statistics'), M=40, Q=3,
                load=False)
    np.random.seed(3)
    opts, gs = Dr.init_statistics(Files, opts)
    assert gs['Z'].shape == (40, 3) and gs['alpha'].shape == (1, 3) and float(gs['sf2']) == 1.0 and float(gs['beta']) == 1.0
    # the same draws by hand: k-means over BOTH shards (the first has fewer than M points, :172-176), then the jitter
    import scipy.cluster.vq as cl
    np.random.seed(3)
    allp = np.concatenate(emb)
    Zk = cl.kmeans(allp, 40)[0]
    if Zk.shape[0] < 40:
        Zk = np.concatenate((Zk, allp[:40 - Zk.shape[0]]))
    np.testing.assert_allclose(gs['Z'], Zk + np.random.randn(40, 3) * 0.05)
    assert len(opts['flat_global_statistics_bounds']) == 40 * 3 + 1 + 3 + 1
    x0 = Dr.initial_flat_vector(opts, gs)
    np.testing.assert_allclose(np.log(1 + np.exp(x0[-1])), 1.0)
    for key in ('Z', 'sf2', 'alpha', 'beta'):
        np.save(str(tmp_path / 'statistics' / ('global_statistics_%s_f.npy' % key)), gs[key] * 2.0)
    opts['load'] = True
    _, gs2 = Dr.init_statistics(Files, opts)
    np.testing.assert_array_equal(gs2['Z'], gs['Z'] * 2.0)
