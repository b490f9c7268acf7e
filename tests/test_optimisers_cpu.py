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
