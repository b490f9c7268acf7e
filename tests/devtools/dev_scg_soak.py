"""Developer soak (GPU box): a GPLVM optimisation with everything resident (ResidentModel + SCG_adapted): the bound must
decrease monotonically over accepted steps and stay finite; prints the time per function evaluation."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from gparml_amd.resident import ResidentCG, ResidentModel
from gparml_amd.scg_adapted import SCG_adapted
from gparml_amd.driver import transform_back

N, D, M, Q = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (50000, 10, 64, 3)))
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 30
fixed = len(sys.argv) > 6 and sys.argv[6] == 'fixed'      # sparse GP regression: embeddings fixed (X_S = 0), hyper-parameters and Z optimised
rs = np.random.RandomState(0)
X = rs.randn(N, Q)
Y = np.sin(X.dot(rs.randn(Q, D))) + 0.1 * rs.randn(N, D)
X_mu = X + 0.3 * rs.randn(N, Q)                         # perturbed start
X_S_raw = np.full((N, Q), float(np.log(np.exp(0.5) - 1)))  # softplus-inverse of 0.5
if fixed:
    X_mu, X_S_raw = X, np.zeros((N, Q))
shards = [(Y[:N // 2], X_mu[:N // 2], X_S_raw[:N // 2]), (Y[N // 2:], X_mu[N // 2:], X_S_raw[N // 2:])]
model = ResidentModel(shards, M, Q, D, fixed_embeddings=fixed)
Z = X_mu[rs.permutation(N)[:M]] + 0.05 * rs.randn(M, Q)
x0 = np.concatenate([Z.ravel(), [1.0], np.full(Q, 1.0), [10.0]])
x0 = np.array([transform_back(b, v) for b, v in zip(model.bounds, x0)])
calls = []
def f_and_g(x, it, step=0):
    t = time.time(); f, g = model.likelihood_and_gradient(x, it, step); calls.append((f, time.time() - t)); return f, g
t0 = time.time()
x, flog, nfe, status = SCG_adapted(f_and_g, x0, ResidentCG(model), fixed_embeddings=fixed, maxiters=iters, xtol=0, ftol=0, gtol=0)
dt = time.time() - t0
fl = [float(f) for f in flog]
print('evaluations %d in %.2f s (%.1f ms each), F: %.6e -> %.6e, monotone=%s finite=%s' % (len(calls), dt, 1e3 * dt / len(calls), fl[0], fl[-1],
      all(b <= a + 1e-9 * abs(a) for a, b in zip(fl, fl[1:])), bool(np.all(np.isfinite(fl)))))
model.close()
