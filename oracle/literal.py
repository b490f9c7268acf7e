"""Literal CPU restatement of the reference algorithm (numpy, float64).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  This module follows the reference's
*algorithm* — per-point psi2 matrices are materialised as an (N_s, M, M) tensor exactly as
partial_terms.py:45 does — so it is only usable at small N.  Every function cites the
reference lines (paths relative to /root/reference) it restates.  It is pinned against
golden vectors generated from the imported reference (tests/golden/).

Conventions: ``alpha`` is the inverse squared lengthscale (alpha = ard**-2), ``sf2`` the
signal variance, ``beta`` the noise precision, ``S`` the diagonal variances of q(X).
"""
import numpy as np


# --------------------------------------------------------------------------- kernel pieces
def rbf_gram(Z, sf2, alpha, Z2=None):
    """ARD-RBF Gram matrix.  kernels.py:72-113 (cdist 'seuclidean' with V = 2*ard**2,
    then sf**2 * exp(-d**2)), i.e. sf2 * exp(-0.5 * sum_q alpha_q (z_q - z'_q)**2)."""
    Z = np.atleast_2d(Z)
    Z2 = Z if Z2 is None else np.atleast_2d(Z2)
    diff = Z[:, None, :] - Z2[None, :, :]
    d2 = np.sum(diff * diff * (0.5 * np.asarray(alpha))[None, None, :], axis=2)
    return sf2 * np.exp(-d2)


def _check_inputs(alpha, mu, S):
    # kernel_exp.py:30-34, 68-72, 128-132
    assert np.all(S >= 0.0)
    assert np.all(np.asarray(alpha) >= 0.0)
    assert mu.ndim == 2 and S.ndim == 2
    assert mu.shape[1] == S.shape[1]


def psi1(Z, sf2, alpha, mu, S):
    """<K_nm>_q(x_n) for every point: (N, M).  kernel_exp.py:51-82 (line 80)."""
    _check_inputs(alpha, mu, S)
    assert mu.shape == S.shape
    alpha = np.asarray(alpha, dtype=float)
    denom = alpha[None, :] * S + 1.0                                   # (N, Q)
    norm = sf2 / np.sqrt(denom).prod(axis=1)                           # (N,)
    delta = Z[None, :, :] - mu[:, None, :]                             # (N, M, Q)
    quad = np.sum(delta * delta * alpha[None, None, :] / denom[:, None, :], axis=2)
    return norm[:, None] * np.exp(-0.5 * quad)


def psi2_point(Z, sf2, alpha, mu_n, S_n):
    """<K_mn K_nm'>_q(x_n) for ONE point: (M, M).  kernel_exp.py:126-148 (lines 143-146);
    the scalar triple-loop twin is kernel_exp.py:84-124."""
    alpha = np.asarray(alpha, dtype=float)
    mu_n = np.asarray(mu_n, dtype=float).reshape(-1)
    S_n = np.asarray(S_n, dtype=float).reshape(-1)
    assert np.all(S_n >= 0.0) and np.all(alpha >= 0.0)
    two_as1 = 2.0 * alpha * S_n + 1.0
    const = sf2 * sf2 / np.sqrt(np.prod(two_as1))
    dz = Z[:, None, :] - Z[None, :, :]
    zbar = 0.5 * (Z[:, None, :] + Z[None, :, :])
    t1 = -0.25 * np.sum(alpha[None, None, :] * dz * dz, axis=2)
    t2 = -np.sum(alpha[None, None, :] * (mu_n[None, None, :] - zbar) ** 2 / two_as1[None, None, :], axis=2)
    return const * np.exp(t1 + t2)


def psi2_point_scalar(Z, sf2, alpha, mu_n, S_n):
    """Scalar twin of psi2_point.  kernel_exp.py:84-124 (the *_old triple loop)."""
    M, Q = Z.shape
    alpha = np.asarray(alpha, dtype=float)
    out = np.zeros((M, M))
    const = sf2 * sf2
    for q in range(Q):
        const /= np.sqrt(2.0 * alpha[q] * S_n[q] + 1.0)
    for a in range(M):
        for b in range(M):
            e = 0.0
            for q in range(Q):
                e -= 0.25 * alpha[q] * (Z[a, q] - Z[b, q]) ** 2
                e -= alpha[q] * (mu_n[q] - 0.5 * Z[a, q] - 0.5 * Z[b, q]) ** 2 / (2.0 * alpha[q] * S_n[q] + 1.0)
            out[a, b] = const * np.exp(e)
    return out


def psi1_T_Y(Z, sf2, alpha, mu, S, Y):
    """Psi1^T Y, (M, D).  kernel_exp.py:13-49 (sum of N outer products, lines 44-47)."""
    P1 = psi1(Z, sf2, alpha, mu, S)
    assert P1.shape[0] == Y.shape[0]
    return P1.T.dot(Y)


# --------------------------------------------------------------------------- the class
class PartialTermsOracle(object):
    """Restatement of ``class partial_terms`` (partial_terms.py:15-473).  Method names match
    the reference so that parity tests read like the reference's own tests (test.py)."""

    def __init__(self, Z, sf2, alpha, beta, M, Q, N, D, update_global_statistics=True):
        # partial_terms.py:16-36
        self.Z = np.array(Z, dtype=float)
        self.M, self.Q, self.N, self.D = M, Q, N, D
        self.beta = float(beta)
        self.sf2 = float(sf2)
        self.alpha = np.atleast_1d(np.array(alpha, dtype=float).squeeze())
        if update_global_statistics:
            self.update_global_statistics()

    # ---- statistics -----------------------------------------------------------------
    def update_global_statistics(self):
        # partial_terms.py:89-95: Kmm and its LU inverse (no jitter)
        self.Kmm = rbf_gram(self.Z, self.sf2, self.alpha)
        self.Kmm_inv = np.linalg.inv(self.Kmm)

    def set_global_statistics(self, Kmm, Kmm_inv):
        # partial_terms.py:70-72
        self.Kmm, self.Kmm_inv = Kmm, Kmm_inv

    def set_data(self, Y, X_mu, X_S, is_set_statistics=True):
        # partial_terms.py:38-52
        self.Y, self.X_mu, self.X_S = Y, X_mu, X_S
        self.sum_YYT = float(np.sum(Y * Y))                               # :40
        self.local_N = X_mu.shape[0]
        self.exp_K_mi_K_im = np.zeros((self.local_N, self.M, self.M))       # :45 (the big one)
        for n in range(self.local_N):
            self.exp_K_mi_K_im[n] = psi2_point(self.Z, self.sf2, self.alpha, X_mu[n], X_S[n])
        self.exp_K_mi = psi1(self.Z, self.sf2, self.alpha, X_mu, X_S)        # :49
        if is_set_statistics:
            self.update_local_statistics()

    def update_local_statistics(self):
        # partial_terms.py:74-87
        self.sum_exp_K_mi_K_im = self.exp_K_mi_K_im.sum(0)
        self.exp_K_miY = psi1_T_Y(self.Z, self.sf2, self.alpha, self.X_mu, self.X_S, self.Y)
        self.sum_exp_K_ii = self.sf2 * self.local_N
        self.Kmm_plus_op_inv = np.linalg.inv(self.Kmm + self.beta * self.sum_exp_K_mi_K_im)
        if not np.all(self.X_S == 0):
            mu2 = np.sum(self.X_mu * self.X_mu, axis=1)
            self.KL = 0.5 * np.sum(np.sum(self.X_S - np.log(self.X_S), 1) + mu2 - self.Q)
        else:
            self.KL = 0

    def set_local_statistics(self, sum_YYT, sum_exp_K_mi_K_im, exp_K_miY, sum_exp_K_ii, KL):
        # partial_terms.py:54-61
        self.sum_YYT = sum_YYT
        self.sum_exp_K_mi_K_im = sum_exp_K_mi_K_im
        self.exp_K_miY = exp_K_miY
        self.sum_exp_K_ii = sum_exp_K_ii
        self.Kmm_plus_op_inv = np.linalg.inv(self.Kmm + self.beta * sum_exp_K_mi_K_im)
        self.KL = KL

    def get_local_statistics(self):
        # partial_terms.py:63-68
        return dict(sum_YYT=self.sum_YYT, sum_exp_K_mi_K_im=self.sum_exp_K_mi_K_im,
                    exp_K_miY=self.exp_K_miY, sum_exp_K_ii=self.sum_exp_K_ii, KL=self.KL)

    # ---- bound ------------------------------------------------------------------------
    def logmarglik(self):
        # partial_terms.py:436-473
        A = self.Kmm + self.beta * self.sum_exp_K_mi_K_im
        s1, ld_K = np.linalg.slogdet(self.Kmm)
        s2, ld_A = np.linalg.slogdet(A)
        if s1 < 0:                                                            # :452-453
            s1, ld_K = np.linalg.slogdet(self.Kmm + 1e-7 * np.eye(self.M))
        if s2 < 0:                                                            # :454-456
            A = A + 1e-7 * np.eye(self.M)
            s2, ld_A = np.linalg.slogdet(A)
        assert s1 >= 0.0 and s2 >= 0.0                                        # :459-461
        C = self.exp_K_miY
        return (-0.5 * self.N * self.D * np.log(2.0 * np.pi)
                + 0.5 * self.D * self.N * np.log(self.beta)
                + 0.5 * self.D * ld_K
                - 0.5 * self.D * ld_A
                - 0.5 * self.beta * self.sum_YYT
                - 0.5 * self.beta * self.D * self.sum_exp_K_ii
                + 0.5 * self.beta * self.D * np.trace(self.Kmm_inv.dot(self.sum_exp_K_mi_K_im))
                + 0.5 * self.beta ** 2 * np.trace(C.T.dot(np.linalg.inv(A).dot(C)))
                - self.KL)

    # ---- partials of F w.r.t. the statistics -----------------------------------------------
    def _PCCtP(self):
        P, C = self.Kmm_plus_op_inv, self.exp_K_miY
        return P.dot(C.dot(C.T.dot(P)))

    def dF_dKmm(self):
        # partial_terms.py:102-113
        Ki, P, b, D = self.Kmm_inv, self.Kmm_plus_op_inv, self.beta, self.D
        return (0.5 * D * Ki - 0.5 * D * P
                - 0.5 * b * D * Ki.dot(self.sum_exp_K_mi_K_im.dot(Ki))
                - 0.5 * b * b * self._PCCtP())

    def dF_dexp_K_miY(self):
        # partial_terms.py:115-121
        return self.beta ** 2 * self.Kmm_plus_op_inv.dot(self.exp_K_miY)

    def dF_dexp_K_mi_K_im(self):
        # partial_terms.py:123-131
        b, D = self.beta, self.D
        return -0.5 * b * D * self.Kmm_plus_op_inv + 0.5 * b * D * self.Kmm_inv - 0.5 * b ** 3 * self._PCCtP()

    def dF_dexp_K_ii(self):
        # partial_terms.py:133-138
        return -0.5 * self.beta * self.D

    # ---- Z ------------------------------------------------------------------------------
    def dKmm_dZ(self):
        # partial_terms.py:146-160 -> (M, Q, M): K[j,m'] * -alpha_k * (z_jk - z_m'k)
        K = rbf_gram(self.Z, self.sf2, self.alpha)
        dz = self.Z[:, :, None] - self.Z.T[None, :, :]
        return K[:, None, :] * (-self.alpha)[None, :, None] * dz

    def dexp_K_miY_dZ(self):
        # partial_terms.py:162-188 -> (M, Q, D)
        a = self.alpha
        fac = a[None, None, :] * (self.X_mu[:, None, :] - self.Z[None, :, :]) / (a[None, None, :] * self.X_S[:, None, :] + 1.0)
        w = self.exp_K_mi[:, :, None] * fac                                   # (N, M, Q)
        return np.einsum('nmq,nd->mqd', w, self.Y)

    def dexp_K_mi_K_im_dZ(self):
        # partial_terms.py:190-205 -> (M, Q, M)
        a = self.alpha
        Zc = self.Z[:, :, None]
        Zr = self.Z.T[None, :, :]
        out = np.zeros((self.M, self.Q, self.M))
        for n in range(self.local_N):
            mu = self.X_mu[n][None, :, None]
            s = self.X_S[n][None, :, None]
            fac = (-0.5 * a[None, :, None] * (Zc - Zr)
                   + 0.5 * a[None, :, None] * (2.0 * mu - Zc - Zr) / (2.0 * a[None, :, None] * s + 1.0))
            out += self.exp_K_mi_K_im[n][:, None, :] * fac
        return out

    def grad_Z(self, dF_dKmm, dKmm_dZ, dF_dexp_K_miY, dexp_K_miY_dZ, dF_dexp_K_mi_K_im, dexp_K_mi_K_im_dZ):
        # partial_terms.py:207-240 (symmetrised Kmm term :227-231, factor 2 at :238)
        sym = dF_dKmm + dF_dKmm.T
        g = np.einsum('jm,jkm->jk', sym, dKmm_dZ)
        # the reference writes row j AND column j of an (M,M) mask; entry (j,j) is counted once,
        # and dKmm_dZ[j,k,j] == 0 anyway.
        g += np.einsum('jd,jkd->jk', dF_dexp_K_miY, dexp_K_miY_dZ)
        g += 2.0 * np.einsum('jm,jkm->jk', dF_dexp_K_mi_K_im, dexp_K_mi_K_im_dZ)
        return g

    # ---- alpha ---------------------------------------------------------------------------
    def dKmm_dalpha(self):
        # partial_terms.py:247-254 -> (Q, M, M)
        dz = self.Z[:, None, :] - self.Z[None, :, :]
        return -0.5 * self.Kmm[None, :, :] * np.transpose(dz * dz, (2, 0, 1))

    def dexp_K_miY_dalpha(self):
        # partial_terms.py:256-271 -> (Q, M, D)
        a = self.alpha
        den = a[None, :] * self.X_S + 1.0                                     # (N, Q)
        d = (self.X_mu[:, None, :] - self.Z[None, :, :]) / den[:, None, :]      # (N, M, Q)
        v = -0.5 * self.exp_K_mi[:, :, None] * (d * d + (self.X_S / den)[:, None, :])
        return np.einsum('nmq,nd->qmd', v, self.Y)

    def dexp_K_mi_K_im_dalpha(self):
        # partial_terms.py:273-284 -> (Q, M, M)
        a = self.alpha
        dz = self.Z[:, None, :] - self.Z[None, :, :]                            # (M, M, Q)
        zs = self.Z[:, None, :] + self.Z[None, :, :]
        out = np.zeros((self.Q, self.M, self.M))
        for n in range(self.local_N):
            den = 2.0 * a * self.X_S[n] + 1.0
            f = (-0.25 * dz * dz
                 - 0.25 * ((2.0 * self.X_mu[n][None, None, :] - zs) / den[None, None, :]) ** 2
                 - (self.X_S[n] / den)[None, None, :])
            out += self.exp_K_mi_K_im[n][None, :, :] * np.transpose(f, (2, 0, 1))
        return out

    def grad_alpha(self, dF_dKmm, dKmm_dalpha, dF_dexp_K_miY, dexp_K_miY_dalpha, dF_dexp_K_mi_K_im, dexp_K_mi_K_im_dalpha):
        # partial_terms.py:286-299 (no factor 2 here)
        return (np.einsum('ab,qab->q', dF_dKmm, dKmm_dalpha)
                + np.einsum('ab,qab->q', dF_dexp_K_miY, dexp_K_miY_dalpha)
                + np.einsum('ab,qab->q', dF_dexp_K_mi_K_im, dexp_K_mi_K_im_dalpha))

    # ---- sf2 -----------------------------------------------------------------------------
    def dKmm_dsf2(self):
        return self.Kmm / self.sf2                                             # :306-308

    def dexp_K_miY_dsf2(self):
        return self.exp_K_miY / self.sf2                                       # :310-312

    def dexp_K_mi_K_im_dsf2(self):
        return 2.0 * self.sum_exp_K_mi_K_im / self.sf2                         # :314-316

    def dexp_K_ii_dsf2(self):
        return self.local_N                                                    # :318-320 (an int)

    def grad_sf2(self, dF_dKmm, dKmm_dsf2, dF_dexp_K_ii, dexp_K_ii_dsf2, dF_dexp_K_miY, dexp_K_miY_dsf2,
                 dF_dexp_K_mi_K_im, dexp_K_mi_K_im_dsf2):
        # partial_terms.py:322-333
        return (np.sum(dF_dKmm * dKmm_dsf2) + dF_dexp_K_ii * dexp_K_ii_dsf2
                + np.sum(dF_dexp_K_miY * dexp_K_miY_dsf2) + np.sum(dF_dexp_K_mi_K_im * dexp_K_mi_K_im_dsf2))

    # ---- beta ----------------------------------------------------------------------------
    def grad_beta(self):
        # partial_terms.py:340-360 (uses the GLOBAL N)
        N, D, b = self.N, self.D, self.beta
        P, Ki, C, Psi2 = self.Kmm_plus_op_inv, self.Kmm_inv, self.exp_K_miY, self.sum_exp_K_mi_K_im
        return (0.5 * N * D / b
                - 0.5 * D * np.trace(P.dot(Psi2))
                - 0.5 * self.sum_YYT
                - 0.5 * D * self.sum_exp_K_ii
                + 0.5 * D * np.trace(Ki.dot(Psi2))
                + b * np.trace(C.T.dot(P.dot(C)))
                - 0.5 * b * b * np.trace(C.T.dot(P.dot(Psi2).dot(P).dot(C))))

    # ---- per-point ----------------------------------------------------------------------
    def grad_X_mu(self):
        # partial_terms.py:367-398
        a = self.alpha
        Abar = self.dF_dexp_K_miY()                                            # (M, D)
        Bbar = self.dF_dexp_K_mi_K_im()                                        # (M, M)
        g = np.zeros((self.local_N, self.Q))
        for n in range(self.local_N):
            g[n] = -self.X_mu[n]                                               # :385 (5.72), also when S == 0
            ay = Abar.dot(self.Y[n])                                           # (M,)
            for q in range(self.Q):
                f1 = -a[q] * (self.X_mu[n, q] - self.Z[:, q]) / (a[q] * self.X_S[n, q] + 1.0)
                zsum = self.Z[:, None, q] + self.Z[None, :, q]
                f2 = -a[q] * (2.0 * self.X_mu[n, q] - zsum) / (2.0 * a[q] * self.X_S[n, q] + 1.0)
                g[n, q] += np.sum(ay * self.exp_K_mi[n] * f1) + np.sum(Bbar * self.exp_K_mi_K_im[n] * f2)
        return g

    def grad_X_S(self):
        # partial_terms.py:400-431
        a = self.alpha
        Abar = self.dF_dexp_K_miY()
        Bbar = self.dF_dexp_K_mi_K_im()
        g = np.zeros((self.local_N, self.Q))
        for n in range(self.local_N):
            g[n] = -0.5 * (1.0 - 1.0 / self.X_S[n])                            # :417 (5.73)
            ay = Abar.dot(self.Y[n])
            for q in range(self.Q):
                den1 = a[q] * self.X_S[n, q] + 1.0
                f1 = 0.5 * (a[q] * (self.X_mu[n, q] - self.Z[:, q]) / den1) ** 2 - 0.5 * a[q] / den1
                zsum = self.Z[:, None, q] + self.Z[None, :, q]
                f2 = (2.0 * (a[q] * (2.0 * self.X_mu[n, q] - zsum) / (4.0 * a[q] * self.X_S[n, q] + 2.0)) ** 2
                      - a[q] / (2.0 * a[q] * self.X_S[n, q] + 1.0))
                g[n, q] += np.sum(ay * self.exp_K_mi[n] * f1) + np.sum(Bbar * self.exp_K_mi_K_im[n] * f2)
        return g


# --------------------------------------------------------------------------- softplus transforms
LIM_VAL = -np.log(np.finfo(float).eps)        # supporting_functions.py:125


def transformVar(x):
    """softplus, supporting_functions.py:153-156"""
    assert np.all(-LIM_VAL < x) and np.all(x < LIM_VAL)
    return np.log(1.0 + np.exp(x))


def transformVar_back(x):
    """softplus inverse, supporting_functions.py:159-162"""
    assert np.all(np.finfo(float).eps < x) and np.all(x < LIM_VAL)
    return np.log(np.exp(x) - 1.0)


def transformVar_grad(x):
    """d softplus / dx, supporting_functions.py:165-168"""
    assert np.all(-LIM_VAL < x) and np.all(x < LIM_VAL)
    return 1.0 / (np.exp(-x) + 1.0)


# --------------------------------------------------------------------------- one full evaluation
def full_evaluation(Z, sf2, alpha, beta, Y, X_mu, X_S, N_global=None, with_embeddings=True):
    """One bound+gradient evaluation for a single shard, following the sequential call sequence of
    scg_adapted-example.py:140-203 / parallel_GPLVM.py:302-369.  Returns a dict with F, the gradients
    w.r.t. Z, sf2, alpha, beta and (optionally) X_mu, X_S, plus the accumulated statistics."""
    M, Q = Z.shape
    N_s, D = Y.shape
    N = N_s if N_global is None else N_global
    pt = PartialTermsOracle(Z, sf2, alpha, beta, M, Q, N, D)
    pt.set_data(Y, X_mu, X_S, True)
    out = {}
    out['F'] = pt.logmarglik()
    dKmm, dC, dPsi2, dPsi0 = pt.dF_dKmm(), pt.dF_dexp_K_miY(), pt.dF_dexp_K_mi_K_im(), pt.dF_dexp_K_ii()
    out['dF_dKmm'], out['dF_dexp_K_miY'], out['dF_dexp_K_mi_K_im'], out['dF_dexp_K_ii'] = dKmm, dC, dPsi2, dPsi0
    out['grad_Z'] = pt.grad_Z(dKmm, pt.dKmm_dZ(), dC, pt.dexp_K_miY_dZ(), dPsi2, pt.dexp_K_mi_K_im_dZ())
    out['grad_alpha'] = pt.grad_alpha(dKmm, pt.dKmm_dalpha(), dC, pt.dexp_K_miY_dalpha(), dPsi2, pt.dexp_K_mi_K_im_dalpha())
    out['grad_sf2'] = pt.grad_sf2(dKmm, pt.dKmm_dsf2(), dPsi0, pt.dexp_K_ii_dsf2(), dC, pt.dexp_K_miY_dsf2(),
                                  dPsi2, pt.dexp_K_mi_K_im_dsf2())
    out['grad_beta'] = pt.grad_beta()
    if with_embeddings:
        out['grad_X_mu'] = pt.grad_X_mu()
        if not np.all(X_S == 0):
            out['grad_X_S'] = pt.grad_X_S()
    out.update(pt.get_local_statistics())
    out['Kmm'], out['Kmm_inv'] = pt.Kmm, pt.Kmm_inv
    return out
