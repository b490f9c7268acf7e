"""Drop-in for the reference's ``partial_terms.partial_terms`` class (partial_terms.py:15-473), computed on the GPU.

Same constructor, method names, argument meaning, array shapes and exception classes as the reference, so that
``test.py``-style code, ``local_MapReduce.load_partial_terms`` (local_MapReduce.py:403-409) and
``parallel_GPLVM.calculate_global_statistics / calculate_global_derivatives`` (parallel_GPLVM.py:302-369) run
against it unchanged.  Every number comes from ``libgparml_hip.so``; there is no CPU fallback.

State model.  The reference keeps mutable attributes (``.Z``, ``.beta``, ``.hyp.sf``, ``.hyp.ard``) that callers
change between ``set_data`` / ``update_global_statistics`` calls (test.py:72,109,171,192).  Here the attributes
are re-read on every call: ``set_data`` (re)computes the Psi statistics with the values current at that moment,
and anything derived from the M x M algebra (``logmarglik``, ``dF_d*``, ``grad_*``) is recomputed lazily when the
statistics or the hyper-parameters changed since the last global step.
"""
import numpy as np

from . import _lib
from .engine import ShardEngine


class ArdHypers(object):
    """kernels.ArdHypers (kernels.py:11-35): sf = sqrt(sf2), ard = lengthscales = alpha**-0.5."""

    def __init__(self, D, sf=1.0, ll=1.0, ard=None):
        self.D = D
        self.sf = sf
        if ard is None:
            self.ard = np.ones(D) * ll
        else:
            self.ard = np.atleast_1d(np.array(ard, dtype=float).squeeze())
            assert self.ard.ndim == 1

    @property
    def ll(self):
        if np.all(self.ard == self.ard[0]):
            return self.ard[0]
        raise ValueError("RBF kernel is not isotropic")

    @ll.setter
    def ll(self, value):
        self.ard = np.ones(self.D) * value


class partial_terms(object):
    def __init__(self, Z, sf2, alpha, beta, M, Q, N, D, update_global_statistics=True, device=0):
        # partial_terms.py:16-36
        self.Z = Z
        self.M, self.Q, self.N, self.D = int(M), int(Q), N, int(D)
        self.beta = beta
        self.hyp = ArdHypers(self.Q, sf=sf2 ** 0.5, ard=np.asarray(alpha, dtype=float) ** -0.5)
        self.device = device
        self._eng = None
        self._have_data = False
        self._stats_on_device = False      # the device stats buffer holds (local or injected) statistics
        self._gstep_key = None             # hyper-parameter snapshot of the last global step
        self._p2_done = False
        self.local_N = None
        if update_global_statistics:
            self.update_global_statistics()

    # ------------------------------------------------------------------ plumbing
    def _alpha(self):
        ard = np.atleast_1d(np.asarray(self.hyp.ard, dtype=float))
        assert np.all(ard >= 0.0)                                      # kernel_exp.py:31
        with np.errstate(divide='ignore'):
            return ard ** -2.0

    def _sf2(self):
        sf = float(np.asarray(self.hyp.sf).reshape(-1)[0])
        assert sf > 0.0                                                # kernels.py:62
        return sf * sf

    def _key(self):
        Z = np.ascontiguousarray(self.Z, dtype=float)
        return (Z.tobytes(), self._sf2(), self._alpha().tobytes(), float(np.asarray(self.beta).reshape(-1)[0]), int(self.N))

    def _engine(self, N_s=None):
        if self._eng is None or (N_s is not None and self._eng.N_s != N_s):
            if self._eng is not None:
                self._eng.close()
            self._eng = ShardEngine(1 if N_s is None else N_s, self.D, self.M, self.Q, device=self.device)
            self._eng.set_timing(0)
            self._have_data = False
            self._stats_on_device = False
            self._gstep_key = None
            self._pushed = None
        return self._eng

    _pushed = None

    def _push_globals(self):
        """gp_set_globals with the current attributes -- once per distinct (engine, Z, sf2, alpha, beta, N): the call starts a new evaluation on the device
        (in the library's poison test mode it refills every per-evaluation buffer with NaN), and predict.py's sequence set_data -> get_local_statistics ->
        set_local_statistics -> logmarglik -> grad_X_mu (predict.py:116-144) reads the Psi1 of ITS set_data after the statistics were replaced: pushing
        unchanged globals a second time in between would declare that Psi1 stale."""
        eng = self._engine()
        Ng = int(self.N) if self.N is not None else eng.N_s
        Z = np.ascontiguousarray(np.asarray(self.Z, dtype=float).reshape(self.M, self.Q))
        sf2, alpha, beta = self._sf2(), self._alpha(), float(np.asarray(self.beta).reshape(-1)[0])
        Nglob = max(Ng, eng.N_s if self._have_data else 1)
        key = (id(eng), eng.h.value if hasattr(eng.h, 'value') else eng.h, Z.tobytes(), sf2, alpha.tobytes(), beta, Nglob)
        if self._pushed == key:
            return
        eng.set_globals(Z, sf2, alpha, beta, N_global=Nglob)
        self._pushed = key

    def _ensure_gstep(self):
        """Global step (Cholesky, F, partials) up to date with the current attributes and statistics."""
        key = self._key()
        if self._gstep_key == key:
            return
        assert self._stats_on_device, 'no statistics: call set_data(...) or set_local_statistics(...) first'
        eng = self._engine()
        if self._stats_from == 'local':
            self._run_phase1()                 # re-reads Z, sf2, alpha, beta, N and recomputes the local statistics
        else:
            self._push_globals()
            eng.set_local_statistics(*self._injected)
        eng.global_step()
        self._gstep_key = key
        self._p2_done = False
        self.Kmm_plus_op_inv = eng.download('KMM_PLUS_OP_INV')

    def _run_phase1(self):
        eng = self._engine()
        self._push_globals()
        eng.phase1()
        k = self._key()
        self._local_key = (k[0], k[1], k[2])
        self._stats_from = 'local'
        self._stats_on_device = True
        self._gstep_key = None

    _stats_from = None
    _local_key = None
    _injected = None

    # ------------------------------------------------------------------ statistics
    def set_data(self, Y, X_mu, X_S, is_set_statistics=True):
        # partial_terms.py:38-52
        Y = np.asarray(Y, dtype=float)
        if Y.ndim == 1:
            Y = Y[:, None]
        X_mu = np.asarray(X_mu, dtype=float)
        X_S = np.asarray(X_S, dtype=float)
        assert np.all(X_S >= 0.0)                                      # kernel_exp.py:30
        assert X_mu.ndim == 2 and X_S.ndim == 2 and X_mu.shape == X_S.shape
        self.Y, self.X_mu, self.X_S = Y, X_mu, X_S
        self.local_N = X_mu.shape[0]
        eng = self._engine(self.local_N)
        eng.upload_shard(Y, X_mu, X_S, xs_is_raw=False)
        self._have_data = True
        self._run_phase1()
        self.sum_YYT = eng.scalars()['sum_YYT']
        if is_set_statistics:
            self.update_local_statistics()
        else:
            # embeddings_mapper path (local_MapReduce.py:348-354): the global sums arrive through set_local_statistics
            self._stats_on_device = False

    @property
    def exp_K_mi(self):
        """(N_s, M) Psi1, partial_terms.py:49"""
        assert self._have_data
        return self._engine().download('PSI1')

    @property
    def exp_K_mi_K_im(self):
        """(N_s, M, M) per-point psi2, partial_terms.py:45-48 (compat: the fast path never stores it)"""
        assert self._have_data
        return self._engine().download('PSI2_POINTS')

    def update_local_statistics(self):
        # partial_terms.py:74-87
        assert self._have_data
        eng = self._engine()
        k = self._key()
        if self._stats_from != 'local' or self._local_key != (k[0], k[1], k[2]):
            self._run_phase1()
        self._stats_from = 'local'
        self._stats_on_device = True
        sc = eng.scalars()
        self.sum_exp_K_mi_K_im = eng.download('PSI2_SUM')
        self.exp_K_miY = eng.download('PSI1TY')
        self.sum_exp_K_ii = sc['sum_exp_K_ii']
        self.sum_YYT = sc['sum_YYT']
        self.KL = sc['KL']
        self._gstep_key = None
        self._ensure_gstep()                                           # Kmm_plus_op_inv (:82)

    def set_local_statistics(self, sum_YYT, sum_exp_K_mi_K_im, exp_K_miY, sum_exp_K_ii, KL):
        # partial_terms.py:54-61
        self.sum_YYT = sum_YYT
        self.sum_exp_K_mi_K_im = np.asarray(sum_exp_K_mi_K_im, dtype=float)
        self.exp_K_miY = np.asarray(exp_K_miY, dtype=float).reshape(self.M, self.D)
        self.sum_exp_K_ii = sum_exp_K_ii
        self.KL = KL
        self._injected = (float(np.asarray(sum_YYT)), self.sum_exp_K_mi_K_im, self.exp_K_miY, float(np.asarray(sum_exp_K_ii)),
                          float(np.asarray(KL)))
        self._stats_from = 'injected'
        self._stats_on_device = True
        self._gstep_key = None
        self._ensure_gstep()

    def get_local_statistics(self):
        # partial_terms.py:63-68
        return {'sum_YYT': self.sum_YYT, 'sum_exp_K_mi_K_im': self.sum_exp_K_mi_K_im, 'exp_K_miY': self.exp_K_miY,
                'sum_exp_K_ii': self.sum_exp_K_ii, 'KL': self.KL}

    def set_global_statistics(self, Kmm, Kmm_inv):
        # partial_terms.py:70-72.  Kept as attributes; the device rebuilds Kmm from Z (cheaper than shipping it).
        self.Kmm = Kmm
        self.Kmm_inv = Kmm_inv

    def update_global_statistics(self):
        # partial_terms.py:89-95
        eng = self._engine()
        self._push_globals()
        if not self._stats_on_device:
            # Kmm alone: run the global step on zero statistics (Kmm + beta*0), only Kmm / Kmm_inv are read
            eng.set_local_statistics(0.0, np.zeros((self.M, self.M)), np.zeros((self.M, self.D)), 0.0, 0.0)
            eng.global_step()
        else:
            self._gstep_key = None
            self._ensure_gstep()
        self.Kmm = eng.download('KMM')
        self.Kmm_inv = eng.download('KMM_INV')

    # ------------------------------------------------------------------ bound and partials
    def logmarglik(self):
        # partial_terms.py:436-473
        self._ensure_gstep()
        return self._engine().scalars()['F']

    def dF_dKmm(self):
        self._ensure_gstep()                                           # partial_terms.py:102-113
        return self._engine().download('DF_DKMM')

    def dF_dexp_K_miY(self):
        self._ensure_gstep()                                           # partial_terms.py:115-121
        return self._engine().download('DF_DPSI1TY')

    def dF_dexp_K_mi_K_im(self):
        self._ensure_gstep()                                           # partial_terms.py:123-131
        return self._engine().download('DF_DPSI2')

    def dF_dexp_K_ii(self):
        return -0.5 * float(np.asarray(self.beta).reshape(-1)[0]) * self.D   # partial_terms.py:133-138

    # ------------------------------------------------------------------ derivative tensors (compat layouts)
    def _compat(self, name):
        self._ensure_gstep()
        return self._engine().download(name)

    def dKmm_dZ(self):
        return self._compat('DKMM_DZ')                                 # partial_terms.py:146-160  (M,Q,M)

    def dexp_K_miY_dZ(self):
        return self._compat('DPSI1TY_DZ')                              # partial_terms.py:162-188  (M,Q,D)

    def dexp_K_mi_K_im_dZ(self):
        return self._compat('DPSI2_DZ')                                # partial_terms.py:190-205  (M,Q,M)

    def dKmm_dalpha(self):
        return self._compat('DKMM_DALPHA')                             # partial_terms.py:247-254  (Q,M,M)

    def dexp_K_miY_dalpha(self):
        return self._compat('DPSI1TY_DALPHA')                          # partial_terms.py:256-271  (Q,M,D)

    def dexp_K_mi_K_im_dalpha(self):
        return self._compat('DPSI2_DALPHA')                            # partial_terms.py:273-284  (Q,M,M)

    def _from_parts(self, which, a, a3, b, b3, c, c3, out):
        eng = self._engine()
        ptrs = []
        keep = []
        for arr, shape in ((a, (self.M, self.M)), (a3, None), (b, (self.M, self.D)), (b3, None), (c, (self.M, self.M)), (c3, None)):
            x = np.ascontiguousarray(arr, dtype=np.float64)
            if shape is not None:
                x = np.ascontiguousarray(x.reshape(shape))
            keep.append(x)
            ptrs.append(x.ctypes.data_as(_lib._dp))
        rc = eng.lib.gp_grad_from_parts(eng.h, which, ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], ptrs[5], out.ctypes.data_as(_lib._dp))
        _lib.raise_for(rc, eng.lib, eng.h, 'gp_grad_from_parts')
        return out

    def grad_Z(self, dF_dKmm, dKmm_dZ, dF_dexp_K_miY, dexp_K_miY_dZ, dF_dexp_K_mi_K_im, dexp_K_mi_K_im_dZ):
        # partial_terms.py:207-240
        assert np.shape(dKmm_dZ) == (self.M, self.Q, self.M) and np.shape(dexp_K_miY_dZ) == (self.M, self.Q, self.D)
        assert np.shape(dexp_K_mi_K_im_dZ) == (self.M, self.Q, self.M)
        return self._from_parts(0, dF_dKmm, dKmm_dZ, dF_dexp_K_miY, dexp_K_miY_dZ, dF_dexp_K_mi_K_im, dexp_K_mi_K_im_dZ,
                                np.empty((self.M, self.Q)))

    def grad_alpha(self, dF_dKmm, dKmm_dalpha, dF_dexp_K_miY, dexp_K_miY_dalpha, dF_dexp_K_mi_K_im, dexp_K_mi_K_im_dalpha):
        # partial_terms.py:286-299
        assert np.shape(dKmm_dalpha) == (self.Q, self.M, self.M) and np.shape(dexp_K_miY_dalpha) == (self.Q, self.M, self.D)
        assert np.shape(dexp_K_mi_K_im_dalpha) == (self.Q, self.M, self.M)
        return self._from_parts(1, dF_dKmm, dKmm_dalpha, dF_dexp_K_miY, dexp_K_miY_dalpha, dF_dexp_K_mi_K_im, dexp_K_mi_K_im_dalpha,
                                np.empty(self.Q))

    def dKmm_dsf2(self):
        self._ensure_gstep()
        return self._engine().download('KMM') / self._sf2()            # partial_terms.py:306-308

    def dexp_K_miY_dsf2(self):
        return self.exp_K_miY / self._sf2()                            # partial_terms.py:310-312

    def dexp_K_mi_K_im_dsf2(self):
        return 2.0 * self.sum_exp_K_mi_K_im / self._sf2()              # partial_terms.py:314-316

    def dexp_K_ii_dsf2(self):
        return self.local_N                                            # partial_terms.py:318-320 (an int)

    def grad_sf2(self, dF_dKmm, dKmm_dsf2, dF_dexp_K_ii, dexp_K_ii_dsf2, dF_dexp_K_miY, dexp_K_miY_dsf2,
                 dF_dexp_K_mi_K_im, dexp_K_mi_K_im_dsf2):
        # partial_terms.py:322-333: four Frobenius inner products of host arrays the caller already holds
        return (np.sum(np.asarray(dF_dKmm) * np.asarray(dKmm_dsf2)) + dF_dexp_K_ii * dexp_K_ii_dsf2
                + np.sum(np.asarray(dF_dexp_K_miY) * np.asarray(dexp_K_miY_dsf2))
                + np.sum(np.asarray(dF_dexp_K_mi_K_im) * np.asarray(dexp_K_mi_K_im_dsf2)))

    def grad_beta(self):
        self._ensure_gstep()                                           # partial_terms.py:340-360
        return self._engine().scalars()['grad_beta']

    # ------------------------------------------------------------------ per-point gradients
    def _ensure_p2(self):
        assert self._have_data
        self._ensure_gstep()
        if not self._p2_done:
            self._engine().phase2(True)
            self._p2_done = True

    def grad_X_mu(self):
        self._ensure_p2()                                              # partial_terms.py:367-398
        return self._engine().download('GRAD_X_MU')

    def grad_X_S(self):
        self._ensure_p2()                                              # partial_terms.py:400-431
        if np.all(self.X_S == 0):
            raise FloatingPointError('grad_X_S with X_S == 0: 1/S in partial_terms.py:417')
        return self._engine().download('GRAD_X_S')

    # ------------------------------------------------------------------ fast path (no 3-tensors)
    def gradients(self, want_embeddings=False):
        """F and all gradients through the two-phase device path (what parallel_GPLVM.calculate_global_derivatives
        computes from the 12 statistics, without materialising them)."""
        assert self._have_data and self._stats_from == 'local'
        self._ensure_gstep()
        eng = self._engine()
        eng.phase2(want_embeddings)
        self._p2_done = want_embeddings
        return eng.finish()
