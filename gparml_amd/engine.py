"""One data shard resident on one GPU: the two-phase bound+gradient evaluation.

    phase1 (statistics_mapper, local_MapReduce.py:183-248)  -> packed statistics  [all-reduce]
    global_step (parallel_GPLVM.py:302-369)                 -> F, dF_d*, grad_beta, Kmm parts
    phase2 (embeddings_mapper, local_MapReduce.py:310-363 + the Z/alpha sums) -> packed gradient sums [all-reduce]
    finish                                                  -> grad_Z, grad_alpha, grad_sf2, grad_beta
"""
import ctypes

import numpy as np

from . import _lib


class ShardEngine(object):
    def __init__(self, N_s, D, M, Q, device=0):
        self.lib = _lib.load()
        self.N_s, self.D, self.M, self.Q, self.device = int(N_s), int(D), int(M), int(Q), int(device)
        h = ctypes.c_void_p()
        rc = self.lib.gp_create(ctypes.byref(h), self.device, self.N_s, self.D, self.M, self.Q)
        _lib.raise_for(rc, self.lib, None, 'gp_create')
        self.h = h
        self._keep = []

    # ---- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, 'h', None):
            self.lib.gp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        _lib.raise_for(rc, self.lib, self.h, what)

    def set_stream(self, stream_ptr):
        self._ck(self.lib.gp_set_stream(self.h, ctypes.c_void_p(stream_ptr)), 'gp_set_stream')

    # ---- data -------------------------------------------------------------------------------------
    def upload_shard(self, Y, X_mu, X_S, xs_is_raw=False):
        Y = np.asarray(Y, dtype=np.float64)
        if Y.ndim == 1:
            Y = Y[:, None]
        assert Y.shape == (self.N_s, self.D), 'Y shape %s != (%d, %d)' % (Y.shape, self.N_s, self.D)
        X_mu = np.asarray(X_mu, dtype=np.float64)
        X_S = np.asarray(X_S, dtype=np.float64)
        assert X_mu.ndim == 2 and X_S.ndim == 2 and X_mu.shape == X_S.shape == (self.N_s, self.Q)   # kernel_exp.py:32-34
        Y, pY = _lib.as_c(Y)
        X_mu, pm = _lib.as_c(X_mu)
        X_S, ps = _lib.as_c(X_S)
        self._ck(self.lib.gp_upload_shard(self.h, pY, pm, ps, 1 if xs_is_raw else 0), 'gp_upload_shard')

    def upload_embeddings(self, X_mu, X_S, xs_is_raw=False):
        X_mu = np.asarray(X_mu, dtype=np.float64)
        X_S = np.asarray(X_S, dtype=np.float64)
        assert X_mu.shape == X_S.shape == (self.N_s, self.Q)
        X_mu, pm = _lib.as_c(X_mu)
        X_S, ps = _lib.as_c(X_S)
        self._ck(self.lib.gp_upload_embeddings(self.h, pm, ps, 1 if xs_is_raw else 0), 'gp_upload_embeddings')

    def set_direction(self, d):
        if d is None:
            self._ck(self.lib.gp_set_direction(self.h, None), 'gp_set_direction')
            return
        d = np.asarray(d, dtype=np.float64)
        assert d.shape == (2, self.N_s, self.Q)
        d, pd = _lib.as_c(d)
        self._ck(self.lib.gp_set_direction(self.h, pd), 'gp_set_direction')

    def set_globals(self, Z, sf2, alpha, beta, N_global=None, step_size=0.0):
        Z = np.asarray(Z, dtype=np.float64)
        assert Z.shape == (self.M, self.Q)
        alpha = np.atleast_1d(np.asarray(alpha, dtype=np.float64).squeeze())
        assert alpha.shape == (self.Q,)
        Z, pZ = _lib.as_c(Z)
        alpha, pa = _lib.as_c(alpha)
        Ng = self.N_s if N_global is None else int(N_global)
        self._ck(self.lib.gp_set_globals(self.h, pZ, float(sf2), pa, float(beta), Ng, float(step_size)), 'gp_set_globals')

    # ---- evaluation ---------------------------------------------------------------------------------
    def phase1(self):
        self._ck(self.lib.gp_phase1(self.h), 'gp_phase1')

    def stats_packed_buffer(self):
        """Psi2 upper triangle | C (M*D) | scalars: the payload of the all-reduce across processes (stats_pack / stats_unpack)."""
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        self._ck(self.lib.gp_stats_packed_buffer(self.h, ctypes.byref(p), ctypes.byref(n)), 'gp_stats_packed_buffer')
        return p.value, n.value

    def stats_pack(self):
        self._ck(self.lib.gp_stats_pack(self.h), 'gp_stats_pack')

    def stats_unpack(self):
        self._ck(self.lib.gp_stats_unpack(self.h), 'gp_stats_unpack')

    def stats_buffer(self):
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        self._ck(self.lib.gp_stats_buffer(self.h, ctypes.byref(p), ctypes.byref(n)), 'gp_stats_buffer')
        return p.value, n.value

    def grads_buffer(self):
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        self._ck(self.lib.gp_grads_buffer(self.h, ctypes.byref(p), ctypes.byref(n)), 'gp_grads_buffer')
        return p.value, n.value

    # ---- the reduce across GPUs inside the library (RCCL resolved with dlopen; include/gparml_hip.h gp_comm_*) -------------------------
    @staticmethod
    def comm_unique_id():
        """128-byte ncclUniqueId (rank 0 calls this and hands the bytes to every other rank)."""
        lib = _lib.load()
        buf = ctypes.create_string_buffer(_lib.GP_COMM_ID_BYTES)
        _lib.raise_for(lib.gp_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), lib, None, 'gp_comm_unique_id')
        return buf.raw

    @staticmethod
    def comm_available():
        """True when RCCL can be resolved in this process (gp_comm_available; nothing collective happens)."""
        return _lib.load().gp_comm_available() == 0

    def comm_info(self, probe=False):
        """What this context's communicator saw: {'ranks', 'rank', 'stats_bytes', 'grads_bytes'[, 'probe_sum']}.  ``probe`` all-reduces one
        1.0 per rank (COLLECTIVE; synchronises the stream): the sum equals the rank count exactly when RCCL connected that many ranks."""
        nr, rk, sb, gb, ps = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double()
        self._ck(self.lib.gp_comm_info(self.h, ctypes.byref(nr), ctypes.byref(rk), ctypes.byref(sb), ctypes.byref(gb),
                                       ctypes.byref(ps) if probe else None), 'gp_comm_info')
        out = {'ranks': nr.value, 'rank': rk.value, 'stats_bytes': sb.value, 'grads_bytes': gb.value}
        if probe:
            out['probe_sum'] = ps.value
        return out

    def comm_init(self, unique_id, nranks, rank):
        assert len(unique_id) == _lib.GP_COMM_ID_BYTES
        buf = ctypes.create_string_buffer(bytes(unique_id), _lib.GP_COMM_ID_BYTES)
        self._ck(self.lib.gp_comm_init(self.h, ctypes.cast(buf, ctypes.c_void_p), int(nranks), int(rank)), 'gp_comm_init')
        self.has_comm = True

    has_comm = False

    def allreduce(self, which='stats'):
        """statistics: pack -> RCCL all-reduce(sum) -> unpack; grads: all-reduce of the gradient sums.  Enqueued on the engine's stream."""
        self._ck(self.lib.gp_allreduce(self.h, 0 if which == 'stats' else 1), 'gp_allreduce')

    def comm_destroy(self):
        self._ck(self.lib.gp_comm_destroy(self.h), 'gp_comm_destroy')
        self.has_comm = False

    def combine(self, src, which='stats', op='add'):
        """dst (self) += src or dst = src for the packed statistics / gradient-sum buffers (same device)."""
        self._ck(self.lib.gp_buffer_combine(self.h, src.h, 0 if which == 'stats' else 1, 0 if op == 'add' else 1), 'gp_buffer_combine')

    # ---- resident CG vectors (scg_adapted_local_MapReduce.py:29-243) ----------------------------------
    CG_RESET_D, CG_UPDATE_D, CG_UPDATE_X, CG_GRAD_OLD, CG_GRAD_NEW, CG_SET_GRADS = range(6)

    def cg_set_grads(self):
        self._ck(self.lib.gp_cg_set_grads(self.h), 'gp_cg_set_grads')

    def cg_dots(self):
        """local sums [mu, kappa, theta, |g_new|^2, g_new.g_old, max|d|]"""
        out = np.zeros(6)
        self._ck(self.lib.gp_cg_dots(self.h, out.ctypes.data_as(_lib._dp)), 'gp_cg_dots')
        return out

    def cg_abs(self):
        """local [sum |grad_now|, max |grad_now|] (gd_local_MapReduce.py:38-61)"""
        out = np.zeros(2)
        self._ck(self.lib.gp_cg_abs(self.h, out.ctypes.data_as(_lib._dp)), 'gp_cg_abs')
        return out

    def cg_update(self, which, a=0.0):
        self._ck(self.lib.gp_cg_update(self.h, int(which), float(a)), 'gp_cg_update')

    def scale_stats(self, factor):
        self._ck(self.lib.gp_scale_stats(self.h, float(factor)), 'gp_scale_stats')

    def scale_buffer(self, which, factor):
        """Scale the packed statistics ('stats') or gradient-sum ('grads') buffer: node drop-out, local_MapReduce.py:119-129, 263-264."""
        self._ck(self.lib.gp_scale_buffer(self.h, 0 if which == 'stats' else 1, float(factor)), 'gp_scale_buffer')

    def global_step(self, sync=True, jitter=0):
        """Enqueue the global step.  ``sync=True`` (the class / MapReduce surfaces) also waits for its outcome, repeats it once with the
        reference's 1e-7 jitter when a factorisation failed (partial_terms.py:452-456) and raises LinAlgError if that fails too;
        ``sync=False`` (the evaluators) defers all of that to finish(), the evaluation's single host synchronisation."""
        self._ck(self.lib.gp_global_step_jitter(self.h, int(jitter)), 'gp_global_step')
        self._jitter_used = int(jitter)         # the mask this step ends up with (last_jitter reports it)
        # An evaluator whose previous evaluation needed the jitter (finish() saw the failed factorisation only after phase 2 had run on its
        # garbage: the whole phase 2 twice per evaluation, +62 % at N = 1e5, M = 512, Q = 5 with free embeddings) checks the outcome HERE the
        # next time: still the reference's order -- first without jitter, then with (partial_terms.py:452-456) -- for one extra host wait
        # instead of a wasted phase 2.  The hint is dropped as soon as a plain factorisation succeeds again.
        if sync or (jitter == 0 and self._jitter_hint):
            used = 0
            for _ in range(2):                  # at most one retry per matrix: the mask only grows (Kmm, then Kmm + beta Psi2)
                try:
                    self.global_status()
                    self._jitter_hint = used
                    return
                except _lib.JitterRetry as r:
                    used = self._jitter_used = r.mask
                    self._ck(self.lib.gp_global_step_jitter(self.h, r.mask), 'gp_global_step')
            self.global_status()                # a third failure is GP_ERR_NOT_PD -> LinAlgError
            self._jitter_hint = used

    def global_status(self):
        mask = ctypes.c_int(0)
        rc = self.lib.gp_global_status(self.h, ctypes.byref(mask))
        if rc == _lib.GP_RETRY_JITTER:
            raise _lib.JitterRetry(mask.value, 'gp_global_status: retry with jitter mask %d' % mask.value)
        self._ck(rc, 'gp_global_status')

    def phase2(self, want_embedding_grads=False):
        self._ck(self.lib.gp_phase2(self.h, 1 if want_embedding_grads else 0), 'gp_phase2')

    def finish(self):
        F = ctypes.c_double()
        gs = ctypes.c_double()
        gb = ctypes.c_double()
        gZ = np.empty((self.M, self.Q))
        ga = np.empty(self.Q)
        rc = self.lib.gp_finish(self.h, ctypes.byref(F), gZ.ctypes.data_as(_lib._dp), ctypes.byref(gs),
                                ga.ctypes.data_as(_lib._dp), ctypes.byref(gb))
        if rc == _lib.GP_RETRY_JITTER:
            try:
                self.global_status()    # raises JitterRetry carrying the mask
            except _lib.JitterRetry as r:
                self._jitter_hint = r.mask
                raise
        self._ck(rc, 'gp_finish')
        return dict(F=F.value, grad_Z=gZ, grad_sf2=gs.value, grad_alpha=ga, grad_beta=gb.value)

    def evaluate(self, want_embedding_grads=False):
        """Single-shard evaluation (no reduction across shards); one host synchronisation, in finish()."""
        self.phase1()
        jitter = 0
        while True:                             # the retry mask only grows (bit 0 Kmm, bit 1 Kmm + beta Psi2): at most two repeats
            self.global_step(sync=False, jitter=jitter)
            self.phase2(want_embedding_grads)
            try:
                out = self.finish()
                break
            except _lib.JitterRetry as r:
                jitter = r.mask
        # 0, or the mask of the matrices that needed the reference's 1e-7 jitter in this evaluation (found by finish(), or -- after an evaluation that needed
        # it -- already inside global_step())
        self.last_jitter = jitter | self._jitter_used
        if want_embedding_grads:
            out['grad_X_mu'] = self.download('GRAD_X_MU')
            if not self.regime_A_hint:
                out['grad_X_S'] = self.download('GRAD_X_S')
        return out

    regime_A_hint = False
    _jitter_used = 0
    _jitter_hint = 0          # the jitter mask the previous evaluation ended up with (global_step)

    def set_local_statistics(self, sum_YYT, Psi2, C, sum_exp_K_ii, KL):
        Psi2, p2 = _lib.as_c(np.asarray(Psi2, dtype=np.float64).reshape(self.M, self.M))
        C, pc = _lib.as_c(np.asarray(C, dtype=np.float64).reshape(self.M, self.D))
        self._ck(self.lib.gp_set_local_statistics(self.h, float(sum_YYT), p2, pc, float(sum_exp_K_ii), float(KL)),
                 'gp_set_local_statistics')

    def i8_status(self):
        """The int8 phase-1 guard (gp_i8_status): {'state': -1 n/a | 0 unchecked | 1 accepted | 2 rejected, 'rel_psi2', 'rel_c', 'cond_lower_bound', 'checks'}."""
        st, ck = ctypes.c_int(), ctypes.c_int64()
        r2, rc, cl = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self._ck(self.lib.gp_i8_status(self.h, ctypes.byref(st), ctypes.byref(r2), ctypes.byref(rc), ctypes.byref(cl), ctypes.byref(ck)), 'gp_i8_status')
        return {'state': st.value, 'rel_psi2': r2.value, 'rel_c': rc.value, 'cond_lower_bound': cl.value, 'checks': ck.value}

    def set_timing(self, level):
        """HIP timing events per evaluation: 2 = every stage and dominant kernel (default), 1 = first and last only (total_ms), 0 = none.
        Each event is ~4-7 us of idle stream; an optimiser on a small problem (BASELINE configs[1]) switches them off."""
        if hasattr(self.lib, "gp_set_timing"):
            self._ck(self.lib.gp_set_timing(self.h, int(level)), "gp_set_timing")

    def timings(self):
        t = np.zeros(8)
        self._ck(self.lib.gp_last_timings(self.h, t.ctypes.data_as(_lib._dp)), 'gp_last_timings')
        return dict(generate_ms=t[0], phase1_ms=t[1], global_ms=t[2], phase2_ms=t[3], total_ms=t[4],
                    psi1_ms=t[5], p1_kernel_ms=t[6], p2_kernel_ms=t[7])

    def memory_info(self):
        """(free, total) bytes of the engine's device right now (hipMemGetInfo)."""
        f, t = ctypes.c_int64(), ctypes.c_int64()
        self._ck(self.lib.gp_memory_info(self.h, ctypes.byref(f), ctypes.byref(t)), 'gp_memory_info')
        return f.value, t.value

    # ---- results ------------------------------------------------------------------------------------
    _SHAPES = {
        'KMM': lambda s: (s.M, s.M), 'KMM_INV': lambda s: (s.M, s.M), 'PSI1': lambda s: (s.N_s, s.M),
        'PSI2_SUM': lambda s: (s.M, s.M), 'PSI1TY': lambda s: (s.M, s.D), 'KMM_PLUS_OP_INV': lambda s: (s.M, s.M),
        'DF_DKMM': lambda s: (s.M, s.M), 'DF_DPSI1TY': lambda s: (s.M, s.D), 'DF_DPSI2': lambda s: (s.M, s.M),
        'GRAD_X_MU': lambda s: (s.N_s, s.Q), 'GRAD_X_S': lambda s: (s.N_s, s.Q), 'SCALARS': lambda s: (8,),
        'PSI2_POINTS': lambda s: (s.N_s, s.M, s.M), 'DKMM_DZ': lambda s: (s.M, s.Q, s.M),
        'DPSI1TY_DZ': lambda s: (s.M, s.Q, s.D), 'DPSI2_DZ': lambda s: (s.M, s.Q, s.M),
        'DKMM_DALPHA': lambda s: (s.Q, s.M, s.M), 'DPSI1TY_DALPHA': lambda s: (s.Q, s.M, s.D),
        'DPSI2_DALPHA': lambda s: (s.Q, s.M, s.M), 'X_MU_TRIAL': lambda s: (s.N_s, s.Q), 'X_S_TRIAL': lambda s: (s.N_s, s.Q),
        'GRAD_LATEST': lambda s: (2, s.N_s, s.Q),
    }

    def download(self, name):
        shape = self._SHAPES[name](self)
        out = np.empty(shape, dtype=np.float64)
        self._ck(self.lib.gp_download(self.h, _lib.ARR[name], out.ctypes.data_as(_lib._dp), out.size), 'gp_download(%s)' % name)
        return out

    def scalars(self):
        s = self.download('SCALARS')
        return dict(sum_YYT=s[0], sum_exp_K_ii=s[1], KL=s[2], logdet_Kmm=s[3], logdet_A=s[4], F=s[5], grad_beta=s[6], grad_sf2=s[7])
