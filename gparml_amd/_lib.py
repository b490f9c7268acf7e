"""ctypes binding of libgparml_hip.so (C ABI: include/gparml_hip.h).  Fails loudly when the library is missing."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPARML_LIB: another build of the same library (tools/ab_bench.sh runs variants side by side without touching the in-tree one)
LIB_PATH = os.environ.get('GPARML_LIB') or os.path.join(_HERE, 'libgparml_hip.so')

GP_OK, GP_ERR_BAD_ARG, GP_ERR_NOT_PD, GP_ERR_NON_FINITE, GP_ERR_HIP, GP_ERR_STATE, GP_ERR_UNSUPPORTED, GP_RETRY_JITTER, GP_ERR_RCCL = range(9)
GP_COMM_ID_BYTES = 128

# gp_download selectors (include/gparml_hip.h)
ARR = dict(KMM=0, KMM_INV=1, PSI1=2, PSI2_SUM=3, PSI1TY=4, KMM_PLUS_OP_INV=5, DF_DKMM=6, DF_DPSI1TY=7, DF_DPSI2=8,
           GRAD_X_MU=9, GRAD_X_S=10, SCALARS=11, PSI2_POINTS=12, DKMM_DZ=13, DPSI1TY_DZ=14, DPSI2_DZ=15, DKMM_DALPHA=16,
           DPSI1TY_DALPHA=17, DPSI2_DALPHA=18, X_MU_TRIAL=19, X_S_TRIAL=20, GRAD_LATEST=21)

_dp = ctypes.POINTER(ctypes.c_double)
_vp = ctypes.c_void_p
_i64 = ctypes.c_int64

# every symbol include/gparml_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    'gp_create': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'gp_destroy': (ctypes.c_int, [_vp]),
    'gp_last_error': (ctypes.c_char_p, [_vp]),
    'gp_version': (ctypes.c_char_p, []),
    'gp_csv_shape': (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    'gp_csv_read': (ctypes.c_int, [ctypes.c_char_p, _dp, _i64, _i64, ctypes.c_int]),
    'gp_set_stream': (ctypes.c_int, [_vp, _vp]),
    'gp_upload_shard': (ctypes.c_int, [_vp, _dp, _dp, _dp, ctypes.c_int]),
    'gp_upload_embeddings': (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_int]),
    'gp_set_direction': (ctypes.c_int, [_vp, _dp]),
    'gp_set_globals': (ctypes.c_int, [_vp, _dp, ctypes.c_double, _dp, ctypes.c_double, _i64, ctypes.c_double]),
    'gp_phase1': (ctypes.c_int, [_vp]),
    'gp_stats_buffer': (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_i64)]),
    'gp_stats_packed_buffer': (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_i64)]),
    'gp_stats_pack': (ctypes.c_int, [_vp]),
    'gp_stats_unpack': (ctypes.c_int, [_vp]),
    'gp_scale_stats': (ctypes.c_int, [_vp, ctypes.c_double]),
    'gp_scale_buffer': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double]),
    'gp_debug_force_staging': (ctypes.c_int, [ctypes.c_int]),
    'gp_buffer_combine': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int]),
    'gp_global_step': (ctypes.c_int, [_vp]),
    'gp_global_step_jitter': (ctypes.c_int, [_vp, ctypes.c_int]),
    'gp_global_status': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int)]),
    'gp_phase2': (ctypes.c_int, [_vp, ctypes.c_int]),
    'gp_comm_unique_id': (ctypes.c_int, [_vp]),
    'gp_comm_init': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int]),
    'gp_allreduce': (ctypes.c_int, [_vp, ctypes.c_int]),
    'gp_comm_destroy': (ctypes.c_int, [_vp]),
    'gp_comm_available': (ctypes.c_int, []),
    'gp_comm_info': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(_i64), ctypes.POINTER(_i64), _dp]),
    'gp_grads_buffer': (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_i64)]),
    'gp_finish': (ctypes.c_int, [_vp, _dp, _dp, _dp, _dp, _dp]),
    'gp_download': (ctypes.c_int, [_vp, ctypes.c_int, _dp, _i64]),
    'gp_set_local_statistics': (ctypes.c_int, [_vp, ctypes.c_double, _dp, _dp, ctypes.c_double, ctypes.c_double]),
    'gp_last_timings': (ctypes.c_int, [_vp, _dp]),
    'gp_set_timing': (ctypes.c_int, [_vp, ctypes.c_int]),
    'gp_i8_status': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int), _dp, _dp, _dp, ctypes.POINTER(_i64)]),
    'gp_memory_info': (ctypes.c_int, [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    'gp_grad_from_parts': (ctypes.c_int, [_vp, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    'gp_cg_set_grads': (ctypes.c_int, [_vp]),
    'gp_cg_dots': (ctypes.c_int, [_vp, _dp]),
    'gp_cg_max_d': (ctypes.c_int, [_vp, ctypes.c_double, _dp]),
    'gp_cg_abs': (ctypes.c_int, [_vp, _dp]),
    'gp_cg_update': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double]),
    'gp_debug_gemm': (ctypes.c_int, [ctypes.c_int] * 6 + [ctypes.c_double, _dp, _dp, ctypes.c_double, _dp]),
    'gp_debug_set_option': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    'gp_debug_potrf_inverse': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp]),
    'gp_debug_gemm_bench': (ctypes.c_int, [ctypes.c_int] * 7 + [_dp]),
    'gp_debug_peek': (ctypes.c_int, [_vp, ctypes.c_char_p, _dp, ctypes.c_long]),
    'gp_debug_operands_overlap': (ctypes.c_int, [ctypes.c_long] * 8),
}

_lib = None


class GparmlHipError(RuntimeError):
    pass


class JitterRetry(Exception):
    """A Cholesky factorisation of the global step failed for the first time: repeat it with 1e-7*I on the matrices in
    ``mask`` (bit 0 Kmm, bit 1 Kmm + beta*Psi2), as partial_terms.logmarglik does (partial_terms.py:452-456)."""

    def __init__(self, mask, msg=''):
        Exception.__init__(self, msg)
        self.mask = int(mask)


def load():
    """Load the HIP library (once).  Raises ImportError with build instructions if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('gparml_amd: %s is missing -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          '(hipcc --offload-arch=gfx950); there is no CPU fallback' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get('GPARML_LIB') and not hasattr(lib, name):
            continue                     # an A/B variant built from an older tree (tools/ab.sh): entry points added since are simply absent
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def as_c(a):
    """C-contiguous float64 view/copy + its ctypes pointer."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def raise_for(rc, lib, ctx, what):
    """Map status codes to the exception classes the reference raises (scg_adapted.py:55)."""
    if rc == GP_OK:
        return
    msg = lib.gp_last_error(ctx)
    msg = '%s: %s' % (what, msg.decode() if msg else 'error %d' % rc)
    if rc == GP_ERR_BAD_ARG:
        raise AssertionError(msg)
    if rc == GP_ERR_NOT_PD:
        raise np.linalg.LinAlgError(msg)
    if rc == GP_ERR_NON_FINITE:
        raise FloatingPointError(msg)
    if rc == GP_RETRY_JITTER:
        # a factorisation failed and the caller of this entry point cannot repeat the global step (the evaluators catch JitterRetry
        # before this): the class the optimiser's failure wrapper handles (scg_adapted.py:55)
        raise np.linalg.LinAlgError(msg)
    raise GparmlHipError(msg)
