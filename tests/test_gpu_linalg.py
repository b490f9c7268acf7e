"""The library's dense FP64 building blocks against numpy (run with -m gpu): the MFMA GEMM core in all four operand layouts -- the
32x32-tile kernel the global step uses for small grids and the 128x128-tile kernel for large ones -- and the blocked Cholesky + inverse."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gemm(ta, tb, m, n, k, alpha, A, B, beta, C):
    from gparml_amd import _lib
    lib = _lib.load()
    A, pa = _lib.as_c(A)
    B, pb = _lib.as_c(B)
    C = np.ascontiguousarray(C, dtype=np.float64).copy()
    rc = lib.gp_debug_gemm(0, ta, tb, m, n, k, alpha, pa, pb, beta, C.ctypes.data_as(_lib._dp))
    _lib.raise_for(rc, lib, None, 'gp_debug_gemm')
    return C


@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('m,n,k', [(100, 70, 50), (512, 512, 512), (300, 129, 1000), (3000, 3000, 40)])
def test_gemm_layouts(ta, tb, m, n, k):
    """(3000, 3000): 24 x 24 = 576 tiles of 128 -> the 128x128-tile kernel; the others (<= 256 such tiles) -> the 32x32-tile kernel."""
    rs = np.random.RandomState(m + n + k + 2 * ta + tb)
    A = rs.randn(k, m) if ta else rs.randn(m, k)
    B = rs.randn(n, k) if tb else rs.randn(k, n)
    C0 = rs.randn(m, n)
    ref = 1.5 * (A.T if ta else A).dot(B.T if tb else B) - 0.5 * C0
    out = _gemm(ta, tb, m, n, k, 1.5, A, B, -0.5, C0)
    assert np.max(np.abs(out - ref)) <= 1e-12 * np.max(np.abs(ref)) * np.sqrt(k)


@pytest.mark.parametrize('n', [1, 100, 128, 300, 640, 700, 900, 1100, 1400])     # 1, 1, 1, 3, 5, 6, 8, 9, 11 panels of 128: the inverse factor by halves meets truncated pairs at every level
def test_cholesky_and_inverse(n):
    from gparml_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(n)
    X = rs.randn(n, n + 5)
    A = X.dot(X.T) / n + 0.1 * np.eye(n)
    A_c, pa = _lib.as_c(A)
    L, Ainv = np.zeros((n, n)), np.zeros((n, n))
    logdet = ctypes.c_double()
    rc = lib.gp_debug_potrf_inverse(0, n, pa, L.ctypes.data_as(_lib._dp), Ainv.ctypes.data_as(_lib._dp), ctypes.byref(logdet))
    _lib.raise_for(rc, lib, None, 'gp_debug_potrf_inverse')
    Lr = np.linalg.cholesky(A)
    assert np.max(np.abs(L - Lr)) <= 1e-11 * np.max(np.abs(Lr))
    assert np.max(np.abs(Ainv - np.linalg.inv(A))) <= 1e-9 * np.max(np.abs(Ainv))
    assert abs(logdet.value - np.linalg.slogdet(A)[1]) <= 1e-10 * max(1.0, abs(logdet.value))
    A[0, 0] = -1.0
    A_c, pa = _lib.as_c(A)
    assert lib.gp_debug_potrf_inverse(0, n, pa, None, None, None) == _lib.GP_ERR_NOT_PD
