"""Developer check of the FP64 GEMM core through the C ABI (run on the GPU box)."""
import ctypes, os, sys, time
import numpy as np
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gparml_amd', 'libgparml_hip.so'))
dp = ctypes.POINTER(ctypes.c_double)
lib.gp_debug_gemm.argtypes = [ctypes.c_int] * 6 + [ctypes.c_double, dp, dp, ctypes.c_double, dp]
lib.gp_debug_gemm_bench.argtypes = [ctypes.c_int] * 7 + [dp]
lib.gp_last_error.restype = ctypes.c_char_p
rs = np.random.RandomState(0)
ok = True
for (m, n, k) in [(5, 7, 3), (128, 128, 16), (130, 250, 37), (300, 64, 200)]:
    for ta in (0, 1):
        for tb in (0, 1):
            A = rs.randn(k, m) if ta else rs.randn(m, k)
            B = rs.randn(n, k) if tb else rs.randn(k, n)
            C0 = rs.randn(m, n)
            C = C0.copy()
            rc = lib.gp_debug_gemm(0, ta, tb, m, n, k, 1.5, A.ctypes.data_as(dp), B.ctypes.data_as(dp), -0.5, C.ctypes.data_as(dp))
            ref = 1.5 * (A.T if ta else A).dot(B.T if tb else B) - 0.5 * C0
            err = np.max(np.abs(C - ref)) / np.max(np.abs(ref))
            good = rc == 0 and err < 1e-13
            ok &= good
            print('gemm m=%d n=%d k=%d ta=%d tb=%d rc=%d relerr=%.2e %s' % (m, n, k, ta, tb, rc, err, 'OK' if good else 'FAIL ' + lib.gp_last_error(None).decode()))
ms = ctypes.c_double()
for (ta, tb, m, n, k) in [(0, 0, 4096, 4096, 4096), (1, 0, 4096, 4096, 4096), (0, 1, 4096, 4096, 4096), (1, 0, 512, 640, 262144), (0, 0, 65536, 512, 640), (0, 0, 512, 512, 512)]:
    rc = lib.gp_debug_gemm_bench(0, ta, tb, m, n, k, 5, ctypes.byref(ms))
    print('bench ta=%d tb=%d m=%d n=%d k=%d: rc=%d %.3f ms  %.1f TFLOP/s' % (ta, tb, m, n, k, rc, ms.value, 2.0 * m * n * k / ms.value / 1e9))
sys.exit(0 if ok else 1)
