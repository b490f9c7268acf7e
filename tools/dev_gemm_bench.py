import ctypes, os, sys
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gparml_amd', os.environ.get('GP_LIB', 'libgparml_hip.so')))
dp = ctypes.POINTER(ctypes.c_double)
lib.gp_debug_gemm_bench.argtypes = [ctypes.c_int] * 7 + [dp]
ms = ctypes.c_double()
shapes = [(0, 0, 4096, 4096, 4096), (1, 0, 4096, 4096, 4096), (0, 0, 8192, 8192, 2048), (0, 0, 65536, 512, 640), (0, 0, 512, 512, 512)]
for (ta, tb, m, n, k) in shapes:
    rc = lib.gp_debug_gemm_bench(0, ta, tb, m, n, k, 5, ctypes.byref(ms))
    print('fill=%s bench ta=%d tb=%d m=%d n=%d k=%d: rc=%d %.3f ms  %.1f TFLOP/s' % (os.environ.get('GP_BENCH_FILL', '0'), ta, tb, m, n, k, rc, ms.value, 2.0 * m * n * k / ms.value / 1e9))
