// Shared host-side declarations of the gparml HIP library (context, error handling, launch helpers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <string>
#include <vector>
#include "../../include/gparml_hip.h"
#include "mma_f64.h"

namespace gp {

inline long round_up(long x, long m) { return (x + m - 1) / m * m; }

// dense batched GEMM on tile-aligned buffers: C = alpha * op(A) op(B) + beta * C
struct GemmP {
  const double* A;
  const double* B;
  double* C;
  long lda, ldb, ldc;
  long sA, sB, sC;  // batch strides (doubles)
  int K;            // multiple of KC
  double alpha, beta;
  int tri;          // 0 all tiles, 1 only tiles with row-tile >= col-tile (lower), 2 only upper
};
// m, n multiples of TILE; la/lb: Layout of A (free index = rows of C) and B (free index = cols of C)
void launch_gemm(hipStream_t st, Layout la, Layout lb, int m, int n, int batch, const GemmP& p);

}  // namespace gp

struct gp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // sizes
  int64_t N = 0;      // local shard rows
  int64_t Np = 0;     // padded to TILE
  int D = 0, M = 0, Q = 0;
  int Mp = 0, Dp = 0, LDK = 0;
  int64_t N_global = 0;
  double sf2 = 1, beta = 1, step = 0;
  bool regime_A = true;
  bool xs_raw = false;
  int state = 0;      // 0 created, 1 data, 2 globals, 3 phase1, 4 global step, 5 phase2
};

namespace gp {
extern thread_local std::string g_create_error;
int fail(gp_ctx* ctx, int code, const char* fmt, ...);
}  // namespace gp

#define GP_HIP(ctx, call)                                                                         \
  do {                                                                                            \
    hipError_t e__ = (call);                                                                      \
    if (e__ != hipSuccess) return gp::fail(ctx, GP_ERR_HIP, "%s failed: %s (%s:%d)", #call,      \
                                            hipGetErrorString(e__), __FILE__, __LINE__);           \
  } while (0)
