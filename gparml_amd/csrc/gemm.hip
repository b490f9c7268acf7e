// Generic FP64 GEMM on the 4x4x4 matrix-core instruction (see mma_f64.h): 128x128 tile per 256-thread
// workgroup, 16-deep k-chunks double-buffered through LDS, two workgroups per CU.
// Used by the global step (M x M algebra) and, through the same building blocks, by the phase kernels.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "gp_common.h"
#include "gemm32.h"
#include <algorithm>
#include <cstdlib>

namespace gp {

template <Layout LA, Layout LB>
__global__ void __launch_bounds__(256, 2) gemm128_kernel(GemmP p) {
  const int bx = blockIdx.x, by = blockIdx.y;
  const int bz = blockIdx.z / p.splits, sp = blockIdx.z % p.splits;
  if (p.tri == 1 && bx > by) return;
  if (p.tri == 2 && bx < by) return;
  __shared__ __attribute__((aligned(16))) double lds[2][2][TILE_LDS_DOUBLES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow0 = (wave >> 1) * WT, wcol0 = (wave & 1) * WT;
  const int bi = bz % p.inner, bo = bz / p.inner;
  const double* A = p.A + (long)bi * p.sA + (long)bo * p.oA;
  const double* B = p.B + (long)bi * p.sB + (long)bo * p.oB;
  double* C = p.C + (long)bi * p.sC + (long)bo * p.oC;
  const long row0 = (long)by * TILE, col0 = (long)bx * TILE;
  // klow (X^T X, X lower-triangular): the tile's k-range starts at its first possibly non-zero k, a multiple of 128; the splits share what is left
  const long klo = p.klow ? (row0 > col0 ? row0 : col0) : 0;
  const int nc = (int)((p.K - klo) / KC / p.splits);          // chunks of this split
  const long k0 = klo + (long)sp * nc * KC;
  const double* Ab = (LA == K_CONTIG) ? A + row0 * p.lda + k0 : A + row0 + k0 * p.lda;
  const double* Bb = (LB == K_CONTIG) ? B + col0 * p.ldb + k0 : B + col0 + k0 * p.ldb;
  const long a_step = (LA == K_CONTIG) ? KC : (long)KC * p.lda;
  const long b_step = (LB == K_CONTIG) ? KC : (long)KC * p.ldb;

  Acc acc;
  acc.zero();
  const LaneOfs ofs = lane_offsets<LA, LB>(wrow0, wcol0, lane);
  tile_dma<LA>(lds[0][0], Ab, p.lda, wave, lane);
  tile_dma<LB>(lds[0][1], Bb, p.ldb, wave, lane);
  dma_wait();
  __syncthreads();
  for (int c = 0; c < nc; ++c) {
    const int cur = c & 1;
    if (c + 1 < nc) {
      tile_dma<LA>(lds[cur ^ 1][0], Ab + (long)(c + 1) * a_step, p.lda, wave, lane);
      tile_dma<LB>(lds[cur ^ 1][1], Bb + (long)(c + 1) * b_step, p.ldb, wave, lane);
    }
    mma_chunk<LA, LB>(lds[cur][0], lds[cur][1], acc, ofs);
    dma_wait();
    __syncthreads();
  }
  acc.drain();
  if (p.splits > 1) {
    // raw partial tile, [z][by][bx][split][128*128]
    double* w = p.ws + ((((long)bz * gridDim.y + by) * gridDim.x + bx) * p.splits + sp) * (TILE * TILE);
#pragma unroll
    for (int ar = 0; ar < 4; ++ar)
#pragma unroll
      for (int bc = 0; bc < 16; ++bc) w[(wrow0 + acc_row(ar, lane)) * TILE + wcol0 + acc_col(bc, lane)] = acc.v[ar][bc];
    return;
  }
  if (p.beta != 0.0) {
#pragma unroll
    for (int ar = 0; ar < 4; ++ar)
#pragma unroll
      for (int bc = 0; bc < 16; ++bc) {
        const long r = row0 + wrow0 + acc_row(ar, lane), cc = col0 + wcol0 + acc_col(bc, lane);
        C[r * p.ldc + cc] = p.alpha * acc.v[ar][bc] + p.beta * C[r * p.ldc + cc];
      }
  } else {
#pragma unroll
    for (int ar = 0; ar < 4; ++ar)
#pragma unroll
      for (int bc = 0; bc < 16; ++bc) {
        const long r = row0 + wrow0 + acc_row(ar, lane), cc = col0 + wcol0 + acc_col(bc, lane);
        C[r * p.ldc + cc] = p.alpha * acc.v[ar][bc];
        if (p.mirror && bx != by) C[cc * p.ldc + r] = p.alpha * acc.v[ar][bc];
      }
  }
}

// Small-tile variant for the latency-bound products of the global step: see gemm32.h (the tile product is a device function shared with the
// fused small-M tail of the global step).
template <Layout LA, Layout LB>
__global__ void __launch_bounds__(256, 1) gemm32_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) double sA[SmallImg<LA>::DOUBLES];
  __shared__ __attribute__((aligned(16))) double sB[SmallImg<LB>::DOUBLES];
  gemm32_tile<LA, LB>(p, blockIdx.x, blockIdx.y, blockIdx.z, sA, sB);
}

__global__ void __launch_bounds__(256) gemm_splitk_reduce_kernel(GemmP p, int tiles_x, int tiles_y) {
  // grid (tiles * 16, batch): 16 blocks of 256 threads per 128x128 tile, one 32 x 32 sub-block each (four elements per thread, rows of 32 doubles);
  // the mirrored copy of an off-diagonal tile (GemmP::mirror) goes through an LDS transposition so that it, too, is written in 256-byte runs
  // (as scattered 8-byte stores the reduce of X^T X took 29 us at M = 1024 against 15 us for the other products)
  __shared__ double tr[32][33];
  const int tile = blockIdx.x >> 4, sub = blockIdx.x & 15;
  const int bx = tile % tiles_x, by = tile / tiles_x, bz = blockIdx.y;
  if (p.tri == 1 && bx > by) return;
  if (p.tri == 2 && bx < by) return;
  const int bi = bz % p.inner, bo = bz / p.inner;
  double* C = p.C + (long)bi * p.sC + (long)bo * p.oC;
  const double* w = p.ws + (((long)bz * tiles_y + by) * tiles_x + bx) * p.splits * (TILE * TILE);
  const int sr = 32 * (sub >> 2), sc = 32 * (sub & 3), lc = threadIdx.x & 31, lr0 = threadIdx.x >> 5;
  const bool mir = p.mirror && bx != by;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lr = 8 * i + lr0;
    const int e = (sr + lr) * TILE + sc + lc;
    double s = 0.0;
    for (int sp = 0; sp < p.splits; ++sp) s += w[(long)sp * (TILE * TILE) + e];
    const long r = (long)by * TILE + sr + lr, cc = (long)bx * TILE + sc + lc;
    const double v = p.alpha * s + (p.beta != 0.0 ? p.beta * C[r * p.ldc + cc] : 0.0);
    C[r * p.ldc + cc] = v;
    if (mir) tr[lr][lc] = v;
  }
  if (mir) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lr = 8 * i + lr0;      // row of the transposed sub-block = column of the original
      C[((long)bx * TILE + sc + lr) * p.ldc + (long)by * TILE + sr + lc] = tr[lc][lr];
    }
  }
}

// Does the stored rectangle of an operand (rows x cols doubles, leading dimension ld, first element at element address x0) share an element with C (m x n, ldc, at c0)?
// Same leading dimension = two windows of one parent matrix: row and column ranges from the address difference; otherwise the address ranges decide (conservative).
bool windows_meet(long x0, long rows, long cols, long ld, long c0, long m, long n, long ldc) {
  const long x1 = x0 + (rows - 1) * ld + cols, c1 = c0 + (m - 1) * ldc + n;
  if (x1 <= c0 || c1 <= x0) return false;
  if (ld != ldc) return true;
  const long d = c0 - x0, dr = (d >= 0 ? d : d - (ld - 1)) / ld, dc = d - dr * ld;   // C(0,0) sits at row dr, column dc of X's frame (floor division)
  auto hit = [&](long r0, long k0) { return r0 < rows && r0 + m > 0 && k0 < cols && k0 + n > 0; };
  return hit(dr, dc) || hit(dr + 1, dc - ld);                                        // the same place one row further down, ld columns to the left
}
static bool operand_meets_c(const double* X, long rows, long cols, long ld, const double* C, long m, long n, long ldc) {
  return windows_meet((long)(reinterpret_cast<uintptr_t>(X) / sizeof(double)), rows, cols, ld, (long)(reinterpret_cast<uintptr_t>(C) / sizeof(double)), m, n, ldc);
}

void launch_gemm(hipStream_t st, Layout la, Layout lb, int m, int n, int batch, const GemmP& p) {
  // No product here may write a tile another workgroup still reads: C must not share an element with A or B (the in-place panel solve of the blocked
  // Cholesky did until r06 -- a race visible only on a cold start, profiles/r06_first_evaluation_race.txt).  A programming error, never a user's: abort.
  if (operand_meets_c(p.A, la == K_CONTIG ? m : p.K, la == K_CONTIG ? p.K : m, p.lda, p.C, m, n, p.ldc) ||
      operand_meets_c(p.B, lb == K_CONTIG ? n : p.K, lb == K_CONTIG ? p.K : n, p.ldb, p.C, m, n, p.ldc)) {
    fprintf(stderr, "gparml: launch_gemm called with C overlapping an operand (m %d n %d k %d)\n", m, n, p.K);
    abort();
  }
  dim3 grid(n / TILE, m / TILE, batch * p.splits), block(256);
  if (!p.big && (long)(n / TILE) * (m / TILE) * batch <= 256) {
    // few tiles (the global step): 32 x 32 tiles spread the product over the chip; split-k is not needed there
    dim3 g32(n / ST, m / ST, batch);
    if (la == K_CONTIG && lb == FREE_CONTIG) hipLaunchKernelGGL((gemm32_kernel<K_CONTIG, FREE_CONTIG>), g32, block, 0, st, p);
    else if (la == K_CONTIG && lb == K_CONTIG) hipLaunchKernelGGL((gemm32_kernel<K_CONTIG, K_CONTIG>), g32, block, 0, st, p);
    else if (la == FREE_CONTIG && lb == FREE_CONTIG) hipLaunchKernelGGL((gemm32_kernel<FREE_CONTIG, FREE_CONTIG>), g32, block, 0, st, p);
    else hipLaunchKernelGGL((gemm32_kernel<FREE_CONTIG, K_CONTIG>), g32, block, 0, st, p);
    return;
  }
  if (la == K_CONTIG && lb == FREE_CONTIG) hipLaunchKernelGGL((gemm128_kernel<K_CONTIG, FREE_CONTIG>), grid, block, 0, st, p);
  else if (la == K_CONTIG && lb == K_CONTIG) hipLaunchKernelGGL((gemm128_kernel<K_CONTIG, K_CONTIG>), grid, block, 0, st, p);
  else if (la == FREE_CONTIG && lb == FREE_CONTIG) hipLaunchKernelGGL((gemm128_kernel<FREE_CONTIG, FREE_CONTIG>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((gemm128_kernel<FREE_CONTIG, K_CONTIG>), grid, block, 0, st, p);
  if (p.splits > 1)
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((n / TILE) * (m / TILE) * 16, batch), block, 0, st, p, n / TILE, m / TILE);
}

}  // namespace gp

// ---- test hook -------------------------------------------------------------------------------------------------
extern "C" int gp_debug_operands_overlap(long x_off, long rows, long cols, long ld, long c_off, long m, long n, long ldc) {
  return gp::windows_meet(x_off, rows, cols, ld, c_off, m, n, ldc) ? 1 : 0;
}

extern "C" int gp_debug_gemm(int device, int ta, int tb, int m, int n, int k, double alpha, const double* A, const double* B,
                             double beta, double* C) {
  using namespace gp;
  gp_ctx* ctx = nullptr;
  if (m <= 0 || n <= 0 || k <= 0 || !A || !B || !C) return fail(ctx, GP_ERR_BAD_ARG, "gp_debug_gemm: bad argument");
  GP_HIP(ctx, hipSetDevice(device));
  const long mp = round_up(m, TILE), np = round_up(n, TILE), kp = round_up(k, KC);
  // A stored (m,k) row-major [K_CONTIG] or (k,m) [FREE_CONTIG]; B stored (k,n) [FREE_CONTIG] or (n,k) [K_CONTIG]
  const long a_rows = ta ? kp : mp, a_cols = ta ? mp : kp, b_rows = tb ? np : kp, b_cols = tb ? kp : np;
  std::vector<double> hA(a_rows * a_cols, 0.0), hB(b_rows * b_cols, 0.0), hC(mp * np, 0.0);
  const long ar = ta ? k : m, ac = ta ? m : k, br = tb ? n : k, bc = tb ? k : n;
  for (long i = 0; i < ar; ++i) for (long j = 0; j < ac; ++j) hA[i * a_cols + j] = A[i * ac + j];
  for (long i = 0; i < br; ++i) for (long j = 0; j < bc; ++j) hB[i * b_cols + j] = B[i * bc + j];
  for (long i = 0; i < m; ++i) for (long j = 0; j < n; ++j) hC[i * np + j] = C[i * n + j];
  double *dA, *dB, *dC;
  GP_HIP(ctx, hipMalloc(&dA, hA.size() * 8));
  GP_HIP(ctx, hipMalloc(&dB, hB.size() * 8));
  GP_HIP(ctx, hipMalloc(&dC, hC.size() * 8));
  GP_HIP(ctx, hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
  GP_HIP(ctx, hipMemcpy(dB, hB.data(), hB.size() * 8, hipMemcpyHostToDevice));
  GP_HIP(ctx, hipMemcpy(dC, hC.data(), hC.size() * 8, hipMemcpyHostToDevice));
  GemmP p{dA, dB, dC, a_cols, b_cols, np, 0, 0, 0, (int)kp, alpha, beta, 0};
  launch_gemm(nullptr, ta ? FREE_CONTIG : K_CONTIG, tb ? K_CONTIG : FREE_CONTIG, (int)mp, (int)np, 1, p);
  GP_HIP(ctx, hipGetLastError());
  GP_HIP(ctx, hipDeviceSynchronize());
  GP_HIP(ctx, hipMemcpy(hC.data(), dC, hC.size() * 8, hipMemcpyDeviceToHost));
  for (long i = 0; i < m; ++i) for (long j = 0; j < n; ++j) C[i * n + j] = hC[i * np + j];
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
  return GP_OK;
}

// device-resident timing of the GEMM core (tools/ only; not part of the public header)
extern "C" int gp_debug_gemm_bench(int device, int ta, int tb, int m, int n, int k, int iters, double* ms_out) {
  const char* fill_env = getenv("GP_BENCH_FILL");
  const int fill = fill_env ? atoi(fill_env) : 0;  // 0 random, 1 zeros, 2 constant
  using namespace gp;
  gp_ctx* ctx = nullptr;
  GP_HIP(ctx, hipSetDevice(device));
  const long mp = round_up(m, TILE), np = round_up(n, TILE), kp = round_up(k, KC);
  const long a_cols = ta ? mp : kp, b_cols = tb ? kp : np;
  const size_t na = (size_t)mp * kp, nb = (size_t)np * kp, ncc = (size_t)mp * np;
  std::vector<double> h(std::max(na, nb));
  unsigned long long s = 88172645463325252ULL;
  for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0; if (fill == 1) x = 0.0; if (fill == 2) x = 1.25; }
  double *dA, *dB, *dC;
  GP_HIP(ctx, hipMalloc(&dA, na * 8)); GP_HIP(ctx, hipMalloc(&dB, nb * 8)); GP_HIP(ctx, hipMalloc(&dC, ncc * 8));
  GP_HIP(ctx, hipMemcpy(dA, h.data(), na * 8, hipMemcpyHostToDevice));
  GP_HIP(ctx, hipMemcpy(dB, h.data(), nb * 8, hipMemcpyHostToDevice));
  GemmP p{dA, dB, dC, a_cols, b_cols, np, 0, 0, 0, (int)kp, 1.0, 0.0, 0};
  hipEvent_t e0, e1;
  GP_HIP(ctx, hipEventCreate(&e0)); GP_HIP(ctx, hipEventCreate(&e1));
  launch_gemm(nullptr, ta ? FREE_CONTIG : K_CONTIG, tb ? K_CONTIG : FREE_CONTIG, (int)mp, (int)np, 1, p);
  GP_HIP(ctx, hipDeviceSynchronize());
  GP_HIP(ctx, hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch_gemm(nullptr, ta ? FREE_CONTIG : K_CONTIG, tb ? K_CONTIG : FREE_CONTIG, (int)mp, (int)np, 1, p);
  GP_HIP(ctx, hipEventRecord(e1));
  GP_HIP(ctx, hipEventSynchronize(e1));
  float ms;
  GP_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
  return GP_OK;
}
