// Global step: the replicated M x M algebra on the all-reduced statistics.
//   Kmm build (kernels.py:72-113), blocked Cholesky of Kmm and A = Kmm + beta*Psi2 with LDS-resident 128x128
//   diagonal panels, triangular inverse by recursive doubling, explicit inverses (partial_terms.py:60, 95),
//   the bound (partial_terms.py:436-473), its partials (partial_terms.py:102-138), grad_beta (:340-360) and the
//   Kmm-dependent parts of grad_Z / grad_alpha / grad_sf2 (:146-160, 207-240, 247-254, 286-333).
// All matrices are padded to multiples of 128 with an identity block, which leaves log-determinants, inverses
// and every trace unchanged.
#include "gp_common.h"
#include "potrf128.h"
#include "gemm32.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace gp {

// switches of the global step's extended-precision pieces (default on; environment at load time, gp_debug_set_option at run time: bench.py
// prices them by timing the global step with and without)
static bool env_on(const char* name) { const char* e = getenv(name); return !(e && e[0] == '0'); }
std::atomic<int> g_opt_dd_kipsi2{env_on("GPARML_DD_KIPSI2") ? 1 : 0};
std::atomic<int> g_opt_refine_E{env_on("GPARML_REFINE_E") ? 1 : 0};
extern std::atomic<int> g_opt_p1_i8;      // p1i8.hip
extern std::atomic<int> g_opt_i8_guard_strict;
extern std::atomic<int> g_opt_gs_i8;      // gsi8.hip

// r05, the global step at M >= 1024 (each switchable for same-box A/B through gp_debug_set_option):
std::atomic<int> g_opt_xtx_tri{1};       // A^-1 = X^T X from its lower tiles, k from the tile's first non-zero row, mirrored store (bit-identical)
std::atomic<int> g_opt_residual_dd{1};   // the refinement residual through ddacc_block (two rows per wave share E's loads) for Mp >= 256
std::atomic<int> g_opt_trtri_rec{1};     // L^-1 by halves: two batched launches per level instead of two per block row
std::atomic<int> g_opt_gemm_big{1};      // the M x M x {M, D} products on the 128 x 128-tile kernel (split-k 8 at M = 1024) for Mp >= 1024

constexpr int kSplitK = 8;
// split-k factor for a product of `tiles` 128 x 128 output tiles (batch included) with contraction length K on the 128-tile kernel: about 512 workgroups
// (two per CU), a power of two <= 8 that divides the number of k-chunks, and whose partial tiles fit the workspace (cap doubles)
static int choose_splits(long tiles, int K, size_t cap) {
  int s = 8;
  while (s > 1 && (tiles * s > 512 || (K / KC) % s != 0 || (size_t)tiles * s * TILE * TILE > cap)) s >>= 1;
  return s;
}   // split-k factor of the M x M x M products of the global step (latency-bound: 16 tiles alone fill 6 % of the chip)



// A: [batch][Mp][Mp] SPD in, lower Cholesky factor out (upper zeroed); Linv: L^-1; Inv: A^-1; Twork: batch * Mp * Mp / 2 doubles
// dst[b][r][0:128] = src[b][r][0:128] for r < rows: the panel solve's result from the work panel into the factor (two doubles per thread)
__global__ void __launch_bounds__(256) panel_copy_kernel(const double* __restrict__ src, long sstride, double* __restrict__ dst, long ld, long dstride, long rows) {
  const double* s = src + (long)blockIdx.y * sstride;
  double* d = dst + (long)blockIdx.y * dstride;
  const long total = rows * (NB / 2);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const long r = i / (NB / 2), c2 = i - r * (NB / 2);
    *reinterpret_cast<double2*>(d + r * ld + 2 * c2) = *reinterpret_cast<const double2*>(s + r * NB + 2 * c2);
  }
}

int potrf_inverse_batched(gp_ctx* c, hipStream_t st, int Mp, int batch, double* A, double* Linv, double* Inv, double* Twork,
                          double* logdet2, double* fail_flag, double* splitk_ws, size_t splitk_cap) {
  const int nt = Mp / NB;
  const long ld = Mp, bs = (long)Mp * Mp;
  // a per-device attribute: set on every call (cheap) rather than once per process -- contexts may live on several GPUs
  GP_HIP(c, hipFuncSetAttribute((const void*)potrf_trinv128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS_DOUBLES * 8));
  // Linv's 128-blocks above the diagonal are never written and are zero from the allocation (gp_create / the test hook)
  for (int j = 0; j < nt; ++j) {
    hipLaunchKernelGGL(potrf_trinv128_kernel, dim3(batch), dim3(512), POTRF_LDS_DOUBLES * 8, st, A, ld, bs, j, Linv, fail_flag, logdet2);
    const int rem = nt - j - 1;
    if (rem > 0) {
      // panel: L[i,j] = A[i,j] * inv(L_jj)^T, i > j   (rows rem*128, cols 128, k 128) -- through the work panel, see below
      GemmP p;
      p.A = A + ((long)(j + 1) * NB) * ld + (long)j * NB; p.lda = ld; p.sA = bs;
      p.B = Linv + ((long)j * NB) * ld + (long)j * NB; p.ldb = ld; p.sB = bs;   // B(k,c) = Xjj[c][k]: stored [c][k] -> K_CONTIG
      p.K = NB; p.alpha = 1.0; p.beta = 0.0; p.tri = 0;
      // NOT in place (r06).  Until then C was A itself, and launch_gemm's small-tile path (32 x 32 tiles for <= 256 tiles) ran it with four workgroups per 32 rows,
      // each reading all 128 columns of the rows and overwriting 32 of them: a race that timing hid -- every workgroup resident and in step, the reads over long
      // before the first store -- except on a cold start: the FIRST evaluation of a fresh process at M = 1024 with free embeddings came back with both
      // factorisations flagged in 40-70 % of the processes (NaN from panel 1 on; the evaluator then repeats the step with the reference's 1e-7 jitter: F 2.9e-8,
      // grad_Z 4.1e-6 off -- inside the parity tolerance on a well-conditioned problem; on round 5's failing test_gpu_tile_phase2 shape 9.973e-5, the
      // logged 9.97e-5: that was this, its evaluations run in a fresh child process).  The product goes to the work panel and a copy kernel puts it
      // in place (+ ~5 us per panel; the 128 x 128-tile kernel in place -- one workgroup owns all columns of its rows -- is safe too but costs 20 us per panel).
      // profiles/r06_first_evaluation_race.txt, tests/test_gpu_first_evaluation.py.
      p.C = Twork; p.ldc = NB; p.sC = (long)rem * NB * NB;
      launch_gemm(st, K_CONTIG, K_CONTIG, rem * NB, NB, batch, p);
      hipLaunchKernelGGL(panel_copy_kernel, dim3((unsigned)std::min<long>(((long)rem * NB * NB / 2 + 255) / 256, 512), batch), dim3(256), 0, st, (const double*)Twork,
                         (long)rem * NB * NB, A + ((long)(j + 1) * NB) * ld + (long)j * NB, ld, bs, (long)rem * NB);
      // trailing update: A[i,k] -= L[i,j] L[k,j]^T for i >= k > j (lower tiles)
      GemmP q;
      q.A = A + ((long)(j + 1) * NB) * ld + (long)j * NB; q.lda = ld; q.sA = bs;
      q.B = q.A; q.ldb = ld; q.sB = bs;                                         // B(k,c) = L[c][k] -> K_CONTIG
      q.C = A + ((long)(j + 1) * NB) * ld + (long)(j + 1) * NB; q.ldc = ld; q.sC = bs;
      q.K = NB; q.alpha = -1.0; q.beta = 1.0; q.tri = 1;
      launch_gemm(st, K_CONTIG, K_CONTIG, rem * NB, rem * NB, batch, q);
    }
  }
  if (g_opt_trtri_rec.load()) {
    // X = L^-1 below the diagonal blocks by halves (r05): with L = [L11 0 ; L21 L22], X21 = -X22 (L21 X11).  Level h = 1, 2, 4, ... (half size in
    // 128-blocks): every pair of halves of that size is independent of the others, so a level is TWO batched launches whatever M -- 2 log2(M / 128)
    // launches with M / 256 ... 1 products each instead of 2 (M / 128 - 1) launches of one growing block row (M = 1024: 6 launches for 14, 115 us before).
    // A pair whose second half runs past the matrix (block counts that are no power of two) is launched on its own with the shorter row count.
    for (int h = 1; h < nt; h *= 2) {
      const long b = (long)h * NB, ps = 2 * b * (ld + 1);
      const int full = nt / (2 * h), rem = nt - full * 2 * h - h;
      auto level = [&](int p0, int np, int rows2) {
        const long o11 = ((long)(2 * p0 * h) * NB) * (ld + 1), o22 = o11 + b * (ld + 1), o21 = o11 + b * ld;
        const long m2 = (long)rows2 * NB;
        GemmP p;                                                        // T = L21 X11
        p.A = A + o21; p.lda = ld; p.sA = ps; p.oA = bs;                // L21 [m][k], K_CONTIG
        p.B = Linv + o11; p.ldb = ld; p.sB = ps; p.oB = bs;             // X11 stored [k][c], FREE_CONTIG
        p.C = Twork; p.ldc = b; p.sC = m2 * b; p.oC = (long)np * m2 * b;
        p.K = (int)b; p.alpha = 1.0; p.beta = 0.0; p.tri = 0; p.inner = np;
        launch_gemm(st, K_CONTIG, FREE_CONTIG, (int)m2, (int)b, np * batch, p);
        GemmP q;                                                        // X21 = -X22 T
        q.A = Linv + o22; q.lda = ld; q.sA = ps; q.oA = bs;             // X22 [m][k], K_CONTIG
        q.B = Twork; q.ldb = b; q.sB = m2 * b; q.oB = (long)np * m2 * b;
        q.C = Linv + o21; q.ldc = ld; q.sC = ps; q.oC = bs;
        q.K = (int)m2; q.alpha = -1.0; q.beta = 0.0; q.tri = 0; q.inner = np;
        launch_gemm(st, K_CONTIG, FREE_CONTIG, (int)m2, (int)b, np * batch, q);
      };
      if (full > 0) level(0, full, h);
      if (rem > 0) level(full, 1, rem);
    }
  } else {
    // block rows of X = L^-1: X[i,0:i] = -X_ii * (L[i,0:i] * X[0:i,0:i])
    for (int i = 1; i < nt; ++i) {
      GemmP p;
      p.A = A + ((long)i * NB) * ld; p.lda = ld; p.sA = bs;            // L row panel (128 x i*128), K_CONTIG
      p.B = Linv; p.ldb = ld; p.sB = bs;                               // X[0:i,0:i] stored [k][c] -> FREE_CONTIG
      p.C = Twork; p.ldc = Mp; p.sC = (long)NB * Mp;
      p.K = i * NB; p.alpha = 1.0; p.beta = 0.0; p.tri = 0;
      launch_gemm(st, K_CONTIG, FREE_CONTIG, NB, i * NB, batch, p);
      GemmP q;
      q.A = Linv + ((long)i * NB) * ld + (long)i * NB; q.lda = ld; q.sA = bs;   // X_ii, K_CONTIG
      q.B = Twork; q.ldb = Mp; q.sB = (long)NB * Mp;                            // T stored [k][c] -> FREE_CONTIG
      q.C = Linv + ((long)i * NB) * ld; q.ldc = ld; q.sC = bs;
      q.K = NB; q.alpha = -1.0; q.beta = 0.0; q.tri = 0;
      launch_gemm(st, K_CONTIG, FREE_CONTIG, NB, i * NB, batch, q);
    }
  }
  // A^-1 = X^T X
  GemmP r;
  r.A = Linv; r.lda = ld; r.sA = bs;   // A(i,k) = X[k][i]: stored [k][i] -> FREE_CONTIG
  r.B = Linv; r.ldb = ld; r.sB = bs;   // B(k,j) = X[k][j] -> FREE_CONTIG
  r.C = Inv; r.ldc = ld; r.sC = bs;
  r.K = Mp; r.alpha = 1.0; r.beta = 0.0; r.tri = 0;
  if (g_opt_xtx_tri.load()) { r.tri = 1; r.klow = 1; r.mirror = 1; }
  if (g_opt_gemm_big.load() && Mp >= 1024 && splitk_ws) {
    const long tiles = (long)(Mp / TILE) * (Mp / TILE) * batch;
    r.splits = choose_splits(tiles, Mp, splitk_cap); r.ws = splitk_ws;
    r.big = tiles * r.splits >= 256 ? 1 : 0;
    if (!r.big) r.splits = 1;
  } else if (splitk_ws && Mp >= 256 && Mp <= 1024 && (Mp / KC) % kSplitK == 0 && (size_t)batch * (Mp / TILE) * (Mp / TILE) * kSplitK * TILE * TILE <= splitk_cap) {
    r.splits = kSplitK; r.ws = splitk_ws;       // (ignored by the small-tile kernel that serves these sizes)
  }
  launch_gemm(st, FREE_CONTIG, FREE_CONTIG, Mp, Mp, batch, r);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

// ---------------------------------------------------------------------------------------------- global step kernels
// Kmm (kernels.py:108-111) and A = Kmm + beta * Psi2, identity in the padded block
// jitK / jitA: the 1e-7 * I the reference adds when a determinant sign is negative (partial_terms.py:452-456); 0 on the first attempt
__global__ void __launch_bounds__(256) build_kmm_kernel(const double* __restrict__ Z, const double* __restrict__ alpha, double sf2,
                                                         double beta, const double* __restrict__ Psi2, int M, int Mp, int Q,
                                                         double* __restrict__ Kmm, double* __restrict__ A, double* __restrict__ Keep,
                                                         double jitK, double jitA, double* __restrict__ gs_zero, double* __restrict__ Acopy) {
  // workgroup = 16 rows x 64 columns: the 64 columns' inducing points transposed into LDS ([q][column], conflict-free), the row's point wave-uniform
  // (scalar loads); one element per lane and row.  (The first form read Z[k * Q + q] with a stride of Q doubles across the lanes: 37 us at M = 1024,
  // Q = 50.)  The exponent is summed over q in the same order as before: same bits.
  __shared__ double zs[64 * 65];
  __shared__ double as[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k0 = blockIdx.x * 64, i0 = blockIdx.y * 16;
  // the global step's device scalars and failure flags start from zero (this used to be a hipMemsetAsync: a blit dispatch with ~10 us of idle
  // stream around it); the panel kernels that write them are later launches
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid < GS_COUNT + 8) gs_zero[tid] = 0.0;
  const int Qs = Q < 64 ? Q : 64;                     // latent dimensions beyond 64 are read from global memory
  for (int e = tid; e < 64 * Qs; e += 256) {
    const int kk = e / Qs, q = e - kk * Qs;
    zs[q * 65 + kk] = (k0 + kk < M) ? Z[(long)(k0 + kk) * Q + q] : 0.0;
  }
  if (tid < Qs) as[tid] = alpha[tid];
  __syncthreads();
  const int k = k0 + lane;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int i = i0 + 4 * wave + rr;
    const long idx = (long)i * Mp + k;
    double v;
    if (i < M && k < M) {
      const double* zi = Z + (long)i * Q;
      double e = 0.0;
      for (int q = 0; q < Qs; ++q) {
        const double d = zi[q] - zs[q * 65 + lane];
        e = fma(as[q] * d, d, e);
      }
      for (int q = Qs; q < Q; ++q) {
        const double d = zi[q] - Z[(long)k * Q + q];
        e = fma(alpha[q] * d, d, e);
      }
      v = sf2 * exp(-0.5 * e);
    } else {
      v = (i == k) ? 1.0 : 0.0;
    }
    const bool diag = (i == k) && i < M;
    Kmm[idx] = v + (diag ? jitK : 0.0);
    Keep[idx] = v;
    const double a = ((i < M && k < M) ? fma(beta, Psi2[idx], v) : v) + (diag ? jitA : 0.0);   // the same expression as in solve_residual_kernel
    A[idx] = a;
    if (Acopy) Acopy[idx] = a;                          // the factorisation overwrites A; the double-double residual reads this copy
  }
}

// One step of iterative refinement for E = A^-1 C (A = Kmm + beta Psi2, cond(A) ~ 1e10 at the benchmark's size): R = C - A E with the
// products and the sum carried in double-double (error-free product by FMA, two-sum accumulation), rounded to double at the end; the caller
// then adds P R.  A is rebuilt from Kmm and Psi2 exactly as build_kmm_kernel rounds it (the factorisation overwrote its copy).  One
// workgroup per row m: A[m][k] is wave-uniform, E[k][:] a coalesced row.  With float64 residuals the step is worthless (the residual IS the
// rounding error); with this one the error of grad_Z against an 80-bit evaluation drops from 1.3e-5 to 7.5e-6 at N = 1e6.
// `identity`: C = I (the residual of an inverse: refining P = A^-1 the same way was measured -- no change in grad_Z, +0.17 ms -- and is not done).
// the double-double partial of row m, column d over k in [k0, k1): the chain every quarter of solve_residual_kernel runs (and the fused tail)
__device__ __forceinline__ void residual_chain(const double* __restrict__ krow, const double* __restrict__ prow, const double* __restrict__ E, double beta,
                                               double jitA, int m, int d, int Dp, int k0, int k1, double& hi, double& lo) {
#pragma clang fp contract(off)   // the error-free transformations below must not be fused (hi + a e as one FMA breaks the two-sum)
  hi = 0.0; lo = 0.0;
#pragma unroll 8
  for (int k = k0; k < k1; ++k) {
    const double a = fma(beta, prow[k], krow[k]) + (k == m ? jitA : 0.0);
    const double e = E[(long)k * Dp + d];
    const double pr = a * e, pe = fma(a, e, -pr);           // a e = pr + pe exactly
    const double t = hi + pr, bb = t - hi;                   // two-sum
    lo += ((hi - (t - bb)) + (pr - bb)) + pe;
    hi = t;
  }
}
// (hi, lo) += (x, xl) in double-double
__device__ __forceinline__ void residual_add(double& hi, double& lo, double x, double xl) {
#pragma clang fp contract(off)
  const double t = hi + x, bb = t - hi;
  lo += ((hi - (t - bb)) + (x - bb)) + xl;
  hi = t;
}
__global__ void __launch_bounds__(512) solve_residual_kernel(const double* __restrict__ Keep, const double* __restrict__ Psi2, double beta, double jitA,
                                                             const double* __restrict__ C, const double* __restrict__ E, int M, int Mp, int Dp,
                                                             double* __restrict__ R, int identity) {
#pragma clang fp contract(off)
  // 128 columns x 4 quarters of the contraction range per workgroup: four waves per SIMD hide the latency of the E loads behind each other's
  // two-sum chains (one thread per column, 1024 waves in all: 44 us at M = 512).  The quarter is wave-uniform (readfirstlane), so A[m][k] stays a
  // scalar load.  The four double-double partials are added in order by the first quarter's thread.
  __shared__ double ph[3][128], pl[3][128];
  const int m = blockIdx.x, dl = threadIdx.x & 127, kq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);
  const double* krow = Keep + (long)m * Mp;
  const double* prow = Psi2 + (long)m * Mp;
  const int kper = (M + 3) / 4, k0 = kq * kper, k1 = min(M, k0 + kper);
  for (int d0 = 0; d0 < Dp; d0 += 128) {
    const int d = d0 + dl;
    double hi = 0.0, lo = 0.0;
    if (d < Dp) residual_chain(krow, prow, E, beta, jitA, m, d, Dp, k0, k1, hi, lo);
    if (kq > 0) { ph[kq - 1][dl] = hi; pl[kq - 1][dl] = lo; }
    __syncthreads();
    if (kq == 0 && d < Dp) {
#pragma unroll
      for (int j = 0; j < 3; ++j) residual_add(hi, lo, ph[j][dl], pl[j][dl]);
      const double c0 = identity ? (d == m ? 1.0 : 0.0) : C[(long)m * Dp + d];
      R[(long)m * Dp + d] = (c0 - hi) - lo;
    }
    __syncthreads();
  }
}
// the same row by a 256-thread workgroup (Dp = 128): thread (dl, h) runs quarters h + 2 and h one after the other -- the arithmetic of every quarter
// and the order in which the four partials are added are those of solve_residual_kernel.  ph / pl: [3][128] doubles of LDS each.
__device__ __forceinline__ void residual_row256(int m, const double* __restrict__ Keep, const double* __restrict__ Psi2, double beta, double jitA,
                                                const double* __restrict__ C, const double* __restrict__ E, int M, int Mp, int Dp,
                                                double* __restrict__ R, double (*ph)[128], double (*pl)[128]) {
#pragma clang fp contract(off)
  const int dl = threadIdx.x & 127, h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);
  const double* krow = Keep + (long)m * Mp;
  const double* prow = Psi2 + (long)m * Mp;
  const int kper = (M + 3) / 4;
  double hi, lo, hi2, lo2;
  { const int kq = h + 2, k0 = kq * kper, k1 = min(M, k0 + kper); residual_chain(krow, prow, E, beta, jitA, m, dl, Dp, k0, k1, hi2, lo2); }
  { const int kq = h, k0 = kq * kper, k1 = min(M, k0 + kper); residual_chain(krow, prow, E, beta, jitA, m, dl, Dp, k0, k1, hi, lo); }
  ph[h + 1][dl] = hi2; pl[h + 1][dl] = lo2;          // quarters 2 and 3 -> slots 1 and 2
  if (h == 1) { ph[0][dl] = hi; pl[0][dl] = lo; }    // quarter 1 -> slot 0
  __syncthreads();
  if (h == 0) {
#pragma unroll
    for (int j = 0; j < 3; ++j) residual_add(hi, lo, ph[j][dl], pl[j][dl]);
    R[(long)m * Dp + dl] = (C[(long)m * Dp + dl] - hi) - lo;
  }
  __syncthreads();
}

// C = A B with the products error-free (FMA) and the sums carried in double-double (two-sum), rounded to double once at the end.
// A [rows][K] row-major with the wave's RB rows wave-uniform (scalar loads), B [K][cols] row-major (lane = column: coalesced), both float64.
// Used for G = K_mm^-1 Psi2, the one product of the global step whose float64 ACCUMULATION carries the whole remaining error of grad_Z at the
// benchmark's conditioning: K_mm^-1 has entries of both signs around 1e5, Psi2 entries around N, and the product is (K_mm^-1 A - I) / beta -- a
// small difference.  With this one product accumulated in double-double (inputs and output float64, everything else as before) grad_Z's
// distance from the 80-bit truth drops from 1.1e-5 to 1.6e-8 at N = 1e5 and stays at 1e-8 .. 2e-8 for every data / inducing-point draw tried
// (the same arithmetic emulated with numpy error-free transformations before it was built).  No extended-precision inverse,
// no Newton step, no double-double storage is needed.  1024 waves at M = 512 with RB = 4; the four waves of a workgroup share their column
// block, so B's rows are read from L2 once per workgroup.
// SUB: the residual of the refinement step, C = (Csub - hi) - lo (solve_residual_kernel's rounding), instead of C = hi + lo.
template <int RB, int KU, bool SUB = false>
__device__ __forceinline__ void ddacc_block(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb, int K,
                                            double* __restrict__ C, long ldc, int vbx, int vby, const double* __restrict__ Csub = nullptr) {
#pragma clang fp contract(off)   // hi + a b as one FMA would break the two-sum
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = vbx * 64 + lane;
  const int i0 = (vby * 4 + wave) * RB;
  double hi[RB], lo[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) { hi[r] = 0.0; lo[r] = 0.0; }
  const double* bp = B + j;
  const double* ap = A + (long)i0 * lda;
  for (int k = 0; k < K; k += KU) {
    double bv[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) bv[u] = bp[(long)(k + u) * ldb];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const double a = ap[(long)r * lda + k + u];              // wave-uniform: a scalar load
        const double pr = a * bv[u], pe = fma(a, bv[u], -pr);    // a b = pr + pe exactly
        const double t = hi[r] + pr, bb = t - hi[r];             // two-sum
        lo[r] += ((hi[r] - (t - bb)) + (pr - bb)) + pe;
        hi[r] = t;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    if constexpr (SUB) C[(long)(i0 + r) * ldc + j] = (Csub[(long)(i0 + r) * ldc + j] - hi[r]) - lo[r];
    else C[(long)(i0 + r) * ldc + j] = hi[r] + lo[r];
  }
}
template <int RB, int KU>
__global__ void __launch_bounds__(256) ddacc_gemm_kernel(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb, int K,
                                                          double* __restrict__ C, long ldc) {
  ddacc_block<RB, KU>(A, lda, B, ldb, K, C, ldc, blockIdx.x, blockIdx.y);
}
// R = C - A E with the sum in double-double: the refinement residual for Mp >= 256 (solve_residual_kernel: one row per workgroup, every lane its own
// load of E per product and two extra instructions to rebuild A[m][k]; 570 us at M = 1024, D = 1000 against the 356 us of the equally large
// product above).  A is the copy build_kmm_kernel keeps, padded rows and columns included: R's padding comes out as exact zeros.
template <int RB, int KU>
__global__ void __launch_bounds__(256) ddacc_residual_kernel(const double* __restrict__ A, long lda, const double* __restrict__ E, long lde, int K,
                                                              const double* __restrict__ Csub, double* __restrict__ R) {
  ddacc_block<RB, KU, true>(A, lda, E, lde, K, R, lde, blockIdx.x, blockIdx.y, Csub);
}

// sum over the M x M (or M x D) block of x o y; one block per pair, results into out[slot]
struct DotJob { const double* x; const double* y; long ld; int rows, cols; int slot; };
struct DotJobs { DotJob j[8]; int n; };
constexpr int DOT_BLOCKS = 64;
// virtual block (vbx of DOT_BLOCKS, job vby) by the calling 256-thread workgroup; red: 256 doubles of LDS
__device__ __forceinline__ void dots_block(const DotJobs& jobs, double* part, int vbx, int vby, double* red) {
  // per-block partial sums part[job][block]; scalars_kernel adds them in a fixed order, so the bound is
  // bit-identical from run to run (an atomicAdd here made the last bits of F depend on the block schedule)
  const DotJob jb = jobs.j[vby];
  double s = 0.0;
  for (int r = vbx; r < jb.rows; r += DOT_BLOCKS)
    for (int c = threadIdx.x; c < jb.cols; c += 256) s += jb.x[(long)r * jb.ld + c] * jb.y[(long)r * jb.ld + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[vby * DOT_BLOCKS + vbx] = red[0];
  __syncthreads();
}
__global__ void __launch_bounds__(256) dots_kernel(DotJobs jobs, double* part) {
  // grid (DOT_BLOCKS, jobs)
  __shared__ double red[256];
  dots_block(jobs, part, blockIdx.x, blockIdx.y, red);
}

// Bbar, dF/dKmm, Abar and the phase-2 operand Bm = [2 Bbar ; Abar^T]
__device__ __forceinline__ void assemble_elem(long idx, const double* __restrict__ Ki, const double* __restrict__ P, const double* __restrict__ EEt,
                                              const double* __restrict__ KPK, const double* __restrict__ E, double beta, double Dd, int Mp, int Dp,
                                              double* __restrict__ Bbar, double* __restrict__ dFdK, double* __restrict__ Abar, double* __restrict__ Bm) {
  const long mm = (long)Mp * Mp;
  if (idx < mm) {
    const double kp = Ki[idx] - P[idx];
    const double b = 0.5 * beta * Dd * kp - 0.5 * beta * beta * beta * EEt[idx];
    Bbar[idx] = b;
    dFdK[idx] = 0.5 * Dd * kp - 0.5 * beta * Dd * KPK[idx] - 0.5 * beta * beta * EEt[idx];
    Bm[idx] = 2.0 * b;
  } else {
    const long e = idx - mm;
    const long m = e / Dp, d = e - m * Dp;
    const double a = beta * beta * E[e];
    Abar[e] = a;
    Bm[mm + d * Mp + m] = a;
  }
}
__global__ void __launch_bounds__(256) assemble_kernel(const double* __restrict__ Ki, const double* __restrict__ P,
                                                        const double* __restrict__ EEt, const double* __restrict__ KPK,
                                                        const double* __restrict__ E, double beta, double Dd, int Mp, int Dp,
                                                        double* __restrict__ Bbar, double* __restrict__ dFdK, double* __restrict__ Abar,
                                                        double* __restrict__ Bm) {
  const long mm = (long)Mp * Mp, md = (long)Mp * Dp;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < mm + md; idx += (long)gridDim.x * 256L)
    assemble_elem(idx, Ki, P, EEt, KPK, E, beta, Dd, Mp, Dp, Bbar, dFdK, Abar, Bm);
}

// F, grad_beta, grad_sf2 from the traces (partial_terms.py:464-472, 346-358, 322-333)
__device__ __forceinline__ void scalars_block(const double* sc, double* gs, const DotJobs& jobs, const double* part, double beta, double sf2, double Dd,
                                              double Nglob) {
  if (threadIdx.x < jobs.n) {
    double s = 0.0;
    for (int b = 0; b < DOT_BLOCKS; ++b) s += part[threadIdx.x * DOT_BLOCKS + b];
    gs[jobs.j[threadIdx.x].slot] = s;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double sumYY = sc[SC_SUM_YYT], Psi0 = sc[SC_PSI0], KL = sc[SC_KL];
  const double ldK = gs[GS_LOGDET_K], ldA = gs[GS_LOGDET_A];
  const double trKi = gs[GS_TR_KIPSI2], trP = gs[GS_TR_PPSI2], trCE = gs[GS_TR_CE], trEPE = gs[GS_TR_EPSI2E];
  const double two_pi = 6.283185307179586476925286766559;
  gs[GS_F] = -0.5 * Nglob * Dd * log(two_pi) + 0.5 * Dd * Nglob * log(beta) + 0.5 * Dd * ldK - 0.5 * Dd * ldA - 0.5 * beta * sumYY -
             0.5 * beta * Dd * Psi0 + 0.5 * beta * Dd * trKi + 0.5 * beta * beta * trCE - KL;
  gs[GS_GRAD_BETA] = 0.5 * Nglob * Dd / beta - 0.5 * Dd * trP - 0.5 * sumYY - 0.5 * Dd * Psi0 + 0.5 * Dd * trKi + beta * trCE -
                     0.5 * beta * beta * trEPE;
  gs[GS_GRAD_SF2] = (gs[GS_SUM_V] + gs[GS_SUM_AC] + 2.0 * gs[GS_SUM_BPSI2] + (-0.5 * beta * Dd) * Psi0) / sf2;
}
__global__ void scalars_kernel(const double* sc, double* gs, DotJobs jobs, const double* part, double beta, double sf2, double Dd, double Nglob) {
  if (blockIdx.x != 0) return;
  scalars_block(sc, gs, jobs, part, beta, sf2, Dd, Nglob);
}

// row j by 128 threads (tid 0..127: two waves); red: 2 * QC doubles of LDS owned by these 128 threads.  `active` false: the threads only keep the
// workgroup barriers in step (the fused tail runs two rows per 256-thread workgroup, and the last pair may be half empty).  This is the form the
// one-panel tail and kmm_grads_kernel run.
constexpr int KG_QC = 8;
__device__ __forceinline__ void kmm_grads_row(int j, bool active, int tid, double* red, const double* __restrict__ dFdK, const double* __restrict__ Kmm,
                                              const double* __restrict__ Bbar, const double* __restrict__ Psi2, const double* __restrict__ Z,
                                              const double* __restrict__ alpha, int M, int Mp, int Q, int regimeA, double* __restrict__ gZ,
                                              double* __restrict__ gapart) {
  // latent dimensions in chunks of 8: the sums of a chunk stay in registers over the row, then one butterfly per sum and one
  // LDS hand-over between the two waves (the first version ran two 7-step workgroup reductions per latent dimension)
  constexpr int QC = KG_QC;
  const int lane = tid & 63, wave = tid >> 6;
  for (int q0 = 0; q0 < Q; q0 += QC) {
    double sz[QC], sa[QC], zj[QC];
#pragma unroll
    for (int u = 0; u < QC; ++u) { sz[u] = 0.0; sa[u] = 0.0; zj[u] = (active && q0 + u < Q) ? Z[(long)j * Q + q0 + u] : 0.0; }
    if (active)
      for (int m = tid; m < M; m += 128) {
        const double k = Kmm[(long)j * Mp + m];
        const double fjm = dFdK[(long)j * Mp + m];
        const double sym = (fjm + dFdK[(long)m * Mp + j]) * k;
        double w = -0.5 * fjm * k;
        if (!regimeA) w += -0.25 * Bbar[(long)j * Mp + m] * Psi2[(long)j * Mp + m];
#pragma unroll
        for (int u = 0; u < QC; ++u) {
          const double dz = zj[u] - ((q0 + u < Q) ? Z[(long)m * Q + q0 + u] : 0.0);
          sz[u] = fma(sym, dz, sz[u]);
          sa[u] = fma(w * dz, dz, sa[u]);
        }
      }
#pragma unroll
    for (int u = 0; u < QC; ++u)
      for (int sh = 32; sh > 0; sh >>= 1) { sz[u] += __shfl_xor(sz[u], sh); sa[u] += __shfl_xor(sa[u], sh); }
    if (wave == 1 && lane == 0) {
#pragma unroll
      for (int u = 0; u < QC; ++u) { red[u] = sz[u]; red[QC + u] = sa[u]; }
    }
    __syncthreads();
    if (active && wave == 0 && lane == 0) {
#pragma unroll
      for (int u = 0; u < QC; ++u)
        if (q0 + u < Q) {
          gZ[(long)j * Q + q0 + u] = -alpha[q0 + u] * (sz[u] + red[u]);
          gapart[(long)j * Q + q0 + u] = sa[u] + red[QC + u];
        }
    }
    __syncthreads();
  }
}
// Kmm-dependent parts: gK[j*Q+k] = -alpha_k sum_m' (dFdK+dFdK^T)[j,m'] Kmm[j,m'] (z_jk - z_m'k)
//                      gK[M*Q+q] = sum_mm' (-1/2 dFdK o Kmm [- 1/4 Bbar o Psi2 if !regimeA]) (z_mq - z_m'q)^2
// one block per row j; alpha parts are accumulated per block into gKpart[j][Q] and summed by the caller's reduce
__global__ void __launch_bounds__(128) kmm_grads_kernel(const double* __restrict__ dFdK, const double* __restrict__ Kmm,
                                                         const double* __restrict__ Bbar, const double* __restrict__ Psi2,
                                                         const double* __restrict__ Z, const double* __restrict__ alpha, int M, int Mp,
                                                         int Q, int regimeA, double* __restrict__ gZ, double* __restrict__ gapart) {
  __shared__ double red[2 * KG_QC];
  kmm_grads_row(blockIdx.x, true, threadIdx.x, red, dFdK, Kmm, Bbar, Psi2, Z, alpha, M, Mp, Q, regimeA, gZ, gapart);
}
// The same row with the q-independent weights computed once (they wait in LDS: [t][tid] and [nm + t][tid], every thread reads back only what it wrote), the
// inducing points read from the transposed copy Zt [Q][Mp] (lanes = consecutive inducing points: one 512-byte run per load instead of 64 cache lines), and
// four inducing points per trip with clamped addresses and zero weights past M (fma(0, dz, s) = s exactly), so a trip's loads are in flight together.
// Same sums in the same order as kmm_grads_row: same bits.  Same-box A/B against the plain form at M = 1024, Q = 50: see run_global_step.
__global__ void __launch_bounds__(128) kmm_grads_lds_kernel(const double* __restrict__ dFdK, const double* __restrict__ Kmm,
                                                             const double* __restrict__ Bbar, const double* __restrict__ Psi2,
                                                             const double* __restrict__ Z, const double* __restrict__ Zt, const double* __restrict__ alpha,
                                                             int M, int Mp, int Q, int regimeA, double* __restrict__ gZ, double* __restrict__ gapart) {
  constexpr int QC = KG_QC;
  __shared__ double red[2 * KG_QC];
  extern __shared__ __attribute__((aligned(16))) double symw[];      // 2 * 128 * ceil(M / 128) doubles
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nm = (M + 127) / 128;
  for (int t0 = 0; t0 < nm; t0 += 4) {
    double k[4], fjm[4], fmj[4], bb[4], pp[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int m = min(tid + 128 * (t0 + tt), M - 1);
      k[tt] = Kmm[(long)j * Mp + m];
      fjm[tt] = dFdK[(long)j * Mp + m];
      fmj[tt] = dFdK[(long)m * Mp + j];
      if (!regimeA) { bb[tt] = Bbar[(long)j * Mp + m]; pp[tt] = Psi2[(long)j * Mp + m]; }
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int t = t0 + tt;
      if (t < nm) {
        const bool ok = tid + 128 * t < M;
        const double sym = (fjm[tt] + fmj[tt]) * k[tt];
        double w = -0.5 * fjm[tt] * k[tt];
        if (!regimeA) w += -0.25 * bb[tt] * pp[tt];
        symw[t * 128 + tid] = ok ? sym : 0.0; symw[(nm + t) * 128 + tid] = ok ? w : 0.0;
      }
    }
  }
  for (int q0 = 0; q0 < Q; q0 += QC) {
    double sz[QC], sa[QC], zj[QC];
#pragma unroll
    for (int u = 0; u < QC; ++u) { sz[u] = 0.0; sa[u] = 0.0; zj[u] = (q0 + u < Q) ? Z[(long)j * Q + q0 + u] : 0.0; }
    for (int t0 = 0; t0 < nm; t0 += 4) {
      double zz[4][QC], sy[4], ww[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int t = t0 + tt, tc = t < nm ? t : nm - 1;
        const int m = min(tid + 128 * tc, M - 1);
        sy[tt] = t < nm ? symw[tc * 128 + tid] : 0.0;
        ww[tt] = t < nm ? symw[(nm + tc) * 128 + tid] : 0.0;
#pragma unroll
        for (int u = 0; u < QC; ++u) zz[tt][u] = (q0 + u < Q) ? Zt[(long)(q0 + u) * Mp + m] : 0.0;
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int u = 0; u < QC; ++u) {
          const double dz = zj[u] - zz[tt][u];
          sz[u] = fma(sy[tt], dz, sz[u]);
          sa[u] = fma(ww[tt] * dz, dz, sa[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < QC; ++u)
      for (int sh = 32; sh > 0; sh >>= 1) { sz[u] += __shfl_xor(sz[u], sh); sa[u] += __shfl_xor(sa[u], sh); }
    if (wave == 1 && lane == 0) {
#pragma unroll
      for (int u = 0; u < QC; ++u) { red[u] = sz[u]; red[QC + u] = sa[u]; }
    }
    __syncthreads();
    if (wave == 0 && lane == 0) {
#pragma unroll
      for (int u = 0; u < QC; ++u)
        if (q0 + u < Q) {
          gZ[(long)j * Q + q0 + u] = -alpha[q0 + u] * (sz[u] + red[u]);
          gapart[(long)j * Q + q0 + u] = sa[u] + red[QC + u];
        }
    }
    __syncthreads();
  }
}
// column q of part [rows][Q] summed by the calling 256-thread workgroup; red: 256 doubles of LDS
__device__ __forceinline__ void colsum_block(const double* __restrict__ part, int rows, int Q, double* __restrict__ out, int q, double* red) {
  double s = 0.0;
  for (int r = threadIdx.x; r < rows; r += 256) s += part[(long)r * Q + q];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) out[q] = red[0];
  __syncthreads();
}
__global__ void __launch_bounds__(256) colsum_kernel(const double* __restrict__ part, int rows, int Q, double* __restrict__ out) {
  __shared__ double red[256];
  colsum_block(part, rows, Q, out, blockIdx.x, red);
}

// ---------------------------------------------------------------------------------------------- short tail for one-panel problems
// M <= 128 and D <= 128 (BASELINE configs[1]: M = 128, D = 10): everything of the global step behind the panel factorisation -- the two inverses,
// E with its refinement step, Psi2 E, E E^T, K_mm^-1 Psi2 in double-double, its product with K_mm^-1, the assembled partials, the seven traces, the
// scalars and the K_mm parts of the gradients -- used to be FIFTEEN launches of 4-12 us each, every one of them at the launch floor
// (profiles/r05_config1_timeline.txt: 80 us for 2e7 flop).  Products that do not depend on each other now share a launch: SEVEN launches of one
// kernel (tail_stage_kernel, stage = 0..6), every workgroup taking one work item of its stage -- the same 32 x 32 tile products, double-double
// blocks, dot-product blocks and rows as the separate kernels (the device functions above are shared, so the results are bit-identical).
//   0: [Ki ; P] = X^T X (32 tiles)                         1: E = P C (16 tiles) | T2 = Ki Psi2 in double-double (64 blocks)
//   2: R = C - A E in double-double (128 rows) | T2 Ki (16 tiles)                    3: E += P R (16 tiles)
//   4: Psi2 E (16 tiles, then Abar and Bm's lower block for the tile) | E E^T (16 tiles, then Bbar, dF/dKmm, Bm's upper block for the tile)
//   5: the seven traces (7 x 64 blocks) | K_mm parts of grad_Z / grad_alpha (two rows per workgroup)        6: scalars | column sums
// Why launches and not one persistent kernel with grid barriers: that was built first (r05) and measured at 94 us against 80 for the fifteen
// launches -- on this part an agent-scope release + acquire (L2 write-back, L1 / L2 invalidate on eight XCDs) costs as much as a kernel boundary,
// which does the same thing in hardware (MI355X_MICROARCH.md, inter-workgroup visibility: 1.7 + 1.7 us of fences + the counter round trips).
constexpr int TAIL_STAGES = 7;
constexpr int TAIL_LDS_DOUBLES = 2 * SmallImg<FREE_CONTIG>::DOUBLES;     // the larger operand image, twice (12288 doubles = 96 KB)
struct TailP {
  const double* Linv; double* Inv;                    // [2][128][128]
  const double* Psi2; const double* C; const double* sc; const double* Keep;
  double *E, *PsiE, *T1, *T2, *dFdK, *Bbar, *Abar, *Bm;
  const double* Z; const double* alpha;
  double* gs; double* gK;
  DotJobs jobs;
  double beta, sf2, Dd, Nglob, jitA;
  int M, Q, regimeA, refine, dd;
};
__host__ __device__ inline int tail_items(int stage, int M, int Q, int refine, int dd) {
  switch (stage) {
    case 0: return 32;
    case 1: return 16 + (dd ? 64 : 16);
    case 2: return 16 + (refine ? 128 : 0);
    case 3: return refine ? 16 : 0;
    case 4: return 32;
    case 5: return 7 * DOT_BLOCKS + (M + 1) / 2;
    default: return 1 + Q;
  }
}
__global__ void __launch_bounds__(256, 1) tail_stage_kernel(TailP p, int stage) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* sA = sm;
  double* sB = sm + SmallImg<FREE_CONTIG>::DOUBLES;
  double* red = sm;                                     // the items without tile products reuse the front of the images
  constexpr int Mp = 128, Dp = 128;
  constexpr long mm = (long)Mp * Mp;
  const int t = blockIdx.x;
  double* Ki = p.Inv;
  double* P = p.Inv + mm;
  GemmP g;
  g.alpha = 1.0; g.beta = 0.0; g.tri = 0; g.sA = g.sB = g.sC = 0;
  if (stage == 0) {
    g.A = p.Linv; g.lda = Mp; g.sA = mm; g.B = p.Linv; g.ldb = Mp; g.sB = mm; g.C = p.Inv; g.ldc = Mp; g.sC = mm; g.K = Mp;
    gemm32_tile<FREE_CONTIG, FREE_CONTIG>(g, t & 3, (t >> 2) & 3, t >> 4, sA, sB);
  } else if (stage == 1) {
    if (t < 16) {
      g.K = Mp; g.A = P; g.lda = Mp; g.B = p.C; g.ldb = Dp; g.C = p.E; g.ldc = Dp;
      gemm32_tile<K_CONTIG, FREE_CONTIG>(g, t & 3, t >> 2, 0, sA, sB);
    } else if (p.dd) {
      // one row per wave: the rows are independent, so the bits are those of the two-rows-per-wave form; 4 us faster at M = 128 (profiles/r04_dd_variants.txt)
      ddacc_block<1, 8>(Ki, (long)Mp, p.Psi2, (long)Mp, Mp, p.T2, (long)Mp, (t - 16) & 1, (t - 16) >> 1);
    } else {
      g.K = Mp; g.A = Ki; g.lda = Mp; g.B = p.Psi2; g.ldb = Mp; g.C = p.T2; g.ldc = Mp;
      gemm32_tile<K_CONTIG, FREE_CONTIG>(g, (t - 16) & 3, (t - 16) >> 2, 0, sA, sB);
    }
  } else if (stage == 2) {
    if (t < 16) {
      g.K = Mp; g.A = p.T2; g.lda = Mp; g.B = Ki; g.ldb = Mp; g.C = p.dFdK; g.ldc = Mp;
      gemm32_tile<K_CONTIG, FREE_CONTIG>(g, t & 3, t >> 2, 0, sA, sB);
    } else {
      const int m = t - 16;
      double (*ph)[128] = reinterpret_cast<double (*)[128]>(red);
      double (*pl)[128] = reinterpret_cast<double (*)[128]>(red + 3 * 128);
      if (m < p.M) residual_row256(m, p.Keep, p.Psi2, p.beta, p.jitA, p.C, p.E, p.M, Mp, Dp, p.PsiE, ph, pl);
      else if (threadIdx.x < Dp) p.PsiE[(long)m * Dp + threadIdx.x] = 0.0;
    }
  } else if (stage == 3) {
    g.K = Mp; g.A = P; g.lda = Mp; g.B = p.PsiE; g.ldb = Dp; g.C = p.E; g.ldc = Dp; g.beta = 1.0;
    gemm32_tile<K_CONTIG, FREE_CONTIG>(g, t & 3, t >> 2, 0, sA, sB);
  } else if (stage == 4) {
    // the tile product, then the assembled outputs that need nothing but this tile and finished inputs: every element of the assembly is
    // computed once, by the expression of assemble_elem
    const int tt = t & 15, bx = tt & 3, by = tt >> 2;
    if (t < 16) {
      g.K = Mp; g.A = p.Psi2; g.lda = Mp; g.B = p.E; g.ldb = Dp; g.C = p.PsiE; g.ldc = Dp;
      gemm32_tile<K_CONTIG, FREE_CONTIG>(g, bx, by, 0, sA, sB);
      for (int e = threadIdx.x; e < ST * ST; e += 256)      // Abar and Bm's lower block: only E (finished in stage 3)
        assemble_elem(mm + (long)(by * ST + (e >> 5)) * Dp + bx * ST + (e & 31), Ki, P, p.T1, p.dFdK, p.E, p.beta, p.Dd, Mp, Dp, p.Bbar, p.dFdK, p.Abar, p.Bm);
    } else {
      g.K = Dp; g.A = p.E; g.lda = Dp; g.B = p.E; g.ldb = Dp; g.C = p.T1; g.ldc = Mp;
      gemm32_tile<K_CONTIG, K_CONTIG>(g, bx, by, 0, sA, sB);
      __syncthreads();                                      // the tile of E E^T this workgroup just stored (same CU: visible after the barrier)
      for (int e = threadIdx.x; e < ST * ST; e += 256)
        assemble_elem((long)(by * ST + (e >> 5)) * Mp + bx * ST + (e & 31), Ki, P, p.T1, p.dFdK, p.E, p.beta, p.Dd, Mp, Dp, p.Bbar, p.dFdK, p.Abar, p.Bm);
    }
  } else if (stage == 5) {
    const int nd = p.jobs.n * DOT_BLOCKS;
    if (t < nd) dots_block(p.jobs, p.gs + GS_COUNT + 8, t % DOT_BLOCKS, t / DOT_BLOCKS, red);
    else {
      const int half = threadIdx.x >> 7, j = 2 * (t - nd) + half;
      kmm_grads_row(j, j < p.M, threadIdx.x & 127, red + 2 * KG_QC * half, p.dFdK, p.Keep, p.Bbar, p.Psi2, p.Z, p.alpha, p.M, Mp, p.Q, p.regimeA,
                    p.gK, p.T2);
    }
  } else {
    if (t == 0) scalars_block(p.sc, p.gs, p.jobs, p.gs + GS_COUNT + 8, p.beta, p.sf2, p.Dd, p.Nglob);
    else colsum_block(p.T2, p.M, p.Q, p.gK + (long)p.M * p.Q, t - 1, red);
  }
}
std::atomic<int> g_opt_gs_tail{env_on("GPARML_GS_TAIL") ? 1 : 0};

// Host side of the global step's outcome: one D2H of the scalars + failure flags, at the first call that needs them
// (gp_global_status, gp_finish, gp_download).  Returns GP_OK, GP_ERR_NOT_PD, GP_ERR_NON_FINITE or GP_RETRY_JITTER.
int check_global(gp_ctx* c) {
  if (c->gs_pending) {
    double h[GS_COUNT + 8];
    GP_HIP(c, hipMemcpyAsync(h, c->gs, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    GP_HIP(c, hipStreamSynchronize(c->stream));
    ++c->sync_epoch;
    return check_global_from(c, h);
  }
  return check_global_from(c, nullptr);
}

int check_global_from(gp_ctx* c, const double* h) {
  if (c->gs_pending && h) {
    for (int i = 0; i < GS_COUNT; ++i) c->h_gs[i] = h[i];
    c->gs_pending = false;
    const int failed = (h[GS_COUNT] != 0.0 ? 1 : 0) | (h[GS_COUNT + 1] != 0.0 ? 2 : 0);
    const bool singular = h[GS_COUNT] == 2.0 || h[GS_COUNT + 1] == 2.0;
    if (singular) {
      // an exactly zero pivot: the reference fails in linalg.inv before any jitter applies (partial_terms.py:60, 95)
      c->gs_status = GP_ERR_NOT_PD;
      c->gs_msg = std::string(h[GS_COUNT] == 2.0 ? "Kmm" : "Kmm + beta*Psi2") + " is singular (numpy.linalg.inv: Singular matrix)";
    } else if (failed & ~c->jitter_mask) {
      // first failure of this matrix: the reference continues with 1e-7 * I added (partial_terms.py:452-456)
      c->retry_mask = c->jitter_mask | failed;
      c->gs_status = GP_RETRY_JITTER;
      c->gs_msg = std::string(failed & 1 ? "Kmm" : "Kmm + beta*Psi2") + " is not positive definite (Cholesky failed); retry with 1e-7 jitter";
    } else if (failed) {
      c->gs_status = GP_ERR_NOT_PD;
      c->gs_msg = std::string(failed & 1 ? "Kmm" : "Kmm + beta*Psi2") + " is not positive definite even with 1e-7 jitter (partial_terms.py:459-461 assertion)";
    } else if (!std::isfinite(h[GS_F])) {
      c->gs_status = GP_ERR_NON_FINITE;
      c->gs_msg = "bound is not finite";
    } else {
      c->gs_status = GP_OK;
    }
  }
  if (c->gs_status != GP_OK) return fail(c, c->gs_status, "%s", c->gs_msg.c_str());
  return GP_OK;
}

int run_global_step(gp_ctx* c) {
  hipStream_t st = c->stream;
  const int Mp = c->Mp, Dp = c->Dp, M = c->M, D = c->D, Q = c->Q;
  const long mm = (long)Mp * Mp;
  double* Psi2 = c->stats;
  double* C = c->stats + mm;
  double* sc = c->stats + mm + (long)Mp * Dp;
  c->gs_status = GP_OK;
  double* failf = c->gs + GS_COUNT;  // [2]
  // T2 is free until G = K_mm^-1 Psi2 is formed: it keeps A for the double-double residual of the refinement step
  const bool gi8 = gs_i8_wanted(c);            // gsi8.hip: both double-double products on the int8 matrix core (M >= 1024)
  const bool res_dd = g_opt_refine_E.load() && ((g_opt_residual_dd.load() && Mp >= 256 && Dp >= 512) || gi8);   // narrow E: too few waves (M = 512, D = 100: +21 us)
  hipLaunchKernelGGL(build_kmm_kernel, dim3(Mp / 64, Mp / 16), dim3(256), 0, st, c->Z, c->alpha, c->sf2, c->beta, Psi2, M, Mp, Q, c->Kmm, c->Kmm + mm,
                     c->KmmKeep, (c->jitter_mask & 1) ? 1e-7 : 0.0, (c->jitter_mask & 2) ? 1e-7 : 0.0, c->gs, res_dd ? c->T2 : (double*)nullptr);
  GP_HIP(c, hipGetLastError());
  // one-panel problems (M, D <= 128): the panel kernel, then seven launches of tail_stage_kernel instead of fifteen kernels
  if (Mp == NB && Dp == NB && g_opt_gs_tail.load()) {
    GP_HIP(c, hipFuncSetAttribute((const void*)potrf_trinv128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS_DOUBLES * 8));
    GP_HIP(c, hipFuncSetAttribute((const void*)tail_stage_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS_DOUBLES * 8));
    hipLaunchKernelGGL(potrf_trinv128_kernel, dim3(2), dim3(512), POTRF_LDS_DOUBLES * 8, st, c->Kmm, (long)Mp, mm, 0, c->Linv, failf, c->gs + GS_LOGDET_K);
    TailP t;
    t.Linv = c->Linv; t.Inv = c->Inv; t.Psi2 = Psi2; t.C = C; t.sc = sc; t.Keep = c->KmmKeep;
    t.E = c->E; t.PsiE = c->PsiE; t.T1 = c->T1; t.T2 = c->T2; t.dFdK = c->dFdK; t.Bbar = c->Bbar; t.Abar = c->Abar; t.Bm = c->Bm;
    t.Z = c->Z; t.alpha = c->alpha; t.gs = c->gs; t.gK = c->gK;
    t.beta = c->beta; t.sf2 = c->sf2; t.Dd = (double)D; t.Nglob = (double)c->N_global; t.jitA = (c->jitter_mask & 2) ? 1e-7 : 0.0;
    t.M = M; t.Q = Q; t.regimeA = c->regime_A ? 1 : 0; t.refine = g_opt_refine_E.load(); t.dd = g_opt_dd_kipsi2.load();
    t.jobs.n = 7;
    t.jobs.j[0] = {c->Inv, Psi2, Mp, M, M, GS_TR_KIPSI2};
    t.jobs.j[1] = {c->Inv + mm, Psi2, Mp, M, M, GS_TR_PPSI2};
    t.jobs.j[2] = {C, c->E, Dp, M, D, GS_TR_CE};
    t.jobs.j[3] = {c->E, c->PsiE, Dp, M, D, GS_TR_EPSI2E};
    t.jobs.j[4] = {c->dFdK, c->KmmKeep, Mp, M, M, GS_SUM_V};
    t.jobs.j[5] = {c->Abar, C, Dp, M, D, GS_SUM_AC};
    t.jobs.j[6] = {c->Bbar, Psi2, Mp, M, M, GS_SUM_BPSI2};
    for (int stage = 0; stage < TAIL_STAGES; ++stage) {
      const int items = tail_items(stage, M, Q, t.refine, t.dd);
      // only the stages with tile products need the operand images
      const size_t lds = (stage <= 4) ? (size_t)TAIL_LDS_DOUBLES * 8 : 4096;
      if (items > 0) hipLaunchKernelGGL(tail_stage_kernel, dim3(items), dim3(256), lds, st, t, stage);
    }
    GP_HIP(c, hipGetLastError());
    c->gs_pending = true;
    return GP_OK;
  }
  // factorise [Kmm ; A] in place, invert.  T1 is the 2 x 128 x Mp work panel.
  // split-k workspace: the phase-1 partial buffer is free during the global step (>= 600 tiles)
  double* ws = c->part;
  const size_t wcap = c->part_doubles;
  int rc = potrf_inverse_batched(c, st, Mp, 2, c->Kmm, c->Linv, c->Inv, c->T1, c->gs + GS_LOGDET_K, failf, ws, wcap);
  if (rc != GP_OK) return rc;
  double* Ki = c->Inv;
  double* P = c->Inv + mm;
  // E = P C ; PsiE = Psi2 E ; T1 = E E^T   and, independent of it,   T2 = Ki Psi2 ; dFdK(tmp) = T2 Ki.   One stream: a side stream for the second
  // chain was measured slower (r03: 0.424 -> 0.455 ms at M = 512 -- every cross-stream event edge costs more than the 5-12 us product it hides)
  // and was removed in r06.
  GemmP g;
  g.K = Mp; g.alpha = 1.0; g.beta = 0.0; g.tri = 0; g.sA = g.sB = g.sC = 0;
  // the 128-tile kernel with split-k where that gives >= 256 workgroups (M >= 1024: 72-74 -> 63 us per product incl. the reduce at M = 1024; at M = 2048 the
  // M x M x M product 616 -> 392 us without a split, the M x M x D ones have 128 tiles and lose to the small tiles unless split); E E^T stays on the small tiles
  const bool bigok = g_opt_gemm_big.load() && Mp >= 1024 && ws;
  const int spMD = bigok ? choose_splits((long)(Mp / TILE) * (Dp / TILE), Mp, wcap) : 1;
  const int spMM = bigok ? choose_splits((long)(Mp / TILE) * (Mp / TILE), Mp, wcap) : 1;
  const int bigMD = (bigok && (long)(Mp / TILE) * (Dp / TILE) * spMD >= 256) ? 1 : 0;
  const int bigMM = (bigok && (long)(Mp / TILE) * (Mp / TILE) * spMM >= 256) ? 1 : 0;
  g.ws = ws;
  g.big = bigMD; g.splits = bigMD ? spMD : 1;
  g.A = P; g.lda = Mp; g.B = C; g.ldb = Dp; g.C = c->E; g.ldc = Dp;
  launch_gemm(st, K_CONTIG, FREE_CONTIG, Mp, Dp, 1, g);
  // one refinement step of E with a double-double residual (PsiE is free until the next product); GPARML_REFINE_E=0 turns it off
  if (g_opt_refine_E.load()) {
    if (gi8) {
      GP_TRY_RC(run_gs_i8_product(c, st, c->T2, (long)Mp, Mp, c->E, (long)Dp, Dp, Mp, c->PsiE, (long)Dp, C));
    } else if (res_dd) {
      hipLaunchKernelGGL((ddacc_residual_kernel<2, 8>), dim3(Dp / 64, Mp / 8), dim3(256), 0, st, c->T2, (long)Mp, c->E, (long)Dp, Mp, C, c->PsiE);
    } else {
      hipLaunchKernelGGL(solve_residual_kernel, dim3(M), dim3(512), 0, st, c->KmmKeep, Psi2, c->beta, (c->jitter_mask & 2) ? 1e-7 : 0.0, C, c->E, M, Mp, Dp,
                         c->PsiE, 0);
      if (M < Mp) GP_HIP(c, hipMemsetAsync(c->PsiE + (long)M * Dp, 0, (size_t)(Mp - M) * Dp * sizeof(double), st));
    }
    GP_HIP(c, hipGetLastError());
    g.A = P; g.lda = Mp; g.B = c->PsiE; g.ldb = Dp; g.C = c->E; g.ldc = Dp; g.beta = 1.0;
    launch_gemm(st, K_CONTIG, FREE_CONTIG, Mp, Dp, 1, g);
    g.beta = 0.0;
  }
  g.A = Psi2; g.lda = Mp; g.B = c->E; g.ldb = Dp; g.C = c->PsiE; g.ldc = Dp;
  launch_gemm(st, K_CONTIG, FREE_CONTIG, Mp, Dp, 1, g);
  g.K = Dp; g.A = c->E; g.lda = Dp; g.B = c->E; g.ldb = Dp; g.C = c->T1; g.ldc = Mp;   // B(k,j) = E[j][k] -> K_CONTIG
  { const int sps = g.splits; g.splits = 1; g.big = 0; launch_gemm(st, K_CONTIG, K_CONTIG, Mp, Mp, 1, g); g.splits = sps; }
  // G = Ki Psi2 with double-double accumulation (ddacc_gemm_kernel above: two rows per wave, eight k per trip -- same-box timing of six shapes
  // in profiles/r04_dd_variants.txt: +50 us at M = 512, +9 us at M = 128, +0.29 ms at M = 1024 over the float64 matrix-core product of r03, which
  // GPARML_DD_KIPSI2=0 or gp_debug_set_option("dd_kipsi2", 0) restores)
  if (g_opt_dd_kipsi2.load() && gi8) {
    GP_TRY_RC(run_gs_i8_product(c, st, Ki, (long)Mp, Mp, Psi2, (long)Mp, Mp, Mp, c->T2, (long)Mp, nullptr));
  } else if (g_opt_dd_kipsi2.load()) {
    hipLaunchKernelGGL((ddacc_gemm_kernel<2, 8>), dim3(Mp / 64, Mp / 8), dim3(256), 0, st, Ki, (long)Mp, Psi2, (long)Mp, Mp, c->T2, (long)Mp);
    GP_HIP(c, hipGetLastError());
  } else {
    g.K = Mp; g.big = bigMM; g.splits = bigMM ? spMM : 1; g.A = Ki; g.lda = Mp; g.B = Psi2; g.ldb = Mp; g.C = c->T2; g.ldc = Mp;
    launch_gemm(st, K_CONTIG, FREE_CONTIG, Mp, Mp, 1, g);
  }
  g.K = Mp; g.big = bigMM; g.splits = bigMM ? spMM : 1;
  g.A = c->T2; g.lda = Mp; g.B = Ki; g.ldb = Mp; g.C = c->dFdK; g.ldc = Mp;
  launch_gemm(st, K_CONTIG, FREE_CONTIG, Mp, Mp, 1, g);
  GP_HIP(c, hipGetLastError());
  // dFdK currently holds Ki Psi2 Ki; assemble in place is unsafe (reads KPK, writes dFdK at the same index: fine, same thread)
  hipLaunchKernelGGL(assemble_kernel, dim3(1024), dim3(256), 0, st, Ki, P, c->T1, c->dFdK, c->E, c->beta, (double)D, Mp, Dp, c->Bbar,
                     c->dFdK, c->Abar, c->Bm);
  GP_HIP(c, hipGetLastError());
  DotJobs jobs;
  jobs.n = 7;
  jobs.j[0] = {Ki, Psi2, Mp, M, M, GS_TR_KIPSI2};
  jobs.j[1] = {P, Psi2, Mp, M, M, GS_TR_PPSI2};
  jobs.j[2] = {C, c->E, Dp, M, D, GS_TR_CE};
  jobs.j[3] = {c->E, c->PsiE, Dp, M, D, GS_TR_EPSI2E};
  jobs.j[4] = {c->dFdK, c->KmmKeep, Mp, M, M, GS_SUM_V};
  jobs.j[5] = {c->Abar, C, Dp, M, D, GS_SUM_AC};
  jobs.j[6] = {c->Bbar, Psi2, Mp, M, M, GS_SUM_BPSI2};
  double* dpart = c->gs + GS_COUNT + 8;   // [jobs][DOT_BLOCKS]
  // the traces / scalars and the Kmm parts of the gradients both start from the assembled partials and do not touch each other's outputs
  hipLaunchKernelGGL(dots_kernel, dim3(DOT_BLOCKS, jobs.n), dim3(256), 0, st, jobs, dpart);
  hipLaunchKernelGGL(scalars_kernel, dim3(1), dim3(64), 0, st, sc, c->gs, jobs, dpart, c->beta, c->sf2, (double)D, (double)c->N_global);
  // Kmm parts of grad_Z / grad_alpha; alpha partials per row go through T2 (free again)
  static const bool kmm_lds = [] { const char* e = getenv("GPARML_KMM_LDS"); return !(e && e[0] == '0'); }();
  if (kmm_lds && M <= 2048)
    hipLaunchKernelGGL(kmm_grads_lds_kernel, dim3(M), dim3(128), (size_t)2 * 128 * ((M + 127) / 128) * sizeof(double), st, c->dFdK, c->KmmKeep, c->Bbar, Psi2, c->Z,
                       c->Zt, c->alpha, M, Mp, Q, c->regime_A ? 1 : 0, c->gK, c->T2);
  else
    hipLaunchKernelGGL(kmm_grads_kernel, dim3(M), dim3(128), 0, st, c->dFdK, c->KmmKeep, c->Bbar, Psi2, c->Z, c->alpha, M, Mp, Q,
                       c->regime_A ? 1 : 0, c->gK, c->T2);
  hipLaunchKernelGGL(colsum_kernel, dim3(Q), dim3(256), 0, st, c->T2, M, Q, c->gK + (long)M * Q);
  GP_HIP(c, hipGetLastError());
  c->gs_pending = true;   // scalars and failure flags are read back at the next host synchronisation point (check_global)
  return GP_OK;
}

}  // namespace gp

// ---- test hooks ------------------------------------------------------------------------------------------------
extern "C" int gp_debug_set_option(const char* name, int value) {
  using namespace gp;
  if (!name) return fail(nullptr, GP_ERR_BAD_ARG, "gp_debug_set_option: NULL name");
  if (!std::strcmp(name, "dd_kipsi2")) { g_opt_dd_kipsi2.store(value ? 1 : 0); return GP_OK; }
  if (!std::strcmp(name, "refine_E")) { g_opt_refine_E.store(value ? 1 : 0); return GP_OK; }
  if (!std::strcmp(name, "p1_i8")) { g_opt_p1_i8.store(value ? 1 : 0); return GP_OK; }
  if (!std::strcmp(name, "gs_tail")) { g_opt_gs_tail.store(value ? 1 : 0); return GP_OK; }
  if (!std::strcmp(name, "i8_guard_strict")) { g_opt_i8_guard_strict.store(value ? 1 : 0); return GP_OK; }
  if (!std::strcmp(name, "xtx_tri")) { g_opt_xtx_tri.store(value ? 1 : 0); return GP_OK; }
  if (!std::strcmp(name, "residual_dd")) { g_opt_residual_dd.store(value ? 1 : 0); return GP_OK; }
  if (!std::strcmp(name, "gemm_big")) { g_opt_gemm_big.store(value ? 1 : 0); return GP_OK; }
  if (!std::strcmp(name, "trtri_rec")) { g_opt_trtri_rec.store(value ? 1 : 0); return GP_OK; }
  if (!std::strcmp(name, "gs_i8")) { g_opt_gs_i8.store(value ? 1 : 0); return GP_OK; }
  if (!std::strcmp(name, "poison_alloc")) { g_opt_poison.store(value ? 1 : 0); return GP_OK; }
  return fail(nullptr, GP_ERR_BAD_ARG, "gp_debug_set_option: unknown option '%s' (dd_kipsi2, refine_E, p1_i8, gs_tail, i8_guard_strict, xtx_tri, residual_dd, gemm_big, trtri_rec, gs_i8, poison_alloc)", name);
}

extern "C" int gp_debug_potrf_inverse(int device, int n, const double* A, double* L, double* Ainv, double* logdet) {
  using namespace gp;
  gp_ctx tmp;
  gp_ctx* c = &tmp;
  if (n <= 0 || !A) return fail(nullptr, GP_ERR_BAD_ARG, "gp_debug_potrf_inverse: bad argument");
  GP_HIP(c, hipSetDevice(device));
  const int Mp = (int)round_up(n, NB);
  const long mm = (long)Mp * Mp;
  std::vector<double> h(mm, 0.0);
  for (int i = 0; i < Mp; ++i) for (int k = 0; k < Mp; ++k) h[(long)i * Mp + k] = (i < n && k < n) ? A[(long)i * n + k] : (i == k ? 1.0 : 0.0);
  double *dA, *dLi, *dInv, *dT, *dS;
  GP_HIP(c, hipMalloc((void**)&dA, mm * 8)); GP_HIP(c, hipMalloc((void**)&dLi, mm * 8)); GP_HIP(c, hipMalloc((void**)&dInv, mm * 8));
  GP_HIP(c, hipMalloc((void**)&dT, mm * 8)); GP_HIP(c, hipMalloc((void**)&dS, 64));
  GP_HIP(c, hipMemcpy(dA, h.data(), mm * 8, hipMemcpyHostToDevice));
  GP_HIP(c, hipMemset(dS, 0, 64));
  GP_HIP(c, hipMemset(dLi, 0, mm * 8));
  int rc = potrf_inverse_batched(c, nullptr, Mp, 1, dA, dLi, dInv, dT, dS, dS + 1, nullptr);
  if (rc == GP_OK) {
    double s[2];
    GP_HIP(c, hipDeviceSynchronize());
    GP_HIP(c, hipMemcpy(s, dS, 16, hipMemcpyDeviceToHost));
    if (logdet) *logdet = s[0];
    if (L) { GP_HIP(c, hipMemcpy(h.data(), dA, mm * 8, hipMemcpyDeviceToHost)); for (int i = 0; i < n; ++i) for (int k = 0; k < n; ++k) L[(long)i * n + k] = (k <= i) ? h[(long)i * Mp + k] : 0.0; }
    if (Ainv) { GP_HIP(c, hipMemcpy(h.data(), dInv, mm * 8, hipMemcpyDeviceToHost)); for (int i = 0; i < n; ++i) for (int k = 0; k < n; ++k) Ainv[(long)i * n + k] = h[(long)i * Mp + k]; }
    if (s[1] != 0.0) rc = fail(nullptr, GP_ERR_NOT_PD, "matrix is not positive definite");
  } else {
    gp::g_create_error = c->err;
  }
  (void)hipFree(dA); (void)hipFree(dLi); (void)hipFree(dInv); (void)hipFree(dT); (void)hipFree(dS);
  return rc;
}

// raw copy of an internal buffer of the global step (developer tool, tests/devtools/dev_tail_diff.py; not part of the public header)
extern "C" int gp_debug_peek(gp_ctx* c, const char* name, double* out, long n) {
  using namespace gp;
  if (!c || !name || !out) return GP_ERR_BAD_ARG;
  GP_HIP(c, hipSetDevice(c->device));
  const long mm = (long)c->Mp * c->Mp, md = (long)c->Mp * c->Dp;
  const double* src = nullptr; long cnt = 0;
  if (!std::strcmp(name, "Linv")) { src = c->Linv; cnt = 2 * mm; }
  else if (!std::strcmp(name, "Inv")) { src = c->Inv; cnt = 2 * mm; }
  else if (!std::strcmp(name, "E")) { src = c->E; cnt = md; }
  else if (!std::strcmp(name, "PsiE")) { src = c->PsiE; cnt = md; }
  else if (!std::strcmp(name, "T1")) { src = c->T1; cnt = mm; }
  else if (!std::strcmp(name, "T2")) { src = c->T2; cnt = mm; }
  else if (!std::strcmp(name, "dFdK")) { src = c->dFdK; cnt = mm; }
  else if (!std::strcmp(name, "Bbar")) { src = c->Bbar; cnt = mm; }
  else if (!std::strcmp(name, "Abar")) { src = c->Abar; cnt = md; }
  else if (!std::strcmp(name, "Bm")) { src = c->Bm; cnt = (long)c->LDK * c->Mp; }
  else if (!std::strcmp(name, "gK")) { src = c->gK; cnt = (long)c->M * c->Q + c->Q; }
  else if (!std::strcmp(name, "gs")) { src = c->gs; cnt = GS_COUNT + 8 + 8 * 64; }
  // r06 (poison probe, tests/devtools/dev_poison_probe.py): the padded device images of the evaluation's other buffers
  else if (!std::strcmp(name, "stats")) { src = c->stats; cnt = mm + md + SC_COUNT; }
  else if (!std::strcmp(name, "Kaug")) { src = c->Kaug; cnt = (long)c->Np * c->LDK; }
  else if (!std::strcmp(name, "Kmm")) { src = c->Kmm; cnt = 2 * mm; }
  else if (!std::strcmp(name, "KmmKeep")) { src = c->KmmKeep; cnt = mm; }
  else if (!std::strcmp(name, "Z")) { src = c->Z; cnt = (long)c->Mp * c->Q; }
  else if (!std::strcmp(name, "Zaug")) { src = c->Zaug; cnt = (long)c->Mp * c->CZp; }
  else if (!std::strcmp(name, "mu")) { src = c->mu; cnt = (long)c->Np * c->Q; }
  else if (!std::strcmp(name, "S")) { src = c->S; cnt = (long)c->Np * c->Q; }
  else if (!std::strcmp(name, "Xa")) { src = c->Xa; cnt = (long)c->Np * c->CXp; }
  else if (!std::strcmp(name, "grads")) { src = c->grads; cnt = (long)c->M * c->Q + c->Q; }
  else if (!std::strcmp(name, "Rpart")) { src = c->Rpart; cnt = (long)2 * (c->p2_slices + 8) * c->Mp * c->CXp; }
  else if (!std::strcmp(name, "LE")) { src = c->LE; cnt = c->LE ? (long)c->Np * c->Mp : 0; }
  else if (!std::strcmp(name, "LEA")) { src = c->LET; cnt = c->LET ? (long)c->Np * c->Mp : 0; }
  else return fail(c, GP_ERR_BAD_ARG, "gp_debug_peek: unknown buffer '%s'", name);
  if (n < cnt) return fail(c, GP_ERR_BAD_ARG, "gp_debug_peek: %ld doubles needed", cnt);
  GP_HIP(c, hipStreamSynchronize(c->stream));
  GP_HIP(c, hipMemcpy(out, src, cnt * 8, hipMemcpyDeviceToHost));
  return (int)GP_OK;
}
