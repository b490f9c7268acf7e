// Generic FP64 GEMM on the 4x4x4 matrix-core instruction (see mma_f64.h): 128x128 tile per 256-thread
// workgroup, 16-deep k-chunks double-buffered through LDS, two workgroups per CU.
// Used by the global step (M x M algebra) and, through the same building blocks, by the phase kernels.
#include "gp_common.h"
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
  const int nc = p.K / KC / p.splits;          // chunks of this split
  const long k0 = (long)sp * nc * KC;
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
      }
  }
}

// Small-tile variant for the latency-bound products of the global step (M x M x {M, D, 128} with M a few hundred).  One CU
// delivers 0.29 TFLOP/s of FP64 MFMA, so a 128 x 128 x 128 tile product takes 15 us on the CU that owns it whatever the
// kernel does; those products have 1 to 16 such tiles and left 94 % of the chip idle (eight-wave 128-tile kernel: 21 us per
// launch).  Here a workgroup owns a 32 x 32 output tile (four waves, 16 x 16 each: one A and four B operand registers, four
// accumulators) and walks K in chunks of 128 staged through LDS from registers (one global round trip per chunk, the next
// chunk's loads in flight during the MFMAs), so a 128^3 product is sixteen workgroups of 0.9 us MFMA time each.
//   LDS image of an operand chunk: K_CONTIG  [32 free][128 k], row stride 130  (16 rows x {k, k+1} cover 32 distinct 8-byte slots)
//                                  FREE_CONTIG [128 k][32 free], row stride 48 (k and k+1 sit 16 slots apart)
// Operand reads are explicit ds_read_b64 with counted lgkmcnt waits (mma_f64.h).

constexpr int ST = 32;     // workgroup tile of the small kernel
constexpr int SKC = 128;   // its k-chunk
template <Layout L> struct SmallImg {
  static constexpr int LD = (L == K_CONTIG) ? SKC + 2 : ST + 16;
  static constexpr int DOUBLES = (L == K_CONTIG) ? ST * LD : SKC * LD;
};

// this thread's 8 x 16 bytes of a [32 x 128] operand chunk; kleft = K - k0 (a multiple of 16; elements beyond it read as zero)
template <Layout L>
__device__ __forceinline__ void small_load(const double* __restrict__ src, long ld, int kleft, int tid, double2 (&r)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = tid + 256 * i;
    const int k = (L == K_CONTIG) ? 2 * (e & 63) : (e >> 4);
    const double* g = (L == K_CONTIG) ? src + (long)(e >> 6) * ld + k : src + (long)k * ld + 2 * (e & 15);
    r[i] = (k < kleft) ? *reinterpret_cast<const double2*>(g) : make_double2(0.0, 0.0);
  }
}
template <Layout L>
__device__ __forceinline__ void small_store(double* img, int tid, const double2 (&r)[8]) {
  constexpr int LD = SmallImg<L>::LD;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = tid + 256 * i;
    double* d = (L == K_CONTIG) ? img + (e >> 6) * LD + 2 * (e & 63) : img + (e >> 4) * LD + 2 * (e & 15);
    *reinterpret_cast<double2*>(d) = r[i];
  }
}

template <Layout LA, Layout LB>
__global__ void __launch_bounds__(256, 1) gemm32_kernel(GemmP p) {
  const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.tri == 1 && bx > by) return;     // 32-tile granularity: every tile that touches the lower triangle is computed
  if (p.tri == 2 && bx < by) return;
  __shared__ __attribute__((aligned(16))) double sA[SmallImg<LA>::DOUBLES];
  __shared__ __attribute__((aligned(16))) double sB[SmallImg<LB>::DOUBLES];
  constexpr int LDA = SmallImg<LA>::LD, LDB = SmallImg<LB>::LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = 16 * (wave >> 1), wc = 16 * (wave & 1);
  const int lr = lane & 15, lk = lane >> 4, lj = lane & 3;
  const int bi = bz % p.inner, bo = bz / p.inner;
  const double* A = p.A + (long)bi * p.sA + (long)bo * p.oA;
  const double* B = p.B + (long)bi * p.sB + (long)bo * p.oB;
  double* C = p.C + (long)bi * p.sC + (long)bo * p.oC;
  const long row0 = (long)by * ST, col0 = (long)bx * ST;
  const double* Ab = (LA == K_CONTIG) ? A + row0 * p.lda : A + row0;
  const double* Bb = (LB == K_CONTIG) ? B + col0 * p.ldb : B + col0;
  const long a_step = (LA == K_CONTIG) ? SKC : (long)SKC * p.lda;
  const long b_step = (LB == K_CONTIG) ? SKC : (long)SKC * p.ldb;
  const int nc = (p.K + SKC - 1) / SKC;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const unsigned aA = lds_byte_addr(sA) + 8u * (unsigned)(LA == K_CONTIG ? (wr + lr) * LDA + lk : lk * LDA + wr + lr);
  const unsigned aB = lds_byte_addr(sB) + 8u * (unsigned)(LB == K_CONTIG ? (wc + lj) * LDB + lk : lk * LDB + wc + lj);
  double2 ra[8], rb[8];
  small_load<LA>(Ab, p.lda, p.K, tid, ra);
  small_load<LB>(Bb, p.ldb, p.K, tid, rb);
  for (int c = 0; c < nc; ++c) {
    small_store<LA>(sA, tid, ra);
    small_store<LB>(sB, tid, rb);
    __syncthreads();
    if (c + 1 < nc) {
      small_load<LA>(Ab + (long)(c + 1) * a_step, p.lda, p.K - (c + 1) * SKC, tid, ra);
      small_load<LB>(Bb + (long)(c + 1) * b_step, p.ldb, p.K - (c + 1) * SKC, tid, rb);
    }
    double a[2], b[2][4];
    auto rd = [&](auto kc, double& av, double (&bv)[4]) {
      constexpr int k4 = decltype(kc)::value;
      av = ds_read64<(LA == K_CONTIG ? 4 * k4 : 4 * k4 * LDA) * 8>(aA);
      static_for<0, 4>([&](auto jc) {
        constexpr int bc = decltype(jc)::value;
        bv[bc] = ds_read64<(LB == K_CONTIG ? 4 * bc * LDB + 4 * k4 : 4 * k4 * LDB + 4 * bc) * 8>(aB);
      });
    };
    rd(IC<0>{}, a[0], b[0]);
    static_for<0, SKC / 4>([&](auto kc) {
      constexpr int k4 = decltype(kc)::value, cur = k4 & 1;
      if constexpr (k4 + 1 < SKC / 4) { rd(IC<k4 + 1>{}, a[cur ^ 1], b[cur ^ 1]); lgkm_wait<5>(); }
      else lgkm_wait<0>();
#pragma unroll
      for (int bc = 0; bc < 4; ++bc) mfma444_acc(acc[bc], a[cur], b[cur][bc]);
    });
    __syncthreads();
  }
  mfma_drain(acc[3]);
  acc_fence(acc);
  const long r = row0 + wr + 4 * ((lane >> 2) & 3) + (lane >> 4);
#pragma unroll
  for (int bc = 0; bc < 4; ++bc) {
    const long cc = col0 + wc + 4 * bc + lj;
    C[r * p.ldc + cc] = p.alpha * acc[bc] + (p.beta != 0.0 ? p.beta * C[r * p.ldc + cc] : 0.0);
  }
}

__global__ void __launch_bounds__(256) gemm_splitk_reduce_kernel(GemmP p, int tiles_x, int tiles_y) {
  // grid (tiles * 16, batch): 16 blocks of 256 threads per 128x128 tile, 4 elements each
  const int tile = blockIdx.x >> 4, sub = blockIdx.x & 15;
  const int bx = tile % tiles_x, by = tile / tiles_x, bz = blockIdx.y;
  if (p.tri == 1 && bx > by) return;
  if (p.tri == 2 && bx < by) return;
  const int bi = bz % p.inner, bo = bz / p.inner;
  double* C = p.C + (long)bi * p.sC + (long)bo * p.oC;
  const double* w = p.ws + (((long)bz * tiles_y + by) * tiles_x + bx) * p.splits * (TILE * TILE);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = sub * 1024 + i * 256 + threadIdx.x;
    double s = 0.0;
    for (int sp = 0; sp < p.splits; ++sp) s += w[(long)sp * (TILE * TILE) + e];
    const long r = (long)by * TILE + (e >> 7), cc = (long)bx * TILE + (e & 127);
    C[r * p.ldc + cc] = p.alpha * s + (p.beta != 0.0 ? p.beta * C[r * p.ldc + cc] : 0.0);
  }
}

void launch_gemm(hipStream_t st, Layout la, Layout lb, int m, int n, int batch, const GemmP& p) {
  dim3 grid(n / TILE, m / TILE, batch * p.splits), block(256);
  if ((long)(n / TILE) * (m / TILE) * batch <= 256) {
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
