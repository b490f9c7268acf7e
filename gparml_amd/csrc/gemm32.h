// The 32 x 32-tile FP64 product of the global step as a device function: gemm32_kernel (gemm.hip) runs one tile per workgroup; the fused
// small-M tail of the global step (linalg.hip: gs_tail128_kernel) walks the same tiles from a persistent grid.  One body, so both give the
// same bits.
#pragma once
#include "gp_common.h"

namespace gp {

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

// one 32 x 32 output tile (bx, by) of batch entry bz by the calling 256-thread workgroup; sA / sB: LDS images (SmallImg<L>::DOUBLES doubles, 16-byte
// aligned).  Ends with the tile stored; the caller synchronises before the images are reused.
template <Layout LA, Layout LB>
__device__ __forceinline__ void gemm32_tile(const GemmP& p, int bx, int by, int bz, double* sA, double* sB) {
  if (p.tri == 1 && bx > by) return;     // 32-tile granularity: every tile that touches the lower triangle is computed
  if (p.tri == 2 && bx < by) return;
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
  const int c0 = p.klow ? (int)((row0 > col0 ? row0 : col0) / SKC) : 0;     // first chunk with a possibly non-zero product
  // NOT zeroed in C++: hipcc rematerialises such zeros as v_mov_b64 right in front of the first asm MFMA of each accumulator, and in the fused
  // tail kernel it put one of them on a register the previous MFMA was still reading as its B operand (columns 4..7 of every tile wrong, r05).
  // The first k-step of the first chunk takes C = 0 as an inline constant instead (mfma444_zero).
  double acc[4];
  const unsigned aA = lds_byte_addr(sA) + 8u * (unsigned)(LA == K_CONTIG ? (wr + lr) * LDA + lk : lk * LDA + wr + lr);
  const unsigned aB = lds_byte_addr(sB) + 8u * (unsigned)(LB == K_CONTIG ? (wc + lj) * LDB + lk : lk * LDB + wc + lj);
  double2 ra[8], rb[8];
  small_load<LA>(Ab + (long)c0 * a_step, p.lda, p.K - c0 * SKC, tid, ra);
  small_load<LB>(Bb + (long)c0 * b_step, p.ldb, p.K - c0 * SKC, tid, rb);
  for (int c = c0; c < nc; ++c) {
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
      if (k4 == 0 && c == c0) {
#pragma unroll
        for (int bc = 0; bc < 4; ++bc) mfma444_zero(acc[bc], a[cur], b[cur][bc]);
      } else {
#pragma unroll
        for (int bc = 0; bc < 4; ++bc) mfma444_acc(acc[bc], a[cur], b[cur][bc]);
      }
    });
    __syncthreads();
  }
  mfma_drain(acc[3]);
  acc_fence(acc);
  const long r = row0 + wr + 4 * ((lane >> 2) & 3) + (lane >> 4);
#pragma unroll
  for (int bc = 0; bc < 4; ++bc) {
    const long cc = col0 + wc + 4 * bc + lj;
    const double v = p.alpha * acc[bc] + (p.beta != 0.0 ? p.beta * C[r * p.ldc + cc] : 0.0);
    C[r * p.ldc + cc] = v;
    if (p.mirror && bx != by) C[cc * p.ldc + r] = v;
  }
}

}  // namespace gp
