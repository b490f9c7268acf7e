// Phase 2 of the fixed-embedding path on the INT8 matrix core (round-4 review item 3(c); opt-in next to the int8 phase 1: GPARML_P2_I8=1 or
// gp_debug_set_option("p2_i8", 1), only together with p1_i8, fixed embeddings, Q <= 10).
//   G = [K | Y] [2 Bbar ; Abar^T],  W = G o K,  R = W^T [mu | 1 | mu^2]        (partial_terms.py:146-160, 207-240, 286-333 contracted)
// K . Bbar cancels ten digits (profiles/r04_ozaki_gate.txt), so both operands carry SEVEN signed 7-bit digits (49 bits below their scale) and the
// 28 digit products with a + b <= 8 are kept: exact integer matrix products on v_mfma_i32_32x32x32_i8, products of equal order in one int32
// accumulator set (7 sets; |d e| <= 4096, <= 7 pairs per order, K = M + D <= 4096 columns: 1.2e8 < 2^31).  Exact CPU emulation of this
// arithmetic: grad_Z 4.7e-8 from the 80-bit truth at N = 1e5 (tests/devtools/dev_ozaki_gate_phase2.py), float64 G: 1.7e-8.
//
// Operands.  A = the rows of [K | Y] scaled to |t| <= 1/2 (K / (2 sf2); Y / its column scale), digits in HBM as
//     SlK[digit][n / 128][k / 32][16-k half][n % 128][16 B]
// -- the contraction runs over the COLUMNS here, so a lane's 16-byte operand is 16 consecutive columns of one row (phase 1 contracts over rows and
// keeps 16 consecutive rows per operand: psi1_kernel writes both layouts from one digit extraction, Y's digits are written once per upload).  One
// k-step (32 columns) of a 128-row tile is 2 x 2 KB contiguous per digit: the LDS image is the HBM image.  The column scale of A (2 sf2, or
// Y's) differs per k, so it is folded into B: B'[k][m] = ksc[k] Bm[k][m], per-column scale bscale[m] = 2^e >= 2 max_k |B'[k][m]|, digits as
//     SlB[digit][k / 32][16-k half][m][16 B]        (i8_slice_b_kernel, after every global step: 2.3 MB at M = 512, D = 100).
// G[n][m] = bscale[m] sum_{a + b <= 8} (dA_a dB_b)[n][m] 128^-(a + b).
//
// Kernel.  Workgroup = 128 rows x 64 inducing columns, eight waves as 4 x 2 of 32 x 32 (one MFMA tile: 7 x 16 accumulator registers, two waves per
// SIMD so that a partner covers every LDS-DMA stall, tools/ubench/dma_wave_ubench.hip); it walks the 128-row tiles of its slice, all eight column
// blocks of a slice on one XCD (the slice's digits are fetched from HBM once).  Per k-step 42 KB of operand panels arrive by LDS-DMA through a
// three-stage ring that runs across tile boundaries.  Epilogue per tile, float64 on the VALU (1.1e10 FMAs per evaluation: 0.3 ms of the chip):
// accumulators -> G, W = G o K with K read from Kaug (float64), R += W^T [mu | 1 | mu^2] with the lane's column of R in registers for the whole slice
// (the tile's feature rows are staged in LDS and read as broadcasts).  R leaves as eight partials per slice (wave row x half wave), summed by
// p2_reduce_kernel in its [mu | 1 | mu^2] form.
#include "gp_common.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>

namespace gp {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
std::atomic<int> g_opt_p2_i8{[] { const char* e = getenv("GPARML_P2_I8"); return (e && e[0] == '1') ? 1 : 0; }()};

constexpr int P2S = 7;                        // digits per operand
constexpr int P2O = 7;                        // accumulator sets: orders a + b = 2 .. 8 (digits numbered from 1)
constexpr int P2_QP = 10;                     // latent dimensions the epilogue is compiled for (features [mu (QP) | 1 | mu^2 (QP)])
constexpr int P2_NF = 2 * P2_QP + 1;

// 7 balanced base-128 digits of t in [-1/2, 1/2] (see psi1_kernel): the 7-bit fields f_k = d_k + (64 - s) of I + C - s T, I = rint(t 2^49), bits
// [7k, 7k + 7) for k < 6 and the leading digit in bits 42 and up; s in {0, 1} from a hash of the element's position (digits in [-64, 63] or [-63, 64]:
// mean zero over the elements, so neither the truncation nor the dropped digit products carry a bias)
struct DigitFields { unsigned long long f; unsigned off; };
__device__ __forceinline__ unsigned digit_hash(unsigned r, unsigned c) { return ((r * 0x9E3779B1u + c * 0x85EBCA6Bu) * 0xC2B2AE35u) >> 31; }
__device__ __forceinline__ DigitFields digit_fields(double t, unsigned s) {
  constexpr unsigned long long DIGC = 64ull * ((1ull << 49) - 1ull) / 127ull, DIGT = ((1ull << 49) - 1ull) / 127ull;
  const long long I = (long long)((unsigned long long)__double_as_longlong(t + 12.0) & ((1ull << 52) - 1ull)) - (1ll << 51);     // signed: |I| <= 2^48
  DigitFields d;
  d.f = (unsigned long long)(I + (long long)(s ? DIGC - DIGT : DIGC));
  d.off = 64u - s;
  return d;
}
__device__ __forceinline__ unsigned digit_byte(const DigitFields& d, int j /* 0 = most significant */) {
  const int k = 6 - j;
  const unsigned v = (k == 6) ? (unsigned)(d.f >> 42) : ((unsigned)(d.f >> (7 * k)) & 127u);
  return (v - d.off) & 0xffu;
}

// ---- digits of Y in the column-contiguous layout (once per upload) and of B' (after every global step) ---------------------------------------
__global__ void __launch_bounds__(256) i8_slice_y2_kernel(const double* __restrict__ Kaug, long ld, long Np, int Mp, int Dp, int KS2,
                                                          const double* __restrict__ yscale, int8_t* __restrict__ SlK, long strideK) {
  // thread = (row n, 16-column block cb of Y)
  const int nblk = Dp / 16;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < Np * nblk; i += (long)gridDim.x * 256L) {
    const long n = i / nblk;
    const int cb = (int)(i - n * nblk);
    unsigned pk[P2S][4];
#pragma unroll
    for (int j = 0; j < P2S; ++j) pk[j][0] = pk[j][1] = pk[j][2] = pk[j][3] = 0u;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int d = cb * 16 + u;
      const DigitFields f = digit_fields(Kaug[n * ld + Mp + d] / yscale[d], digit_hash((unsigned)n, (unsigned)(Mp + d)));
#pragma unroll
      for (int j = 0; j < P2S; ++j) pk[j][u >> 2] |= digit_byte(f, j) << (8 * (u & 3));
    }
    const int cbi = Mp / 16 + cb;
    int8_t* dst = SlK + ((((n >> 7) * KS2 + (cbi >> 1)) * 2 + (cbi & 1)) * 128 + (n & 127)) * 16;
#pragma unroll
    for (int j = 0; j < P2S; ++j) { uint4 v = {pk[j][0], pk[j][1], pk[j][2], pk[j][3]}; *(uint4*)(dst + (long)j * strideK) = v; }
  }
}
__global__ void __launch_bounds__(256) i8_bscale_kernel(const double* __restrict__ Bm, int LDK, int Mp, double ksc_k, const double* __restrict__ yscale,
                                                        double* __restrict__ bscale) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= Mp) return;
  double mx = 0.0;
  for (int k = 0; k < LDK; ++k) mx = fmax(mx, fabs(Bm[(long)k * Mp + m]) * (k < Mp ? ksc_k : yscale[k - Mp]));
  int e = 0;
  if (mx > 0.0) { (void)frexp(mx, &e); e += 1; }          // mx = f 2^e', f in [0.5, 1)  ->  |b'| 2^-(e'+1) <= 1/2
  bscale[m] = ldexp(1.0, e);
}
__global__ void __launch_bounds__(256) i8_slice_b_kernel(const double* __restrict__ Bm, int LDK, int Mp, double ksc_k, const double* __restrict__ yscale,
                                                         const double* __restrict__ bscale, int8_t* __restrict__ SlB, long strideB) {
  // thread = (16-k block kb, column m): sixteen k of one column -> one 16-byte operand per digit
  const long total = (long)(LDK / 16) * Mp;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int kb = (int)(i / Mp), m = (int)(i - (long)kb * Mp);
    const double inv = 1.0 / bscale[m];
    unsigned pk[P2S][4];
#pragma unroll
    for (int j = 0; j < P2S; ++j) pk[j][0] = pk[j][1] = pk[j][2] = pk[j][3] = 0u;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = kb * 16 + u;
      const DigitFields f = digit_fields(Bm[(long)k * Mp + m] * (k < Mp ? ksc_k : yscale[k - Mp]) * inv, digit_hash((unsigned)k, (unsigned)m));
#pragma unroll
      for (int j = 0; j < P2S; ++j) pk[j][u >> 2] |= digit_byte(f, j) << (8 * (u & 3));
    }
    int8_t* dst = SlB + ((long)kb * Mp + m) * 16;          // [k / 32][half] = kb
#pragma unroll
    for (int j = 0; j < P2S; ++j) { uint4 v = {pk[j][0], pk[j][1], pk[j][2], pk[j][3]}; *(uint4*)(dst + (long)j * strideB) = v; }
  }
}

struct P2I8Args {
  const int8_t* SlK; long strideK; int KS2;            // digits of [K | Y]; KS2 = LDK / 32 k-steps per 128-row tile in the layout
  const int8_t* SlB; long strideB; const double* bscale;
  const double* Kaug; long ld; const double* mu;        // float64 Psi1 (the Hadamard factor) and the points' means [Np][Q]
  double* Rpart; int Mp, CXp, Q, MT2, S, tps, ntiles, KS;   // ntiles: 128-row tiles; KS: k-steps that hold real columns (Psi1's Mp / 32 + Y's ceil(D / 32))
};

// Workgroup = 128 rows x 64 inducing columns, eight waves as 4 x 2 of 32 x 32, one workgroup per CU (126 KB of operand ring + 21 KB of features).
// (A 64 x 64 form with four waves and two workgroups per CU -- so that one workgroup's epilogue runs under the other's MFMAs -- was tried: its
// two-stage ring does not cover the HBM latency of the A panel, 7 DMA slots + the feature prefetch spilled 37 registers: 11.3 ms against 9.8.)
constexpr int P2_TN = 128;
constexpr int P2_APANEL = P2S * 2 * P2_TN * 16;   // bytes of one k-step of the A panel: [digit][half][128 rows][16 B]
constexpr int P2_BPANEL = P2S * 2 * 64 * 16;      //                       B panel: [digit][half][64 columns][16 B]
constexpr int P2_STAGE = P2_APANEL + P2_BPANEL;
constexpr int P2_STAGES = 3;
constexpr int P2_LDS = P2_STAGES * P2_STAGE + P2_TN * P2_NF * 8;

__global__ void __launch_bounds__(512, 1) p2i8_kernel(P2I8Args a) {
  extern __shared__ __attribute__((aligned(16))) int8_t p2lds[];      // [3 stages][A panel | B panel] | feature tile [128][NF] doubles
  const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int slice = xcd + 8 * (bi / a.MT2), mt = bi % a.MT2;
  if (slice >= a.S) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;                            // 4 x 2 waves of 32 x 32
  const int kg = lane >> 5, r32 = lane & 31;
  double* xs = reinterpret_cast<double*>(p2lds + P2_STAGES * P2_STAGE);
  const int t0 = slice * a.tps, t1 = min(a.ntiles, t0 + a.tps);
  const long nsteps = (long)(t1 - t0) * a.KS;
  // LDS-DMA: 42 instructions of 1 KB per k-step (A: 7 digits x 2 halves x 2 x 64 rows; B: 7 x 2 x 64 columns), six slots per wave
  constexpr int NSLOT = 6, NE = 4 * P2S + 2 * P2S;
  const int8_t* dsrc[NSLOT];
  int ddst[NSLOT];
  bool isA[NSLOT];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    int e = wave + 8 * s;
    if (e >= NE) e = NE - 1;                                          // padding slots repeat the last transfer
    isA[s] = e < 4 * P2S;
    if (isA[s]) {
      const int j = e >> 2, h = (e >> 1) & 1, half = e & 1;
      dsrc[s] = a.SlK + (long)j * a.strideK + (((long)t0 * a.KS2 * 2 + h) * 128 + 64 * half + lane) * 16;
      ddst[s] = ((j * 2 + h) * 128 + 64 * half) * 16;                 // the instruction writes lane i at base + 16 i
    } else {
      const int f = e - 4 * P2S, j = f >> 1, h = f & 1;
      dsrc[s] = a.SlB + (long)j * a.strideB + ((long)h * a.Mp + mt * 64 + lane) * 16;
      ddst[s] = P2_APANEL + ((j * 2 + h) * 64) * 16;
    }
  }
  const long stepA = 2L * 128 * 16, wrapA = (long)(a.KS2 - a.KS + 1) * stepA;       // next k-step of the tile / first k-step of the next tile
  const long stepB = 2L * a.Mp * 16, wrapB = -(long)(a.KS - 1) * stepB;
  int ks_issue = 0;                                                   // k-step (within its tile) of the next transfer to be issued
  auto dma = [&](int stage) {
    const bool last = ks_issue + 1 == a.KS;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
#if !defined(P2I8_ABLATE) || !(P2I8_ABLATE & 2)      // ablation builds (tests/devtools/dev_p2i8.py; wrong results by construction): 2 = no operand staging
      __builtin_amdgcn_global_load_lds((gbl_void*)dsrc[s], (lds_void*)(p2lds + stage * P2_STAGE + ddst[s]), 16, 0, 0);
#endif
      dsrc[s] += isA[s] ? (last ? wrapA : stepA) : (last ? wrapB : stepB);
    }
    ks_issue = last ? 0 : ks_issue + 1;
  };
  v16i acc[P2O];
#pragma unroll
  for (int o = 0; o < P2O; ++o)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[o][i] = 0;
  double rr[P2_NF];
#pragma unroll
  for (int c = 0; c < P2_NF; ++c) rr[c] = 0.0;
  const int mcol = mt * 64 + wc * 32 + r32;
  const double bs = a.bscale[mcol];
  if (nsteps > 0) dma(0);
  if (nsteps > 1) dma(1);
  int ks = 0, nt = t0;
  for (long it = 0; it < nsteps; ++it) {
    const int stage = (int)(it % P2_STAGES);
    if (it + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSLOT) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage `it` has landed for every wave; stage it + 2 is no longer being read
    const bool early = wave < 4;                                      // the two waves of a SIMD (w, w + 4) stage at different times
    if (early && it + 2 < nsteps) dma((int)((it + 2) % P2_STAGES));
    if (ks == 0) {
      // this tile's feature rows [mu | 1 | mu^2] (behind the barrier: every wave has finished the previous tile's epilogue)
      const long n0 = (long)nt * P2_TN;
      for (int e = tid; e < P2_TN * P2_NF; e += 512) {
        const int r = e / P2_NF, c = e - r * P2_NF;
        const int q = c < P2_QP ? c : c - P2_QP - 1;
        double v = 1.0;
        if (c != P2_QP) { const double m = q < a.Q ? a.mu[(n0 + r) * a.Q + q] : 0.0; v = c < P2_QP ? m : m * m; }
        xs[e] = v;
      }
    }
    const int8_t* pa = p2lds + stage * P2_STAGE;
    const int8_t* pb = pa + P2_APANEL;
    v4i bv[P2S], av[2];
#if defined(P2I8_ABLATE) && (P2I8_ABLATE & 4)          // 4 = no operand reads from LDS (register constants)
#pragma unroll
    for (int b = 0; b < P2S; ++b) bv[b] = v4i{lane + b, 1, 2, (int)it};
    av[0] = v4i{lane, 3, (int)it, 5}; av[1] = v4i{lane + 1, 3, (int)it, 7};
    (void)pa; (void)pb;
#pragma unroll
    for (int d = 0; d < P2S; ++d) {
#pragma unroll
      for (int b = 0; b < P2S; ++b)
        if (d + b < P2O) acc[d + b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[d & 1], bv[b], acc[d + b], 0, 0, 0);
    }
#else
#pragma unroll
    for (int b = 0; b < P2S; ++b) bv[b] = *(const v4i*)(pb + ((b * 2 + kg) * 64 + wc * 32 + r32) * 16);
    av[0] = *(const v4i*)(pa + ((0 * 2 + kg) * P2_TN + wr * 32 + r32) * 16);
#pragma unroll
    for (int d = 0; d < P2S; ++d) {
      if (d == 2 && !early && it + 2 < nsteps) dma((int)((it + 2) % P2_STAGES));
      // the next digit's A operand is requested BEFORE this digit's MFMAs (the scheduling barrier keeps hipcc from sinking the read behind them:
      // left alone it issued read, wait, MFMAs per digit -- one LDS round trip of idle matrix core per digit, 1.2 of 6.3 ms in the ablated loop)
      if (d + 1 < P2S) av[(d + 1) & 1] = *(const v4i*)(pa + (((d + 1) * 2 + kg) * P2_TN + wr * 32 + r32) * 16);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < P2S; ++b)
        if (d + b < P2O) acc[d + b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[d & 1], bv[b], acc[d + b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
#if defined(P2I8_ABLATE) && (P2I8_ABLATE & 1)          // 1 = no epilogue (the accumulators are folded into one number at the very end)
    if (++ks == a.KS) { ks = 0; ++nt; }
    if (false) {
#else
    if (++ks == a.KS) {
#endif
      // ---- epilogue of tile nt: G, W = G o K, R += W^T [mu | 1 | mu^2]; C/D map of the 32 x 32 MFMA: column = lane & 31,
      // row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  The feature tile was written at the tile's first k-step (KS >= 2: a barrier lies between)
      const long n0 = (long)nt * P2_TN + wr * 32;
      const double* kcol = a.Kaug + n0 * a.ld + mcol;
      // two batches of eight rows: eight loads of K in flight, not sixteen (the accumulators, the lane's column of R and the operand registers
      // leave ~20 registers for the epilogue)
#pragma unroll
      for (int h8 = 0; h8 < 2; ++h8) {
        double kv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = 8 * h8 + u; kv[u] = kcol[(long)((i & 3) + 8 * (i >> 2) + 4 * kg) * a.ld]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = 8 * h8 + u;
          double g = 0.0, w = 1.0 / (128.0 * 128.0);
          double ws[P2O];
#pragma unroll
          for (int o = 0; o < P2O; ++o) { ws[o] = w; w *= 1.0 / 128.0; }
#pragma unroll
          for (int o = P2O - 1; o >= 0; --o) g = fma((double)acc[o][i], ws[o], g);     // small terms first
          const double wv = g * bs * kv[u];
          const double* xrow = xs + (wr * 32 + (i & 3) + 8 * (i >> 2) + 4 * kg) * P2_NF;
#pragma unroll
          for (int c = 0; c < P2_NF; ++c) rr[c] = fma(wv, xrow[c], rr[c]);
        }
      }
#pragma unroll
      for (int o = 0; o < P2O; ++o)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[o][i] = 0;
      ks = 0; ++nt;
    }
  }
#if defined(P2I8_ABLATE) && (P2I8_ABLATE & 1)
#pragma unroll
  for (int o = 0; o < P2O; ++o)
#pragma unroll
    for (int i = 0; i < 16; ++i) rr[0] += (double)acc[o][i];
#endif
  // eight partials per slice: (wave row, half wave); columns [mu (Q) | 1 | mu^2 (Q)] of p2_reduce_kernel's fixedA = 2 form
  double* out = a.Rpart + ((long)(slice * 8 + wr * 2 + kg) * a.Mp + mcol) * a.CXp;
#pragma unroll
  for (int q = 0; q < P2_QP; ++q)                        // static indices: a run-time index would put the accumulators in scratch
    if (q < a.Q) { out[q] = rr[q]; out[a.Q + 1 + q] = rr[P2_QP + 1 + q]; }
  out[a.Q] = rr[P2_QP];
}

// ---- host -----------------------------------------------------------------------------------------------------------------------------------
struct P2I8Plan {
  int8_t* SlK = nullptr; long strideK = 0; int KS2 = 0;
  int8_t* SlB = nullptr; long strideB = 0;
  double* bscale = nullptr;
  bool y_valid = false;
};

bool p2i8_wanted(const gp_ctx* c) {
  return g_opt_p2_i8.load() != 0 && c->Q <= P2_QP && c->LDK <= 4096 && !c->p2i8_unsupported;
}

// the second digit layout for psi1_kernel (allocated on first use); Y's digits are written when the shard changed
int p2i8_prepare(gp_ctx* c, const double* yscale, int8_t** SlK, long* strideK, int* KS2) {
  P2I8Plan* pl = static_cast<P2I8Plan*>(c->p2i8plan);
  if (!pl) {
    pl = new P2I8Plan();
    pl->KS2 = c->LDK / 32;
    pl->strideK = c->Np * (long)c->LDK;                     // bytes per digit: one byte per element
    pl->strideB = (long)c->LDK * c->Mp;
    hipError_t e = hipMalloc((void**)&pl->SlK, (size_t)P2S * pl->strideK);
    if (e == hipSuccess) e = hipMalloc((void**)&pl->SlB, (size_t)P2S * pl->strideB);
    if (e == hipSuccess) e = hipMalloc((void**)&pl->bscale, (size_t)c->Mp * sizeof(double));
    if (e != hipSuccess) {
      for (void* p : {(void*)pl->SlK, (void*)pl->SlB, (void*)pl->bscale}) if (p) (void)hipFree(p);
      delete pl;
      c->p2i8_unsupported = true;
      return fail(c, GP_ERR_UNSUPPORTED, "int8 phase 2: digit buffers could not be allocated (%s)", hipGetErrorString(e));
    }
    c->p2i8plan = pl;
  }
  if (!c->p2i8_y_valid) {
    hipLaunchKernelGGL(i8_slice_y2_kernel, dim3(4096), dim3(256), 0, c->stream, (const double*)c->Kaug, (long)c->LDK, (long)c->Np, c->Mp, c->Dp, pl->KS2, yscale,
                       pl->SlK, pl->strideK);
    GP_HIP(c, hipGetLastError());
    c->p2i8_y_valid = true;
  }
  *SlK = pl->SlK; *strideK = pl->strideK; *KS2 = pl->KS2;
  return GP_OK;
}

int run_phase2_i8(gp_ctx* c, const double* yscale, int* nparts) {
  P2I8Plan* pl = static_cast<P2I8Plan*>(c->p2i8plan);
  if (!pl) return fail(c, GP_ERR_STATE, "int8 phase 2 without its digit buffers (psi1 did not write them)");
  const double ksc = 2.0 * c->sf2;
  hipLaunchKernelGGL(i8_bscale_kernel, dim3((c->Mp + 255) / 256), dim3(256), 0, c->stream, (const double*)c->Bm, c->LDK, c->Mp, ksc, yscale, pl->bscale);
  hipLaunchKernelGGL(i8_slice_b_kernel, dim3((unsigned)std::min<long>(((long)(c->LDK / 16) * c->Mp + 255) / 256, 4096)), dim3(256), 0, c->stream,
                     (const double*)c->Bm, c->LDK, c->Mp, ksc, yscale, (const double*)pl->bscale, pl->SlB, pl->strideB);
  GP_HIP(c, hipGetLastError());
  P2I8Args a;
  a.SlK = pl->SlK; a.strideK = pl->strideK; a.KS2 = pl->KS2; a.SlB = pl->SlB; a.strideB = pl->strideB; a.bscale = pl->bscale;
  a.Kaug = c->Kaug; a.ld = c->LDK; a.mu = c->mu; a.Rpart = c->Rpart; a.Mp = c->Mp; a.CXp = c->CXp; a.Q = c->Q;
  a.MT2 = c->Mp / 64;
  a.ntiles = (int)(c->Np / P2_TN);
  a.KS = c->Mp / 32 + (c->D + 31) / 32;          // >= 17: the path needs Mp >= 512 (p1i8_applicable)
  // one workgroup per CU (147 KB of LDS): 256 / MT2 slices, all column blocks of a slice on one XCD; the partial buffer holds 2 (p2_slices + 8) rows
  int S = std::max(1, std::min(std::min(256 / a.MT2, a.ntiles), (2 * (c->p2_slices + 8)) / 8));
  a.tps = (a.ntiles + S - 1) / S;
  S = (a.ntiles + a.tps - 1) / a.tps;
  a.S = S;
  const int blocks = 8 * ((S + 7) / 8) * a.MT2;
  GP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(p2i8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS));
  GP_EV(c, 12);
  hipLaunchKernelGGL(p2i8_kernel, dim3(blocks), dim3(512), P2_LDS, c->stream, a);
  GP_EV(c, 13);
  GP_HIP(c, hipGetLastError());
  *nparts = 8 * S;
  return GP_OK;
}

void p2i8_free(gp_ctx* c) {
  P2I8Plan* pl = static_cast<P2I8Plan*>(c->p2i8plan);
  if (!pl) return;
  for (void* p : {(void*)pl->SlK, (void*)pl->SlB, (void*)pl->bscale}) if (p) (void)hipFree(p);
  delete pl;
  c->p2i8plan = nullptr;
}

}  // namespace gp
