// Phase 1 of the fixed-embedding path on the INT8 matrix core (Ozaki-style FP64 on int8; DESIGN.md section 6 has the gate's
// numbers).  OPT-IN: GPARML_P1_I8=1 or gp_debug_set_option("p1_i8", 1) switches it on where it applies (p1i8_applicable: fixed embeddings,
// N >= 65536, M >= 512, Q <= 16); the default stays the float64 p1v2_kernel -- the gain is 2-8 % of an evaluation, and the statistics, while good
// enough for float64-grade gradients on every workload tried, are 42-bit quantities whose margin on an arbitrary problem is not known.
//   [Psi2 | C] = K^T [K | Y]   (partial_terms.py:45-52, 79-80; kernel_exp.py:13-49)
// with the float64 operands split into signed 7-bit digits,
//   x = scale * sum_{j=1..S} d_j 128^-j,   d_j in [-64, 64],  |x| <= scale / 2,   S = 6 (42 bits below the operand's scale)
//   x y = scale_x scale_y sum_{a,b} d_a e_b 128^-(a+b),  digit products with a + b <= L = 7 kept (21 of 36),
// each an EXACT integer matrix product on v_mfma_i32_32x32x32_i8 (4.6 POPS measured, 58x the FP64 rate: profiles/r04_i8_ubench.txt);
// products of equal order a + b share one int32 accumulator.  |d e| <= 4096 and at most S pairs per order, so an accumulator holds
// 2^31 / (6 * 4096) = 87381 rows: one slice of the shard per workgroup, converted to float64 once at the end.  Integer sums do not depend on
// the order of accumulation: the digit products are bit-identical for any slicing of the shard.
//
// What the gate found (exact CPU emulation of this arithmetic, tests/devtools/dev_ozaki_gate.py, and the kernel itself -- they agree to the digit):
//   * which products are dropped matters more than how many digits are kept.  Psi1's entries span ten decades inside a column and their
//     density falls with magnitude, so inside a digit cell the remainder has a negative mean: neighbouring digits of ONE number are
//     correlated, and dropping their cross products biases the DIAGONAL of Psi2 (+1e-12 relative with S = 5 / L = 6, 15 products) -- a jitter
//     on K_mm + beta Psi2 (cond 1e10) that moves grad_Z by 1.2e-5 .. 1.7e-5 from the 80-bit truth at N = 1e6 (measured): outside the contract.
//     S = 6 / L = 7 (21 products): 3.65e-6 (emulation and kernel, identical); S = 6 / L = 8 (26 products): 1.6e-8.
//   * the bias lives on the diagonal, and the diagonal is cheap: psi1_kernel adds up the squares of its two columns per lane while it writes the
//     digits (float64, one FMA per element), the reduce kernel puts those sums on Psi2's diagonal.  S = 6 / L = 7 with the exact diagonal:
//     grad_Z 1.1e-7 (N = 1e5) / 4.0e-8 (N = 1e6) from the truth, F 7e-12 -- as good as float64 statistics (1.4e-8 / 3.4e-8), with 21 products
//     and six accumulator sets (192 registers: two waves per SIMD without spilling; 26 products need seven: one spilled register's reload put a
//     vmcnt(0) into the loop and serialised the DMA ring).
//   * speed, same box, N = 1e6: p1v2_kernel 5.8-6.0 ms; this kernel 3.6 ms (64 % of the int8 MFMA rate), psi1_kernel + 0.65 ms for the digits
//     (1.61 against 0.96: the digit arithmetic, 33 VALU instructions per element, makes it VALU-bound), evaluation - 0.4 .. - 1.4 ms depending on
//     the box (on some boxes p2_fast8_kernel runs 2-3 % slower behind the int8 kernel).  What lifted the kernel from 46 %: DMA slot arithmetic
//     precomputed (200 -> 37 scalar instructions per k-step and wave), no spill, a raw s_barrier (__syncthreads() waits for vmcnt(0)), and staging
//     at different times by the two waves of a SIMD -- an LDS-DMA instruction stalls the wave that issues it for ~65 cycles and nobody else
//     (tools/ubench/dma_wave_ubench.hip).
//
// Digits live in HBM as Sl[j][n / 16][col][16]: the 16 consecutive rows a lane feeds to the matrix core as ONE 16-byte operand, columns =
// the Mp columns of Psi1 followed by the Dp columns of Y (the layout of Kaug's rows).  Psi1's digits are written by psi1_kernel while it
// generates Psi1 (psi.hip: each lane already walks 16 rows of its two columns), Y's once per upload.  3.8 GB at N = 1e6, M = 512, D = 100.
//
// Kernel: one workgroup = one 128 x 128 output tile x one n-slice, 8 waves as 2 x 4 of 64 x 32 (two 32 x 32 MFMA tiles, L - 1 orders), k-step =
// 32 rows; both operand panels (S digits x 2 row blocks x 128 columns x 16 B) arrive by LDS-DMA straight in operand order (the digit
// layout IS the LDS image: no swizzle, no padding) through a three-stage ring.
#include "gp_common.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace gp {

constexpr int I8S = GP_I8_DIGITS;          // digits per operand (gp_common.h)
constexpr long I8_MAX_ROWS = 81920;        // rows per workgroup slice (int32 accumulators: 2^31 / (6 * 4096) = 87381)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

std::atomic<int> g_opt_p1_i8{[] { const char* e = getenv("GPARML_P1_I8"); return (e && e[0] == '1') ? 1 : 0; }()};   // opt-in (see the header)
constexpr int I8L = 7;                     // digit products with a + b <= I8L (digits numbered from 1) are kept: 21 of 36
constexpr int I8O = I8L - 1;               // accumulators: orders a + b = 2 .. I8L

struct I8Job { int ci, cj; int ks0, ks1; int part; int pad0, pad1, pad2; };   // column blocks (128 combined columns), k-steps [ks0, ks1) of 32 rows; ci < 0: idle
struct I8Args { const int8_t* Sl; long strideJ; int LDK; const I8Job* jobs; double* part; };

constexpr int I8_PANEL = I8S * 2 * 128 * 16;          // bytes of one operand panel of one k-step: [digit][row block of 16][128 columns][16 B]
constexpr int I8_STAGE = 2 * I8_PANEL;                // A panel | B panel
constexpr int I8_STAGES = 3;

// Workgroup = one 128 x 128 output tile x one n-slice; 8 waves as 2 x 4, wave tile 64 x 32 = two 32x32 MFMA tiles: 32 accumulators per order.
// Three-stage LDS ring, one barrier per k-step: the DMA of k-step i + 2 is issued before the MFMAs of k-step i (two k-steps = ~4k cycles cover
// an HBM round trip; with a two-stage ring the kernel ran at 45 % of the MFMA rate).
__global__ void __launch_bounds__(512, 1) p1i8_kernel(I8Args a) {
  extern __shared__ __attribute__((aligned(16))) int8_t i8lds[];     // [3 stages][A panel | B panel]
  const I8Job job = a.jobs[blockIdx.x];
  if (job.ci < 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const bool diag = job.ci == job.cj;
  v16i acc[I8O][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
      for (int o = 0; o < I8O; ++o) acc[o][t][i] = 0;
    }
  // LDS-DMA of one k-step: 4 I8S instructions of 1 KB per panel (a diagonal tile's B panel is its A panel), a fixed set of NDMA slots per wave:
  // slot e = wave + 8 t covers (panel, digit, row block, column half); its global pointer only advances by one k-step (2 LDK 16 bytes) per
  // iteration and its LDS offset is constant, so nothing is recomputed in the loop (the first version spent 200 scalar instructions per k-step
  // and wave on the slot arithmetic).  Every wave issues the same number per k-step: `s_waitcnt vmcnt(NDMA)` = "all but the newest k-step".
  constexpr int NDMA_OFF = 8 * I8S / 8, NDMA_DIAG = (4 * I8S + 7) / 8;
  const int8_t* dsrc[NDMA_OFF];
  int ddst[NDMA_OFF];
#pragma unroll
  for (int t = 0; t < NDMA_OFF; ++t) {
    const int ne = diag ? 4 * I8S : 8 * I8S;
    int e = wave + 8 * t;
    if (e >= ne) e = ne - 1;                              // padding slots repeat the last transfer (same data, same place)
    const int op = e / (4 * I8S), rem = e - op * (4 * I8S), j = rem >> 2, h = (rem >> 1) & 1, half = rem & 1;
    dsrc[t] = a.Sl + (long)j * a.strideJ + (((long)(2 * job.ks0 + h)) * a.LDK + (op ? job.cj : job.ci) * 128 + 64 * half + lane) * 16;
    ddst[t] = op * I8_PANEL + ((j * 2 + h) * 128 + 64 * half) * 16;
  }
  const long kstep_bytes = 2L * a.LDK * 16;
  auto dma = [&](int stage) {
#pragma unroll
    for (int t = 0; t < NDMA_OFF; ++t)
      if (t < NDMA_DIAG || !diag) {
        __builtin_amdgcn_global_load_lds((gbl_void*)dsrc[t], (lds_void*)(i8lds + stage * I8_STAGE + ddst[t]), 16, 0, 0);
        dsrc[t] += kstep_bytes;
      }
  };
  const int kg = lane >> 5, r32 = lane & 31;
  const int nks = job.ks1 - job.ks0;
  dma(0);
  if (nks > 1) dma(1);
  for (int i = 0; i < nks; ++i) {
    const int stage = i % I8_STAGES;
    if (i + 1 < nks) { if (diag) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA_DIAG) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA_OFF) : "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // stage i has landed for every wave; stage (i + 2) % 3 = (i - 1) % 3 is no longer being read.  A raw barrier: __syncthreads() also waits
    // for vmcnt(0), i.e. for the DMA of k-step i + 1 -- the ring would be one stage deep
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // An LDS-DMA instruction stalls the wave that issues it for ~65 cycles and nobody else (tools/ubench/dma_wave_ubench.hip: the partner wave on
    // the same SIMD keeps its full MFMA rate).  The two waves of a SIMD (w and w + 4) therefore stage at different times: the first right behind
    // the barrier, the second in the middle of its MFMAs -- one of them always feeds the matrix core.
    const bool early = wave < 4;
    if (early && i + 2 < nks) dma((i + 2) % I8_STAGES);
    const int8_t* pa = i8lds + stage * I8_STAGE;
    const int8_t* pb = diag ? pa : pa + I8_PANEL;
    v4i bv[I8S], av[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) av[0][ti] = *(const v4i*)(pa + ((0 * 2 + kg) * 128 + wr * 64 + ti * 32 + r32) * 16);
#pragma unroll
    for (int b = 0; b < I8S; ++b) bv[b] = *(const v4i*)(pb + ((b * 2 + kg) * 128 + wc * 32 + r32) * 16);
    // digit a of the A panel against digits b <= I8O - 1 - a of the B panel; the next digit's A operands are requested before this digit's MFMAs
#pragma unroll
    for (int av_ = 0; av_ < I8S; ++av_) {
      if (av_ == 2 && !early && i + 2 < nks) dma((i + 2) % I8_STAGES);
      if (av_ + 1 < I8S) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) av[(av_ + 1) & 1][ti] = *(const v4i*)(pa + (((av_ + 1) * 2 + kg) * 128 + wr * 64 + ti * 32 + r32) * 16);
      }
#pragma unroll
      for (int b = 0; b < I8S; ++b)
        if (b + av_ < I8O) {
#pragma unroll
          for (int ti = 0; ti < 2; ++ti)
            acc[av_ + b][ti] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[av_ & 1][ti], bv[b], acc[av_ + b][ti], 0, 0, 0);
        }
    }
  }
  // int32 -> float64: sum_o acc[o] 128^-(o + 2) (digit j has weight 128^-j, j = 1 .. S); C/D map of the 32x32 MFMA: col = lane & 31,
  // row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  double* out = a.part + (long)job.part * (TILE * TILE);
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      double v = 0.0, w = 1.0 / (128.0 * 128.0);
      double ws[I8O];
#pragma unroll
      for (int o = 0; o < I8O; ++o) { ws[o] = w; w *= 1.0 / 128.0; }
#pragma unroll
      for (int o = I8O - 1; o >= 0; --o) v = fma((double)acc[o][ti][i], ws[o], v);     // small terms first
      const int row = wr * 64 + ti * 32 + (i & 3) + 8 * (i >> 2) + 4 * kg, col = wc * 32 + r32;
      out[row * TILE + col] = v;
    }
}

// The diagonal of Psi2 from psi1_kernel's float64 sums of squares, dpart [row_blocks][Mp] -> diag [Mp]: eight interleaved partial sums per column
// (row block b goes to sum b & 7, in order), combined as ((0+1)+(2+3))+((4+5)+(6+7)) -- the order of the loop that used to sit in the reduce kernel,
// where ONE thread per diagonal element walked all row blocks (1954 dependent loads at N = 1e6: 0.8 ms for a 36 MB reduction, r04).  Here eight
// threads per column own one partial sum each, eight loads in flight.
__global__ void __launch_bounds__(256) p1i8_diag_kernel(const double* __restrict__ dpart, int row_blocks, int Mp, double* __restrict__ diag) {
  __shared__ double comb[8][32];
  const int cl = threadIdx.x & 31, u = threadIdx.x >> 5, col = blockIdx.x * 32 + cl;
  double acc = 0.0;
  if (col < Mp) {
    const double* src = dpart + col;
    int b = u;
    for (; b + 8 * 7 < row_blocks; b += 8 * 8) {
      double x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = src[(long)(b + 8 * j) * Mp];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += x[j];
    }
    for (; b < row_blocks; b += 8) acc += src[(long)b * Mp];
  }
  comb[u][cl] = acc;
  __syncthreads();
  if (u == 0 && col < Mp)
    diag[col] = ((comb[0][cl] + comb[1][cl]) + (comb[2][cl] + comb[3][cl])) + ((comb[4][cl] + comb[5][cl]) + (comb[6][cl] + comb[7][cl]));
}

// sum the slices' partial tiles in a fixed order, apply the operands' scales and write the statistics (both triangles of Psi2)
struct I8Out { int ci, cj, first, nslices, stride, isC, pad0, pad1; };
__global__ void __launch_bounds__(256) p1i8_reduce_kernel(const double* __restrict__ part, const I8Out* __restrict__ outs, const double* __restrict__ yscale,
                                                          double kscale2, double kscale, double* __restrict__ Psi2, double* __restrict__ C, int Mp, int Dp,
                                                          const double* __restrict__ diag) {
  const I8Out o = outs[blockIdx.y];
  const int e = blockIdx.x * 256 + threadIdx.x, r = e >> 7, c = e & 127;
  const double* src = part + (long)o.first * (TILE * TILE) + e;
  const long step = (long)o.stride * (TILE * TILE);
  double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int sl = 0;
  for (; sl + 8 <= o.nslices; sl += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += src[(long)(sl + u) * step];
  }
  for (int u = 0; sl < o.nslices; ++sl, ++u) acc[u] += src[(long)sl * step];
  const double s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  if (o.isC) {
    const int d = (o.cj - Mp / TILE) * TILE + c;
    if (d < Dp) C[((long)o.ci * TILE + r) * Dp + d] = s * kscale * yscale[d];
  } else {
    const long R = (long)o.ci * TILE + r, Cc = (long)o.cj * TILE + c;
    double v = s * kscale2;
    if (R == Cc) {
      // the diagonal of Psi2 from psi1_kernel's float64 sums of squares: the dropped digit products of ONE number with itself do not average out
      // (neighbouring digits are correlated), and a bias on the diagonal is a jitter on K_mm + beta Psi2
      v = diag[R];                                          // p1i8_diag_kernel
    }
    Psi2[R * Mp + Cc] = v;
    if (o.ci != o.cj) Psi2[Cc * Mp + R] = v;               // a diagonal tile is computed whole: integer sums, exactly symmetric
  }
}

// ---- digits of Y (once per upload): per-column scale 2^e with max |y| <= 2^(e-1)
__global__ void __launch_bounds__(256) i8_colmax_kernel(const double* __restrict__ Kaug, long ld, long Np, int Mp, int Dp, double* __restrict__ pmax /*[blocks][Dp]*/) {
  const int d = threadIdx.x & 127, sub = threadIdx.x >> 7;
  __shared__ double red[256];
  for (int d0 = 0; d0 < Dp; d0 += 128) {
    double m = 0.0;
    for (long n = blockIdx.x * 2L + sub; n < Np; n += gridDim.x * 2L) m = fmax(m, fabs(Kaug[n * ld + Mp + d0 + d]));
    red[threadIdx.x] = m;
    __syncthreads();
    if (sub == 0) pmax[(long)blockIdx.x * Dp + d0 + d] = fmax(red[d], red[128 + d]);
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) i8_yscale_kernel(const double* __restrict__ pmax, int blocks, int Dp, double* __restrict__ yscale) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= Dp) return;
  double m = 0.0;
  for (int b = 0; b < blocks; ++b) m = fmax(m, pmax[(long)b * Dp + d]);
  int e = 0;
  if (m > 0.0) { (void)frexp(m, &e); e += 1; }          // m = f 2^e, f in [0.5, 1)  ->  m <= 2^e  ->  |y| 2^-(e+1) <= 1/2
  yscale[d] = ldexp(1.0, e);
}
__global__ void __launch_bounds__(256) i8_slice_y_kernel(const double* __restrict__ Kaug, long ld, long Np, int Mp, int Dp, int LDK,
                                                         const double* __restrict__ yscale, int8_t* __restrict__ Sl, long strideJ) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= Dp) return;
  const double inv = 1.0 / yscale[d];
  for (long nb = blockIdx.y; nb < Np / 16; nb += gridDim.y) {
    unsigned pk[I8S][4];
#pragma unroll
    for (int j = 0; j < I8S; ++j) { pk[j][0] = pk[j][1] = pk[j][2] = pk[j][3] = 0u; }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      double t = Kaug[(nb * 16 + i) * ld + Mp + d] * inv;
#pragma unroll
      for (int j = 0; j < I8S; ++j) {
        t *= 128.0;
        const double dg = __builtin_rint(t);
        t -= dg;
        pk[j][i >> 2] |= ((unsigned)(int)dg & 0xffu) << (8 * (i & 3));
      }
    }
#pragma unroll
    for (int j = 0; j < I8S; ++j) {
      uint4 v = {pk[j][0], pk[j][1], pk[j][2], pk[j][3]};
      *(uint4*)(Sl + (long)j * strideJ + ((nb * (long)LDK) + Mp + d) * 16) = v;
    }
  }
}

// ---- host ---------------------------------------------------------------------------------------------------------------------------
struct I8Plan {
  int8_t* Sl = nullptr; long strideJ = 0;
  double* yscale = nullptr; double* pmax = nullptr;
  double* dpart = nullptr; int row_blocks = 0;     // [row_blocks][Mp] sums of squares of Psi1's columns per psi1_kernel workgroup (the exact diagonal of Psi2)
  double* diag = nullptr;                          // [Mp] their sums (p1i8_diag_kernel)
  I8Job* jobs = nullptr; I8Out* outs = nullptr;
  int blocks = 0, nouts = 0;
  bool y_valid = false;
};

// Psi1's digits come from psi1_kernel's four-waves-across-the-columns form (Mp >= 512, Q <= 16) with fixed embeddings
bool p1i8_applicable_static(const gp_ctx* c) {
  return !c->i8_unsupported && c->regime_A && c->N >= 65536 && c->Mp >= 512 && psi1_qp(c->Q) > 0 && psi1_qp(c->Q) <= 16;
}
bool p1i8_applicable(const gp_ctx* c) {
  // from 65536 rows on: below that a workgroup's slice is a few hundred k-steps and the float64 kernels are as fast; and with few rows per
  // inducing point the truncation of the operands weighs more (N = 5e3, M = 600: 2.7e-5 on grad_Z with five digits)
  return g_opt_p1_i8.load() != 0 && !c->i8_unsupported && c->i8_guard != 2 && c->regime_A && c->N >= 65536 && c->Mp >= 512 && psi1_qp(c->Q) > 0 && psi1_qp(c->Q) <= 16 && !c->want_emb;
}

int p1i8_prepare(gp_ctx* c, int8_t** Sl, long* strideJ, double** Dpart, int row_blocks) {
  I8Plan* pl = static_cast<I8Plan*>(c->i8plan);
  if (c->i8_unsupported) return GP_ERR_UNSUPPORTED;
  if (!pl) {
    // published only when complete: a failure below frees what was allocated and switches the path off for this context
    struct Guard { gp_ctx* c; I8Plan* p; ~Guard() { if (p) { c->i8plan = p; p1i8_free(c); c->i8_unsupported = true; } } };
    pl = new I8Plan();
    Guard guard{c, pl};
    const int MT = c->Mp / TILE, DT = c->Dp / TILE;
    pl->strideJ = (c->Np / 16) * (long)c->LDK * 16;
    GP_TRY_RC(dalloc_bytes(c, (void**)&pl->Sl, (size_t)I8S * pl->strideJ, DA_RAW));
    GP_TRY_RC(dalloc_bytes(c, (void**)&pl->yscale, (size_t)c->Dp * sizeof(double), DA_RAW));
    GP_TRY_RC(dalloc_bytes(c, (void**)&pl->pmax, (size_t)1024 * c->Dp * sizeof(double), DA_RAW));
    pl->row_blocks = row_blocks;
    GP_TRY_RC(dalloc_bytes(c, (void**)&pl->dpart, (size_t)row_blocks * c->Mp * sizeof(double), DA_RAW));
    GP_TRY_RC(dalloc_bytes(c, (void**)&pl->diag, (size_t)c->Mp * sizeof(double), DA_RAW));
    // tiles of one n-slice: Psi2 upper tiles, then the C tiles; slices per XCD chosen for whole rounds of the XCD's 32 CUs (one workgroup
    // per CU: 320 accumulator registers), every tile of a slice on ONE XCD so that the slice's digits are fetched from HBM once
    std::vector<int> tiles;                                 // (row block, column block) of 128 combined columns [Psi1 | Y]
    for (int i = 0; i < MT; ++i) for (int j = i; j < MT; ++j) { tiles.push_back(i); tiles.push_back(j); }
    for (int i = 0; i < MT; ++i) for (int j = 0; j < DT; ++j) { tiles.push_back(i); tiles.push_back(MT + j); }
    const int T = (int)tiles.size() / 2;
    constexpr int SLOTS = 32;                               // resident workgroups per XCD: one per CU (120 KB of LDS)
    const long ksteps = c->Np / 32;
    const int s8min = (int)std::max<long>(1, (c->Np + 8 * I8_MAX_ROWS - 1) / (8 * I8_MAX_ROWS));
    int s8 = s8min; double best = 1e30;
    for (int t = s8min; t <= s8min + 8; ++t) {
      const double cost = std::ceil((double)t * T / (double)SLOTS) / t;
      if (cost < best - 1e-9) { best = cost; s8 = t; }
    }
    const int Smax = (int)(c->part_doubles / ((size_t)T * TILE * TILE));          // the partial buffer holds S x T tiles
    int S = (int)std::min<long>(std::min<long>(8L * s8, ksteps), Smax);
    if (S < 1 || (c->Np + S - 1) / S + 32 > 87000) return fail(c, GP_ERR_UNSUPPORTED, "int8 phase 1: partial buffer too small for %d tiles", T);
    // placement: block b runs on XCD b % 8, one workgroup per CU, 32 CUs per XCD.  Whole slices first (every tile of a slice on one XCD: the
    // slice's digits are fetched from HBM once); when a single round fits (T <= 32) the CUs left over on each XCD are pooled into "shared"
    // slices whose tiles are spread over the XCDs (their digits are fetched by several XCDs: 2 of 18 slices at M = 512, D = 100)
    std::vector<std::vector<I8Job>> per_xcd(8);
    int n_shared = 0;
    if (T <= SLOTS && S == 8 * (SLOTS / T) && s8 == SLOTS / T) {
      const int left = SLOTS - (SLOTS / T) * T;
      n_shared = (int)std::min<long>((8 * left) / T, std::min<long>(Smax - S, ksteps - S));
      if (n_shared < 0) n_shared = 0;
    }
    const int Stot = S + n_shared;
    for (int s = 0; s < S; ++s) {
      const int k0 = (int)((long)s * ksteps / Stot), k1 = (int)((long)(s + 1) * ksteps / Stot);
      for (int t = 0; t < T; ++t) per_xcd[s % 8].push_back(I8Job{tiles[2 * t], tiles[2 * t + 1], k0, k1, s * T + t, 0, 0, 0});
    }
    {
      int x = 0;
      for (int s = S; s < Stot; ++s) {
        const int k0 = (int)((long)s * ksteps / Stot), k1 = (int)((long)(s + 1) * ksteps / Stot);
        for (int t = 0; t < T; ++t) {
          while ((int)per_xcd[x].size() >= SLOTS) x = (x + 1) & 7;
          per_xcd[x].push_back(I8Job{tiles[2 * t], tiles[2 * t + 1], k0, k1, s * T + t, 0, 0, 0});
        }
      }
    }
    S = Stot;
    size_t depth = 0;
    for (auto& v : per_xcd) depth = std::max(depth, v.size());
    std::vector<I8Job> jobs(depth * 8, I8Job{-1, 0, 0, 0, 0, 0, 0, 0});
    for (int x = 0; x < 8; ++x) for (size_t j = 0; j < per_xcd[x].size(); ++j) jobs[j * 8 + x] = per_xcd[x][j];   // block b runs on XCD b % 8
    std::vector<I8Out> outs;
    for (int t = 0; t < T; ++t) outs.push_back(I8Out{tiles[2 * t], tiles[2 * t + 1], t, S, T, tiles[2 * t + 1] >= MT ? 1 : 0, 0, 0});
    GP_HIP(c, hipMalloc((void**)&pl->jobs, jobs.size() * sizeof(I8Job)));
    GP_HIP(c, hipMalloc((void**)&pl->outs, outs.size() * sizeof(I8Out)));
    GP_HIP(c, hipMemcpyAsync(pl->jobs, jobs.data(), jobs.size() * sizeof(I8Job), hipMemcpyHostToDevice, c->stream));
    GP_HIP(c, hipMemcpyAsync(pl->outs, outs.data(), outs.size() * sizeof(I8Out), hipMemcpyHostToDevice, c->stream));
    GP_HIP(c, hipStreamSynchronize(c->stream));
    pl->blocks = (int)jobs.size(); pl->nouts = (int)outs.size();
    pl->y_valid = false;
    guard.p = nullptr;
    c->i8plan = pl;
  }
  if (!c->i8_y_valid) pl->y_valid = false;
  if (!pl->y_valid) {
    const int nb = 1024;
    hipLaunchKernelGGL(i8_colmax_kernel, dim3(nb), dim3(256), 0, c->stream, c->Kaug, (long)c->LDK, (long)c->Np, c->Mp, c->Dp, pl->pmax);
    hipLaunchKernelGGL(i8_yscale_kernel, dim3((c->Dp + 255) / 256), dim3(256), 0, c->stream, pl->pmax, nb, c->Dp, pl->yscale);
    hipLaunchKernelGGL(i8_slice_y_kernel, dim3((c->Dp + 255) / 256, (unsigned)std::min<long>(c->Np / 16, 4096)), dim3(256), 0, c->stream, c->Kaug, (long)c->LDK,
                       (long)c->Np, c->Mp, c->Dp, c->LDK, pl->yscale, pl->Sl, pl->strideJ);
    GP_HIP(c, hipGetLastError());
    pl->y_valid = true;
    c->i8_y_valid = true;
  }
  if (pl->row_blocks != row_blocks) return fail(c, GP_ERR_STATE, "int8 phase 1: psi1_kernel's row blocking changed");
  *Sl = pl->Sl; *strideJ = pl->strideJ; *Dpart = pl->dpart;
  return GP_OK;
}

int run_phase1_i8(gp_ctx* c) {
  I8Plan* pl = static_cast<I8Plan*>(c->i8plan);
  if (!pl || !pl->y_valid) return fail(c, GP_ERR_STATE, "int8 phase 1 without its digit buffers (psi1 did not write them)");
  I8Args a;
  a.Sl = pl->Sl; a.strideJ = pl->strideJ; a.LDK = c->LDK; a.jobs = pl->jobs; a.part = c->part;
  constexpr int lds = I8_STAGES * I8_STAGE;
  GP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(p1i8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  GP_EV(c, 10);
  hipLaunchKernelGGL(p1i8_kernel, dim3(pl->blocks), dim3(512), lds, c->stream, a);
  GP_EV(c, 11);
  GP_HIP(c, hipGetLastError());
  double* Psi2 = c->stats;
  double* C = c->stats + (long)c->Mp * c->Mp;
  // K = 2 sf2 t  (t = the sliced value, |t| <= 1/2)
  hipLaunchKernelGGL(p1i8_diag_kernel, dim3((c->Mp + 31) / 32), dim3(256), 0, c->stream, pl->dpart, pl->row_blocks, c->Mp, pl->diag);
  hipLaunchKernelGGL(p1i8_reduce_kernel, dim3(TILE * TILE / 256, pl->nouts), dim3(256), 0, c->stream, c->part, pl->outs, pl->yscale, 4.0 * c->sf2 * c->sf2,
                     2.0 * c->sf2, Psi2, C, c->Mp, c->Dp, pl->diag);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

// ---- guard ----------------------------------------------------------------------------------------------------------------------------
// The int8 statistics are 42-bit operands and 21 of 36 digit products: how good they are on THIS problem is measured, not assumed (DESIGN.md
// section 6 has the error model and why an a-priori bound is vacuous: cond(K_mm + beta Psi2) x relative perturbation overestimates the effect on the
// gradients a hundredfold).  The first int8 evaluation after an upload and every 64th one run BOTH phase-1 paths on the same Psi1; the evaluation
// itself uses the float64 statistics; on the device
//     r2 = |Psi2_int8 - Psi2|_F / |Psi2|_F,   rC = |C_int8 - C|_F / |C|_F,   cond_lb = (sf2 + beta max_i Psi2_ii) max_i P_ii  <=  cond_2(K_mm + beta Psi2)
// (max diag(A) <= lambda_max and max diag(A^-1) <= 1 / lambda_min).  The context stays on the int8 path while cond_lb max(r2, rC) <= I8_GUARD_TAU.
// Calibration on the benchmark workload against the 80-bit truth (profiles/r05_int8_guard.txt): six digits score 7.6e-6 (N = 1e5, grad_Z 1.1e-7 from
// the truth) and 2.5e-6 (N = 1e6, 4.0e-8); a five-digit build 3.8e-4 (N = 1e5) and 1.3e-4 (N = 1e6, grad_Z 5.8e-7): grad_Z's error stays below
// 0.015 x score, so tau = 1e-4 keeps it below 1.5e-6 with a 13-fold margin over what the six-digit path shows on this workload.  A rejected context runs the float64 kernels until
// the next upload; gp_i8_status reports the state and the three numbers.
constexpr double I8_GUARD_TAU = 1e-4;
std::atomic<int> g_opt_i8_guard_strict{0};     // test hook (gp_debug_set_option("i8_guard_strict", 1)): threshold 0 -- every check rejects
constexpr int I8_CMP_BLOCKS = 64;
__global__ void __launch_bounds__(256) i8_compare_kernel(const double* __restrict__ a, const double* __restrict__ b, long n2, long nc, double* __restrict__ cmp) {
  // blocks [0, 64): Psi2 part, [64, 128): C part; per-block partials of |a - b|^2 and |b|^2 at cmp[8 + 2 block], fixed order
  __shared__ double r0[256], r1[256];
  const int half = blockIdx.x / I8_CMP_BLOCKS, blk = blockIdx.x % I8_CMP_BLOCKS;
  const long lo = half ? n2 : 0, n = half ? nc : n2;
  double d2 = 0.0, s2 = 0.0;
  for (long i = blk * 256L + threadIdx.x; i < n; i += I8_CMP_BLOCKS * 256L) { const double x = a[lo + i], y = b[lo + i]; d2 = fma(x - y, x - y, d2); s2 = fma(y, y, s2); }
  r0[threadIdx.x] = d2; r1[threadIdx.x] = s2;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) { r0[threadIdx.x] += r0[threadIdx.x + k]; r1[threadIdx.x] += r1[threadIdx.x + k]; } __syncthreads(); }
  if (threadIdx.x == 0) { cmp[8 + 2 * blockIdx.x] = r0[0]; cmp[8 + 2 * blockIdx.x + 1] = r1[0]; }
}
__global__ void __launch_bounds__(256) i8_compare_final_kernel(double* __restrict__ cmp, const double* __restrict__ Psi2, const double* __restrict__ P, int M, int Mp) {
  // after the global step: the four norms in block order, max diag(Psi2), max diag(P)
  __shared__ double r0[256], r1[256];
  if (threadIdx.x < 4) {
    const int half = threadIdx.x >> 1, which = threadIdx.x & 1;
    double s = 0.0;
    for (int b = 0; b < I8_CMP_BLOCKS; ++b) s += cmp[8 + 2 * (half * I8_CMP_BLOCKS + b) + which];
    cmp[threadIdx.x] = s;                                   // [dPsi2^2, Psi2^2, dC^2, C^2]
  }
  double m2 = 0.0, mp = 0.0;
  for (int i = threadIdx.x; i < M; i += 256) { m2 = fmax(m2, Psi2[(long)i * Mp + i]); mp = fmax(mp, P[(long)i * Mp + i]); }
  r0[threadIdx.x] = m2; r1[threadIdx.x] = mp;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) { r0[threadIdx.x] = fmax(r0[threadIdx.x], r0[threadIdx.x + k]); r1[threadIdx.x] = fmax(r1[threadIdx.x], r1[threadIdx.x + k]); } __syncthreads(); }
  if (threadIdx.x == 0) { cmp[4] = r0[0]; cmp[5] = r1[0]; }
}

int p1i8_check_begin(gp_ctx* c) {
  const size_t n = (size_t)c->Mp * c->Mp + (size_t)c->Mp * c->Dp;
  if (!c->i8_cmp) GP_TRY_RC(dalloc_bytes(c, (void**)&c->i8_cmp, (8 + 4 * I8_CMP_BLOCKS + n) * sizeof(double), DA_RAW));
  // the int8 statistics aside (behind the comparison scalars): the float64 phase 1 overwrites the statistics buffer
  GP_HIP(c, hipMemcpyAsync(c->i8_cmp + 8 + 4 * I8_CMP_BLOCKS, c->stats, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  return GP_OK;
}
int p1i8_check_compare(gp_ctx* c) {
  const long n2 = (long)c->Mp * c->Mp, nc = (long)c->Mp * c->Dp;
  hipLaunchKernelGGL(i8_compare_kernel, dim3(2 * I8_CMP_BLOCKS), dim3(256), 0, c->stream, (const double*)(c->i8_cmp + 8 + 4 * I8_CMP_BLOCKS), (const double*)c->stats, n2, nc,
                     c->i8_cmp);
  GP_HIP(c, hipGetLastError());
  c->i8_check_pending = true;
  c->i8_since_check = 0;
  return GP_OK;
}
int p1i8_check_finish(gp_ctx* c) {
  // called by gp_finish of a checked evaluation, after the global step (P = (K_mm + beta Psi2)^-1 exists) and before its synchronisation is over:
  // one more small kernel and a 48-byte copy, once per 64 evaluations
  const long mm = (long)c->Mp * c->Mp;
  hipLaunchKernelGGL(i8_compare_final_kernel, dim3(1), dim3(256), 0, c->stream, c->i8_cmp, (const double*)c->stats, (const double*)(c->Inv + mm), c->M, c->Mp);
  double h[6];
  GP_HIP(c, hipMemcpyAsync(h, c->i8_cmp, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  GP_HIP(c, hipStreamSynchronize(c->stream));
  ++c->sync_epoch;
  c->i8_check_pending = false;
  c->i8_rel_psi2 = h[1] > 0.0 ? std::sqrt(h[0] / h[1]) : 0.0;
  c->i8_rel_c = h[3] > 0.0 ? std::sqrt(h[2] / h[3]) : 0.0;
  c->i8_cond_lb = (c->sf2 + c->beta * h[4]) * h[5];
  ++c->i8_checks;
  const double score = c->i8_cond_lb * std::max(c->i8_rel_psi2, c->i8_rel_c);
  c->i8_guard = (std::isfinite(score) && score <= (g_opt_i8_guard_strict.load() ? 0.0 : I8_GUARD_TAU)) ? 1 : 2;
  return GP_OK;
}

void p1i8_free(gp_ctx* c) {
  if (c->i8_cmp) { (void)hipFree(c->i8_cmp); c->i8_cmp = nullptr; }
  I8Plan* pl = static_cast<I8Plan*>(c->i8plan);
  if (!pl) return;
  for (void* p : {(void*)pl->Sl, (void*)pl->yscale, (void*)pl->pmax, (void*)pl->dpart, (void*)pl->diag, (void*)pl->jobs, (void*)pl->outs}) if (p) (void)hipFree(p);
  delete pl;
  c->i8plan = nullptr;
}

}  // namespace gp
