// Regime B (Bayesian GPLVM, variances > 0): the pairwise psi2 statistics.
//   psi2_n[m,m'] = exp( LE[n,m] + LE[n,m'] + sum_q V[n,q] * dz2[m,m',q] )            (kernel_exp.py:143-146, factorised)
//     LE[n,m]    = 1/2 ln c2_n - 1/2 sum_q w_nq (mu_nq - z_mq)^2,  w = alpha/(2 alpha S + 1),  c2 = sf2^2 prod (2 alpha S + 1)^-1/2
//     V[n,q]     = -1/4 (alpha_q - w_nq),   dz2[m,m',q] = (z_mq - z_m'q)^2
// Phase 1: Psi2 = sum_n psi2_n with one THREAD per (m,m') pair (upper triangle), the point index is wave-uniform.
// Phase 2: T_n = Bbar o psi2_n, r_n = T_n 1, t_n = T_n Z with one LANE per point, the pair index is wave-uniform
//          (every pair quantity is a scalar operand), giving the psi2 parts of grad_Z / grad_alpha / grad_X_mu / grad_X_S
//          (partial_terms.py:190-205, 273-284, 388-394, 421-427).
#include "gp_common.h"
#include "mma_f64.h"
#include "fexp.h"
#include "quad_mma.h"
#include "lane_reduce.h"
#include <algorithm>

namespace gp {

// ---------------------------------------------------------------------------------------------- tables
// per-point tables and LE in both layouts; thread = point for LET (coalesced along n), thread = column for LE
__global__ void __launch_bounds__(256) b_tables_kernel(const double* __restrict__ mu, const double* __restrict__ S,
                                                        const double* __restrict__ alpha, long N, long Np, int Q, double sf2,
                                                        double* __restrict__ Vn, double* __restrict__ Wn, double* __restrict__ lnc2h,
                                                        double* __restrict__ V2P, int QB, double* __restrict__ WP,
                                                        double* __restrict__ MUP) {
  for (long n = blockIdx.x * 256L + threadIdx.x; n < Np; n += (long)gridDim.x * 256L) {
    double l = log(sf2);   // half of ln c2 = ln sf2 - 1/4 sum ln(2 a S + 1)
    for (int q = 0; q < Q; ++q) {
      const double a = alpha[q], s = S[n * Q + q];
      const double d2 = 2.0 * a * s + 1.0, w = a / d2;
      Wn[n * Q + q] = w;
      Vn[n * Q + q] = -0.25 * (a - w);
      V2P[n * QB + q] = 0.5 * (a - w);      // -2 V_nq (columns >= Q stay zero from the allocation)
      WP[n * QB + q] = w;
      MUP[n * QB + q] = mu[n * Q + q];
      l -= 0.25 * log(d2);
    }
    lnc2h[n] = l;
  }
}

// LE and LEA (n-major [Np][Mp], padded entries = kPadLog so that their exp is exactly 0).  Same shape as psi1_kernel: a wave
// owns 64*CPL columns (z_m in registers), the point's padded rows [mu | w | -2V] are wave-uniform scalar loads, a workgroup
// writes whole rows (HBM-write bound: 16 B per (n, m)).
//   LE  = 1/2 ln c2_n - 1/2 sum_q w_nq (mu_nq - z_mq)^2
//   LEA = LE + sum_q V_nq z_mq^2: with it the pair exponent is LEA_nm + LEA_nm' - 2 sum_q V_nq z_mq z_m'q
// Layout of LE (r06).  Up to the 16-wide latent tables psi2_pairs_kernel is its only reader in the hot path, four points per trip: element (n, m) sits at
// ((n / 4) Mp + m) 4 + n % 4, so a lane reads the four values of its row m (and of its column m') with two 16-byte loads instead of four 8-byte ones.  The kernel
// was bound by the NUMBER of its vector-memory instructions as much as by the FP64 pipe (the texture addresser takes 16 cycles per wave-instruction whatever
// the width: a timing build with one of the two 8-byte loads per point dropped ran 10 % faster at Q = 10, 20 % at Q = 5; profiles/r06_gplvm_experiments.txt item 16).
// Wider tables (the matrix-core pair kernel reads LEA) keep LE point-major for the compat path.

template <int QT, int CPL>
__global__ void __launch_bounds__(256) b_le_kernel(const double* __restrict__ MUP, const double* __restrict__ WP, const double* __restrict__ V2P,
                                                    const double* __restrict__ lnc2h, const double* __restrict__ ZP, long N, int M, int Mp,
                                                    double* __restrict__ LE, double* __restrict__ LEA) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = (blockIdx.x * 4 + wave) * (64 * CPL) + CPL * lane;
  if (col >= Mp) return;
  double z[CPL][QT];
#pragma unroll
  for (int c = 0; c < CPL; ++c)
#pragma unroll
    for (int q = 0; q < QT; ++q) z[c][q] = ZP[(long)(col + c) * QT + q];
  const long row0 = blockIdx.y * 16L;
  constexpr bool IL = le_interleaved(QT);
  double e4[IL ? 4 : 1][CPL];
#pragma unroll 2
  for (int r = 0; r < 16; ++r) {
    const long n = row0 + r;                    // < Np (a multiple of 128)
    const double* mu = MUP + n * QT;            // wave-uniform
    const double* w = WP + n * QT;
    const double* v2 = V2P + n * QT;
    const double l0 = lnc2h[n];
    double e[CPL], ea[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      double s = 0.0, t = 0.0;
#pragma unroll
      for (int q = 0; q < QT; ++q) {
        const double d = mu[q] - z[c][q];
        s = fma(w[q] * d, d, s);
        t = fma(v2[q] * z[c][q], z[c][q], t);
      }
      const bool live = n < N && col + c < M;
      e[c] = live ? l0 - 0.5 * s : kPadLog;
      ea[c] = live ? e[c] - 0.5 * t : kPadLog;   // V = -V2P / 2
    }
    if (CPL == 2) {
      double2 b2;
      b2.x = ea[0]; b2.y = ea[CPL - 1];
      *reinterpret_cast<double2*>(&LEA[n * Mp + col]) = b2;
    } else {
      LEA[n * Mp + col] = ea[0];
    }
    if (IL) {
      // four points of one column side by side (le_index): psi2_pairs_kernel reads a trip's four values with two 16-byte loads per side
#pragma unroll
      for (int c = 0; c < CPL; ++c) e4[IL ? (r & 3) : 0][c] = e[c];
      if ((r & 3) == 3) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          double* dst = LE + (((n >> 2) * Mp + col + c) << 2);
          double2 lo2, hi2;
          lo2.x = e4[0][c]; lo2.y = e4[IL ? 1 : 0][c]; hi2.x = e4[IL ? 2 : 0][c]; hi2.y = e4[IL ? 3 : 0][c];
          *reinterpret_cast<double2*>(dst) = lo2;
          *reinterpret_cast<double2*>(dst + 2) = hi2;
        }
      }
    } else if (CPL == 2) {
      double2 a;
      a.x = e[0]; a.y = e[CPL - 1];
      *reinterpret_cast<double2*>(&LE[n * Mp + col]) = a;
    } else {
      LE[n * Mp + col] = e[0];
    }
  }
}

__global__ void __launch_bounds__(256) dz2_kernel(const double* __restrict__ Z, int M, int Q, double* __restrict__ DZ2) {
  const long total = (long)M * M * Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int q = (int)(i % Q);
    const long mm = i / Q;
    const int m2 = (int)(mm % M), m1 = (int)(mm / M);
    const double d = Z[(long)m1 * Q + q] - Z[(long)m2 * Q + q];
    DZ2[i] = d * d;
  }
}

__global__ void __launch_bounds__(256) zpad_kernel(const double* __restrict__ Z, int M, int Mp, int Q, int QB, double* __restrict__ ZP,
                                                    double* __restrict__ Z1P, double* __restrict__ Z1S, int RT) {
  const long total = (long)Mp * QB;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int q = (int)(i % QB), m = (int)(i / QB);
    const double z = (m < M && q < Q) ? Z[(long)m * Q + q] : 0.0;
    ZP[i] = z;
    Z1P[i] = (q == Q) ? 1.0 : z;          // Z with a column of ones at index Q (the MFMA kernel's row-sum column)
  }
  if (Z1S) {
    // [Z | 1 at index QB | 0]: the tile-pair kernel's row-side operand (padded rows are all zero except the ones column, which
    // only ever multiplies T = 0 there)
    const long tot2 = (long)Mp * RT;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < tot2; i += (long)gridDim.x * 256L) {
      const int f = (int)(i % RT), m = (int)(i / RT);
      Z1S[i] = (f < Q && m < M) ? Z[(long)m * Q + f] : (f == QB ? 1.0 : 0.0);
    }
  }
}

// ---------------------------------------------------------------------------------------------- phase 1
// grid (pair tiles, n slices); thread (i,j) of a 16x16 tile owns the pair (m = I*16+i, m' = J*16+j), J >= I.
template <int QT>
__global__ void __launch_bounds__(256) psi2_pairs_kernel(const double* __restrict__ LE, const double* __restrict__ V2P,
                                                          const double* __restrict__ ZP, const int* __restrict__ ptiles, long N,
                                                          int Mp, int S, double* __restrict__ part, int T) {
  // exponent = LE[n,m] + LE[n,m'] + sum_q V_nq dz2_q = LE + LE' + sum_q (-2 V_nq) * (-1/2 dz2_q): the per-pair vector lives in
  // registers, the per-point vector (-2V, zero-padded to QT) is wave-uniform -> scalar loads; four points per trip so
  // the loads of a trip are in flight together.  Padded rows of LE hold kPadLog (exp -> 0), padded pairs are dropped by
  // the reduce kernel.
  const int tile = blockIdx.x, slice = blockIdx.y;
  const int I = ptiles[2 * tile], J = ptiles[2 * tile + 1];
  const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
  const int m1 = I * 16 + i, m2 = J * 16 + j;
  double dz[QT];
#pragma unroll
  for (int q = 0; q < QT; ++q) {
    const double d = ZP[(long)m1 * QT + q] - ZP[(long)m2 * QT + q];
    dz[q] = -0.5 * d * d;
  }
  const long per = ((N + S - 1) / S + 3) / 4 * 4;       // slices start at multiples of four points (the layout of LE)
  const long n0 = slice * per, n1 = min(N, n0 + per);
  const ExpTab xt = exp_tab_lane();
  unsigned o1 = 32u * (unsigned)m1, o2 = 32u * (unsigned)m2;
  double acc0 = 0.0, acc1 = 0.0;
  long n = n0;
  // (r06: the eight LE values of the NEXT trip requested at the top of the current one -- 16 more VGPRs, seven waves per SIMD instead of eight -- measured slower:
  // 13.7 -> 14.3 ms per 1e5 points at Q = 10, 15.6 -> 23.1 at Q = 16: eight waves already hide that latency; profiles/r06_gplvm_experiments.txt)
  for (; n + 4 <= n1; n += 4) {
    double e[4];
    {
      // wave-uniform group base + 32-bit BYTE offsets of the lane: scalar-base addressing (global_load v, voff, s[base]), no per-lane 64-bit address
      // arithmetic (the offsets pass through an empty asm, otherwise base + offset is hoisted out of the loop as a per-lane 64-bit pointer again).
      // n is a multiple of four (the slices are): the four points' values of a column are 32 contiguous bytes (le_index)
      const char* grp = reinterpret_cast<const char*>(LE + (n >> 2) * Mp * 4);
      asm volatile("" : "+v"(o1), "+v"(o2));
      const double2 a0 = *reinterpret_cast<const double2*>(grp + o1), a1 = *reinterpret_cast<const double2*>(grp + o1 + 16);
#ifdef GPARML_PAIRS_HALFLOADS   // timing build (WRONG results): the column side's two loads dropped -- what would two instead of four vector loads per trip buy?
      const double2 b0 = a0, b1 = a1;
#else
      const double2 b0 = *reinterpret_cast<const double2*>(grp + o2), b1 = *reinterpret_cast<const double2*>(grp + o2 + 16);
#endif
      e[0] = a0.x + b0.x; e[1] = a0.y + b0.y; e[2] = a1.x + b1.x; e[3] = a1.y + b1.y;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double* v = V2P + (n + u) * QT;      // wave-uniform
#pragma unroll
      for (int q = 0; q < QT; ++q) e[u] = fma(v[q], dz[q], e[u]);
    }
#ifdef GPARML_PAIRS_FEXP      // timing build: the table-free exp (17 instead of 12 FP64 instructions, no ds_bpermute): is the loop bound by the LDS crossbar?
    acc0 += fexp(e[0]) + fexp(e[2]);
    acc1 += fexp(e[1]) + fexp(e[3]);
#else
    acc0 += fexp_t(e[0], xt) + fexp_t(e[2], xt);
    acc1 += fexp_t(e[1], xt) + fexp_t(e[3], xt);
#endif
  }
  for (; n < n1; ++n) {
    double e = LE[le_index(true, n, m1, Mp)] + LE[le_index(true, n, m2, Mp)];
    const double* v = V2P + n * QT;
#pragma unroll
    for (int q = 0; q < QT; ++q) e = fma(v[q], dz[q], e);
    acc0 += fexp_t(e, xt);
  }
  part[((long)slice * T + tile) * 256 + threadIdx.x] = acc0 + acc1;
}

__global__ void __launch_bounds__(256) psi2_reduce_kernel(const double* __restrict__ part, const int* __restrict__ ptiles, int T, int S,
                                                          int M, int Mp, double* __restrict__ Psi2) {
  const int tile = blockIdx.x;
  const int I = ptiles[2 * tile], J = ptiles[2 * tile + 1];
  const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
  const int m1 = I * 16 + i, m2 = J * 16 + j;
  double s = 0.0;
  for (int sl = 0; sl < S; ++sl) s += part[((long)sl * T + tile) * 256 + threadIdx.x];
  if (m1 < M && m2 < M) {
    if (I != J || m2 >= m1) {
      Psi2[(long)m1 * Mp + m2] = s;
      Psi2[(long)m2 * Mp + m1] = s;
    }
  }
}

// ---- wide latent spaces (Q >= 25): phase 1 on the matrix core.  A workgroup owns a 64 x 64 tile (I <= J) of Psi2 and an
// n-slice; its four waves take 16 columns each.  Per point: E = Z_I . ZZ_n(J)^T (K = QT) by 4x4x4 MFMAs (rows from LDS,
// staged once per workgroup; the wave's ZZ_n written to its own LDS slab per point), then acc += exp(E + LEA[n,m] + LEA[n,m'])
// on the result registers.  The accumulators stay in registers for the whole slice.
template <int QT>
__global__ void __launch_bounds__(256, 2) psi2_pairs_mfma_kernel(const double* __restrict__ LEA, const double* __restrict__ V2P,
                                                                 const double* __restrict__ ZP, const int* __restrict__ tiles64, long N, int Mp,
                                                                 int S, double* __restrict__ part, int T) {
  constexpr int NQ = QT / 4, LDZ = QT <= 32 ? 34 : 66, ZPL = (16 * QT + 63) / 64;
  __shared__ double zr[64 * LDZ];                  // rows of tile I (the A operand), fixed
  __shared__ double zzs[4][16 * LDZ];              // per wave: ZZ_n of its 16 columns (written and read by that wave only, in order)
  const int tile = blockIdx.x, slice = blockIdx.y;
  const int I = tiles64[2 * tile], J = tiles64[2 * tile + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 3, lb = (lane >> 2) & 3, lk = lane >> 4;
  const int r0 = 64 * I, c0 = 64 * J + 16 * wave;
  for (int e = tid; e < 64 * QT; e += 256) zr[(e / QT) * LDZ + (e % QT)] = ZP[(long)r0 * QT + e];
  // this lane's share of the wave's 16 x QT column block of Z (element e = lane + 64 i)
  double zc[ZPL];
#pragma unroll
  for (int i = 0; i < ZPL; ++i) { const int e = lane + 64 * i; zc[i] = e < 16 * QT ? ZP[(long)c0 * QT + e] : 0.0; }
  __syncthreads();
  const long per = (N + S - 1) / S;
  const long na = slice * per, nb = min(N, na + per);
  double acc[4][4];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cq = 0; cq < 4; ++cq) acc[rb][cq] = 0.0;
  const int aofs = (4 * lb + li) * LDZ + lk, bofs = li * LDZ + lk;
  // operands of the first point
  double v2[ZPL], lr[4], lc[4];
  if (na < nb) {
#pragma unroll
    for (int i = 0; i < ZPL; ++i) { const int e = lane + 64 * i; v2[i] = e < 16 * QT ? V2P[na * QT + (e % QT)] : 0.0; }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) lr[rb] = LEA[na * Mp + r0 + 16 * rb + 4 * lb + lk];
#pragma unroll
    for (int cq = 0; cq < 4; ++cq) lc[cq] = LEA[na * Mp + c0 + 4 * cq + li];
  }
  for (long n = na; n < nb; ++n) {
    double* zzw = zzs[wave];
#pragma unroll
    for (int i = 0; i < ZPL; ++i) { const int e = lane + 64 * i; if (e < 16 * QT) zzw[(e / QT) * LDZ + (e % QT)] = v2[i] * zc[i]; }
    double lrc[4], lcc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { lrc[k] = lr[k]; lcc[k] = lc[k]; }
    if (n + 1 < nb) {                              // the next point's operands travel while this point computes
#pragma unroll
      for (int i = 0; i < ZPL; ++i) { const int e = lane + 64 * i; v2[i] = e < 16 * QT ? V2P[(n + 1) * QT + (e % QT)] : 0.0; }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) lr[rb] = LEA[(n + 1) * Mp + r0 + 16 * rb + 4 * lb + lk];
#pragma unroll
      for (int cq = 0; cq < 4; ++cq) lc[cq] = LEA[(n + 1) * Mp + c0 + 4 * cq + li];
    }
    double E[4][4];      // the first k-step's MFMAs take C = 0 as an inline constant (mfma444_zero): no zeroing moves next to asm MFMAs
    {
      // explicit ds_read_b64 operand reads with counted waits, asm MFMAs (see psi2_cols_mfma_kernel: hipcc's ds_read2_b64 merge
      // makes the A reads 2-way bank conflicts)
      const unsigned aA = lds_byte_addr(zr) + 8u * (unsigned)aofs, aB = lds_byte_addr(zzw) + 8u * (unsigned)bofs;
      double av[2][4], bv[2][4];
      auto rd = [&](auto kc, double (&a_)[4], double (&b_)[4]) {
        constexpr int k4 = decltype(kc)::value;
        static_for<0, 4>([&](auto rc) { constexpr int rb = decltype(rc)::value; a_[rb] = ds_read64<(16 * rb * LDZ + 4 * k4) * 8>(aA); });
        static_for<0, 4>([&](auto cc) { constexpr int cq = decltype(cc)::value; b_[cq] = ds_read64<(4 * cq * LDZ + 4 * k4) * 8>(aB); });
      };
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's own ZZ_n stores above have landed
      rd(IC<0>{}, av[0], bv[0]);
      static_for<0, NQ>([&](auto kc) {
        constexpr int k4 = decltype(kc)::value, cur = k4 & 1;
        if constexpr (k4 + 1 < NQ) { rd(IC<k4 + 1>{}, av[cur ^ 1], bv[cur ^ 1]); lgkm_wait<8>(); }
        else lgkm_wait<0>();
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int cq = 0; cq < 4; ++cq) {
            if constexpr (k4 == 0) mfma444_zero(E[rb][cq], av[cur][rb], bv[cur][cq]);
            else mfma444_acc(E[rb][cq], av[cur][rb], bv[cur][cq]);
          }
      });
      mfma_drain(E[3][3]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc_fence<4>(E[rb]);
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cq = 0; cq < 4; ++cq) acc[rb][cq] += fexp(E[rb][cq] + lrc[rb] + lcc[cq]);
  }
  // partial tile of this slice, in the register layout: element (rb, cq) of lane l is row 16 rb + 4 lb + lk, column 16 wave + 4 cq + li
  double* dst = part + ((long)slice * T + tile) * 4096;
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cq = 0; cq < 4; ++cq) dst[(16 * rb + 4 * lb + lk) * 64 + 16 * wave + 4 * cq + li] = acc[rb][cq];
}

__global__ void __launch_bounds__(256) psi2_reduce64_kernel(const double* __restrict__ part, const int* __restrict__ tiles64, int T, int S,
                                                            int M, int Mp, double* __restrict__ Psi2) {
  const int tile = blockIdx.x;
  const int I = tiles64[2 * tile], J = tiles64[2 * tile + 1];
  for (int e = threadIdx.x; e < 4096; e += 256) {
    const int m1 = 64 * I + (e >> 6), m2 = 64 * J + (e & 63);
    double s = 0.0;
    for (int sl = 0; sl < S; ++sl) s += part[((long)sl * T + tile) * 4096 + e];
    if (m1 < M && m2 < M && (I != J || m2 >= m1)) {
      Psi2[(long)m1 * Mp + m2] = s;
      Psi2[(long)m2 * Mp + m1] = s;
    }
  }
}

// Psi2's padding (rows / columns M .. Mp - 1): the global step's products run over Mp, and the reduce kernels above write the M x M block only.  Until r06 these
// zeros were whatever the allocation left (the poison run found it: K_mm^-1 Psi2 all NaN); one small launch per evaluation, nothing when M = Mp.
__global__ void __launch_bounds__(256) psi2_pad_zero_kernel(double* __restrict__ Psi2, int M, int Mp) {
  const long total = (long)Mp * Mp;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int r = (int)(i / Mp), cidx = (int)(i - (long)r * Mp);
    if (r >= M || cidx >= M) Psi2[i] = 0.0;
  }
}
int psi2_zero_pads(gp_ctx* c) {
  if (c->M == c->Mp) return GP_OK;
  hipLaunchKernelGGL(psi2_pad_zero_kernel, dim3((unsigned)std::min<long>(((long)c->Mp * c->Mp + 255) / 256, 1024)), dim3(256), 0, c->stream, c->stats, c->M, c->Mp);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

// ---------------------------------------------------------------------------------------------- phase 2
// Bbar with four rows interleaved: element (m, m') at ((m / 4) Mp + m') 4 + m % 4.  The phase-2 kernels below walk down the rows of a lane's column four (two)
// at a time: one 16-byte load per TWO rows instead of an 8-byte load per row -- the vector-memory instruction count is what the texture addresser prices
// (16 cycles per wave-instruction whatever its width; profiles/r06_gplvm_experiments.txt items 16, 17).
__global__ void __launch_bounds__(256) bbar_interleave_kernel(const double* __restrict__ Bbar, int Mp, double* __restrict__ B4) {
  const long total = (long)Mp * Mp;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const long m = i / Mp, mc = i - m * Mp;
    B4[(((m >> 2) * Mp + mc) << 2) + (m & 3)] = Bbar[i];
  }
}

// T_n = Bbar o psi2_n (symmetric M x M per point), r_n = T_n 1, t_n = T_n Z give the psi2 parts of every gradient:
//   grad_Z psi2 part  G[m,k] += -a_k z_mk r + a_k t_k + w_k (2 mu_k r - z_mk r - t_k)          (partial_terms.py:190-205, x2 at :238)
//   per point: sr, zr_q, z2r_q, zt_q -> quad = 4 mu^2 sr - 8 mu zr + 2 z2r + 2 zt
//   grad_alpha += -1/4 quad/d2^2 - (S/d2) sr ; grad_X_mu += -w (2 mu sr - 2 zr) ; grad_X_S += 1/2 w^2 quad - w sr
//                                                                               (partial_terms.py:273-284, 388-394, 421-427)
// Layout: a wave owns 64 inducing COLUMNS m' (one per lane) and walks over points n and rows m, both wave-uniform:
//   exponent(n; m, m') = LEA[n,m] + LEA[n,m'] + sum_q z_mq * (-2 V_nq z_m'q)
// so the row operands (z_m, LEA[n,m]) are scalar loads, the column operands (z_m', zz = -2 V_n z_m', LEA[n,m']) live in
// the lane's registers, Bbar[m][m'] is one coalesced load per step, and by the symmetry of T the lane accumulates its own
// column sums r[m'] = sum_m T, t[m'][q] = sum_m T z_mq with no cross-lane traffic.  After the M rows of a point: grad_Z of
// the lane's column accumulates in registers over all points; the per-point sums are wave-reduced, combined over the
// workgroup's four column slabs in LDS and written once per (point, slab group).  Nothing is re-streamed per row block
// and the register need is 8 QT (KEEP) or 4 QT (!KEEP: z_m' re-read per point, grad_Z accumulated in memory) VGPRs.
struct PB2Args {
  const double* Wn; const double* mu; const double* S; const double* alpha;
  double* Gpart; double* gapart2; double* gmu; double* gS; double* pp;
  long N, Np; int M, Mp, Q, QB, nslab, ppb, ngrp;   // nslab = ceil(M/64) column slabs in ngrp groups of <= 4; ppb points per workgroup
};

// the lane's index inside its wave from the execution mask (v_mbcnt): no register has to carry threadIdx.x through a kernel's hot loop for the few places
// behind it that need the lane (psi2_cols_kernel<10, true> spilled it: the library's last scratch allocation, r06)
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }


// Four waves per SIMD (128 VGPRs) up to QT = 14: r06, same box, phase 2 per 1e5 points at M = 512: Q = 13 / 14 44.7 / 45.1 -> 42.1 / 42.5 ms (the cap costs QT = 12 / 14
// 20 / 28 B of scratch per lane -- values spilled in the prologue and reloaded once per POINT, outside the row loop -- and buys a fourth wave: FP64 issue 6.0 -> 5.6
// cycles); QT = 16 gains nothing from it (47.6 -> 47.2; re-measured with the reduce-scatter sums: 47.7 -> 47.4 at M = 512 with 12 B of scratch, 3.96 -> 4.12
// at M = 128) and keeps its 164 registers without scratch.
// which instantiations of psi2_cols_kernel read the row-interleaved Bbar (QT = 8 came out with a 36-byte scratch allocation with it and keeps the plain table)
__host__ __device__ constexpr bool cols_b4(int QT) { return QT <= 10 && QT != 8; }
template <int QT, bool KEEP>
__global__ void __launch_bounds__(256, QT <= 14 ? 4 : 2) psi2_cols_kernel(PB2Args a, const double* __restrict__ ZP, const double* __restrict__ Bbar,
                                                        const double* __restrict__ LEA, const double* __restrict__ V2P,
                                                        const double* __restrict__ WP, const double* __restrict__ MUP,
                                                        const double* __restrict__ alphaP) {
  constexpr int PW = 3 * QT + 1;
  __shared__ double red[4][PW];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (the lane index comes from lane_id() wherever it is needed: nothing per-lane lives across the row loop but z, g, zz, t)
  const int nw = blockDim.x >> 6;                        // waves per workgroup = min(4, nslab)
  const int slab = blockIdx.y * nw + wave;
  const bool active = slab < a.nslab;                    // idle waves (nslab not a multiple of nw) only join the barriers
  const int mc0 = (active ? slab : 0) * 64;              // wave-uniform; this lane's column m' = mc0 + lane (< Mp; columns >= M: LEA = kPadLog, ZP = 0)
  double* G = a.Gpart + (long)blockIdx.x * a.M * a.Q;    // this workgroup's grad_Z partial (rows of its slabs)
  double z[KEEP ? QT : 1], g[KEEP ? QT : 1];
  if (KEEP) {
    const int mc = mc0 + lane_id();
#pragma unroll
    for (int q = 0; q < QT; ++q) { z[q] = ZP[(long)mc * QT + q]; g[q] = 0.0; }
  }
  const long n0 = (long)blockIdx.x * a.ppb, n1 = min(a.N, n0 + a.ppb);
  const int Mr = (a.M + 3) & ~3;                          // rows >= M: ZP rows are zero, LEA entries kPadLog
  for (long n = n0; n < n1; ++n) {
    double zz[QT], t[QT], r = 0.0;
    if (active) {
      const int mc = mc0 + lane_id();
      const double* v2 = V2P + n * QT;                   // wave-uniform
#pragma unroll
      for (int q = 0; q < QT; ++q) { zz[q] = v2[q] * (KEEP ? z[q] : ZP[(long)mc * QT + q]); t[q] = 0.0; }
      const double* lrow = LEA + n * a.Mp;               // wave-uniform row of this point
      const double lea = lrow[mc];
      constexpr bool B4 = cols_b4(QT);                   // Bbar with four rows interleaved (launch_cols passes that table): one 16-byte load per two rows; from QT = 12 on
      const double* bcol = Bbar + (B4 ? 4 * mc : mc);    // (two rows per trip) it measured 1.2-1.4 % slower than the plain row-major table: profiles/r06_gplvm_experiments.txt item 17
      constexpr int U = QT <= 10 ? 4 : 2;   // rows per trip: U z-rows (2 QT SGPRs each) must fit the scalar file
      // (r06: Bbar of the next trip's rows requested one trip ahead, as psi2_sym_kernel does, for QT > 10 where registers are to spare: slower -- phase 2
      // 42.2 -> 43.6 / 45.7 -> 47.9 / 48.2 -> 65.9 ms per 1e5 points at Q = 12 / 14 / 16; the loop is bound by FP64 issue, not by that latency)
      for (int m = 0; m < Mr; m += U) {
        double bb[U];
        if constexpr (B4) {
          const double* b4 = bcol + (long)(m >> 2) * a.Mp * 4;                  // m is a multiple of four
          const double2 x = *reinterpret_cast<const double2*>(b4), y = *reinterpret_cast<const double2*>(b4 + 2);
          bb[0] = x.x; bb[1] = x.y; bb[U - 2] = y.x; bb[U - 1] = y.y;
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) bb[u] = bcol[(long)(m + u) * a.Mp];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const double* zm = ZP + (long)(m + u) * QT;    // wave-uniform
          double e = lrow[m + u] + lea;
#pragma unroll
          for (int q = 0; q < QT; ++q) e = fma(zm[q], zz[q], e);
          const double T = bb[u] * fexp(e);
          r += T;
#pragma unroll
          for (int q = 0; q < QT; ++q) t[q] = fma(T, zm[q], t[q]);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < QT; ++q) { zz[q] = 0.0; t[q] = 0.0; }
    }
    // grad_Z of this lane's column and the per-point sums
    const int lane = lane_id(), mc = mc0 + lane;
    const double* wn = WP + n * QT;                      // wave-uniform
    const double* mun = MUP + n * QT;
    // The wave's 3 QT + 1 sums as reduce-scatters (lane_reduce.h, r06): about one cross-lane move and one add per value where butterfly sums take six of each, and
    // one LDS store by the lanes that own a sum.  QT <= 10: two passes of half the latent dimensions (phase 2 at M = 128, Q = 10: 2.75 -> 2.56 ms per 1e5 points,
    // M = 1024 neutral).  QT >= 12: passes of TWO dimensions -- with QT / 2 per pass the values next to t[] changed the allocation of the row loop (M = 512, Q = 16:
    // 47.3 -> 50.7 ms); with two: Q = 16 47.85 -> 47.3, Q = 14 42.5 -> 42.9, Q = 13 unchanged, M = 128 / Q = 14 3.80 -> 3.68.  Either way no instantiation of this
    // kernel has a scratch allocation any more (<10, true> 16 B, <12 / 14, false> 20 / 28 B before).  profiles/r06_gplvm_experiments.txt item 15.
    constexpr int QC = QT <= 10 ? QT / 2 : 2;
    static_for<0, QT / QC>([&](auto cc) {
      constexpr int q0 = decltype(cc)::value * QC;
      constexpr bool first = q0 == 0;
      constexpr int NV = 3 * QC + (first ? 1 : 0), O = first ? 1 : 0;
      double v[NV];
      if (first) v[0] = r;
#pragma unroll
      for (int qi = 0; qi < QC; ++qi) {
        const int q = q0 + qi;
        const double zq = KEEP ? z[KEEP ? q : 0] : ZP[(long)mc * QT + q];
        const double gq = -alphaP[q] * (zq * r - t[q]) + wn[q] * (2.0 * mun[q] * r - zq * r - t[q]);
        if (KEEP) g[KEEP ? q : 0] += gq;
        else if (active && mc < a.M && q < a.Q) { double* dst = G + (long)mc * a.Q + q; *dst = ((n == n0) ? 0.0 : *dst) + gq; }
        v[O + qi] = zq * r; v[O + QC + qi] = zq * zq * r; v[O + 2 * QC + qi] = zq * t[q];
      }
      const double tot = wave_reduce_scatter<NV>(v, lane);
      const int j = lane - O, kq = (j >= 2 * QC) ? 2 : (j >= QC ? 1 : 0);
      const int row = (first && lane == 0) ? 0 : 1 + kq * QT + q0 + (j - kq * QC);
      if (lane < NV) red[wave][row] = tot;
    });
    __syncthreads();
    for (int i = 64 * wave + lane; i < PW; i += blockDim.x) {
      double sum = red[0][i];
      for (int w = 1; w < nw; ++w) sum += red[w][i];
      a.pp[((long)blockIdx.y * PW + i) * a.Np + n] = sum;
    }
    __syncthreads();
  }
  const int mcf = mc0 + lane_id();
  if (KEEP && active && mcf < a.M) {
#pragma unroll
    for (int q = 0; q < QT; ++q) if (q < a.Q) G[(long)mcf * a.Q + q] = g[q];
  }
}

// ---- the same on tile PAIRS (Q <= 10, M <= 1024): T_n is symmetric, so a 64 x 64 tile (I < J) of it serves the column sums of
// slab J AND the row sums of slab I; only the exponent, the exp and the product with Bbar -- half of the work of the column
// kernel above -- are shared, the Q + 1 accumulations per pair are needed on both sides.
//   column side: as above, lane = column of slab J, rows of slab I as scalar operands (2 Q + 20 issue slots per pair);
//   row side:    r_m, t_m[q] = sum over the lanes -- a reduction across the wave, done on the matrix core: the T values of four
//                rows are transposed inside each lane quad (quad_perm moves), which makes them the A operand of
//                v_mfma_f64_4x4x4_4b (A_b[i][k] = T[m0 + i][16 k + 4 b + v]); B_b[k][j] = [Z | 1][16 k + 4 b + v][4 qq + j] comes
//                from registers; four blocks b x four v cover the 64 columns, the block partials are added with two row
//                rotations.  4 (Q + 2) / 4 MFMAs per 4 rows x 64 columns = the VALU cost of the accumulation, no LDS traffic.
// Both sides add into one LDS array rt[M][Q + 2] per point.  The tiles of a point are scheduled as a round-robin tournament over
// the slabs (a round = disjoint slab pairs, one per wave, rounds separated by a barrier), so no two waves ever touch the same rows
// of rt at the same time: plain read-add-write in a fixed order, no atomics, results bit-identical from run to run.  When all
// rounds are done every thread finishes its rows (grad_Z of the row, the per-point sums) exactly as the column kernel does.
// r06, latent width 12 (Q = 11, 12): the kernel as it stands for QT <= 10 needs 168 VGPRs + 100-132 B of scratch there (4 NQ = 16 operand registers of the row
// side, 2 x 24 registers of zz / t) and 64 KB of LDS (two workgroups per CU): measured no faster than the column kernel (40.9 against 41.0 ms per 1e5 points), and
// 14 / 16 slower (52.7 / 64.1 against 45.1 / 47.7).  What made 12 work (35.3 against 42.5 ms, Q = 11: 34.6 against 42.2): (1) rt rows QT + 1 doubles apart
// (52 KB: three workgroups per CU); (2) at QT = 12 the ones column opens a feature quad of its own, [1 0 0 0] for every column -- ONE operand register built from
// the lane id instead of four loaded ones; (3) no one-group-ahead request of Bbar; (4) the four rows of a group two at a time (a scheduling barrier between the
// pairs: four interleaved exp chains do not fit); (5) the per-point finish in two passes of six latent dimensions (3 x 12 running sums in one pass spilled).
// 128 VGPRs, no scratch.  14 and 16 stay on the column kernel: 15 / 17 doubles per row of rt are two workgroups per CU whatever the registers do
// (profiles/r06_gplvm_experiments.txt).
// the "register diet" (r06): compact rt rows, no one-group-ahead request of Bbar, the four rows of a group two at a time, the finish in two passes.  QT = 12 needs it to
// exist at all; QT = 8 gets a fourth workgroup per CU from it (rt 36 KB, 96 VGPRs): phase 2 30.5 -> 28.0 ms per 1e5 points at Q = 8, 30.5 -> 27.5 at Q = 7 (same box).
// QT <= 6: a fifth workgroup per CU (28.7 KB of rt, 82 VGPRs): 24.1 -> 23.4 ms at Q = 6.  QT = 10 -- the one width where compact rows do not buy a workgroup (45 KB) -- kept the
// round-3 layout (one-group-ahead request of Bbar, padded rows, one finish pass: 167 VGPRs) while the diet's pieces measured neutral there; with Bbar read through the
// row-interleaved table (one 16-byte load per two rows) they pay: 114 VGPRs, no scratch, phase 2 30.3 -> 29.5 ms per 1e5 points at Q = 10, 30.4 -> 29.2 at Q = 9 (same box,
// profiles/r06_gplvm_experiments.txt item 18).  Every width runs the diet now.
__host__ __device__ constexpr int sym_rs(int QT) { return QT + 1; }

template <int QT>
__global__ void __launch_bounds__(512, QT == 8 ? 4 : (QT <= 6 ? 5 : 3)) psi2_sym_kernel(PB2Args a, const double* __restrict__ ZP, const double* __restrict__ Z1S,
                                                       const double* __restrict__ Bbar, const double* __restrict__ LEA,
                                                       const double* __restrict__ V2P, const double* __restrict__ WP,
                                                       const double* __restrict__ MUP, const double* __restrict__ alphaP,
                                                       const int* __restrict__ sched, int nrounds) {
  // RT: feature columns of the row-side MFMAs ([Z | 1] padded to four); RS: row stride of rt -- RT up to QT = 10, QT + 1 at QT = 12 (52 KB at M = 512: three
  // workgroups per CU; with the padded stride two).  ONESQ (QT = 12): the ones column opens a feature quad of its own, [1 0 0 0] -- the same B operand for every
  // column, so it is ONE register built from the lane id instead of four loaded ones, and the kernel keeps its 3 x 4 operand registers of Z like QT = 10.
  constexpr int RT = (QT + 1 + 3) / 4 * 4, NQ = RT / 4, PW = 3 * QT + 1, RS = sym_rs(QT);
  constexpr bool ONESQ = (QT % 4) == 0;
  extern __shared__ double smem[];
  double* rt = smem;                       // [Mp][RS]   t_m[q] (q < QT), r_m (index QT) of the current point
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  double* G = a.Gpart + (long)blockIdx.x * a.M * a.Q;
  const long n0 = (long)blockIdx.x * a.ppb, n1 = min(a.N, n0 + a.ppb);
  for (int i = tid; i < a.Mp * RS; i += blockDim.x) rt[i] = 0.0;
  __syncthreads();
  for (long n = n0; n < n1; ++n) {
    const double* v2 = V2P + n * QT;                   // wave-uniform
    const double* lrow = LEA + n * a.Mp;
    for (int rd = 0; rd < nrounds; ++rd) {
      const int code = sched[rd * nw + wave];          // wave-uniform: I | J << 16, -1 = idle
      if (code >= 0) {
        // lane coordinates from an opaque copy of the lane id at their point of use: as loop invariants they were spilled (32 B of scratch per lane)
        int le = lane;
        asm volatile("" : "+v"(le));
        const int lq = le & 3, lb = (le >> 2) & 3, lk = le >> 4;
        const int I = code & 0xffff, J = code >> 16;
        const bool offd = I != J;
        const int mc = 64 * J + lane;
        double zz[QT], t[QT], r = 0.0;
#pragma unroll
        for (int q = 0; q < QT; ++q) { zz[q] = v2[q] * ZP[(long)mc * QT + q]; t[q] = 0.0; }
        const double lea = lrow[mc];     // (r06 timing build with these per-tile loads dropped: no faster -- their latency is hidden behind the other waves)
        const double* bcol = Bbar + 4 * mc;                // Bbar4: four rows of this column side by side
        // the row side's B operand [Z | 1] of slab J: 4 NQ values per lane, in registers for the whole tile
        double ZB[4][NQ];
        const double* zbp = Z1S + (long)(64 * J + 16 * lk + 4 * lb) * RT + lq;
        if (offd) {
          const double onesq = (lq == 0) ? 1.0 : 0.0;
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq) ZB[v][qq] = (ONESQ && qq == NQ - 1) ? onesq : zbp[v * RT + 4 * qq];
        }
        for (int g = 0; g < 16; ++g) {
          const int m0 = 64 * I + 4 * g;
          double bb[4], T[4];
#pragma unroll
          for (int u = 0; u < 4; u += 2) { const double2 x = *reinterpret_cast<const double2*>(bcol + (long)(m0 >> 2) * a.Mp * 4 + u); bb[u] = x.x; bb[u + 1] = x.y; }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const double* zm = ZP + (long)(m0 + u) * QT;    // wave-uniform: scalar loads (an explicit one-row-ahead request of
            double e = lrow[m0 + u] + lea;                  // these operands made hipcc spill SGPRs through v_writelane: +20 %)
#pragma unroll
            for (int q = 0; q < QT; ++q) e = fma(zm[q], zz[q], e);
            T[u] = bb[u] * fexp(e);
            r += T[u];
#pragma unroll
            for (int q = 0; q < QT; ++q) t[q] = fma(T[u], zm[q], t[q]);
            // QT = 12: the four rows two at a time (four interleaved exp chains with their temporaries do not fit next to 2 x 24 + 26 long-lived registers)
            if (u == 1) __builtin_amdgcn_sched_barrier(0);
          }
          if (offd) {
            double acc[NQ];
            wave_rows_times_features<NQ>(T, ZB, acc);
            if (lb == 0) {
              double* dst = rt + (m0 + lk) * RS + lq;
#pragma unroll
              for (int qq = 0; qq < NQ; ++qq) if (4 * qq + lq <= QT) dst[4 * qq] += acc[qq];      // (compact stride: features beyond the ones column do not exist)
            }
          }
        }
        // column side of this tile into rt (rows of slab J)
        double* dst = rt + mc * RS;
#pragma unroll
        for (int q = 0; q < QT; ++q) dst[q] += t[q];
        dst[QT] += r;
      }
      __syncthreads();
    }
    // ---- finish the point: every thread takes rows tid, tid + blockDim.x, ... (and clears them for the next point)
    const double* wn = WP + n * QT;                      // wave-uniform
    const double* mun = MUP + n * QT;
    // QC latent dimensions at a time (all of them up to QT = 10; six at QT = 12, where the 3 QT running sums of one pass did not fit the registers): the rows' r stays
    // in place until the last pass
    constexpr int QC = QT / 2;
    static_assert(QT % QC == 0, "q chunks");
    double* ppw = a.pp + (long)wave * PW * a.Np + n;
    static_for<0, QT / QC>([&](auto cc) {
      constexpr int q0 = decltype(cc)::value * QC;
      constexpr bool first = q0 == 0, last = q0 + QC == QT;
      double s0 = 0.0, s1[QC], s2[QC], s3[QC];
#pragma unroll
      for (int q = 0; q < QC; ++q) { s1[q] = 0.0; s2[q] = 0.0; s3[q] = 0.0; }
      for (int m = tid; m < a.Mp; m += blockDim.x) {
        double* src = rt + m * RS;
        const double r = src[QT];
        if (last) src[QT] = 0.0;
        if (first) s0 += r;
#pragma unroll
        for (int qi = 0; qi < QC; ++qi) {
          const int q = q0 + qi;
          const double tq = src[q];
          src[q] = 0.0;
          const double zq = ZP[(long)m * QT + q];
          const double gq = -alphaP[q] * (zq * r - tq) + wn[q] * (2.0 * mun[q] * r - zq * r - tq);
          if (m < a.M && q < a.Q) { double* d = G + (long)m * a.Q + q; *d = ((n == n0) ? 0.0 : *d) + gq; }
          s1[qi] = fma(zq, r, s1[qi]); s2[qi] = fma(zq * zq, r, s2[qi]); s3[qi] = fma(zq, tq, s3[qi]);
        }
      }
      // the waves' sums go to pp as they are, one group per wave (a.ngrp = waves): psi2_points_finish_kernel adds the groups in wave order, and the [waves][PW]
      // LDS array of r03-r05 with its second barrier is gone (r06).  The sums across the wave as ONE reduce-scatter of the pass's 3 QC (+ 1) values (lane_reduce.h):
      // about one cross-lane move and one add per value where 3 QC + 1 butterfly sums took six of each (372 ds_bpermute per point at QT = 10), and one store
      // by the lanes that end up owning a sum instead of 3 QC + 1 single-lane stores.
      constexpr int NV = 3 * QC + (first ? 1 : 0);
      double v[NV];
      if (first) v[0] = s0;
#pragma unroll
      for (int qi = 0; qi < QC; ++qi) { v[(first ? 1 : 0) + qi] = s1[qi]; v[(first ? 1 : 0) + QC + qi] = s2[qi]; v[(first ? 1 : 0) + 2 * QC + qi] = s3[qi]; }
      const double tot = wave_reduce_scatter<NV>(v, lane);
      const int j = lane - (first ? 1 : 0), kq = (j >= 2 * QC) ? 2 : (j >= QC ? 1 : 0);
      const int row = (first && lane == 0) ? 0 : 1 + kq * QT + q0 + (j - kq * QC);
      if (lane < NV) ppw[(long)row * a.Np] = tot;
    });
    __syncthreads();       // every row of rt has been read and cleared before the next point's tiles add to it
  }
}

// per-point finish of the psi2 part from the running sums pp[n] = [sr, zr_q, z2r_q, zt_q]
__global__ void __launch_bounds__(256) psi2_points_finish_kernel(PB2Args a) {
  __shared__ double redq[256];
  for (int q = 0; q < a.Q; ++q) {
    double ga = 0.0;
    for (long n = blockIdx.x * 256L + threadIdx.x; n < a.N; n += (long)gridDim.x * 256L) {
      double sr = 0.0, zr = 0.0, z2r = 0.0, zt = 0.0;
      for (int sp = 0; sp < a.ngrp; ++sp) {                     // one slab of sums per group of column slabs (workgroup row)
        const double* ppn = a.pp + (long)sp * (3 * a.QB + 1) * a.Np + n;
        sr += ppn[0]; zr += ppn[(long)(1 + q) * a.Np]; z2r += ppn[(long)(1 + a.QB + q) * a.Np]; zt += ppn[(long)(1 + 2 * a.QB + q) * a.Np];
      }
      const double s = a.S[n * a.Q + q], al = a.alpha[q], w = a.Wn[n * a.Q + q], mu = a.mu[n * a.Q + q];
      const double d2 = 2.0 * al * s + 1.0;
      const double quad = 4.0 * mu * mu * sr - 8.0 * mu * zr + 2.0 * z2r + 2.0 * zt;
      ga += -0.25 * quad / (d2 * d2) - (s / d2) * sr;
      a.gmu[n * a.Q + q] += -w * (2.0 * mu * sr - 2.0 * zr);
      a.gS[n * a.Q + q] += 0.5 * w * w * quad - w * sr;
    }
    redq[threadIdx.x] = ga;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) redq[threadIdx.x] += redq[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) a.gapart2[(long)blockIdx.x * a.Q + q] = redq[0];
    __syncthreads();
  }
}

// stage 1 of the grad_Z partial sum: tmp[s][e] = sum over the s-th share of the workgroup partials (coalesced 512-byte rows)
__global__ void __launch_bounds__(256) pb2_reduce1_kernel(const double* __restrict__ Gpart, int nb, long E, int S2, double* __restrict__ tmp) {
  __shared__ double red[4][64];
  const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long e = blockIdx.x * 64L + el;
  const int per = (nb + S2 - 1) / S2;
  const int b0 = blockIdx.y * per, b1 = min(nb, b0 + per);
  double s = 0.0;
  if (e < E) for (int b = b0 + g; b < b1; b += 4) s += Gpart[(long)b * E + e];
  red[g][el] = s;
  __syncthreads();
  if (g == 0 && e < E) tmp[(long)blockIdx.y * E + e] = red[0][el] + red[1][el] + red[2][el] + red[3][el];
}

// grads[0:M*Q] += sum_s tmp[s] ; grads[M*Q + q] += sum_blocks gapart2
__global__ void __launch_bounds__(256) pb2_reduce_kernel(const double* __restrict__ Gpart, const double* __restrict__ gapart2, int nb, int nb2,
                                                         long MQ, int Q, double* __restrict__ grads) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < MQ + Q; i += (long)gridDim.x * 256L) {
    double s = 0.0;
    if (i < MQ) for (int b = 0; b < nb; ++b) s += Gpart[(long)b * MQ + i];
    else for (int b = 0; b < nb2; ++b) s += gapart2[(long)b * Q + (i - MQ)];
    grads[i] += s;
  }
}

template <typename T>
static int balloc(gp_ctx* c, T** p, size_t count, int mode = DA_INIT) {
  return dalloc_bytes(c, (void**)p, count * sizeof(T), mode);
}

int ensure_regime_b_buffers(gp_ctx* c) {
  if (c->b_alloc) return GP_OK;
  const long Np = c->Np, Mp = c->Mp, M = c->M, Q = c->Q;
  int rc = GP_OK;
  auto A = [&](auto** p, size_t n) { if (rc == GP_OK) rc = balloc(c, p, n); };
  // compiled latent widths; from 25 on the phase-2 kernel is the MFMA one and needs a spare column (QB > Q) for the ones.  r04: 6 and 8 next to 10 -- every
  // pair of every point pays 2 QB + 20 issue slots whatever Q is (N = 1e5, M = 512, same box: Q = 5, 6: 44.6 -> 39.0 ms per evaluation; Q = 7, 8: -4 %, the
  // tile-pair kernel's row tables are 12 wide at 8 as at 10); 12 and 14 next to 16 (the column kernel): Q = 12: 63.9 -> 57.1 ms, Q = 14: 65.8 -> 61.8 ms
  c->QB = Q <= 4 ? 4 : Q <= 6 ? 6 : Q <= 8 ? 8 : Q <= 10 ? 10 : Q <= 12 ? 12 : Q <= 14 ? 14 : Q <= 16 ? 16 : Q <= 24 ? 24 : Q <= 31 ? 32 : Q <= 51 ? 52 : 64;
  if (b_generic(c)) c->QB = (int)Q;       // psi2_generic.hip: the tables are exactly Q wide
  c->b_mfma = Q >= 25 && Q < c->QB;
  A(&c->LE, (size_t)Np * Mp); A(&c->LET, (size_t)Np * Mp); A(&c->Vn, (size_t)Np * Q); A(&c->Wn, (size_t)Np * Q);
  A(&c->ZP, (size_t)Mp * c->QB); A(&c->Z1P, (size_t)Mp * c->QB);
  // zero contract: columns Q .. QB - 1 of the per-point tables and of alphaP are never written (b_tables_kernel fills q < Q) and every kernel runs its q loops to QB
  auto A0 = [&](auto** p, size_t n) { if (rc == GP_OK) rc = balloc(c, p, n, DA_ZERO); };
  A0(&c->V2P, (size_t)Np * c->QB); A0(&c->WP, (size_t)Np * c->QB); A0(&c->MUP, (size_t)Np * c->QB);
  A0(&c->alphaP, (size_t)c->QB);
  A(&c->lnc2h, (size_t)Np);
  // phase-2 pair kernel: grid (point chunks, groups of <= 4 64-column slabs); >= 16 points per workgroup, <= 4096 chunks
  c->nslab = (int)((M + 63) / 64);
  c->ppb = (int)std::max<long>(16, (c->N + 4095) / 4096);
  c->pb_blocks = (int)((c->N + c->ppb - 1) / c->ppb);
  A(&c->Bbar4, (size_t)Mp * Mp);
  A(&c->Gpart, (size_t)c->pb_blocks * M * Q); A(&c->gapart2, (size_t)c->pb_blocks * Q); A(&c->Gtmp, (size_t)64 * M * Q);
  std::vector<int> t;
  const int Mt = (int)((M + 15) / 16);
  for (int i = 0; i < Mt; ++i) for (int j = i; j < Mt; ++j) { t.push_back(i); t.push_back(j); }
  c->n_ptiles = (int)t.size() / 2;
  A(&c->ptiles, t.size());
  std::vector<int> t64;
  const int Mt64 = (int)((M + 63) / 64);
  for (int i = 0; i < Mt64; ++i) for (int j = i; j < Mt64; ++j) { t64.push_back(i); t64.push_back(j); }
  c->n_tiles64 = (int)t64.size() / 2;
  A(&c->tiles64, t64.size());
  // tile-pair phase 2 (psi2_sym_kernel): Q <= 10, three to sixteen 64-column slabs (two slabs = a single wave per workgroup
  // running three tiles one after the other: slower than the column kernel, configs[1] 4.9 -> 5.2 ms).  Schedule = round-robin tournament over the
  // slabs (circle method; an odd count gets a bye) followed by the diagonal tiles, one tile per wave and round.
  // ... and its per-point array rt (Mp x RT doubles of LDS per workgroup) must leave room for twelve waves per CU: with fewer
  // the scalar-operand latency is exposed and the column kernel (four waves per SIMD) is faster (M = 1024: one workgroup per CU)
  {
    const int nv = (c->nslab + 1) / 2 * 2, nw = nv / 2;
    const size_t smem = (size_t)Mp * sym_rs(c->QB) * sizeof(double);
    static const int maxq = [] { const char* e = getenv("GPARML_B_SYM_MAXQ"); return e ? atoi(e) : 12; }();     // 10: the column kernel at Q = 11, 12 (same-box A/B)
    // (>= 12 waves per CU re-measured in r06: M = 1024, Q = 10 -- one eight-wave workgroup per CU -- 75.5 ms per 5e4 points on this kernel against 69.8 on the column kernel)
    c->b_sym = !c->b_mfma && c->QB <= std::min(maxq, 12) && c->nslab >= 3 && c->nslab <= 16 && smem <= 160 * 1024 && (160 * 1024 / smem) * nw >= 12;
    c->sym_nw = nw;
  }
  // the matrix-core tile-pair phase 2 (psi2_tile.hip) wherever psi2_sym_kernel does not apply; it keeps its own per-launch sums buffer
  c->b_tile = !b_generic(c) && pt2_applicable(c, c->b_sym);
  if (c->b_tile) c->b_sym = false;
  c->pp_doubles = c->b_tile ? 1 : (size_t)Np * (3 * c->QB + 1) * (c->b_sym ? c->sym_nw : (c->nslab + 3) / 4);     // one group of sums per wave (sym) / per four slabs (cols)
  A(&c->pp, c->pp_doubles);
  std::vector<int> sch;
  if (c->b_sym) {
    const int nv = (c->nslab + 1) / 2 * 2, nw = nv / 2;
    for (int r = 0; r < nv - 1; ++r)
      for (int k = 0; k < nw; ++k) {
        const int x = (k == 0) ? nv - 1 : (r + k) % (nv - 1), y = (k == 0) ? r : (r - k + (nv - 1)) % (nv - 1);
        const int I = std::min(x, y), J = std::max(x, y);
        sch.push_back(J < c->nslab ? (I | (J << 16)) : -1);
      }
    for (int d = 0; d < c->nslab; d += nw)
      for (int k = 0; k < nw; ++k) sch.push_back(d + k < c->nslab ? ((d + k) | ((d + k) << 16)) : -1);
    c->sym_nw = nw; c->sym_rounds = (int)sch.size() / nw;
    A(&c->Z1S, (size_t)Mp * ((c->QB + 1 + 3) / 4 * 4));
    A(&c->sym_sched, sch.size());
  }
  if (rc != GP_OK) return rc;
  if (c->b_sym) GP_HIP(c, hipMemcpyAsync(c->sym_sched, sch.data(), sch.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  GP_HIP(c, hipMemcpyAsync(c->ptiles, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  GP_HIP(c, hipMemcpyAsync(c->tiles64, t64.data(), t64.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  GP_HIP(c, hipStreamSynchronize(c->stream));
  // the pair kernel's split-n partials live in c->part: make sure it is large enough
  const size_t need = std::max((size_t)c->n_ptiles * 256 * 64, (size_t)c->n_tiles64 * 4096 * 32);
  if (need > c->part_doubles) {
    (void)hipFree(c->part);
    c->part = nullptr;
    GP_TRY_RC(dalloc_bytes(c, (void**)&c->part, need * 8, DA_RAW));
    c->part_doubles = need;
  }
  c->b_alloc = true;
  return GP_OK;
}

// (z_mq - z_m'q)^2 for the compat path's per-point psi2 tensor only (the pair kernels use the padded Z tables)
int run_dz2(gp_ctx* c) {
  const long total = (long)c->M * c->M * c->Q;
  if (!c->DZ2) GP_TRY_RC(dalloc_bytes(c, (void**)&c->DZ2, (size_t)std::max<long>(total, 1) * sizeof(double), DA_RAW));
  hipLaunchKernelGGL(dz2_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, c->stream, c->Z, c->M, c->Q, c->DZ2);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

template <int QT, int CPL>
static void launch_le(gp_ctx* c) {
  dim3 grid((c->Mp + 256 * CPL - 1) / (256 * CPL), (unsigned)(c->Np / 16));
  hipLaunchKernelGGL((b_le_kernel<QT, CPL>), grid, dim3(256), 0, c->stream, (const double*)c->MUP, (const double*)c->WP, (const double*)c->V2P,
                     (const double*)c->lnc2h, (const double*)c->ZP, (long)c->N, c->M, c->Mp, c->LE, c->LET);
}

int run_generate_b(gp_ctx* c) {
  int rc = ensure_regime_b_buffers(c);
  if (rc != GP_OK) return rc;
  hipLaunchKernelGGL(b_tables_kernel, dim3(c->kl_blocks), dim3(256), 0, c->stream, c->mu, c->S, c->alpha, (long)c->N, (long)c->Np, c->Q,
                     c->sf2, c->Vn, c->Wn, c->lnc2h, c->V2P, c->QB, c->WP, c->MUP);
  GP_HIP(c, hipMemcpyAsync(c->alphaP, c->alpha, (size_t)c->Q * 8, hipMemcpyDeviceToDevice, c->stream));
  hipLaunchKernelGGL(zpad_kernel, dim3((unsigned)(((long)c->Mp * (c->QB + 6) + 255) / 256)), dim3(256), 0, c->stream, c->Z, c->M, c->Mp, c->Q, c->QB,
                     c->ZP, c->Z1P, c->b_sym ? c->Z1S : nullptr, (c->QB + 1 + 3) / 4 * 4);
  if (b_generic(c)) return run_le_generic(c);
  switch (c->QB) {
    case 4: launch_le<4, 2>(c); break;
    case 6: launch_le<6, 2>(c); break;
    case 8: launch_le<8, 2>(c); break;
    case 10: launch_le<10, 2>(c); break;
    case 12: launch_le<12, 2>(c); break;
    case 14: launch_le<14, 2>(c); break;
    case 16: launch_le<16, 2>(c); break;
    case 24: launch_le<24, 2>(c); break;
    case 32: launch_le<32, 1>(c); break;
    case 52: launch_le<52, 1>(c); break;
    default: launch_le<64, 1>(c); break;
  }
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

template <int QT>
static void launch_pairs(gp_ctx* c, int S) {
  hipLaunchKernelGGL((psi2_pairs_kernel<QT>), dim3(c->n_ptiles, S), dim3(256), 0, c->stream, c->LE, c->V2P, c->ZP, c->ptiles, (long)c->N,
                     c->Mp, S, c->part, c->n_ptiles);
}

template <int QT>
static void launch_pairs_mfma(gp_ctx* c, int S) {
  hipLaunchKernelGGL((psi2_pairs_mfma_kernel<QT>), dim3(c->n_tiles64, S), dim3(256), 0, c->stream, (const double*)c->LET, (const double*)c->V2P,
                     (const double*)c->ZP, (const int*)c->tiles64, (long)c->N, c->Mp, S, c->part, c->n_tiles64);
}

int run_phase1_b(gp_ctx* c) {
  if (b_generic(c)) { GP_TRY_RC(run_phase1_b_generic(c)); return psi2_zero_pads(c); }
  // gp_last_timings' "p1 kernel" slot: in regime B the Psi2 pair kernel (the C tiles' p1_kernel8 launch recorded the events before)
  GP_EV(c, 10);
  // the matrix-core pair kernel from the 24-wide latent tables on (17 <= Q): same-box, N = 1e5, M = 512, ms of this kernel at Q = 17 / 20 / 24:
  // 21.4 / 21.4 / 21.8 against 30.5 for psi2_pairs_kernel<24>; at 16 columns the VALU kernel is still ahead (15.3 ms measured against ~17 by the instruction count)
  if (c->b_mfma || c->QB == 24) {
    // 64 x 64 tiles x n-slices: several rounds of workgroups over the 512 resident slots, >= 256 points per slice
    int S = (int)std::max<long>(1, std::min<long>(32, std::max<long>((2048 + c->n_tiles64 - 1) / c->n_tiles64, c->N / 4096)));
    S = (int)std::min<long>(S, std::max<long>(1, c->N / 256));
    switch (c->QB) {
      case 24: launch_pairs_mfma<24>(c, S); break;
      case 32: launch_pairs_mfma<32>(c, S); break;
      case 52: launch_pairs_mfma<52>(c, S); break;
      default: launch_pairs_mfma<64>(c, S); break;
    }
    GP_EV(c, 11);
    GP_HIP(c, hipGetLastError());
    hipLaunchKernelGGL(psi2_reduce64_kernel, dim3(c->n_tiles64), dim3(256), 0, c->stream, c->part, c->tiles64, c->n_tiles64, S, c->M, c->Mp, c->stats);
    GP_HIP(c, hipGetLastError());
    return psi2_zero_pads(c);
  }
  // n-slices: many more workgroups than resident slots (256 CUs x 7) so the last round is short, >= 1024 points per slice
  int S = (int)std::max<long>(1, std::min<long>(64, std::max<long>((4096 + c->n_ptiles - 1) / c->n_ptiles, c->N / 1024)));
  S = (int)std::min<long>(S, c->N);
  switch (c->QB) {
    case 4: launch_pairs<4>(c, S); break;
    case 6: launch_pairs<6>(c, S); break;
    case 8: launch_pairs<8>(c, S); break;
    case 10: launch_pairs<10>(c, S); break;
    case 12: launch_pairs<12>(c, S); break;
    case 14: launch_pairs<14>(c, S); break;
    case 16: launch_pairs<16>(c, S); break;
    default: return fail(c, GP_ERR_UNSUPPORTED, "regime-B pair kernel: no instantiation for the latent table width %d", c->QB);
  }
  GP_EV(c, 11);
  GP_HIP(c, hipGetLastError());
  hipLaunchKernelGGL(psi2_reduce_kernel, dim3(c->n_ptiles), dim3(256), 0, c->stream, c->part, c->ptiles, c->n_ptiles, S, c->M, c->Mp, c->stats);
  GP_HIP(c, hipGetLastError());
  return psi2_zero_pads(c);
}

template <int QT, bool KEEP>
static void launch_cols(gp_ctx* c, const PB2Args& a) {     // (the Bbar argument: the row-interleaved table up to QT = 10, the plain one beyond)
  const int nw = std::min(4, c->nslab);
  hipLaunchKernelGGL((psi2_cols_kernel<QT, KEEP>), dim3(c->pb_blocks, (c->nslab + nw - 1) / nw), dim3(64 * nw), 0, c->stream, a, (const double*)c->ZP,
                     (const double*)(cols_b4(QT) ? c->Bbar4 : c->Bbar), (const double*)c->LET, (const double*)c->V2P, (const double*)c->WP, (const double*)c->MUP,
                     (const double*)c->alphaP);
}

template <int QT>
static int launch_sym(gp_ctx* c, const PB2Args& a) {
  const size_t smem = (size_t)c->Mp * sym_rs(QT) * sizeof(double);
  GP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(psi2_sym_kernel<QT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((psi2_sym_kernel<QT>), dim3(c->pb_blocks), dim3(64 * c->sym_nw), smem, c->stream, a, (const double*)c->ZP, (const double*)c->Z1S,
                     (const double*)c->Bbar4, (const double*)c->LET, (const double*)c->V2P, (const double*)c->WP, (const double*)c->MUP,
                     (const double*)c->alphaP, (const int*)c->sym_sched, c->sym_rounds);
  return GP_OK;
}

int run_phase2_b(gp_ctx* c) {
  if (c->b_tile) return run_phase2_b_tiles(c);
  PB2Args a;
  a.Wn = c->Wn; a.mu = c->mu; a.S = c->S; a.alpha = c->alpha;
  a.Gpart = c->Gpart; a.gapart2 = c->gapart2; a.gmu = c->gXmu; a.gS = c->gXs; a.pp = c->pp;
  a.N = c->N; a.Np = c->Np; a.M = c->M; a.Mp = c->Mp; a.Q = c->Q; a.QB = c->QB; a.nslab = c->nslab; a.ppb = c->ppb; a.ngrp = (c->nslab + std::min(4, c->nslab) - 1) / std::min(4, c->nslab);
  if (!b_generic(c)) hipLaunchKernelGGL(bbar_interleave_kernel, dim3((unsigned)std::min<long>(((long)c->Mp * c->Mp + 255) / 256, 2048)), dim3(256), 0, c->stream,
                                        (const double*)c->Bbar, c->Mp, c->Bbar4);
  GP_EV(c, 12);   // gp_last_timings' "p2 kernel" slot: in regime B the T_n = Bbar o psi2_n kernel
  if (b_generic(c)) {
    a.ngrp = 1;
    GP_TRY_RC(run_phase2_b_generic(c));
  } else if (c->b_sym) {
    a.ngrp = c->sym_nw;
    int rc = c->QB == 4 ? launch_sym<4>(c, a) : c->QB == 6 ? launch_sym<6>(c, a) : c->QB == 8 ? launch_sym<8>(c, a) : c->QB == 10 ? launch_sym<10>(c, a) : launch_sym<12>(c, a);
    if (rc != GP_OK) return rc;
  } else
  switch (c->QB) {
    case 4: launch_cols<4, true>(c, a); break;
    case 6: launch_cols<6, true>(c, a); break;
    case 8: launch_cols<8, true>(c, a); break;
    // (the <10, false> form -- z re-read and grad_Z accumulated in memory per point -- is slower than <10, true>: same box, N = 1e5, M = 128: 2.75 -> 2.98 ms,
    // M = 1024 (5e4 points): 69.2 -> 69.9 ms; profiles/r06_gplvm_experiments.txt)
    case 10: launch_cols<10, true>(c, a); break;
    case 12: launch_cols<12, false>(c, a); break;
    case 14: launch_cols<14, false>(c, a); break;
    case 16: launch_cols<16, false>(c, a); break;
    default: return fail(c, GP_ERR_UNSUPPORTED, "regime-B column kernel: no instantiation for the latent table width %d (Q = %d runs on psi2_tile_kernel)", c->QB, c->Q);
  }
  GP_EV(c, 13);
  GP_HIP(c, hipGetLastError());
  hipLaunchKernelGGL(psi2_points_finish_kernel, dim3((unsigned)std::min<long>(c->pb_blocks, 256)), dim3(256), 0, c->stream, a);
  GP_HIP(c, hipGetLastError());
  const long MQ = (long)c->M * c->Q;
  const int fin_blocks_g = (int)std::min<long>(c->pb_blocks, 256);
  if (b_generic(c)) {
    // grad_Z's psi2 part is in c->grads already: only the alpha partials of the points' finish are left
    hipLaunchKernelGGL(pb2_reduce_kernel, dim3((unsigned)std::min<long>((MQ + c->Q + 255) / 256, 1024)), dim3(256), 0, c->stream, c->Gtmp, c->gapart2, 0, fin_blocks_g, MQ,
                       c->Q, c->grads);
    GP_HIP(c, hipGetLastError());
    return GP_OK;
  }
  const int S2 = std::max(1, std::min(64, c->pb_blocks / 16));
  hipLaunchKernelGGL(pb2_reduce1_kernel, dim3((unsigned)((MQ + 63) / 64), S2), dim3(256), 0, c->stream, c->Gpart, c->pb_blocks, MQ, S2, c->Gtmp);
  const int fin_blocks = (int)std::min<long>(c->pb_blocks, 256);      // psi2_points_finish_kernel's grid (one gapart2 row per workgroup)
  hipLaunchKernelGGL(pb2_reduce_kernel, dim3((unsigned)std::min<long>((MQ + c->Q + 255) / 256, 1024)), dim3(256), 0, c->stream, c->Gtmp,
                     c->gapart2, S2, fin_blocks, MQ, c->Q, c->grads);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

}  // namespace gp
