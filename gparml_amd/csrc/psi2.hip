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
#include <algorithm>

namespace gp {

// ---------------------------------------------------------------------------------------------- tables
// per-point tables and LE in both layouts; thread = point for LET (coalesced along n), thread = column for LE
__global__ void __launch_bounds__(256) b_tables_kernel(const double* __restrict__ mu, const double* __restrict__ S,
                                                        const double* __restrict__ alpha, long N, long Np, int Q, double sf2,
                                                        double* __restrict__ Vn, double* __restrict__ Wn, double* __restrict__ lnc2h,
                                                        double* __restrict__ V2P, int QB, double* __restrict__ V2T,
                                                        double* __restrict__ WT, double* __restrict__ MUT) {
  for (long n = blockIdx.x * 256L + threadIdx.x; n < Np; n += (long)gridDim.x * 256L) {
    double l = log(sf2);   // half of ln c2 = ln sf2 - 1/4 sum ln(2 a S + 1)
    for (int q = 0; q < Q; ++q) {
      const double a = alpha[q], s = S[n * Q + q];
      const double d2 = 2.0 * a * s + 1.0, w = a / d2;
      Wn[n * Q + q] = w;
      Vn[n * Q + q] = -0.25 * (a - w);
      V2P[n * QB + q] = 0.5 * (a - w);      // -2 V_nq (columns >= Q stay zero from the allocation)
      V2T[q * Np + n] = 0.5 * (a - w);      // the same, q-major (coalesced per-lane reads in the rows kernel)
      WT[q * Np + n] = w;
      MUT[q * Np + n] = mu[n * Q + q];
      l -= 0.25 * log(d2);
    }
    lnc2h[n] = l;
  }
}

__global__ void __launch_bounds__(256) b_le_kernel(const double* __restrict__ mu, const double* __restrict__ Wn, const double* __restrict__ Vn,
                                                    const double* __restrict__ lnc2h, const double* __restrict__ Z, long N, long Np,
                                                    int M, int Mp, int Q, double* __restrict__ LE, double* __restrict__ LET) {
  // block: 16 rows (n) x 16 cols (m) tile computed once, written in both layouts through LDS
  __shared__ double tile[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const long n = blockIdx.y * 16L + ty;
  const int m = blockIdx.x * 16 + tx;
  double e = 0.0;
  if (n < N && m < M) {
    for (int q = 0; q < Q; ++q) {
      const double d = mu[n * Q + q] - Z[(long)m * Q + q];
      e = fma(Wn[n * Q + q] * d, d, e);
    }
    e = lnc2h[n] - 0.5 * e;
  } else {
    e = -1e300;   // exp() of a padded entry is exactly 0
  }
  LE[n * Mp + m] = e;
  // LEA = LE + sum_q V_nq z_mq^2: with it the pair exponent is LEA_nm + LEA_nm' - 2 sum_q V_nq z_mq z_m'q
  double ea = e;
  if (n < N && m < M) {
    for (int q = 0; q < Q; ++q) { const double z = Z[(long)m * Q + q]; ea = fma(Vn[n * Q + q] * z, z, ea); }
  }
  tile[ty][tx] = ea;
  __syncthreads();
  const long n2 = blockIdx.y * 16L + tx;
  const int m2 = blockIdx.x * 16 + ty;
  LET[((n2 >> 6) * (long)Mp + m2) * 64 + (n2 & 63)] = tile[tx][ty];   // tiled [Np/64][Mp][64]: a wave's stream is contiguous
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

__global__ void __launch_bounds__(256) zpad_kernel(const double* __restrict__ Z, int M, int Mp, int Q, int QB, double* __restrict__ ZP) {
  const long total = (long)Mp * QB;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int q = (int)(i % QB), m = (int)(i / QB);
    ZP[i] = (m < M && q < Q) ? Z[(long)m * Q + q] : 0.0;
  }
}

// exp for the pair kernels: 20 FP64 instructions, no table, no special cases beyond underflow (the argument is a finite
// log-density or the -1e300 padding marker).  x = k ln2 + r, |r| <= ln2/2; degree-13 Taylor (truncation r^14/14! < 5e-18
// relative); the result is exact to ~2 ulp, far inside the 1e-6 / 1e-5 parity budget.
__device__ __forceinline__ double fexp(double x) {
  x = fmax(x, -745.5);
  const double k = rint(x * 1.4426950408889634074);
  double r = fma(k, -6.93147180369123816490e-01, x);
  r = fma(k, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821614599e-10;              // 1/13!
  p = fma(p, r, 2.0876756987868098979e-09);          // 1/12!
  p = fma(p, r, 2.5052108385441718775e-08);          // 1/11!
  p = fma(p, r, 2.7557319223985890653e-07);          // 1/10!
  p = fma(p, r, 2.7557319223985892511e-06);          // 1/9!
  p = fma(p, r, 2.4801587301587301566e-05);          // 1/8!
  p = fma(p, r, 1.9841269841269841253e-04);          // 1/7!
  p = fma(p, r, 1.3888888888888889419e-03);          // 1/6!
  p = fma(p, r, 8.3333333333333332177e-03);          // 1/5!
  p = fma(p, r, 4.1666666666666664354e-02);          // 1/4!
  p = fma(p, r, 1.6666666666666665741e-01);          // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)k);
}

// ---------------------------------------------------------------------------------------------- phase 1
// grid (pair tiles, n slices); thread (i,j) of a 16x16 tile owns the pair (m = I*16+i, m' = J*16+j), J >= I.
template <int QT>
__global__ void __launch_bounds__(256) psi2_pairs_kernel(const double* __restrict__ LE, const double* __restrict__ V2P,
                                                          const double* __restrict__ ZP, const int* __restrict__ ptiles, long N,
                                                          int Mp, int S, double* __restrict__ part, int T) {
  // exponent = LE[n,m] + LE[n,m'] + sum_q V_nq dz2_q = LE + LE' + sum_q (-2 V_nq) * (-1/2 dz2_q): the per-pair vector lives in
  // registers, the per-point vector (-2V, zero-padded to QT) is wave-uniform -> scalar loads; four points per trip so
  // the loads of a trip are in flight together.  Padded rows of LE hold -1e300 (exp -> 0), padded pairs are dropped by
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
  const long per = (N + S - 1) / S;
  const long n0 = slice * per, n1 = min(N, n0 + per);
  const double* l1 = LE + m1;
  const double* l2 = LE + m2;
  double acc0 = 0.0, acc1 = 0.0;
  long n = n0;
  for (; n + 4 <= n1; n += 4) {
    double e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) e[u] = l1[(n + u) * Mp] + l2[(n + u) * Mp];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double* v = V2P + (n + u) * QT;      // wave-uniform
#pragma unroll
      for (int q = 0; q < QT; ++q) e[u] = fma(v[q], dz[q], e[u]);
    }
    acc0 += fexp(e[0]) + fexp(e[2]);
    acc1 += fexp(e[1]) + fexp(e[3]);
  }
  for (; n < n1; ++n) {
    double e = l1[n * Mp] + l2[n * Mp];
    const double* v = V2P + n * QT;
#pragma unroll
    for (int q = 0; q < QT; ++q) e = fma(v[q], dz[q], e);
    acc0 += fexp(e);
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

// zero the M x M block (and pads) of Psi2 before the pair reduce writes it
__global__ void fill_kernel(double* x, long n, double v) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) x[i] = v;
}

// ---------------------------------------------------------------------------------------------- phase 2
// lane = point.  For every m: r = sum_m' T[m,m'], t_q = sum_m' T[m,m'] z_m'q with T = Bbar[m,m'] psi2_n[m,m'].
//   grad_Z psi2 part  G[m,k] += -a_k z_mk r + a_k t_k + w_k (2 mu_k r - z_mk r - t_k)          (partial_terms.py:190-205, x2 at :238)
//   per point: sr, zr_q, z2r_q, zt_q -> quad = 4 mu^2 sr - 8 mu zr + 2 z2r + 2 zt
//   grad_alpha += -1/4 quad/d2^2 - (S/d2) sr ; grad_X_mu += -w (2 mu sr - 2 zr) ; grad_X_S += 1/2 w^2 quad - w sr
struct PB2Args {
  const double* LET; const double* Vn; const double* Wn; const double* mu; const double* S; const double* DZ2; const double* Z;
  const double* Bbar; const double* alpha; double* Gpart; double* gapart2; double* gmu; double* gS; double* pp;
  const double* V2P; const double* ZP; const double* V2T; const double* WT; const double* MUT;
  long N, Np; int M, Mp, Q, QB, groups_per_block;   // QB: padded Q (row stride of the q-major per-point tables and of pp)
};

// MC inducing rows per pass: one LEA[m'][n] load and one uniform z_m' vector serve MC pair terms (MC = 4 for Q <= 16; the
// register arrays p[MC][Q], t[MC][Q] force MC = 1 for larger Q)
template <int QT, int MC>
__global__ void __launch_bounds__(256, 2) psi2_rows_wide_kernel(PB2Args a) {
  // LET holds LEA (m-major): exponent(n; m, m') = LEA[m][n] + LEA[m'][n] + sum_q p_mq z_m'q with p_mq = -2 V_nq z_mq.
  // Per point the running sums sr, zr_q, z2r_q, zt_q live in a.pp (global, touched once per MC rows), so the inner loop
  // keeps only p[MC][Q], t[MC][Q] and r[MC] in registers.
  __shared__ double red[4][MC][QT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* G = a.Gpart + (long)blockIdx.x * a.M * a.Q;
  const int PW = 3 * a.QB + 1;
  for (int grp = 0; grp < a.groups_per_block; ++grp) {
    const long n = ((long)blockIdx.x * a.groups_per_block + grp) * 256 + tid;
    const bool live = n < a.N;
    const long nn = live ? n : 0;
    double* ppn = a.pp + nn;                       // running sums, one row of Np doubles per quantity
    if (live) for (int k = 0; k < PW; ++k) ppn[(long)k * a.Np] = 0.0;
    const double* lcol = a.LET + (nn >> 6) * (long)a.Mp * 64 + (nn & 63);   // element m at lcol[m * 64]
    for (int m0 = 0; m0 < a.M; m0 += MC) {
      double p[MC][QT], t[MC][QT], r[MC], lem[MC];
#pragma unroll
      for (int k = 0; k < MC; ++k) {
        const double* zm = a.Z + (long)(m0 + k) * a.Q;               // wave-uniform (rows >= M are zero)
        lem[k] = (live && m0 + k < a.M) ? lcol[(long)(m0 + k) * 64] : -1e300;
        r[k] = 0.0;
#pragma unroll
        for (int q = 0; q < QT; ++q) { p[k][q] = (q < a.Q) ? -2.0 * a.Vn[nn * a.Q + q] * zm[q] : 0.0; t[k][q] = 0.0; }
      }
      const double* brow = a.Bbar + (long)m0 * a.Mp;                 // wave-uniform, MC consecutive rows
#pragma unroll 2
      for (int m2 = 0; m2 < a.M; ++m2) {
        const double* z2 = a.Z + (long)m2 * a.Q;                     // wave-uniform
        const double l2 = lcol[(long)m2 * 64];
        double zz[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) zz[q] = (q < a.Q) ? z2[q] : 0.0;
#pragma unroll
        for (int k = 0; k < MC; ++k) {
          double e = lem[k] + l2;
#pragma unroll
          for (int q = 0; q < QT; ++q) e = fma(p[k][q], zz[q], e);
          const double T = brow[(long)k * a.Mp + m2] * exp(e);
          r[k] += T;
#pragma unroll
          for (int q = 0; q < QT; ++q) t[k][q] = fma(T, zz[q], t[k][q]);
        }
      }
      // fold the MC rows into the per-point sums and the block-level grad_Z contribution
      double dsr = 0.0;
#pragma unroll
      for (int k = 0; k < MC; ++k) {
        const double* zm = a.Z + (long)(m0 + k) * a.Q;
        const double rk = live ? r[k] : 0.0;
        dsr += rk;
#pragma unroll
        for (int q = 0; q < QT; ++q) {
          if (q < a.Q) {
            const double z = zm[q];
            const double tq = live ? t[k][q] : 0.0;
            if (live) {
              double* p1 = ppn + (long)(1 + q) * a.Np;
              double* p2 = ppn + (long)(1 + a.QB + q) * a.Np;
              double* p3 = ppn + (long)(1 + 2 * a.QB + q) * a.Np;
              *p1 = fma(z, rk, *p1);
              *p2 = fma(z * z, rk, *p2);
              *p3 = fma(z, tq, *p3);
            }
            const double w = a.Wn[nn * a.Q + q], mu = a.mu[nn * a.Q + q];
            double g = -a.alpha[q] * (z * rk - tq) + w * (2.0 * mu * rk - z * rk - tq);
            for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o);
            if (lane == 0) red[wave][k][q] = g;
          }
        }
      }
      if (live) ppn[0] += dsr;
      __syncthreads();
      if (tid < MC * a.Q) {
        const int k = tid / a.Q, q = tid - k * a.Q;
        if (m0 + k < a.M) {
          const double s = red[0][k][q] + red[1][k][q] + red[2][k][q] + red[3][k][q];
          double* dst = G + (long)(m0 + k) * a.Q + q;
          *dst = ((grp == 0) ? 0.0 : *dst) + s;
        }
      }
      __syncthreads();
    }
  }
}

// Q <= 16: the same scheme on the zero-padded tables (no q guards in the pair loop), the B-bar row block read through the
// matrix' symmetry as MC consecutive doubles (one scalar load), the hand-rolled exp.  The LEA[m'][n] stream (one 512-byte
// row piece per wave and m', re-read for every row block, far larger than L2 across the resident waves) is brought in
// by LDS-DMA, RCH rows ahead of its use, each wave feeding its own double buffer: without it every m' step waits a full
// HBM round trip (measured: 115 ms of 137 ms at N = 2e5, M = 512).
constexpr int RCH = 8;
typedef double dbl2 __attribute__((ext_vector_type(2)));
// two rows (j, j+1) of the wave's LDS chunk for this lane.  Inline asm on purpose: a compiler-visible LDS read makes
// hipcc wait for every outstanding LDS-DMA (s_waitcnt vmcnt(0)) first, which would expose the prefetch it is there to hide.
template <int J>
__device__ __forceinline__ dbl2 lds_rows2(unsigned byte_addr) {
  dbl2 v;
  asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_addr), "n"(J), "n"(J + 1));
  return v;
}
template <int QT, int MC>
__global__ void __launch_bounds__(256, MC >= 4 ? 2 : 4) psi2_rows_kernel(PB2Args a, const double* __restrict__ ZP, const double* __restrict__ Bbar,
                                                           const double* __restrict__ LET, const double* __restrict__ V2P) {
  // (the read-only tables are separate __restrict__ kernel arguments so that the wave-uniform reads become scalar loads)
  __shared__ double red[4][MC][QT];
  __shared__ __attribute__((aligned(16))) double lbuf[4][MC >= 4 ? 3 : 2][RCH][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* G = a.Gpart + (long)blockIdx.x * a.M * a.Q;
  constexpr int PW = 3 * QT + 1;                   // rows of pp: [sr | zr_q | z2r_q | zt_q], q padded to QT = QB
  const int Mr = (a.M + RCH - 1) / RCH * RCH;       // <= Mp; rows >= M of LET hold -1e300, of ZP / Bbar zero
  for (int grp = 0; grp < a.groups_per_block; ++grp) {
    const long nw0 = ((long)blockIdx.x * a.groups_per_block + grp) * 256 + wave * 64;
    const long n = nw0 + lane;
    const bool live = n < a.N;
    const long nn = live ? n : 0;
    double* ppn = a.pp + nn;                       // running sums [sr | zr_q | z2r_q | zt_q], one row of Np doubles each
    if (live) for (int k = 0; k < PW; ++k) ppn[(long)k * a.Np] = 0.0;
    const double* lcol = LET + (nn >> 6) * (long)a.Mp * 64 + (nn & 63);     // element m at lcol[m * 64]
    const double* v2 = a.V2T + nn;
    // DMA source of this lane: LET is tiled [Np/64][Mp][64], so rows mb .. mb+RCH-1 of the wave's 64 points are RCH*512
    // contiguous bytes; each instruction moves two rows (16 bytes per lane)
    const double* dsrc = LET + min(nw0 >> 6, a.Np / 64 - 1) * (long)a.Mp * 64 + 2 * lane;
    auto dma = [&](int buf, int mb) {
#pragma unroll
      for (int j = 0; j < RCH / 2; ++j)
        __builtin_amdgcn_global_load_lds((gp::gbl_void*)(dsrc + (long)(mb + 2 * j) * 64), (gp::lds_void*)&lbuf[wave][buf][2 * j][0], 16, 0, 2);
      // aux = 2 (nt): the stream must not evict the scalar operands (Z, Bbar) from L2 -- measured 158 -> 134 ms
    };
    const unsigned lbuf_addr = (unsigned)(unsigned long)(gp::lds_void*)&lbuf[wave][0][0][lane];
    for (int m0 = 0; m0 < a.M; m0 += MC) {          // rows m0 .. m0+MC-1 (rows >= M: ZP rows are zero, LET rows are -1e300)
      double p[MC][QT], t[MC][QT], r[MC], lem[MC];
      dma(0, 0);
      if (MC >= 4) dma(1, RCH);
#pragma unroll
      for (int k = 0; k < MC; ++k) {
        const double* zm = ZP + (long)(m0 + k) * QT;                   // wave-uniform
        lem[k] = live ? lcol[(long)(m0 + k) * 64] : -1e300;
        r[k] = 0.0;
#pragma unroll
        for (int q = 0; q < QT; ++q) { p[k][q] = v2[(long)q * a.Np] * zm[q]; t[k][q] = 0.0; }
      }
      const double* bcol = Bbar + m0;                                  // Bbar[m'][m0 + k] = Bbar[m0 + k][m'] (symmetric)
      // the uniform operands of step m' (z_m', Bbar[m'][m0..]) are scalar loads issued one step ahead of their use
      double zc[QT], bc[MC];
#pragma unroll
      for (int q = 0; q < QT; ++q) zc[q] = ZP[q];
#pragma unroll
      for (int k = 0; k < MC; ++k) bc[k] = bcol[k];
      for (int mb = 0; mb < Mr; mb += RCH) {
        constexpr int NB = MC >= 4 ? 3 : 2;         // LDS chunk buffers per wave (NB - 1 chunks in flight)
        const int buf = (mb / RCH) % NB;
        if (NB == 3 && mb + RCH < Mr) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else gp::dma_wait();
        if (mb + (NB - 1) * RCH < Mr) dma((buf + NB - 1) % NB, mb + (NB - 1) * RCH);
        const unsigned lds_addr = lbuf_addr + buf * (RCH * 64 * 8);
        dbl2 lv;
#pragma unroll
        for (int j = 0; j < RCH; ++j) {
          if (j == 0) lv = lds_rows2<0>(lds_addr);
          if (j == 2) lv = lds_rows2<2>(lds_addr);
          if (j == 4) lv = lds_rows2<4>(lds_addr);
          if (j == 6) lv = lds_rows2<6>(lds_addr);
          const double l2 = (j & 1) ? lv.y : lv.x;
          // SMEM returns out of order, so only lgkmcnt(0) is usable: touch the current operands first (that wait
          // retires the loads issued one step ago), THEN issue the loads of the next step, then do the bulk of the work
          double e[MC];
#pragma unroll
          for (int k = 0; k < MC; ++k) e[k] = fma(p[k][0], zc[0], lem[k] + l2);
          __builtin_amdgcn_sched_barrier(0);
          const int m2n = min(mb + j + 1, a.Mp - 1);
          const double* z2 = ZP + (long)m2n * QT;                      // wave-uniform
          const double* bb = bcol + (long)m2n * a.Mp;                  // wave-uniform, MC consecutive doubles
          double zn[QT], bn[MC];
#pragma unroll
          for (int q = 0; q < QT; ++q) zn[q] = z2[q];
#pragma unroll
          for (int k = 0; k < MC; ++k) bn[k] = bb[k];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < MC; ++k) {
#pragma unroll
            for (int q = 1; q < QT; ++q) e[k] = fma(p[k][q], zc[q], e[k]);
            const double T = bc[k] * fexp(e[k]);
            r[k] += T;
#pragma unroll
            for (int q = 0; q < QT; ++q) t[k][q] = fma(T, zc[q], t[k][q]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < QT; ++q) zc[q] = zn[q];
#pragma unroll
          for (int k = 0; k < MC; ++k) bc[k] = bn[k];
        }
      }
      // fold the MC rows into the per-point sums and the block-level grad_Z contribution.  All loads of the epilogue are
      // issued together (one memory round trip per row block, not one per quantity).
      double sv[PW], wv[QT], muv[QT];
#pragma unroll
      for (int i = 0; i < PW; ++i) sv[i] = ppn[(long)i * a.Np];
#pragma unroll
      for (int q = 0; q < QT; ++q) { wv[q] = a.WT[(long)q * a.Np + nn]; muv[q] = a.MUT[(long)q * a.Np + nn]; }
#pragma unroll
      for (int k = 0; k < MC; ++k) {
        const double* zm = ZP + (long)(m0 + k) * QT;
        const double rk = r[k];                                        // dead lanes: exp(-1e300) = 0 -> r = t = 0
        sv[0] += rk;
#pragma unroll
        for (int q = 0; q < QT; ++q) {
          const double z = zm[q];
          const double tq = t[k][q];
          sv[1 + q] = fma(z, rk, sv[1 + q]);
          sv[1 + QT + q] = fma(z * z, rk, sv[1 + QT + q]);
          sv[1 + 2 * QT + q] = fma(z, tq, sv[1 + 2 * QT + q]);
          if (q < a.Q) {
            double g = -a.alpha[q] * (z * rk - tq) + wv[q] * (2.0 * muv[q] * rk - z * rk - tq);
            for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o);
            if (lane == 0) red[wave][k][q] = g;
          }
        }
      }
      if (live) {
#pragma unroll
        for (int i = 0; i < PW; ++i) ppn[(long)i * a.Np] = sv[i];
      }
      __syncthreads();
      if (tid < MC * a.Q) {
        const int k = tid / a.Q, q = tid - k * a.Q;
        if (m0 + k < a.M) {
          const double s = red[0][k][q] + red[1][k][q] + red[2][k][q] + red[3][k][q];
          double* dst = G + (long)(m0 + k) * a.Q + q;
          *dst = ((grp == 0) ? 0.0 : *dst) + s;
        }
      }
      __syncthreads();
    }
  }
}

// per-point finish of the psi2 part from the running sums pp[n] = [sr, zr_q, z2r_q, zt_q]
__global__ void __launch_bounds__(256) psi2_points_finish_kernel(PB2Args a) {
  __shared__ double redq[256];
  const int PW = 3 * a.Q + 1;
  for (int q = 0; q < a.Q; ++q) {
    double ga = 0.0;
    for (long n = blockIdx.x * 256L + threadIdx.x; n < a.N; n += (long)gridDim.x * 256L) {
      const double* ppn = a.pp + n;
      const double sr = ppn[0], zr = ppn[(long)(1 + q) * a.Np], z2r = ppn[(long)(1 + a.QB + q) * a.Np], zt = ppn[(long)(1 + 2 * a.QB + q) * a.Np];
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

// grads[0:M*Q] += sum_blocks Gpart ; grads[M*Q + q] += sum_blocks gapart2
__global__ void __launch_bounds__(256) pb2_reduce_kernel(const double* __restrict__ Gpart, const double* __restrict__ gapart2, int nb, int nb2,
                                                         long MQ, int Q, double* __restrict__ grads) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < MQ + Q; i += (long)gridDim.x * 256L) {
    double s = 0.0;
    if (i < MQ) for (int b = 0; b < nb; ++b) s += Gpart[(long)b * MQ + i];
    else for (int b = 0; b < nb2; ++b) s += gapart2[(long)b * Q + (i - MQ)];
    grads[i] += s;
  }
}

// ---------------------------------------------------------------------------------------------- host side
template <typename T>
static int balloc(gp_ctx* c, T** p, size_t count) {
  GP_HIP(c, hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)));
  GP_HIP(c, hipMemsetAsync(*p, 0, std::max<size_t>(count, 1) * sizeof(T), c->stream));
  return GP_OK;
}

int ensure_regime_b_buffers(gp_ctx* c) {
  if (c->b_alloc) return GP_OK;
  const long Np = c->Np, Mp = c->Mp, M = c->M, Q = c->Q;
  int rc = GP_OK;
  auto A = [&](auto** p, size_t n) { if (rc == GP_OK) rc = balloc(c, p, n); };
  c->QB = Q <= 4 ? 4 : Q <= 10 ? 10 : Q <= 16 ? 16 : Q <= 32 ? 32 : 64;
  A(&c->LE, (size_t)Np * Mp); A(&c->LET, (size_t)Mp * Np); A(&c->Vn, (size_t)Np * Q); A(&c->Wn, (size_t)Np * Q);
  A(&c->V2P, (size_t)Np * c->QB); A(&c->ZP, (size_t)Mp * c->QB);
  A(&c->V2T, (size_t)Np * c->QB); A(&c->WT, (size_t)Np * c->QB); A(&c->MUT, (size_t)Np * c->QB);
  A(&c->DZ2, (size_t)M * M * Q); A(&c->lnc2h, (size_t)Np);
  const long groups = (c->N + 255) / 256;
  c->pb_blocks = (int)std::min<long>(groups, 2048);
  A(&c->Gpart, (size_t)c->pb_blocks * M * Q); A(&c->gapart2, (size_t)c->pb_blocks * Q); A(&c->pp, (size_t)Np * (3 * c->QB + 1));
  std::vector<int> t;
  const int Mt = (int)((M + 15) / 16);
  for (int i = 0; i < Mt; ++i) for (int j = i; j < Mt; ++j) { t.push_back(i); t.push_back(j); }
  c->n_ptiles = (int)t.size() / 2;
  A(&c->ptiles, t.size());
  if (rc != GP_OK) return rc;
  GP_HIP(c, hipMemcpyAsync(c->ptiles, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  GP_HIP(c, hipStreamSynchronize(c->stream));
  // the pair kernel's split-n partials live in c->part: make sure it is large enough
  const size_t need = (size_t)c->n_ptiles * 256 * 64;
  if (need > c->part_doubles) {
    (void)hipFree(c->part);
    c->part = nullptr;
    GP_HIP(c, hipMalloc((void**)&c->part, need * 8));
    c->part_doubles = need;
  }
  c->b_alloc = true;
  return GP_OK;
}

int run_dz2(gp_ctx* c) {
  const long total = (long)c->M * c->M * c->Q;
  hipLaunchKernelGGL(dz2_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, c->stream, c->Z, c->M, c->Q, c->DZ2);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

int run_generate_b(gp_ctx* c) {
  int rc = ensure_regime_b_buffers(c);
  if (rc != GP_OK) return rc;
  rc = run_dz2(c);
  if (rc != GP_OK) return rc;
  hipLaunchKernelGGL(b_tables_kernel, dim3(c->kl_blocks), dim3(256), 0, c->stream, c->mu, c->S, c->alpha, (long)c->N, (long)c->Np, c->Q,
                     c->sf2, c->Vn, c->Wn, c->lnc2h, c->V2P, c->QB, c->V2T, c->WT, c->MUT);
  hipLaunchKernelGGL(zpad_kernel, dim3((unsigned)(((long)c->Mp * c->QB + 255) / 256)), dim3(256), 0, c->stream, c->Z, c->M, c->Mp, c->Q, c->QB,
                     c->ZP);
  dim3 grid(c->Mp / 16, (unsigned)(c->Np / 16));
  hipLaunchKernelGGL(b_le_kernel, grid, dim3(256), 0, c->stream, c->mu, c->Wn, c->Vn, c->lnc2h, c->Z, (long)c->N, (long)c->Np, c->M, c->Mp, c->Q,
                     c->LE, c->LET);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

template <int QT>
static void launch_pairs(gp_ctx* c, int S) {
  hipLaunchKernelGGL((psi2_pairs_kernel<QT>), dim3(c->n_ptiles, S), dim3(256), 0, c->stream, c->LE, c->V2P, c->ZP, c->ptiles, (long)c->N,
                     c->Mp, S, c->part, c->n_ptiles);
}

int run_phase1_b(gp_ctx* c) {
  int S = (int)std::max<long>(1, std::min<long>(64, std::min<long>(c->N, (4096 + c->n_ptiles - 1) / c->n_ptiles)));
  if (c->Q > 64) return fail(c, GP_ERR_UNSUPPORTED, "regime B supports Q <= 64 (got %d)", c->Q);
  switch (c->QB) {
    case 4: launch_pairs<4>(c, S); break;
    case 10: launch_pairs<10>(c, S); break;
    case 16: launch_pairs<16>(c, S); break;
    case 32: launch_pairs<32>(c, S); break;
    default: launch_pairs<64>(c, S); break;
  }
  GP_HIP(c, hipGetLastError());
  hipLaunchKernelGGL(psi2_reduce_kernel, dim3(c->n_ptiles), dim3(256), 0, c->stream, c->part, c->ptiles, c->n_ptiles, S, c->M, c->Mp, c->stats);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

template <int QT, int MC>
static void launch_rows(gp_ctx* c, const PB2Args& a, int blocks) {
  hipLaunchKernelGGL((psi2_rows_kernel<QT, MC>), dim3(blocks), dim3(256), 0, c->stream, a, (const double*)c->ZP, (const double*)c->Bbar,
                     (const double*)c->LET, (const double*)c->V2P);
}

int run_phase2_b(gp_ctx* c) {
  if (c->Q > 64) return fail(c, GP_ERR_UNSUPPORTED, "regime B supports Q <= 64 (got %d)", c->Q);
  PB2Args a;
  a.LET = c->LET; a.Vn = c->Vn; a.Wn = c->Wn; a.mu = c->mu; a.S = c->S; a.DZ2 = c->DZ2; a.Z = c->Z; a.Bbar = c->Bbar; a.alpha = c->alpha;
  a.Gpart = c->Gpart; a.gapart2 = c->gapart2; a.gmu = c->gXmu; a.gS = c->gXs; a.pp = c->pp; a.V2P = c->V2P; a.ZP = c->ZP; a.V2T = c->V2T; a.WT = c->WT; a.MUT = c->MUT;
  a.N = c->N; a.Np = c->Np; a.M = c->M; a.Mp = c->Mp; a.Q = c->Q; a.QB = c->QB;
  const long groups = (c->N + 255) / 256;
  a.groups_per_block = (int)((groups + c->pb_blocks - 1) / c->pb_blocks);
  const int blocks = (int)((groups + a.groups_per_block - 1) / a.groups_per_block);
  if (c->Q <= 4) launch_rows<4, 4>(c, a, blocks);
  else if (c->Q <= 10) launch_rows<10, 4>(c, a, blocks);
  else if (c->Q <= 16) launch_rows<16, 2>(c, a, blocks);
  else if (c->Q <= 32) hipLaunchKernelGGL((psi2_rows_wide_kernel<32, 1>), dim3(blocks), dim3(256), 0, c->stream, a);
  else hipLaunchKernelGGL((psi2_rows_wide_kernel<64, 1>), dim3(blocks), dim3(256), 0, c->stream, a);
  GP_HIP(c, hipGetLastError());
  hipLaunchKernelGGL(psi2_points_finish_kernel, dim3(c->pb_blocks), dim3(256), 0, c->stream, a);
  GP_HIP(c, hipGetLastError());
  const long MQ = (long)c->M * c->Q;
  hipLaunchKernelGGL(pb2_reduce_kernel, dim3((unsigned)std::min<long>((MQ + c->Q + 255) / 256, 1024)), dim3(256), 0, c->stream, c->Gpart,
                     c->gapart2, blocks, c->pb_blocks, MQ, c->Q, c->grads);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

}  // namespace gp
