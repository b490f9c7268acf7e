// Per-shard data kernels: trial-point preparation, Psi1 generation, phase-1 statistics (Psi2 = sum_n psi2_n,
// C = Psi1^T Y, KL), phase-2 gradient sums.  Reference: kernel_exp.py:13-148, partial_terms.py:38-87, 162-431,
// local_MapReduce.py:183-248, 310-363.
#include "gp_common.h"
#include <vector>
#include "fexp.h"
#include <algorithm>
#include <cstdlib>

namespace gp {


// ------------------------------------------------------------------------------------------------ Y upload
// Kaug[n][Mp + d] = Y[n][d] (zero padded); one block per row group
__global__ void __launch_bounds__(256) copy_y_kernel(const double* __restrict__ Y, double* __restrict__ Kaug, long N, long Np, int D,
                                                     int Dp, int Mp, long ld) {
  const long total = Np * (long)Dp;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const long n = i / Dp;
    const int d = (int)(i - n * Dp);
    Kaug[n * ld + Mp + d] = (n < N && d < D) ? Y[n * D + d] : 0.0;
  }
}

// ------------------------------------------------------------------------------------------------ prep
// Trial point (local_MapReduce.py:205-214): mu = X_mu + step*d_mu ; S = softplus(X_S_raw + step*d_S) when the
// variances are stored raw, else S = X_S.  Also ln c1_n (Psi1 normaliser, kernel_exp.py:80), the per-point
// features Xa = [f1, f2, 1] and the per-block KL partial sums (partial_terms.py:83-85).
struct PrepArgs {
  const double* Xmu; const double* Xs; const double* dir; const double* alpha;
  double* mu; double* S; double* U; double* lnc1; double* Xa; double* klpart;
  double* PU; int QP;   // packed per-point records [mu (QP) | u (QP) | ln c1 | 0] for psi1_kernel (PU == nullptr: not used); padding stays zero from the allocation
  long N, Np; int Q, CXp; double step, sf2; int raw, regimeA, fixedA;
};

__device__ __forceinline__ double softplus(double x) { return log(1.0 + exp(x)); }

// element-wise part: one thread per (n, q), fully coalesced
__global__ void __launch_bounds__(256) prep_elem_kernel(PrepArgs a) {
  const long total = a.Np * a.Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const long n = i / a.Q;
    const int q = (int)(i - n * a.Q);
    double m = 0.0, s = 0.0;
    if (n < a.N) {
      m = a.Xmu[i];
      s = a.Xs[i];
      if (a.raw) {
        if (a.dir && a.step != 0.0) {
          m += a.step * a.dir[i];
          s += a.step * a.dir[a.N * a.Q + i];
        }
        s = softplus(s);
      }
    }
    a.mu[i] = m;
    a.S[i] = s;
    const double al = a.alpha[q];
    const double u = al / (al * s + 1.0);
    a.U[i] = u;
    if (a.PU) { double* rec = a.PU + n * (2 * a.QP + 2); rec[q] = m; rec[a.QP + q] = u; }
    // per-point features of the n-contraction: fixed embeddings [mu (Q) | 1 | 0 ...] (fixedA = 1: Q + 1 <= 12, p2_fast8_kernel) or
    // [mu (Q) | 1 | mu^2 (Q) | 0 ...] (fixedA = 2: wider latent spaces on p2_gen8_kernel<false>), otherwise [u mu (Q) | u (Q) | 1 | 0 ...]
    if (a.fixedA) {
      a.Xa[n * a.CXp + q] = (n < a.N) ? m : 0.0;
      if (a.fixedA == 2) a.Xa[n * a.CXp + a.Q + 1 + q] = (n < a.N) ? m * m : 0.0;
    } else {
      a.Xa[n * a.CXp + q] = (n < a.N) ? u * m : 0.0;
      a.Xa[n * a.CXp + a.Q + q] = (n < a.N) ? u : 0.0;
    }
  }
}

// per-row part: ln c1_n, the constant/padding feature columns, KL partial sums.  Regime A never reads S (it is zero).
__global__ void __launch_bounds__(256) prep_row_kernel(PrepArgs a) {
  __shared__ double red[256];
  double kl = 0.0;
  const double ln_sf2 = log(a.sf2);
  for (long n = blockIdx.x * 256L + threadIdx.x; n < a.Np; n += (long)gridDim.x * 256L) {
    double lnc = ln_sf2;
    if (!a.regimeA) {
      double klrow = 0.0;
      for (int q = 0; q < a.Q; ++q) {
        const double s = a.S[n * a.Q + q], m = a.mu[n * a.Q + q];
        lnc -= 0.5 * log(a.alpha[q] * s + 1.0);
        if (n < a.N) klrow += s - log(s) + m * m - 1.0;
      }
      kl += 0.5 * klrow;
    }
    a.lnc1[n] = lnc;
    if (a.PU) a.PU[n * (2 * a.QP + 2) + 2 * a.QP] = lnc;
    const int c1 = a.fixedA ? a.Q : 2 * a.Q;   // column of ones
    a.Xa[n * a.CXp + c1] = (n < a.N) ? 1.0 : 0.0;
    for (int c = (a.fixedA == 2 ? 2 * a.Q : c1) + 1; c < a.CXp; ++c) a.Xa[n * a.CXp + c] = 0.0;   // fixedA = 2: columns Q + 1 .. 2 Q hold mu^2
  }
  red[threadIdx.x] = kl;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.klpart[blockIdx.x] = red[0];
}

// ------------------------------------------------------------------------------------------------ Psi1
// Kaug[n][m] = exp(ln c1_n - 1/2 sum_q u_nq (mu_nq - z_mq)^2), u = alpha/(alpha S + 1)   (kernel_exp.py:80)
// z_m lives in registers (QP = Q rounded up to 2, zero padded); the packed per-point records PU[n] = [mu_n | u_n | ln c1_n]
// (written by the prep kernels) are staged through LDS; no guards in the q loop (padding has u = 0).
constexpr int PSI1_ROWS = 128;   // row granule of psi1_kernel (Np is a multiple of 128); a workgroup takes `nblk` granules
// FIXA (fixed embeddings, every variance zero): u_nq = alpha_q and ln c1 = ln sf2 for every point, so the records carry only
// sqrt(alpha) o mu (scaled while they are staged), z is scaled once per lane and the exponent is -1/2 sum_q (mu' - z')^2:
// 2 Q + 18 issue slots per element instead of 3 Q + 20, and nothing per point depends on the hyper-parameters (the prep
// kernels are skipped from the second evaluation on).
// SL > 0 (int8 phase 1, p1i8.hip; only the WC = 4 form, where a wave walks all 16 rows of a group): the element is also written as SL signed
// 7-bit digits of t = Psi1 / (2 sf2) in (0, 1/2], sixteen consecutive rows packed into one 16-byte store per column and digit.
template <int QP, bool FIXA, int SL = 0, bool TEMPORAL = false>
__global__ void __launch_bounds__(256) psi1_kernel(const double* __restrict__ PU, const double* __restrict__ Z, double* __restrict__ Kaug,
                                                   long N, long Np, int M, int Mp, int Q, long ld, int WC, const double* __restrict__ alpha,
                                                   double lnsf2, int nblk, int8_t* __restrict__ Sl = nullptr, long strideJ = 0, double hscale = 0.0,
                                                   double* __restrict__ Dpart = nullptr) {
  // A workgroup writes 16 rows x 512 columns: wave w owns 128 columns (two adjacent per lane -> one 16-byte store per lane),
  // so a row's 4 KB leave the CU together (one DRAM page) instead of from four workgroups on four XCDs.  The rows' packed
  // [mu | u | lnc1] records are staged in LDS (below).
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // WC = waves across the columns (4 for Mp >= 512; 1 or 2 for narrower matrices, where the other 4 / WC waves take every
  // (4 / WC)-th row of a 16-row group instead of idling: M = 128 ran at a quarter of the rate)
  const int RG = 4 / WC, rg = wave / WC;
  const int col = (blockIdx.x * WC + (wave % WC)) * 128 + 2 * lane;
  const long row0 = blockIdx.y * 16L * nblk;      // nblk = 16-row groups per workgroup (8 per 128-row granule): the per-lane set-up (z, sqrt(alpha)) is paid once
  const bool ok0 = col < M, ok1 = col + 1 < M;
  const ExpTab xt = exp_tab_lane();
  double z0[QP], z1[QP];
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    const double sa = (FIXA && q < Q) ? sqrt(alpha[q]) : 1.0;
    z0[q] = (q < Q && ok0) ? sa * Z[(long)col * Q + q] : 0.0;
    z1[q] = (q < Q && ok1) ? sa * Z[(long)(col + 1) * Q + q] : 0.0;
  }
  constexpr int W = 2 * QP + 2;          // row width of PU (doubles), a multiple of 2
  constexpr int WS = FIXA ? QP : W;      // staged doubles per row
  // Records of 16 rows at a time: one coalesced load into LDS, read back as broadcast operands; the next group's records
  // travel while the current group is computed.  (Per-row scalar loads cost a serial memory round trip per row and held the
  // kernel at 1.9 ms although plain stores reach 5.5 TB/s: tools/ubench/store_ubench.hip.)
  constexpr int GR = 16, RPT = (GR * WS + 255) / 256;
  const int NG = (int)min((long)nblk, (Np - row0) / GR);   // Np is a multiple of 128: whole groups only
  __shared__ double rec_s[2][GR * WS];
  double stage[RPT];
  double dsq0 = 0.0, dsq1 = 0.0;            // SL > 0: sum of squares of the lane's two columns over this workgroup's rows (the exact diagonal of Psi2, p1i8.hip)
  auto fetch = [&](int g, int i) -> double {
    const int e = threadIdx.x + 256 * i;
    if (e >= GR * WS) return 0.0;
    if (!FIXA) return PU[(row0 + GR * g) * W + e];
    const int r = e / QP, q = e - r * QP;
    return (q < Q ? sqrt(alpha[q]) : 0.0) * PU[(row0 + GR * g + r) * W + q];
  };
#pragma unroll
  for (int i = 0; i < RPT; ++i) { const int e = threadIdx.x + 256 * i; if (e < GR * WS) rec_s[0][e] = fetch(0, i); }
  __syncthreads();
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) {
#pragma unroll
      for (int i = 0; i < RPT; ++i) stage[i] = fetch(g + 1, i);
    }
    const double* recs = rec_s[g & 1];
    if constexpr (SL > 0) {
      // all 16 rows of the group by this wave (RG = 1), unrolled: the digits of the two columns collect in 2 SL x 4 registers, one 16-byte store
      // per digit and column
      unsigned pk0[SL][GR / 4], pk1[SL][GR / 4];
#pragma unroll
      for (int j = 0; j < SL; ++j)
#pragma unroll
        for (int w = 0; w < GR / 4; ++w) { pk0[j][w] = 0u; pk1[j][w] = 0u; }
#pragma unroll
      for (int r = 0; r < GR; ++r) {
      const long n = row0 + GR * g + r;       // Np is a multiple of 128: always in range
      const double* row = recs + r * WS;
      double e0 = 0.0, e1 = 0.0;
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const double d0 = row[q] - z0[q], d1 = row[q] - z1[q];
        if (FIXA) {
          e0 = fma(d0, d0, e0);
          e1 = fma(d1, d1, e1);
        } else {
          e0 = fma(row[QP + q] * d0, d0, e0);
          e1 = fma(row[QP + q] * d1, d1, e1);
        }
      }
      const double l0 = FIXA ? lnsf2 : row[2 * QP];
      const double x0 = fexp_t(fma(-0.5, e0, l0), xt), x1 = fexp_t(fma(-0.5, e1, l0), xt);   // all lanes (cross-lane table lookup), then select
      double2 v;
      v.x = (n < N && ok0) ? x0 : 0.0;
      v.y = (n < N && ok1) ? x1 : 0.0;
      // non-temporal (one global_store_dwordx4 ... nt per lane): the 4.1 GB of Psi1 is next read after the whole array has been written,
      // so keeping it in L2 / MALL only evicts what the statistics kernels are about to use -- same-box A/B (three rounds): this kernel
      // 0.912 -> 0.932 ms, but p1v2_kernel 5.860 -> 5.795 and p2_fast8_kernel 10.051 -> 9.995 ms: evaluation 17.457 -> 17.354 ms
      if (col < Mp) { __builtin_nontemporal_store(v.x, &Kaug[n * ld + col]); __builtin_nontemporal_store(v.y, &Kaug[n * ld + col + 1]); }
      dsq0 = fma(v.x, v.x, dsq0); dsq1 = fma(v.y, v.y, dsq1);
      // SL rounds of multiply, round to nearest, subtract: symmetric digits in [-64, 64] (mean zero: the dropped tail does not bias the sums; digits
      // cut from the bit fields of one integer lie in [-64, 63] and need a per-element offset hash for that -- r05, removed with the int8 phase 2)
      double t0 = v.x * hscale, t1 = v.y * hscale;
#pragma unroll
      for (int j = 0; j < SL; ++j) {
        t0 *= 128.0; t1 *= 128.0;
        const double g0 = __builtin_rint(t0), g1 = __builtin_rint(t1);
        t0 -= g0; t1 -= g1;
        pk0[j][r >> 2] |= ((unsigned)(int)g0 & 0xffu) << (8 * (r & 3));
        pk1[j][r >> 2] |= ((unsigned)(int)g1 & 0xffu) << (8 * (r & 3));
      }
      }
      if (col < Mp) {
        int8_t* dst = Sl + (((row0 + GR * g) / 16) * ld + col) * 16;
#pragma unroll
        for (int j = 0; j < SL; ++j) {
          uint4 a0 = {pk0[j][0], pk0[j][1], pk0[j][2], pk0[j][3]}, a1 = {pk1[j][0], pk1[j][1], pk1[j][2], pk1[j][3]};
          *(uint4*)(dst + (long)j * strideJ) = a0;
          *(uint4*)(dst + (long)j * strideJ + 16) = a1;
        }
      }
    } else {
#pragma unroll 1
    for (int r = rg; r < GR; r += RG) {
      const long n = row0 + GR * g + r;       // Np is a multiple of 128: always in range
      const double* row = recs + r * WS;
      double e0 = 0.0, e1 = 0.0;
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const double d0 = row[q] - z0[q], d1 = row[q] - z1[q];
        if (FIXA) {
          e0 = fma(d0, d0, e0);
          e1 = fma(d1, d1, e1);
        } else {
          e0 = fma(row[QP + q] * d0, d0, e0);
          e1 = fma(row[QP + q] * d1, d1, e1);
        }
      }
      const double l0 = FIXA ? lnsf2 : row[2 * QP];
      const double x0 = fexp_t(fma(-0.5, e0, l0), xt), x1 = fexp_t(fma(-0.5, e1, l0), xt);   // all lanes (cross-lane table lookup), then select
      double2 v;
      v.x = (n < N && ok0) ? x0 : 0.0;
      v.y = (n < N && ok1) ? x1 : 0.0;
      // non-temporal (one global_store_dwordx4 ... nt per lane): the 4.1 GB of Psi1 is next read after the whole array has been written,
      // so keeping it in L2 / MALL only evicts what the statistics kernels are about to use -- same-box A/B (three rounds): this kernel
      // 0.912 -> 0.932 ms, but p1v2_kernel 5.860 -> 5.795 and p2_fast8_kernel 10.051 -> 9.995 ms: evaluation 17.457 -> 17.354 ms
      // ... on a SHORT shard (temporal: [Psi1 | Y] <= 200 MB) plain stores: 50.6 -> 45.3 us at N = 1e5, M = 128 on one box (the kernels behind it do not change)
      // (a compile-time switch: as a run-time branch it broke the merge of the two non-temporal stores into one 16-byte store -- 0.96 -> 2.28 ms at N = 1e6)
      if (col < Mp) {
        if constexpr (TEMPORAL) *reinterpret_cast<double2*>(&Kaug[n * ld + col]) = v;
        else { __builtin_nontemporal_store(v.x, &Kaug[n * ld + col]); __builtin_nontemporal_store(v.y, &Kaug[n * ld + col + 1]); }
      }
    }
    }
    if (g + 1 < NG) {
#pragma unroll
      for (int i = 0; i < RPT; ++i) { const int e = threadIdx.x + 256 * i; if (e < GR * WS) rec_s[(g + 1) & 1][e] = stage[i]; }
    }
    __syncthreads();
  }
  if constexpr (SL > 0) {
    if (Dpart && col < Mp) { Dpart[(long)blockIdx.y * Mp + col] = dsq0; Dpart[(long)blockIdx.y * Mp + col + 1] = dsq1; }
  }
}

// Wide latent spaces (33 <= Q <= 64, records padded to QP = 52 / 64): the same kernel with ONE column per lane (z_m is QP
// registers) and four waves = 256 columns per workgroup.  17 <= Q <= 32 (QP = 24 / 32) still fit two columns per lane and use psi1_kernel (142 / 170 VGPRs):
// N = 1e6, M = 512, Q = 30: 2.73 -> 2.47 ms.  Tried here and dropped (r04), all inside 3 % of this form once the fixed-variance arithmetic was used
// (the kernel is VALU-bound, not LDS-bound): the records as scalar loads / SGPR operands instead of LDS broadcasts (4.7 ms at Q = 30: hipcc waits for each
// s_load_dwordx16 right behind its issue); a lane PAIR per column pair with the latent dimensions split between the two lanes and one DPP exchange per
// row (half the LDS reads per output, z in 2 x QP/2 registers: 1.89 vs 1.89 ms at Q = 30, 1.49 vs 1.41 at Q = 20, 1.85 vs 2.03 at Q = 60); the same with
// explicit ds_read_b64 one chunk ahead (slower: 2.9 ms).
template <int QP, bool FIXA>
__global__ void __launch_bounds__(256) psi1_wide_kernel(const double* __restrict__ PU, const double* __restrict__ Z, double* __restrict__ Kaug,
                                                        long N, long Np, int M, int Mp, int Q, long ld, const double* __restrict__ alpha,
                                                        double lnsf2, int nblk) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = (blockIdx.x * 4 + wave) * 64 + lane;
  const long row0 = blockIdx.y * (long)PSI1_ROWS * nblk;
  const bool ok = col < M;
  const ExpTab xt = exp_tab_lane();
  double z[QP];
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    const double sa = (FIXA && q < Q) ? sqrt(alpha[q]) : 1.0;
    z[q] = (q < Q && ok) ? sa * Z[(long)col * Q + q] : 0.0;
  }
  constexpr int W = 2 * QP + 2, WS = FIXA ? QP : W;
  constexpr int GR = 16, RPT = (GR * WS + 255) / 256;
  const int NG = (int)min((long)nblk * (PSI1_ROWS / GR), (Np - row0) / GR);
  __shared__ double rec_s[2][GR * WS];
  double stage[RPT];
  auto fetch = [&](int g, int i) -> double {
    const int e = threadIdx.x + 256 * i;
    if (e >= GR * WS) return 0.0;
    if (!FIXA) return PU[(row0 + GR * g) * W + e];
    const int r = e / QP, q = e - r * QP;
    return (q < Q ? sqrt(alpha[q]) : 0.0) * PU[(row0 + GR * g + r) * W + q];
  };
#pragma unroll
  for (int i = 0; i < RPT; ++i) { const int e = threadIdx.x + 256 * i; if (e < GR * WS) rec_s[0][e] = fetch(0, i); }
  __syncthreads();
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) {
#pragma unroll
      for (int i = 0; i < RPT; ++i) stage[i] = fetch(g + 1, i);
    }
    const double* recs = rec_s[g & 1];
#pragma unroll 1
    for (int r = 0; r < GR; ++r) {
      const long n = row0 + GR * g + r;
      const double* row = recs + r * WS;
      double e0 = 0.0, e1 = 0.0;               // two partial sums: the chain of QP dependent FMAs is the latency here
#pragma unroll
      for (int q = 0; q < QP; q += 2) {
        const double d0 = row[q] - z[q], d1 = row[q + 1] - z[q + 1];
        if (FIXA) { e0 = fma(d0, d0, e0); e1 = fma(d1, d1, e1); }
        else { e0 = fma(row[QP + q] * d0, d0, e0); e1 = fma(row[QP + q + 1] * d1, d1, e1); }
      }
      const double x = fexp_t(fma(-0.5, e0 + e1, FIXA ? lnsf2 : row[2 * QP]), xt);   // all lanes, then select
      if (col < Mp) Kaug[n * ld + col] = (n < N && ok) ? x : 0.0;
    }
    if (g + 1 < NG) {
#pragma unroll
      for (int i = 0; i < RPT; ++i) { const int e = threadIdx.x + 256 * i; if (e < GR * WS) rec_s[(g + 1) & 1][e] = stage[i]; }
    }
    __syncthreads();
  }
}

// generic fallback (any Q): operands from global memory
__global__ void __launch_bounds__(256) psi1_generic_kernel(const double* __restrict__ mu, const double* __restrict__ U,
                                                           const double* __restrict__ lnc1, const double* __restrict__ Z,
                                                           double* __restrict__ Kaug, long N, long Np, int M, int Q, long ld) {
  const int col = blockIdx.x * 128 + (threadIdx.x & 127);
  const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);
  const long row0 = blockIdx.y * 64L + half * 32;
  const bool colok = col < M;
  for (int r = 0; r < 32; ++r) {
    const long n = row0 + r;
    double e = 0.0;
    for (int q = 0; q < Q; ++q) {
      const double d = mu[n * Q + q] - (colok ? Z[(long)col * Q + q] : 0.0);
      e = fma(U[n * Q + q] * d, d, e);
    }
    Kaug[n * ld + col] = (n < N && colok) ? exp(lnc1[n] - 0.5 * e) : 0.0;
  }
}

// ------------------------------------------------------------------------------------------------ phase 1
// out[i][j] = sum_n Kaug[n][ti*128 + i] * Kaug[n][tj*128 + j] over the slice's rows: Psi2 tiles (tj < Mp/128,
// only tj >= ti) and C = Psi1^T Y tiles (tj >= Mp/128).  Split over n into slices; partial tiles are summed by
// p1_reduce_kernel.  All tile types of one slice sit on one XCD (block b runs on XCD b % 8) so the slice's rows
// are fetched from HBM once and re-read from that XCD's L2.
struct P1Args {
  const double* Kaug; long ld; const int* tiles; int T; int S; int cps; int total_chunks; double* part; const int* bmap;
};

// Eight waves per 128x128 workgroup tile: each 64x64 quadrant is shared by two waves (32 columns each): 32 accumulators per wave,
// 100 VGPRs, four waves per SIMD to cover the barrier / LDS latency that two waves per SIMD leave exposed (a four-wave variant with
// 64x64 wave tiles measured 7.22 ms against 6.90 ms).
__global__ void __launch_bounds__(512, 4) p1_kernel8(P1Args p) {
  const int slice = p.bmap[2 * blockIdx.x], type = p.bmap[2 * blockIdx.x + 1];
  if (slice < 0) return;
  const int ti = p.tiles[2 * type], tj = p.tiles[2 * type + 1];
  __shared__ __attribute__((aligned(16))) double lds[2][2][TILE_LDS_DOUBLES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int quad = wave & 3, half = wave >> 2;
  const int wrow0 = (quad >> 1) * WT, wcol0 = (quad & 1) * WT;
  const bool skip = (ti == tj) && (quad == 2);
  const int c0 = slice * p.cps, c1 = min(p.total_chunks, c0 + p.cps);
  const double* Ab = p.Kaug + (long)ti * TILE + (long)c0 * KC * p.ld;
  const double* Bb = p.Kaug + (long)tj * TILE + (long)c0 * KC * p.ld;
  const long step = (long)KC * p.ld;
  const int nc = c1 - c0;
  double acc[4][8];
#pragma unroll
  for (int ar = 0; ar < 4; ++ar)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[ar][j] = 0.0;
  const int lr = lane & 15, lk = lane >> 4, lj = lane & 3;
  const int aofs = lk * LDS_RC + wrow0 + lr;
  const int bofs = lk * LDS_RC + wcol0 + 32 * half + lj;
  auto dma = [&](int buf, const double* a, const double* b) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wave * 2 + i;
      glds16(a + (long)row * p.ld + 2 * lane, lds[buf][0] + row * LDS_RC);
      glds16(b + (long)row * p.ld + 2 * lane, lds[buf][1] + row * LDS_RC);
    }
  };
  dma(0, Ab, Bb);
  dma_wait();
  __syncthreads();
  for (int c = 0; c < nc; ++c) {
    const int cur = c & 1;
    if (c + 1 < nc) dma(cur ^ 1, Ab + (long)(c + 1) * step, Bb + (long)(c + 1) * step);
    if (!skip) {
      // operand reads as explicit ds_read_b64 (mma_f64.h): twice the LDS rate of the ds_read2_b64 pairs hipcc would form
      const unsigned aA = lds_byte_addr(lds[cur][0]) + 8u * (unsigned)aofs;
      const unsigned aB = lds_byte_addr(lds[cur][1]) + 8u * (unsigned)bofs;
      static_for<0, KC / 4>([&](auto k4c) {
        constexpr int k4 = decltype(k4c)::value;
        double a[4], b[8];
        static_for<0, 4>([&](auto ic) { constexpr int ar = decltype(ic)::value; a[ar] = ds_read64<k4 * 4 * LDS_RC * 8 + 128 * ar>(aA); });
        static_for<0, 8>([&](auto jc) { constexpr int j = decltype(jc)::value; b[j] = ds_read64<k4 * 4 * LDS_RC * 8 + 32 * j>(aB); });
        static_for<0, 8>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          lgkm_wait<7 - j>();
#pragma unroll
          for (int ar = 0; ar < 4; ++ar) mfma444_acc(acc[ar][j], a[ar], b[j]);
        });
      });
    }
    dma_wait();
    __syncthreads();
  }
  mfma_drain(acc[3][7]);
#pragma unroll
  for (int ar = 0; ar < 4; ++ar) acc_fence8(acc[ar]);
  double* out = p.part + ((long)slice * p.T + type) * (TILE * TILE);
#pragma unroll
  for (int ar = 0; ar < 4; ++ar)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      out[(wrow0 + acc_row(ar, lane)) * TILE + wcol0 + 32 * half + acc_col(j, lane)] = acc[ar][j];
}

// sums the slices; writes Psi2 (both triangles) and C into the packed statistics buffer
__global__ void __launch_bounds__(256) p1_reduce_kernel(const double* __restrict__ part, const int* __restrict__ tiles, int T, int S,
                                                        double* __restrict__ Psi2, double* __restrict__ C, int Mp, int Dp) {
  const int type = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;  // element of the 128x128 tile
  const int r = e >> 7, c = e & 127;
  const int ti = tiles[2 * type], tj = tiles[2 * type + 1];
  const int mt = Mp / TILE;
  if (ti == tj && r >= WT && c < WT) return;  // region skipped by p1_kernel, filled by its mirror
  double s = 0.0;
  for (int sl = 0; sl < S; ++sl) s += part[((long)sl * T + type) * (TILE * TILE) + e];
  if (tj < mt) {
    const long R = (long)ti * TILE + r, Cc = (long)tj * TILE + c;
    Psi2[R * Mp + Cc] = s;
    if (ti != tj || (r < WT && c >= WT)) Psi2[Cc * Mp + R] = s;
  } else {
    C[((long)ti * TILE + r) * Dp + (long)(tj - mt) * TILE + c] = s;
  }
}

// scalars at the tail of the stats buffer: sum_YYT, Psi0 = sf2 * N_local, KL, N_local
__global__ void p1_scalars_kernel(const double* klpart, int nblocks, double sumYY, double sf2, double nlocal, int regimeA, double* sc) {
  __shared__ double red[256];
  double part = 0.0;
  if (!regimeA) for (int i = threadIdx.x; i < nblocks; i += 256) part += klpart[i];
  red[threadIdx.x] = part;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double kl = red[0];
    sc[SC_SUM_YYT] = sumYY;
    sc[SC_PSI0] = sf2 * nlocal;
    sc[SC_KL] = regimeA ? 0.0 : kl;
    sc[SC_NLOCAL] = nlocal;
    for (int i = 4; i < SC_COUNT; ++i) sc[i] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------------ host side
bool p2_fast_mode(const gp_ctx* c);
bool p2_wide_fixed_mode(const gp_ctx* c);

int run_upload_y(gp_ctx* c, const double* dY) {
  const long total = c->Np * (long)c->Dp;
  const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(copy_y_kernel, dim3(blocks), dim3(256), 0, c->stream, dY, c->Kaug, (long)c->N, (long)c->Np, c->D, c->Dp, c->Mp,
                     (long)c->LDK);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

template <int QP>
static void launch_psi1(gp_ctx* c, bool fixa) {
  const int WC = c->Mp >= 512 ? 4 : (c->Mp >= 256 ? 2 : 1);
  // 16-row groups per workgroup: 512 rows on large shards (still >= 7 workgroups per CU at N = 1e6), 128 rows below 2^17 rows (64-row workgroups on
  // configs[1]'s 1e5 x 128 shard -- every workgroup resident at once -- changed nothing: 43-44 us, r05)
  const int ngrp = c->Np >= (1L << 17) ? 32 : 8;
  const int nblk = ngrp;
  static const int temporal_env = [] { const char* e = getenv("GPARML_PSI1_TEMPORAL"); return e ? atoi(e) : -1; }();
  const int temporal = temporal_env >= 0 ? temporal_env : ((size_t)c->Np * c->LDK * sizeof(double) <= ((size_t)200 << 20) ? 1 : 0);
  dim3 grid((c->Mp + 128 * WC - 1) / (128 * WC), (unsigned)((c->Np / 16 + ngrp - 1) / ngrp));
  if constexpr (QP <= 16) if (fixa && WC == 4 && c->i8_active) {
    // int8 phase 1 (p1i8.hip): Psi1's digits are written next to Psi1 itself
    int8_t* Sl = nullptr; long strideJ = 0; double* Dpart = nullptr;
    if (p1i8_prepare(c, &Sl, &strideJ, &Dpart, (int)grid.y) == GP_OK) {
      hipLaunchKernelGGL((psi1_kernel<QP, true, GP_I8_DIGITS>), grid, dim3(256), 0, c->stream, c->PU, c->Z, c->Kaug, (long)c->N, (long)c->Np, c->M, c->Mp, c->Q,
                         (long)c->LDK, WC, (const double*)c->alpha, log(c->sf2), nblk, Sl, strideJ, 0.5 / c->sf2, Dpart);
      return;
    }
    c->i8_active = false;
  }
  if (fixa)
  {
    if (temporal)
      hipLaunchKernelGGL((psi1_kernel<QP, true, 0, true>), grid, dim3(256), 0, c->stream, c->PU, c->Z, c->Kaug, (long)c->N, (long)c->Np, c->M, c->Mp, c->Q,
                         (long)c->LDK, WC, (const double*)c->alpha, log(c->sf2), nblk);
    else
      hipLaunchKernelGGL((psi1_kernel<QP, true>), grid, dim3(256), 0, c->stream, c->PU, c->Z, c->Kaug, (long)c->N, (long)c->Np, c->M, c->Mp, c->Q,
                         (long)c->LDK, WC, (const double*)c->alpha, log(c->sf2), nblk);
  }
  else
    hipLaunchKernelGGL((psi1_kernel<QP, false>), grid, dim3(256), 0, c->stream, c->PU, c->Z, c->Kaug, (long)c->N, (long)c->Np, c->M, c->Mp, c->Q,
                       (long)c->LDK, WC, (const double*)c->alpha, 0.0, nblk);
}

template <int QP>
static void launch_psi1_wide(gp_ctx* c, bool fixa) {
  const int nblk = c->Np >= (1L << 17) ? 4 : 1;
  dim3 grid((c->Mp + 255) / 256, (unsigned)((c->Np / PSI1_ROWS + nblk - 1) / nblk));
  if (fixa)
    hipLaunchKernelGGL((psi1_wide_kernel<QP, true>), grid, dim3(256), 0, c->stream, c->PU, c->Z, c->Kaug, (long)c->N, (long)c->Np, c->M, c->Mp, c->Q,
                       (long)c->LDK, (const double*)c->alpha, log(c->sf2), nblk);
  else
    hipLaunchKernelGGL((psi1_wide_kernel<QP, false>), grid, dim3(256), 0, c->stream, c->PU, c->Z, c->Kaug, (long)c->N, (long)c->Np, c->M, c->Mp, c->Q,
                       (long)c->LDK, (const double*)c->alpha, 0.0, nblk);
}

int run_prep_and_generate(gp_ctx* c) {
  PrepArgs a;
  a.Xmu = c->Xmu; a.Xs = c->Xs; a.dir = c->have_dir ? c->dir : nullptr; a.alpha = c->alpha;
  a.mu = c->mu; a.S = c->S; a.U = c->U; a.lnc1 = c->lnc1; a.Xa = c->Xa; a.klpart = c->klpart;
  a.N = c->N; a.Np = c->Np; a.Q = c->Q; a.CXp = c->CXp; a.step = c->step; a.sf2 = c->sf2;
  a.raw = c->xs_raw ? 1 : 0; a.regimeA = c->regime_A ? 1 : 0; a.fixedA = p2_fast_mode(c) ? 1 : (p2_wide_fixed_mode(c) ? 2 : 0);
  a.QP = psi1_qp(c->Q); a.PU = a.QP > 0 ? c->PU : nullptr;
  // Fixed embeddings with every variance zero (regime A, no embedding gradients): the trial point is X_mu itself, S = 0, the
  // feature matrix is [mu | 1] and KL = 0 -- nothing the prep kernels write depends on the hyper-parameters, so they run once per
  // upload / mode switch; Psi1 then takes alpha and sf2 as arguments (psi1_kernel<QP, true>).
  const bool fixa = a.fixedA && a.PU != nullptr;
  if (!(fixa && c->prep_fixa_valid)) {
    hipLaunchKernelGGL(prep_elem_kernel, dim3((unsigned)std::min<long>((c->Np * c->Q + 255) / 256, 16384)), dim3(256), 0, c->stream, a);
    hipLaunchKernelGGL(prep_row_kernel, dim3(c->kl_blocks), dim3(256), 0, c->stream, a);
    GP_HIP(c, hipGetLastError());
  }
  c->prep_fixa_valid = fixa;
  c->i8_active = fixa && p1i8_applicable(c);     // decided per evaluation (gp_debug_set_option("p1_i8", ...) switches it at run time)
  GP_EV(c, 8);
  const int QP = psi1_qp(c->Q);
  // The Psi1 kernels' fixed-variance form (u = alpha, ln c1 = ln sf2: 2 Q + 14 issue slots per element instead of 3 Q + 14) only needs every variance
  // to be zero -- not the fixed-embedding FEATURE layout, which the fast kernel uses for Q + 1 <= 12 -- so it also serves regime A with embedding gradients and
  // wider latent spaces (when measured: N = 1e6, M = 512, Q = 30: 2.70 -> 1.89 ms).  The int8 digits (launch_psi1) stay tied to `fixa` through c->i8_active.
  const bool kfix = c->regime_A && a.PU != nullptr;
  if (QP > 16) {
    switch (QP) {
      case 24: launch_psi1<24>(c, kfix); break;       // two columns per lane as long as 4 QP registers of z fit (psi1_wide_kernel's comment)
      case 32: launch_psi1<32>(c, kfix); break;
      case 52: launch_psi1_wide<52>(c, kfix); break;
      default: launch_psi1_wide<64>(c, kfix); break;
    }
  } else if (QP > 0) {
    switch (QP) {
      case 2: launch_psi1<2>(c, kfix); break;
      case 4: launch_psi1<4>(c, kfix); break;
      case 6: launch_psi1<6>(c, kfix); break;
      case 8: launch_psi1<8>(c, kfix); break;
      case 10: launch_psi1<10>(c, kfix); break;
      case 12: launch_psi1<12>(c, kfix); break;
      case 14: launch_psi1<14>(c, kfix); break;
      default: launch_psi1<16>(c, kfix); break;
    }
  } else {
    dim3 grid(c->Mp / 128, (unsigned)(c->Np / 64));
    hipLaunchKernelGGL(psi1_generic_kernel, grid, dim3(256), 0, c->stream, c->mu, c->U, c->lnc1, c->Z, c->Kaug, (long)c->N, (long)c->Np, c->M,
                       c->Q, (long)c->LDK);
  }
  GP_EV(c, 9);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

int run_phase1(gp_ctx* c) {
  if (c->i8_active) {
    // fixed embeddings on the int8 matrix core: exact integer products of the 35-bit digits psi1_kernel wrote (p1i8.hip)
    int rc = run_phase1_i8(c);
    if (rc != GP_OK) return rc;
    // the guard (p1i8.hip): the first int8 evaluation after an upload and every 64th one also run the float64 phase 1, compare the two sets of
    // statistics on the device and carry on with the float64 ones; gp_finish decides whether the context stays on the int8 path
    if (c->i8_guard == 0 || ++c->i8_since_check >= 64) {
      GP_TRY_RC(p1i8_check_begin(c));
      c->i8_active = false;                         // the float64 phase 1 this shape takes by default (p1v2_kernel or the tile kernel)
      const int rc64 = run_phase1(c);
      c->i8_active = true;
      if (rc64 != GP_OK) return rc64;
      return p1i8_check_compare(c);
    }
    hipLaunchKernelGGL(p1_scalars_kernel, dim3(1), dim3(256), 0, c->stream, c->klpart, c->kl_blocks, c->sumYY, c->sf2, (double)c->N,
                       1, c->stats + (long)c->Mp * c->Mp + (long)c->Mp * c->Dp);
    GP_HIP(c, hipGetLastError());
    return GP_OK;
  }
  if (p1v2_applicable(c)) {
    // fixed embeddings / sparse GP: the decomposition without wasted tile slots (p1v2.hip)
    // (the four scalars of regime A -- sum_YYT, Psi0 = sf2 N, KL = 0, n -- are written by p1v2_reduce_kernel: one launch less)
    return run_phase1_v2(c);
  }
  const int mt = c->Mp / TILE;
  // regime A: Psi2 = Psi1^T Psi1 and C tiles; regime B: only the C tiles here (Psi2 comes from the pair kernel)
  const int first = c->regime_A ? 0 : c->n_tiles - mt * (c->Dp / TILE);
  const int T = c->n_tiles - first;
  P1Args p;
  p.Kaug = c->Kaug; p.ld = c->LDK; p.tiles = c->tiles + 2 * first; p.T = T;
  p.total_chunks = (int)(c->Np / KC);
  // Placement (block b runs on XCD b % 8, 64 resident workgroups per XCD at 2 per CU): every XCD gets L = 64/T whole
  // slices (all T tile types of a slice share the XCD's L2, so the slice's rows are fetched from HBM once); the 64 - L*T
  // left-over slots per XCD are pooled into "shared" slices whose tile types are spread over neighbouring XCDs.  All
  // workgroups are resident at once -- a 65th workgroup on an XCD would wait for a whole first round.
  const int L = 64 / std::max(T, 1);
  const int left = 64 - L * T;
  const int n_shared = (T <= 64) ? (8 * left) / T : std::max(1, 512 / T);
  int S = std::max(1, std::min(8 * L + n_shared, p.total_chunks));
  p.cps = (p.total_chunks + S - 1) / S;
  S = (p.total_chunks + p.cps - 1) / p.cps;
  p.S = S; p.part = c->part;
  if (c->bmap_T != T || c->bmap_S != S) {
    const int per_xcd = (T <= 64) ? 64 : (S * T + 7) / 8;
    std::vector<int> slot(8 * per_xcd * 2, -1);     // [xcd][j] -> (slice, type)
    std::vector<int> fill(8, 0);
    int sl = 0;
    for (int x = 0; x < 8 && T <= 64; ++x)
      for (int l = 0; l < L && sl < S; ++l, ++sl)
        for (int t = 0; t < T; ++t) { const int j = fill[x]++; slot[(x * per_xcd + j) * 2] = sl; slot[(x * per_xcd + j) * 2 + 1] = t; }
    int x = 0;
    for (; sl < S; ++sl)
      for (int t = 0; t < T; ++t) {
        while (fill[x] >= per_xcd) x = (x + 1) & 7;
        const int j = fill[x]++;
        slot[(x * per_xcd + j) * 2] = sl; slot[(x * per_xcd + j) * 2 + 1] = t;
        if (fill[x] >= per_xcd || (T <= 64 && fill[x] - L * T >= (left + 1) / 2 * 2 && false)) x = (x + 1) & 7;
      }
    std::vector<int> bm(8 * per_xcd * 2);
    for (int j = 0; j < per_xcd; ++j)
      for (int xx = 0; xx < 8; ++xx) { bm[(j * 8 + xx) * 2] = slot[(xx * per_xcd + j) * 2]; bm[(j * 8 + xx) * 2 + 1] = slot[(xx * per_xcd + j) * 2 + 1]; }
    if (c->bmap) (void)hipFree(c->bmap);
    GP_HIP(c, hipMalloc((void**)&c->bmap, bm.size() * sizeof(int)));
    GP_HIP(c, hipMemcpyAsync(c->bmap, bm.data(), bm.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    GP_HIP(c, hipStreamSynchronize(c->stream));
    c->bmap_T = T; c->bmap_S = S; c->bmap_blocks = 8 * per_xcd;
  }
  p.bmap = c->bmap;
  const int blocks = c->bmap_blocks;
  GP_EV(c, 10);
  hipLaunchKernelGGL(p1_kernel8, dim3(blocks), dim3(512), 0, c->stream, p);
  GP_EV(c, 11);
  GP_HIP(c, hipGetLastError());
  double* Psi2 = c->stats;
  double* C = c->stats + (long)c->Mp * c->Mp;
  hipLaunchKernelGGL(p1_reduce_kernel, dim3(TILE * TILE / 256, T), dim3(256), 0, c->stream, c->part, p.tiles, T, S, Psi2, C, c->Mp, c->Dp);
  GP_HIP(c, hipGetLastError());
  hipLaunchKernelGGL(p1_scalars_kernel, dim3(1), dim3(256), 0, c->stream, c->klpart, c->kl_blocks, c->sumYY, c->sf2, (double)c->N,
                     c->regime_A ? 1 : 0, c->stats + (long)c->Mp * c->Mp + (long)c->Mp * c->Dp);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

// ------------------------------------------------------------------------------------------------ phase 2
// G = [Psi1 | Y] * [2 Bbar ; Abar^T]  (regime A; regime B uses only the Y block: G = Y Abar^T), W = G o Psi1, then
//   R[m][c]  = sum_n W[n][m] Xa[n][c]    (n-contraction: grad_Z / grad_alpha sums, partial_terms.py:162-188, 256-271)
//   HZ[n][c] = sum_m W[n][m] Zaug[m][c]  (m-contraction: per-point terms of grad_X_mu / grad_X_S / grad_alpha, :367-431)
// A workgroup owns one 128-wide m tile and walks the 128-row n tiles of its slice; the four m tiles of a slice share
// an XCD so the Kaug rows come from HBM once.  W goes through LDS in 16-row slabs to become an MFMA operand.
struct P2Args {
  const double* Kaug; long ld; const double* Bm; const double* Xa; const double* Zaug;
  double* Rpart; double* HZp;
  int Mp, CXp, CZp, MT, S, tps, ntiles, kbeg, kend, klast; long Np;   // klast: k-steps (of 4) of the final chunk that hold real Y columns
  double* gapart;   // eight-wave fast kernel: [blocks * 8][4 NRB] per-wave partials of grad_alpha's mu^2 term
  // r06, p2_fast8_kernel: the MT workgroups of a row slice keep in step tile by tile (prog[slice][mt] = base + tiles started), so that the slice's
  // 128 x (Mp + Dp) rows are fetched from HBM once and found in the XCD's L2 by the other MT - 1 (0 / NULL: free-running)
  unsigned long long* prog; unsigned long long prog_base;
  long long* dbg;   // timing build (GPARML_GEN8_TIMING): [blocks][8 waves][8 sections] s_memtime totals
};

constexpr int SLAB_LD = 66;   // 16 x 64 slab row stride (doubles)


// Fast variant for the fixed-embedding regime (no per-point outputs), Q + 1 <= 4 * NRB <= 12 (r04: the four-wave kernel that served Q + 1 <= 24 is gone --
// from Q = 12 on p2_gen8_kernel<false> on [mu | 1 | mu^2] is faster: same box, N = 1e6, M = 512, ms of the phase-2 kernel at Q = 12 / 16 / 23: 10.43 / 10.74 / 11.30
// against 11.27 / 11.35 / 11.51).  Per-point features are
// Xa = [mu (Q) | 1 | 0..]: R[m][:] = sum_n W[n][m] Xa[n][:] gives W^T mu and W^T 1 (grad_Z, and the z-dependent terms of
// grad_alpha); the remaining term of grad_alpha, sum_nm W[n][m] mu_nq^2 = sum_n h_n mu_nq^2, only needs the row sums
// h_n = sum_m W[n][m], which each wave forms from its accumulator registers.  Compared with
// carrying [mu, mu^2, 1] through the MFMAs this halves the epilogue and the resident R accumulators (no scratch spills).
// (Running the k-loop in rotated order, own Psi1 columns last, so that the epilogue's re-read of that tile hits L2 was tried
// and dropped: the four m-tile workgroups of a slice then stream different k-chunks at any moment and stop sharing the slice's
// rows in L2 -- fabric traffic 15 -> 25 GB per launch, +0.37 ms.)
__device__ __forceinline__ double quad_sum(double v) {   // sum over the four lanes l&3 (same accumulator row)
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  return v;
}
// Eight waves (4 * NRB <= 12, i.e. Q <= 11 -- BASELINE configs[1..3]): a 128 x 128 workgroup tile with every 64 x 64
// quadrant shared by two waves (64 rows x 32 columns: 32 accumulators), four waves per SIMD, and an epilogue that touches
// no global memory from registers: the workgroup's Psi1 tile arrives by LDS-DMA in four 16-row slabs per wave (the first
// one and the tile's Xa rows travel during the last k-chunk, the others while the previous slab is processed), W = G o Psi1
// overwrites the slab in place, and the same slab is read back transposed as the MFMA A operand.  The slab image is
// permuted through the DMA's SOURCE address -- 16-byte pair p of slab row r sits at pair p ^ 2 g(r),
// g(r) = ((r & 1) << 2) | ((r >> 2) & 3) -- so that the accumulator-layout accesses (8 rows x 4 columns per 32-lane group)
// and the transposed operand reads (2 rows x 16 columns) are both free of bank conflicts.
// LDS map (doubles): buf0 [0, 4608) | buf1 [4608, 9216) | extra [9216, 10240).  The last k-chunk always computes from buf0, so
// slab set A (8 waves x 512) + the Xa tile (128 x 4 NRB) occupy buf1 + extra, and slab set B goes to buf0 after the loop.
constexpr int P2W8_LDS = 10240;
__device__ __forceinline__ int slab_g(int r) { return ((r & 1) << 2) | ((r >> 2) & 3); }

template <int NRB>
__global__ void __launch_bounds__(512, 4) p2_fast8_kernel(P2Args p) {
  const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int slice = xcd + 8 * (bi / p.MT), mt = bi % p.MT;
  constexpr int XS = 4 * NRB;
  if (slice >= p.S) {
    // a block past the last slice: its eight rows of the mu^2 partials are read by colsum2_kernel and must be zero (this replaces a
    // hipMemsetAsync of the whole array in front of every launch: a blit dispatch with ~15 us of idle stream around it)
    if (threadIdx.x < 8 * XS) p.gapart[(long)blockIdx.x * 8 * XS + threadIdx.x] = 0.0;
    return;
  }
  __shared__ __attribute__((aligned(16))) double lds[P2W8_LDS];
  static_assert(4096 + TILE * XS <= 4608 + 1024, "Xa tile does not fit next to slab set A");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Static priority for the second-dispatched half of the workgroup: with both halves at priority 0 the younger waves lose the VALU / LDS
  // issue arbitration in every k-chunk (MI355X_MICROARCH.md, "Two waves per SIMD", item 4); same-box A/B: 10.27 -> 10.14 ms
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  const int quad = wave & 3, half = wave >> 2;
  const int wr = quad >> 1, wc = quad & 1;
  const int wrow0 = wr * WT, wcol0 = wc * WT + 32 * half;
  const int lr = lane & 15, lk = lane >> 4, lj = lane & 3;
  const int srow = 4 * ((lane >> 2) & 3) + (lane >> 4);
  const LaneOfs ofs = lane_offsets<K_CONTIG, FREE_CONTIG>(wrow0, wcol0, lane);
  const int nc = p.kend - p.kbeg;
  const int t0 = slice * p.tps, t1 = min(p.ntiles, t0 + p.tps);
  const unsigned lds_base = lds_byte_addr(lds);
  double* const setA = lds + 4608 + wave * 512;
  double* const setB = lds + wave * 512;
  double* const xa_s = lds + 4608 + 4096;
  double r[2][NRB], gq[NRB];
#pragma unroll
  for (int g = 0; g < NRB; ++g) { r[0][g] = 0.0; r[1][g] = 0.0; gq[g] = 0.0; }
  // chunk staging: 16 DMA instructions per operand tile, two per wave; lane offsets are 32-bit (uniform base + offset addressing)
  auto chunk_dma = [&](double* buf, const double* a, const double* b) {
    int ld_ = lane;
    asm volatile("" : "+v"(ld_));                             // recomputed per chunk (a handful of integer ops) instead of held in registers
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int I = wave * 2 + i;
      const int row = swap03(I * 8 + (ld_ >> 3));
      // byte offset as an unsigned 32-bit value: uniform base + 32-bit lane offset addressing (one address register instead of two)
#if !defined(GPARML_ABLATE_P2_DMA) || GPARML_ABLATE_P2_DMA < 2
      glds16(reinterpret_cast<const double*>(reinterpret_cast<const char*>(a) + (unsigned)(8 * (row * (int)p.ld + 2 * ((ld_ & 7) ^ (row & 7))))),
             buf + I * 8 * KC);
#endif
      // ablation builds (tools/r03_p2_ablate.sh; wrong results, same instruction stream otherwise): 1 = the B tile is staged for every
      // second chunk only -- 25 % fewer LDS-DMA instructions per flop, what a 256 x 128 workgroup tile would issue; 2 = no staging at all
#if defined(GPARML_ABLATE_P2_DMA) && GPARML_ABLATE_P2_DMA == 1
      if ((reinterpret_cast<size_t>(b) / (KC * 8 * (size_t)p.Mp)) & 1)
#endif
#if !defined(GPARML_ABLATE_P2_DMA) || GPARML_ABLATE_P2_DMA < 2
      glds16(b + (long)I * p.Mp + 2u * ld_, buf + TILE_LDS_DOUBLES + I * LDS_RC);
#endif
    }
  };
  // Epilogue addressing is derived from an OPAQUE copy of the lane id at its point of use: as loop invariants these values
  // would stay live across the k-loop and push the kernel over its 128 registers (scratch spills).
  struct Epi { int lr, lk, lj, srow, sg, sbase, abase, aflip, dpair; };
  auto epi = [&]() {
    int le = lane;
    asm volatile("" : "+v"(le));
    Epi e;
    e.lr = le & 15; e.lk = le >> 4; e.lj = le & 3;
    e.srow = 4 * ((le >> 2) & 3) + (le >> 4);
    // slab addressing (doubles): accumulator layout (row srow, column 4 bc + lj) -> srow*32 + 4*(bc ^ g(srow)) + lj
    e.sg = slab_g(e.srow);
    e.sbase = e.srow * 32 + e.lj;
    // operand reads (row 4 k4 + lk, column 16 am + lr) -> (4 k4 + lk)*32 + 2*(((am ^ (lk & 1)) << 3) | ((lr >> 1) ^ (2 k4))) + (lr & 1)
    e.abase = e.lk * 32 + (e.lr & 1) + 16 * (e.lk & 1);     // am = 0; am = 1 adds aflip
    e.aflip = 16 - 32 * (e.lk & 1);
    // slab DMA source: instruction i moves slab rows 4 i + (lane >> 4); lane & 15 = physical pair, g(4 i + lk) = ((lk & 1) << 2) | i
    e.dpair = (le & 15) ^ ((e.lk & 1) << 3);
    return e;
  };
  auto slab_dma = [&](const Epi& e, const double* Ktile, int ar, double* slab) {   // Ktile: row 0, column 0 of the wave's Psi1 block
#pragma unroll
    for (int i = 0; i < 4; ++i)
      glds16(reinterpret_cast<const double*>(reinterpret_cast<const char*>(Ktile) +
                                             (unsigned)(8 * ((16 * ar + 4 * i + e.lk) * (int)p.ld + 2 * (e.dpair ^ (2 * i))))), slab + i * 128);
  };
  bool in_step = p.prog != nullptr;
  for (int nt = t0; nt < t1; ++nt) {
    const long n0 = (long)nt * TILE;
    const double* Ab = p.Kaug + n0 * p.ld + (long)p.kbeg * KC;
    const double* Bb = p.Bm + (long)p.kbeg * KC * p.Mp + (long)mt * TILE;
    double acc[4][8];
#pragma unroll
    for (int ar = 0; ar < 4; ++ar)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[ar][j] = 0.0;
    if (p.prog && nt > t0) {
      // Keep the slice's MT workgroups in step: nothing synchronises them otherwise, they drift apart over the ~60 tiles of a slice, and once the distance
      // exceeds what the XCD's 4 MB L2 holds for its sixteen slices every workgroup fetches the slice's rows from HBM itself (same-box FETCH_SIZE: 11.6 GB per
      // launch in round 4, 14.9 GB in round 5 for the same code, 5.3 GB algorithmic).  Wave 0 publishes "tile nt started" and waits until the others have
      // started it too -- they are resident (the grid is the chip's capacity: two workgroups per CU), on the same XCD (block id mod 8), and at most one tile
      // behind.  The wait is bounded: if a partner does not show up within ~0.1 ms (a grid that is not fully resident), this workgroup keeps publishing but stops waiting for good.
      if (wave == 0) {
        unsigned long long* mine = p.prog + (long)slice * p.MT;
        const unsigned long long want = p.prog_base + (unsigned long long)(nt - t0);
        if (lane == 0) __hip_atomic_store(mine + mt, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (entries of earlier launches carry a smaller base and read as "behind"; the early-exit blocks past the last slice never get here; the other
        // seven waves meet wave 0 at the first chunk's barrier)
        for (int spins = 0; in_step; ++spins) {
          const unsigned long long v = (lane < p.MT) ? __hip_atomic_load(mine + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
          if (__builtin_amdgcn_ballot_w64(v < want) == 0) break;
          if (spins > 2048) in_step = false;
          __builtin_amdgcn_s_sleep(2);
        }
      }
    }
    int kc = 0;
    const int b0 = (nc + 1) & 1;                                // buffer of chunk 0; chunk c uses (c + nc + 1) & 1, the last one buf0
    chunk_dma(lds + b0 * 4608, Ab + (long)kc * KC, Bb + (long)kc * KC * p.Mp);
    dma_wait();
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
      const int cur = (c + nc + 1) & 1;
      // the next chunk's staging (or, in the last chunk, the epilogue's first slab and feature rows) is requested by the first half of
      // the workgroup before k-step 0 and by the second half before k-step 2: the two waves a SIMD hosts from this workgroup are
      // then never in VMEM issue at the same time
      auto issue_next = [&]() {
        if (c + 1 < nc) {
          ++kc;
          chunk_dma(lds + (cur ^ 1) * 4608, Ab + (long)kc * KC, Bb + (long)kc * KC * p.Mp);
        } else {
          // last chunk (computing from buf0): slab 0 of this wave and the tile's Xa rows into buf1 + extra
          const Epi e0 = epi();
          slab_dma(e0, Ab - (long)p.kbeg * KC + (long)wrow0 * p.ld + (long)mt * TILE + wcol0, 0, setA);
          for (int I = wave; I < TILE * XS / 128; I += 8) {       // 128 doubles (64 x 16 B) per instruction
            const int piece = I * 64 + lane;                      // 16-byte piece of the compact [128][XS] tile
            const int row = piece / (XS / 2), c2 = piece - row * (XS / 2);
            glds16(p.Xa + (n0 + row) * p.CXp + 2 * c2, xa_s + I * 128);
          }
        }
      };
      if (half == 0) issue_next();
      // operand reads as explicit ds_read_b64 (see mma_f64.h): 4 A + 8 B per k-step, MFMAs start as soon as A and the first B landed
      const unsigned sbase_b = lds_base + (unsigned)cur * (4608u * 8u);
      const unsigned aB = sbase_b + TILE_LDS_DOUBLES * 8 + 8u * (unsigned)ofs.b[0];
      const bool tail = (c == nc - 1);
      static_for<0, KC / 4>([&](auto k4c) {
        constexpr int k4 = decltype(k4c)::value;
        if constexpr (k4 == 2) { if (half == 1) issue_next(); }
#ifdef GPARML_FAST8_KSKIP
        // Skipping the k-steps that only multiply Y's zero padding (three of 156 at D = 100) is NOT done here: same kernel time (the last chunk of a
        // tile waits for the epilogue's first DMA anyway), but FETCH_SIZE 11.55 -> 14.2 GB per launch, reproducibly (tools/r03_traffic_ab.sh): the
        // shortened last chunk lets the four m-tile workgroups of a slice drift apart and they stop finding each other's rows in L2.
        // p2_gen8_kernel, whose k-loop at D = 100 is seven chunks, keeps the skip (-10 % of its loop).
        if (k4 > 0 && tail && k4 >= p.klast) return;
#endif
        const unsigned aA = sbase_b + 8u * (unsigned)ofs.a[k4];
        double a[4], b[8];
        a[0] = ds_read64<0>(aA); a[1] = ds_read64<2048>(aA); a[2] = ds_read64<4096>(aA); a[3] = ds_read64<6144>(aA);
        static_for<0, 8>([&](auto jc) { constexpr int j = decltype(jc)::value; b[j] = ds_read64<k4 * 4 * LDS_RC * 8 + 32 * j>(aB); });
        static_for<0, 8>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          lgkm_wait<7 - j>();
#pragma unroll
          for (int ar = 0; ar < 4; ++ar) mfma444_acc(acc[ar][j], a[ar], b[j]);
        });
      });
      dma_wait();
      __syncthreads();
    }
    const Epi e = epi();
    const double* Ktile = p.Kaug + (n0 + wrow0) * p.ld + (long)mt * TILE + wcol0;
    slab_dma(e, Ktile, 1, setB);           // buf0 is free now
    mfma_drain(acc[3][7]);
#pragma unroll
    for (int ar = 0; ar < 4; ++ar) acc_fence8(acc[ar]);
#pragma unroll
    for (int ar = 0; ar < 4; ++ar) {
      double* slab = (ar & 1) ? setB : setA;
      if (ar < 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // slab ar has landed; slab ar + 1 (4 DMAs) may still fly
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      double hs = 0.0;
#pragma unroll
      for (int bc = 0; bc < 8; ++bc) {
        const int idx = e.sbase + 4 * (bc ^ e.sg);
        const double w = acc[ar][bc] * slab[idx];
        slab[idx] = w;
        hs += w;
      }
      // grad_alpha's mu^2 term: -1/2 sum_n h_n mu_nq^2 with h_n = this wave's row sum of W; lane (row, lj) takes q = lj + 4 i
      // (column Q of Xa is the constant 1 and the padding is 0: those accumulators are never read)
      hs = -0.5 * quad_sum(hs);
#pragma unroll
      for (int i = 0; i < NRB; ++i) {
        const double x = xa_s[(wrow0 + 16 * ar + e.srow) * XS + 4 * i + e.lj];
        gq[i] = fma(hs * x, x, gq[i]);
      }
      const double* xrow = xa_s + (wrow0 + 16 * ar + e.lk) * XS + e.lj;
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        double a[2], b[NRB];
        const int ai = e.abase + 128 * k4 + 2 * ((e.lr >> 1) ^ (2 * k4));
        a[0] = slab[ai];
        a[1] = slab[ai + e.aflip];
#pragma unroll
        for (int g = 0; g < NRB; ++g) b[g] = xrow[4 * k4 * XS + 4 * g];
#pragma unroll
        for (int am = 0; am < 2; ++am)
#pragma unroll
          for (int g = 0; g < NRB; ++g) mfma444_acc(r[am][g], a[am], b[g]);
      }
      if (ar < 2) {
        // this slab buffer is free again (its reads were consumed by the MFMAs above): slab ar + 2
        __builtin_amdgcn_sched_barrier(0);
        slab_dma(e, Ktile, ar + 2, slab);
      }
    }
    mfma_drain(r[1][NRB - 1]);
    acc_fence<NRB>(r[0]); acc_fence<NRB>(r[1]);
    __syncthreads();   // slabs / xa_s live in the staging buffers the next tile's DMA overwrites
  }
  double* Rmine = p.Rpart + ((long)(slice * 2 + wr) * p.Mp + (long)mt * TILE + wcol0) * XS;
#pragma unroll
  for (int am = 0; am < 2; ++am)
#pragma unroll
    for (int g = 0; g < NRB; ++g) Rmine[(long)(16 * am + srow) * XS + 4 * g + lj] = (t1 > t0) ? r[am][g] : 0.0;
  // per-wave partial of the mu^2 term: fixed butterfly over the 16 rows (lane bits 2..5), one row of gapart per wave
#pragma unroll
  for (int g = 0; g < NRB; ++g) {
    double v = gq[g];
    v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    if (lane < 4) p.gapart[((long)blockIdx.x * 8 + wave) * XS + 4 * g + lane] = v;
  }
}

// General phase 2 on the eight-wave structure (r03; replaces the four-wave p2_kernel<true>: 256 VGPRs + 320 B of scratch, 0.19 of peak): any
// number of feature columns, with the per-point m-contraction HZ = W Zaug (embedding gradients, partial_terms.py:367-431).  The k-loop is
// p2_fast8_kernel's (64 x 32 wave tiles, LDS-DMA staging, explicit ds_read_b64 operand reads).  The epilogue keeps W = G o Psi1 in the 32
// accumulator registers (Psi1 arrives by LDS-DMA in 16-row slabs) and contracts it twice:
//   n-contraction  R[m][c]  += sum_n W[n][m] Xa[n][c]      straight from the registers: the accumulator layout IS the B operand of the
//                  four-block MFMA, the blocks' partial sums are added through LDS (no transposed copy of W)
//   m-contraction  HZ[n][c]  = sum_m W[n][m] Zaug[m][c]    W through a per-wave LDS slab pair (32 rows), read back transposed as the A
//                  operand; B = the Zaug rows of the wave's 32 columns, eight features at a time, staged by LDS-DMA one group ahead; the four
//                  32-column partials of a row are exchanged through LDS and added in column order, so HZp holds one partial array per
//                  128-column tile
// 127 VGPRs, no scratch; configs[2] shape with free embeddings: 8.19 -> 3.66 ms per 1e6 points (918 -> 403 us per 1e5), point_kernel 2.4 -> 0.21 ms.
constexpr int GSLD = 34;     // slab row stride (doubles): 16 x 32 values per wave
template <bool PPATH>
__global__ void __launch_bounds__(512, 4) p2_gen8_kernel(P2Args p) {
  const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int slice = xcd + 8 * (bi / p.MT), mt = bi % p.MT;
  if (slice >= p.S) return;
  __shared__ __attribute__((aligned(16))) double lds[P2W8_LDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  const int quad = wave & 3, half = wave >> 2;
  const int wr = quad >> 1, wc = quad & 1;
  const int wrow0 = wr * WT, wcol0 = wc * WT + 32 * half;
  const int nc = p.kend - p.kbeg;
  const int t0 = slice * p.tps, t1 = min(p.ntiles, t0 + p.tps);
  const unsigned lds_base = lds_byte_addr(lds);
  // per wave a private 1280-double area: [0, 512) Psi1 slab A | [512, 1056) the W slab, which doubles as Psi1 slab B while W is being formed
  double* const area = lds + wave * 1280;
#ifdef GPARML_GEN8_TIMING
  long long tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define G8SEC(k) { const long long tn_ = __builtin_amdgcn_s_memtime(); tsec[k] += tn_ - tlast; tlast = tn_; }
#else
#define G8SEC(k)
#endif
  double* const slab = area + 512;
  auto chunk_dma = [&](double* buf, const double* a, const double* b) {
    int ld_ = lane;
    asm volatile("" : "+v"(ld_));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int I = wave * 2 + i;
      const int row = swap03(I * 8 + (ld_ >> 3));
      glds16(reinterpret_cast<const double*>(reinterpret_cast<const char*>(a) + (unsigned)(8 * (row * (int)p.ld + 2 * ((ld_ & 7) ^ (row & 7))))),
             buf + I * 8 * KC);
      glds16(b + (long)I * p.Mp + 2u * ld_, buf + TILE_LDS_DOUBLES + I * LDS_RC);
    }
  };
  for (int nt = t0; nt < t1; ++nt) {
    const long n0 = (long)nt * TILE;
    const double* Ab = p.Kaug + n0 * p.ld + (long)p.kbeg * KC;
    const double* Bb = p.Bm + (long)p.kbeg * KC * p.Mp + (long)mt * TILE;
    // the k-loop's lane offsets are recomputed per tile from an opaque copy of the lane id: held across the epilogue they are spilled
    int lo_ = lane;
    asm volatile("" : "+v"(lo_));
    const LaneOfs ofs = lane_offsets<K_CONTIG, FREE_CONTIG>(wrow0, wcol0, lo_);
    double acc[4][8];
#pragma unroll
    for (int ar = 0; ar < 4; ++ar)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[ar][j] = 0.0;
#pragma unroll
    for (int ar = 0; ar < 4; ++ar) acc_fence8(acc[ar]);                  // asm-defined zeros: no rematerialised v_mov in front of an asm MFMA
    G8SEC(7)
    chunk_dma(lds, Ab, Bb);
    dma_wait();
    __syncthreads();
    G8SEC(0)
    for (int c = 0; c < nc; ++c) {
      const int cur = c & 1;
      auto issue_next = [&]() { if (c + 1 < nc) chunk_dma(lds + (cur ^ 1) * 4608, Ab + (long)(c + 1) * KC, Bb + (long)(c + 1) * KC * p.Mp); };
      if (half == 0) issue_next();
      const unsigned sbase_b = lds_base + (unsigned)cur * (4608u * 8u);
      const unsigned aB = sbase_b + TILE_LDS_DOUBLES * 8 + 8u * (unsigned)ofs.b[0];
      const bool tail = (c == nc - 1);
      static_for<0, KC / 4>([&](auto k4c) {
        constexpr int k4 = decltype(k4c)::value;
        if constexpr (k4 == 2) { if (half == 1) issue_next(); }
        if (k4 > 0 && tail && k4 >= p.klast) return;   // k-steps past the last real Y column multiply zeros
        const unsigned aA = sbase_b + 8u * (unsigned)ofs.a[k4];
        double a[4], b[8];
        a[0] = ds_read64<0>(aA); a[1] = ds_read64<2048>(aA); a[2] = ds_read64<4096>(aA); a[3] = ds_read64<6144>(aA);
        static_for<0, 8>([&](auto jc) { constexpr int j = decltype(jc)::value; b[j] = ds_read64<k4 * 4 * LDS_RC * 8 + 32 * j>(aB); });
        static_for<0, 8>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          lgkm_wait<7 - j>();
#pragma unroll
          for (int ar = 0; ar < 4; ++ar) mfma444_acc(acc[ar][j], a[ar], b[j]);
        });
      });
      dma_wait();
      __syncthreads();
    }
    mfma_drain(acc[3][7]);
#pragma unroll
    for (int ar = 0; ar < 4; ++ar) acc_fence8(acc[ar]);
    G8SEC(1)
    // ---- epilogue: the staging buffers are free.  Lane coordinates from an opaque copy (not held across the k-loop).
    int le = lane;
    asm volatile("" : "+v"(le));
    const int lk = le >> 4, lj = le & 3;
    const int srow = 4 * ((le >> 2) & 3) + (le >> 4);
    // W = G o Psi1 in the accumulator layout (row 16 ar + srow, column 4 bc + lj of the wave's 64 x 32 block).  The wave's Psi1 block
    // arrives by LDS-DMA in four 16-row slabs, two in flight (register-staged loads of the 32 values do not fit next to the accumulators:
    // the compiler serialised them, one L2 round trip per value); slab image permuted as in p2_fast8_kernel (pair p of row r at p ^ 2 g(r))
#if defined(GPARML_GEN8_ABLATE) && (GPARML_GEN8_ABLATE & 4)
    if (p.MT < 0)
#endif
    {
      const int sg = slab_g(srow), sbase = srow * 32 + lj;
      const int dpair = (le & 15) ^ ((lk & 1) << 3);
      const double* Ktile = p.Kaug + (n0 + wrow0) * p.ld + (long)mt * TILE + wcol0;
      auto slab_dma = [&](int ar, double* dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          glds16(reinterpret_cast<const double*>(reinterpret_cast<const char*>(Ktile) +
                                                 (unsigned)(8 * ((16 * ar + 4 * i + lk) * (int)p.ld + 2 * (dpair ^ (2 * i))))), dst + i * 128);
      };
      slab_dma(0, area);
      slab_dma(1, slab);
#pragma unroll
      for (int ar = 0; ar < 4; ++ar) {
        double* ks = (ar & 1) ? slab : area;
        if (ar < 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // slab ar has landed; slab ar + 1 (4 DMAs) may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int bc = 0; bc < 8; ++bc) acc[ar][bc] *= ks[sbase + 4 * (bc ^ sg)];
        acc_fence8(acc[ar]);
        if (ar < 2) {
          __builtin_amdgcn_sched_barrier(0);                            // the slab's values are in the accumulators: the buffer is free again
          slab_dma(ar + 2, ks);
        }
      }
    }
    G8SEC(2)
#ifdef GPARML_GEN8_ABLATE   // timing experiments only (results wrong): bit 0 = no m-contraction, bit 1 = no n-contraction, bit 2 = no Psi1 product
    if (p.MT < 0) {
#pragma unroll
      for (int ar = 0; ar < 4; ++ar)
#pragma unroll
        for (int bc = 0; bc < 8; ++bc) p.HZp[(ar * 8 + bc) * 64 + le] = acc[ar][bc];
    }
#endif
    // ---- n-contraction, straight from the accumulators: R[m][c] += sum_n W[n][m] Xa[n][c], this wave's 32 columns m.
    // v_mfma_f64_4x4x4_4b computes four independent 4x4x4 blocks; lane l = (k = l >> 4, block b = (l >> 2) & 3, j = l & 3) holds
    // W[16 ar + 4 b + k][4 bc + j] in acc[ar][bc] -- exactly the B operand B_b[k][j] of block b.  With A_b[i][k] = Xa[16 ar + 4 b + k][c0 + i]
    // (lane (i = l & 3, b, k): the same row 16 ar + srow, column c0 + lj) block b accumulates its four points' share of R[4 bc + j][c0 + i];
    // the four block partials are added through the wave's LDS area.  No transposed copy of W is needed.
#if defined(GPARML_GEN8_ABLATE) && (GPARML_GEN8_ABLATE & 2)
    if (p.MT < 0)
#endif
    {
      double* Rmine = p.Rpart + ((long)(slice * 2 + wr) * p.Mp + (long)mt * TILE + wcol0) * p.CXp;
      const double* xbase = p.Xa + (n0 + wrow0) * p.CXp;          // uniform bases + 32-bit lane offsets (scalar-base addressing)
      const int xo = srow * p.CXp + lj;
      const int nqx = p.CXp / 4, bsel = (le >> 2) & 3;
      const int ro = (8 * bsel + lj) * p.CXp + lk;
      const bool first = (nt == t0);
      // A wave issues in order, so whatever is to run under a quad's 32 MFMAs must be REQUESTED before them: quad q - 1's old R values and
      // quad q + 1's feature rows are issued first, the MFMAs of quad q follow, and quad q - 1's partial reads, sums and stores come behind them.
      // (Two quads' accumulators in flight would need 16 more VGPRs than there are: seven W registers spilled.)
      double x[4], xn[4], d[8];
      auto load_x = [&](double (&xv)[4], int q) {
#pragma unroll
        for (int ar = 0; ar < 4; ++ar) xv[ar] = xbase[xo + 16 * ar * p.CXp + 4 * q];
      };
      auto mfmas = [&]() {
        asm volatile("s_nop 1");                           // x may come from a register copy: VALU write -> MFMA operand read needs two wait states
#pragma unroll
        for (int bc = 0; bc < 8; ++bc) mfma444_zero(d[bc], x[0], acc[0][bc]);
#pragma unroll
        for (int ar = 1; ar < 4; ++ar)
#pragma unroll
          for (int bc = 0; bc < 8; ++bc) mfma444_acc(d[bc], x[ar], acc[ar][bc]);
      };
      auto park = [&]() {                                  // the quad's block partials to LDS (the registers are free again once the writes are issued)
        mfma_drain(d[7]);
        acc_fence8(d);
#pragma unroll
        for (int bc = 0; bc < 8; ++bc) area[bc * 66 + le] = d[bc];
      };
      load_x(x, 0);
#pragma unroll
      for (int ar = 0; ar < 4; ++ar) xn[ar] = 0.0;
      if (nqx > 1) load_x(xn, 1);
      mfmas();
      park();
      for (int q = 1; q <= nqx; ++q) {
        // quad q - 1: this lane finishes R[4 (2 bsel + s) + lj][4 (q - 1) + lk], s = 0, 1
        double* dst = Rmine + (ro + 4 * (q - 1));
        double o0 = 0.0, o1 = 0.0;
        if (!first) { o0 = dst[0]; o1 = dst[4 * p.CXp]; }
        double r0[4], r1[4];                                // (the LDS reads would also fit under the MFMAs, but not in the register file)
        if (q < nqx) {
#pragma unroll
          for (int ar = 0; ar < 4; ++ar) x[ar] = xn[ar];
          if (q + 1 < nqx) load_x(xn, q + 1);
          // the parked partials must be out of d before the MFMAs overwrite it: the ds_writes have read their data when they are issued
          mfmas();
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          r0[b] = area[(2 * bsel) * 66 + 16 * lk + 4 * b + lj];
          r1[b] = area[(2 * bsel + 1) * 66 + 16 * lk + 4 * b + lj];
        }
        dst[0] = o0 + ((r0[0] + r0[1]) + (r0[2] + r0[3]));
        dst[4 * p.CXp] = o1 + ((r1[0] + r1[1]) + (r1[2] + r1[3]));
        if (q < nqx) park();
      }
    }
    G8SEC(3)
    // ---- m-contraction: HZ[n][c] = sum_m W[n][m] Zaug[m][c].  W goes through the wave's LDS area 32 rows at a time (the transposed read is
    // the MFMA A operand: lane (row l & 15, k l >> 4)) and stays in 16 registers; B = Zaug rows of the wave's 32 columns, eight feature
    // columns per group, staged in LDS by DMA one group ahead (read from global memory it would be one replicated 512-byte load per MFMA).
    // Each wave contracts its own 32 columns; the four partials of a 32-row half are exchanged through LDS and added in column order, so HZp
    // holds ONE array per 128-column tile.  (v_mfma_f64_16x16x4 with 64 distinct B values per load was tried: half rate, same kernel time.)
#if defined(GPARML_GEN8_ABLATE) && (GPARML_GEN8_ABLATE & 1)
    if (p.MT < 0)
#else
    if (PPATH)
#endif
    {
      int lm = lane;                                        // fresh lane coordinates: carried over from above they are spilled
      asm volatile("" : "+v"(lm));
      const int lr = lm & 15, lk = lm >> 4, lj = lm & 3, le = lm;
      const int srow = 4 * ((lm >> 2) & 3) + (lm >> 4);
      const int ng = (p.CZp + 7) / 8;                       // feature columns in groups of eight (two quads)
      // LDS area of the wave once the slabs are in registers: [0, 512) the Zaug rows of the wave's 32 columns x 8 features, staged by
      // LDS-DMA one group ahead (two buffers) | [512, 1088) exchange buffers [2][32 rows][stride 9: 8 features] (stride 8 puts the four
      // row quads of a 16-lane write group on the same banks: SQ_LDS_BANK_CONFLICT was 70 % of the kernel's active LDS cycles)
      double* const zst = area;
      double* const xbuf = area + 512;
      const double* zsrc = p.Zaug + ((long)mt * TILE + wcol0 + (lm >> 2)) * p.CZp + 2 * (lm & 3);   // DMA i: rows 16 i + (lane >> 2), 16-byte piece lane & 3
      auto stage = [&](int g, double* dst) {
        glds16(zsrc + 8 * g, dst);
        glds16(zsrc + 16 * p.CZp + 8 * g, dst + 128);
      };
      const int wi = 2 * wc + half;
      const int rrow = 8 * wi + (le >> 3), rcol = le & 7;
      int xpar = 0;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        if (pr == 1) __syncthreads();   // the other waves are done with this wave's exchange buffers, which the slab writes overwrite
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int bc = 0; bc < 8; ++bc) area[t * 544 + srow * GSLD + 4 * bc + lj] = acc[2 * pr + t][bc];
        double a[2][8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int k4 = 0; k4 < 8; ++k4) a[t][k4] = area[t * 544 + lr * GSLD + 4 * k4 + lk];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slabs are in registers: their LDS is free for the staged rows
        acc_fence8(a[0]); acc_fence8(a[1]);
        stage(0, zst);
        for (int g = 0; g < ng; ++g) {
          dma_wait();                                       // group g's rows have landed (and the previous group's HZ stores retired)
          if (g + 1 < ng) stage(g + 1, zst + ((g + 1) & 1) * 256);
          const double* zb = zst + (g & 1) * 256 + lk * 8 + lj;
          double h[2][2];
          {
            const double b0 = zb[0], b1 = zb[4];                    // Zaug[wcol0 + lk][8 g + lj], [... + 4]
            mfma444_zero(h[0][0], a[0][0], b0);
            mfma444_zero(h[1][0], a[1][0], b0);
            mfma444_zero(h[0][1], a[0][0], b1);
            mfma444_zero(h[1][1], a[1][0], b1);
          }
#pragma unroll
          for (int k4 = 1; k4 < 8; ++k4) {
            const double b0 = zb[32 * k4], b1 = zb[32 * k4 + 4];   // Zaug[wcol0 + 4 k4 + lk][8 g + lj], [... + 4]
            mfma444_acc(h[0][0], a[0][k4], b0);
            mfma444_acc(h[1][0], a[1][k4], b0);
            mfma444_acc(h[0][1], a[0][k4], b1);
            mfma444_acc(h[1][1], a[1][k4], b1);
          }
          mfma_drain(h[1][1]);
          acc_fence<2>(h[0]); acc_fence<2>(h[1]);
          double* xb = xbuf + xpar * 288;                   // [32 rows][8 features, row stride 9]: this wave's partial over its 32 columns
#pragma unroll
          for (int t = 0; t < 2; ++t) { xb[(16 * t + srow) * 9 + lj] = h[t][0]; xb[(16 * t + srow) * 9 + 4 + lj] = h[t][1]; }
          __syncthreads();
          double s0 = 0.0;
#pragma unroll
          for (int w4 = 0; w4 < 4; ++w4)                    // partials in column order: (wc, half) = (0,0), (0,1), (1,0), (1,1)
            s0 += lds[(2 * wr + (w4 >> 1) + 4 * (w4 & 1)) * 1280 + 512 + xpar * 288 + rrow * 9 + rcol];
          if (8 * g + rcol < p.CZp) p.HZp[((long)mt * p.Np + n0 + wrow0 + 32 * pr + rrow) * p.CZp + 8 * g + rcol] = s0;
          xpar ^= 1;
        }
      }
    }
    G8SEC(4)
    __syncthreads();   // the wave areas live in the staging buffers the next tile's DMA overwrites
    G8SEC(5)
  }
#ifdef GPARML_GEN8_TIMING
  if (p.dbg && lane == 0) for (int k = 0; k < 8; ++k) p.dbg[((long)blockIdx.x * 8 + wave) * 8 + k] = tsec[k];
#endif
}

// R = sum of the (slice, wave-row) partials; then the data parts of grad_Z / grad_alpha
//   fixedA (Xa = [mu, 1]):  gZ = a (R1 - Z R0),  ga = -1/2 sum_m (-2 Z R1 + Z^2 R0) + the fast kernel's mu^2 term   [regime A, fixed embeddings]
//   fixedA = 2 (Xa = [mu, 1, mu^2]): the same with the mu^2 term from the third block, ga = -1/2 sum_m (R2 - 2 Z R1 + Z^2 R0), R2 = W^T(mu o mu)
//   general (Xa = [u mu, u, 1]):  gZ = R1 - Z R2'  with R1 = W^T(u mu), R2' = W^T u; ga comes from the per-point kernel
__global__ void __launch_bounds__(256) p2_reduce_kernel(const double* __restrict__ Rpart, int nparts, int Mp, int CXp, int M, int Q,
                                                        const double* __restrict__ Z, const double* __restrict__ alpha, int fixedA,
                                                        double* __restrict__ gZ, double* __restrict__ gapart) {
  // one block per inducing row m: 256 threads split the partial index, then tree-reduce per column
  const int m = blockIdx.x;
  __shared__ double red[256];
  __shared__ double R[512];
  for (int c0 = 0; c0 < CXp; c0 += 8) {
    // 8 columns x 32 part-lanes per pass
    const int c = c0 + (threadIdx.x & 7), pl = threadIdx.x >> 3;
    double s = 0.0;
    if (c < CXp) {
      // eight loads in flight, added in the order of the plain loop (hipcc compiled that one as load, wait, add: one memory round trip per
      // partial, 24 in a row at configs[1]'s size)
      const double* src = Rpart + (long)m * CXp + c;
      const long step = (long)Mp * CXp;
      int i = pl;
      for (; i + 32 * 7 < nparts; i += 32 * 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = src[(long)(i + 32 * u) * step];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += x[u];
      }
      for (; i < nparts; i += 32) s += src[(long)i * step];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k >= 8; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x < 8 && c0 + threadIdx.x < CXp) R[c0 + threadIdx.x] = red[threadIdx.x];
    __syncthreads();
  }
  for (int q = threadIdx.x; q < Q; q += 256) {
    const double z = Z[(long)m * Q + q];
    if (fixedA) {
      const double r1 = R[q], r0 = R[Q];
      gZ[(long)m * Q + q] = alpha[q] * (r1 - z * r0);
      gapart[(long)m * Q + q] = -0.5 * ((fixedA == 2 ? R[Q + 1 + q] : 0.0) + z * z * r0 - 2.0 * z * r1);
    } else {
      gZ[(long)m * Q + q] = R[q] - z * R[Q + q];
      gapart[(long)m * Q + q] = 0.0;
    }
  }
}

// per-point finish (general mode): HZ = sum of partials; grad_X_mu, grad_X_S, and the per-point part of grad_alpha
struct PtArgs {
  const double* HZp; int nparts; long N, Np; int Q, CZp; const double* mu; const double* S; const double* alpha;
  double* gmu; double* gS; double* gapart; int regimeA, pb;
};
// points per pass (their HZ rows are contiguous in every partial array, so the part sums are flat coalesced reads); LDS: pb (CZp + Q) + Q doubles
static int point_pb(int CZp, int Q) { return std::max(1, std::min(32, (7000 - Q) / (CZp + Q))); }
__global__ void __launch_bounds__(256) point_kernel(PtArgs a) {
  extern __shared__ double sm[];      // [pb][CZp] summed HZ rows | [pb][Q] grad_alpha contributions | [Q] this block's column sums (points in order)
  double* hz = sm;
  double* contrib = sm + a.pb * a.CZp;
  double* gacc = contrib + a.pb * a.Q;
  const int t = threadIdx.x;
  for (int q = t; q < a.Q; q += 256) gacc[q] = 0.0;   // each column is only ever touched by the same thread
  const long nchunks = (a.N + a.pb - 1) / a.pb;
  for (long ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const long n0 = ch * a.pb;
    const int np = (int)min((long)a.pb, a.N - n0);
    for (int e = t; e < np * a.CZp; e += 256) {
      double s = 0.0;
      for (int i = 0; i < a.nparts; ++i) s += a.HZp[((long)i * a.Np + n0) * a.CZp + e];
      hz[e] = s;
    }
    __syncthreads();
    for (int e = t; e < np * a.Q; e += 256) {
      const int pi = e / a.Q, q = e - pi * a.Q;
      const double* row = hz + pi * a.CZp;
      const double h = row[0], hzq = row[1 + q], hz2 = row[1 + a.Q + q];
      const double m = a.mu[n0 * a.Q + e], s = a.S[n0 * a.Q + e], al = a.alpha[q];
      const double d1 = al * s + 1.0, u = al / d1;
      const double quad = m * m * h - 2.0 * m * hzq + hz2;
      contrib[e] = -0.5 * (quad / (d1 * d1) + (s / d1) * h);
      a.gmu[n0 * a.Q + e] = -m - u * (m * h - hzq);
      // (fixed variances S = 0: the reference's expression divides by S, partial_terms.py:400-431; the library defines the entry as 0 instead of leaving the buffer as it was)
      a.gS[n0 * a.Q + e] = a.regimeA ? 0.0 : -0.5 * (1.0 - 1.0 / s) + 0.5 * u * u * quad - 0.5 * u * h;
    }
    __syncthreads();
    for (int q = t; q < a.Q; q += 256) {
      double ga = gacc[q];
      for (int pi = 0; pi < np; ++pi) ga += contrib[pi * a.Q + q];
      gacc[q] = ga;
    }
  }
  for (int q = t; q < a.Q; q += 256) a.gapart[(long)blockIdx.x * a.Q + q] = gacc[q];
}

__global__ void colsum2_kernel(const double* __restrict__ a, int rows_a, int lda, const double* __restrict__ b, int rows_b, int ldb, int Q,
                               double* __restrict__ out) {
  const int q = blockIdx.x;
  __shared__ double red[256];
  double s = 0.0;
  // eight loads in flight, added in the order of the plain loop (which compiles to one memory round trip per element)
  auto strided = [&](const double* __restrict__ x, int rows, int ld) {
    int r = threadIdx.x;
    for (; r + 256 * 7 < rows; r += 256 * 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[(long)(r + 256 * u) * ld + q];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < rows; r += 256) s += x[(long)r * ld + q];
  };
  strided(a, rows_a, lda);
  if (b) strided(b, rows_b, ldb);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) out[q] = red[0];
}

// fixed-embedding fast path: regime A without embedding gradients and Q + 1 <= 12 feature columns (p2_fast8_kernel)
bool p2_fast_mode(const gp_ctx* c) { return c->regime_A && !c->want_emb && c->Q + 1 <= 12; }
// fixed embeddings with a wider latent space: the general eight-wave kernel WITHOUT its per-point m-contraction, on the hyper-parameter independent
// features [mu | 1 | mu^2] (the mu^2 term of grad_alpha comes out of the same n-contraction; no point_kernel, the prep kernels run once per upload).
// N = 1e6, D = 100, M = 512, Q = 30: phase-2 kernel 13.7 -> 12.0 ms, evaluation 23.6 -> 20.4 ms (same box, with the fixed-variance Psi1 kernel; profiles/r04_shape_sweep.txt)
bool p2_wide_fixed_mode(const gp_ctx* c) { return c->regime_A && !c->want_emb && c->Q + 1 > 12; }

int run_phase2(gp_ctx* c) {
  const bool fast = p2_fast_mode(c), widefix = p2_wide_fixed_mode(c);
  const bool ppath = !fast && !widefix;
  P2Args p;
  p.Kaug = c->Kaug; p.ld = c->LDK; p.Bm = c->Bm; p.Xa = c->Xa; p.Zaug = c->Zaug; p.Rpart = c->Rpart; p.HZp = c->HZp;
  p.Mp = c->Mp; p.CXp = c->CXp; p.CZp = c->CZp; p.MT = c->Mp / TILE; p.Np = c->Np;
  p.ntiles = (int)(c->Np / TILE);
  int S = std::max(1, std::min(c->p2_slices, p.ntiles));
  p.tps = (p.ntiles + S - 1) / S;
  S = (p.ntiles + p.tps - 1) / p.tps;
  p.S = S;
  p.kbeg = c->regime_A ? 0 : c->Mp / KC;
  p.kend = (c->Mp + (int)round_up(c->D, KC)) / KC;   // chunks beyond the last real Y column are all zero
  p.klast = ((c->D - 1) % KC) / 4 + 1;
  const int blocks = 8 * ((S + 7) / 8) * p.MT;
  const int nrb = (c->Q + 1 + 3) / 4;                // fast path: feature columns [mu (Q) | 1] in groups of four
  p.prog = nullptr; p.prog_base = 0;
  static const bool p2_sync = [] { const char* e = getenv("GPARML_P2_SYNC"); return !(e && e[0] == '0'); }();
  // the in-step wait needs every workgroup of the launch resident at once: the grid is sized for two workgroups per CU on 256 CUs
  if (fast && p2_sync && p.MT > 1 && blocks <= 512) {
    if (!c->p2prog) GP_TRY_RC(dalloc_bytes(c, (void**)&c->p2prog, (size_t)(c->p2_slices + 8) * p.MT * sizeof(unsigned long long), DA_ZERO));   // zero contract: bases only grow
    p.prog = c->p2prog; p.prog_base = (unsigned long long)(++c->p2_epoch) << 32;
  }
  GP_EV(c, 12);
  p.dbg = nullptr;
#ifdef GPARML_GEN8_TIMING
  static long long* g8dbg = nullptr;
  if (!g8dbg) GP_HIP(c, hipMalloc((void**)&g8dbg, (size_t)65536 * 64 * sizeof(long long)));
  p.dbg = g8dbg;
#endif
  if (ppath) hipLaunchKernelGGL((p2_gen8_kernel<true>), dim3(blocks), dim3(512), 0, c->stream, p);
  if (widefix) hipLaunchKernelGGL((p2_gen8_kernel<false>), dim3(blocks), dim3(512), 0, c->stream, p);
#ifdef GPARML_GEN8_TIMING
  if (ppath || widefix) {
    std::vector<long long> h((size_t)blocks * 64);
    GP_HIP(c, hipMemcpy(h.data(), g8dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    const char* names[8] = {"first chunk", "k-loop", "Psi1 product", "n-contraction", "m-contraction", "tile barrier", "-", "tile setup"};
    for (int blk : {0, blocks / 2}) for (int wv : {0, 5}) {
      fprintf(stderr, "[gen8 timing] block %d wave %d, %d tiles (s_memtime ticks, 100 MHz):", blk, wv, p.tps);
      for (int k = 0; k < 8; ++k) fprintf(stderr, " %s=%lld", names[k], h[((size_t)blk * 8 + wv) * 8 + k]);
      fprintf(stderr, "\n");
    }
  }
#endif
  if (fast) {                                        // nrb <= 3
    p.gapart = c->hgpart;
    // (blocks past the last slice zero their own rows of hgpart)
    switch (nrb) {
      case 1: hipLaunchKernelGGL((p2_fast8_kernel<1>), dim3(blocks), dim3(512), 0, c->stream, p); break;
      case 2: hipLaunchKernelGGL((p2_fast8_kernel<2>), dim3(blocks), dim3(512), 0, c->stream, p); break;
      default: hipLaunchKernelGGL((p2_fast8_kernel<3>), dim3(blocks), dim3(512), 0, c->stream, p); break;
    }
  }
  GP_EV(c, 13);
  GP_HIP(c, hipGetLastError());
  double* gZ = c->grads;
  double* ga = c->grads + (long)c->M * c->Q;
  // T2 is free after the global step: per-row alpha partials [M][Q]
  hipLaunchKernelGGL(p2_reduce_kernel, dim3(c->M), dim3(256), 0, c->stream, c->Rpart, 2 * S, c->Mp, fast ? 4 * nrb : c->CXp, c->M, c->Q, c->Z,
                     c->alpha, fast ? 1 : (widefix ? 2 : 0), gZ, c->T2);
  GP_HIP(c, hipGetLastError());
  if (ppath) {
    PtArgs a;
    a.HZp = c->HZp; a.nparts = p.MT;            // p2_gen8_kernel: one partial array per 128 inducing columns
    a.N = c->N; a.Np = c->Np; a.Q = c->Q; a.CZp = c->CZp; a.mu = c->mu; a.S = c->S; a.alpha = c->alpha;
    a.gmu = c->gXmu; a.gS = c->gXs; a.gapart = c->gapart; a.regimeA = c->regime_A ? 1 : 0;
    a.pb = point_pb(c->CZp, c->Q);
    hipLaunchKernelGGL(point_kernel, dim3(c->ga_blocks), dim3(256), (size_t)(a.pb * (c->CZp + c->Q) + c->Q) * sizeof(double), c->stream, a);
    GP_HIP(c, hipGetLastError());
    hipLaunchKernelGGL(colsum2_kernel, dim3(c->Q), dim3(256), 0, c->stream, c->T2, c->M, c->Q, c->gapart, c->ga_blocks, c->Q, c->Q, ga);
  } else if (widefix) {
    hipLaunchKernelGGL(colsum2_kernel, dim3(c->Q), dim3(256), 0, c->stream, c->T2, c->M, c->Q, (const double*)nullptr, 0, 0, c->Q, ga);
  } else {
    const int hb = blocks * 8, hstride = 4 * nrb;    // one partial row of grad_alpha's mu^2 term per wave
    hipLaunchKernelGGL(colsum2_kernel, dim3(c->Q), dim3(256), 0, c->stream, c->T2, c->M, c->Q, c->hgpart, hb, hstride, c->Q, ga);
  }
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

}  // namespace gp
