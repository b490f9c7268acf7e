// Regime B (Bayesian GPLVM), phase 2 on TILE PAIRS of the symmetric per-point matrix T_n = Bbar o psi2_n  (Q <= 63).
//   reference: partial_terms.py:190-205, 273-284 (psi2 parts of grad_Z / grad_alpha), 388-394, 421-427 (grad_X_mu / grad_X_S),
//   kernel_exp.py:126-148 (psi2_n); formulation: SURVEY.md section 7 / oracle/factorised.py phase2.
//
// A workgroup owns ONE 64 x 64 tile (I <= J) of T_n and streams the points of its slice through it (the order of psi2_pairs_mfma_kernel):
// Z_I, Z_J (LDS) and the tile of Bbar (registers) are loaded once, per point only LEA[n, I], LEA[n, J] and three Q-vectors arrive.
//   GEMM1   E[m, m'] = sum_q (v2_nq z_mq) z_m'q            4x4x4 FP64 MFMAs, the per-point scale v2 = -2 V_n applied to the A operand
//   T       = Bbar o exp(E + LEA[n, m] + LEA[n, m'])       on the accumulator registers
//   column side   t[m', q] = sum_{m in I} T[m, m'] Z1[m, q]     (the T registers ARE the transposed A operand, psi2.hip)
//   row side      t[m, q]  = sum_{m' in J} T[m, m'] Z1[m', q]   (I < J only: T_n is symmetric, so the tile also serves slab I)
// Z1 = [Z | 0 | 1]: the column of ones (index QT - 1, a compile-time position) yields r = T 1.  For the row side the tile goes through
// LDS once (64 x 64 doubles) and every wave takes 16 ROWS across all 64 columns, so its sums are complete: no cross-wave partials.
// Everything a point needs from t is linear in it, so the tile's contributions are folded at once:
//   grad_Z[m', q] += kappa_nq t + r (2 w_nq mu_nq - z_m'q (alpha_q + w_nq))      kappa = alpha - w = 2 v2           (registers, all points)
//   per point: s0 = sum r, s1_q = sum z r, s2_q = sum z^2 r, s3_q = sum z t  -> pp[tile][n]  (s3 of the two sides is equal by symmetry)
// Register layout after the column-side MFMAs is four row-quad partials per column (the instruction's four blocks); they are
// summed with a reduce-scatter over the four lanes (each lane ends up owning ONE column and every fourth q), and the per-point
// sums are reduce-scattered over sixteen lanes the same way (lane q ends up holding s_q): ~1 add per value instead of log2(lanes).
// Workgroup = 8 waves = two point streams (even / odd points of the slice) of four waves; LDS: Z_I, Z_J, two T tiles, tables.
#include "gp_common.h"
#include "fexp.h"
#include "quad_mma.h"
#include "lane_reduce.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace gp {

struct PT2Args {
  const double* ZP; const double* Bbar; const double* LEA; const double* V2P; const double* WP; const double* MUP; const double* alphaP;
  const int* tiles;     // [T][2] (I, J), I <= J, 64-row slabs
  double* Gt;           // [S][T][2][64][Q] grad_Z partials per workgroup: side 0 = rows of slab J (column side), 1 = rows of slab I (row side)
  double* pp;           // [T][3Q+1][CH] per-point sums of every tile for the points of this launch
  long n0, n1, CH, Np;
  int Mp, M, Q, QB, T, S, accumulate;
  long long* dbg;       // timing build (GPARML_TILE_TIMING): [blocks][8 waves][8 sections] cycle totals
};

__device__ __forceinline__ double mul_asm(double a, double b) {   // ordered with the asm MFMAs around it (the compiler may not move it)
  double r;
  asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}


// LDS image of the current point's tile for the two contractions: element (row, col) of T at tx[col * TXS + (row ^ 16 (col & 1))].
// With TXS = 66 the column side's operand reads (16 columns x rows 4 s + {0, 1} per 32-lane group) are conflict-free and the row side's
// (columns 4 s + {0, 1} x 16 rows) collide in 2 of 32 lanes.
constexpr int TXS = 66;
#ifndef GPARML_TILE_FOLD_DEPTH
#define GPARML_TILE_FOLD_DEPTH 1
#endif

template <int QT>
__global__ void __launch_bounds__(512, 2) psi2_tile_kernel(PT2Args a) {
  constexpr int NQ = QT / 4, LDZ = QT + 2, RW = 3 * QT, NG = (NQ + 3) / 4;
  extern __shared__ double smem[];
  double* zr = smem;                          // [64][LDZ]  Z1 of slab I (rows of the tile)
  double* zj = zr + 64 * LDZ;                 // [64][LDZ]  Z1 of slab J (columns)
  double* txb = zj + 64 * LDZ;                // [2 streams][64 * TXS]  the current point's tile
  double* qtb = txb + 2 * 64 * TXS;           // [2 streams][2 parities][3][QT]  kappa | alpha + w | 2 w mu
  double* redb = qtb + 2 * 2 * 3 * QT;        // [2 streams][4 waves][RW]  per-wave s1 | s2 | s3
  double* rrb = redb + 2 * 4 * RW;            // [8 waves][2 sides][16]  r of the wave's 16 columns / 16 rows
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, w = wave & 3, ts = tid & 255;
  const int li = lane & 3, lb = (lane >> 2) & 3, lk = lane >> 4, lr = lane & 15;
  const int I = a.tiles[2 * blockIdx.x], J = a.tiles[2 * blockIdx.x + 1];
  const bool offd = I != J;
  const long per = (a.n1 - a.n0 + a.S - 1) / a.S;
  const long na = a.n0 + (long)blockIdx.y * per, nb = min(a.n1, na + per);
  const int PW = 3 * a.Q + 1;
#ifdef GPARML_TILE_TIMING
  long long tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define TSEC(k) { const long long tn_ = __builtin_amdgcn_s_memtime(); tsec[k] += tn_ - tlast; tlast = tn_; }
#else
#define TSEC(k)
#endif
  for (int e = tid; e < 64 * QT; e += 512) {
    const int r = e / QT, q = e - r * QT;
    const double one = (q == QT - 1) ? 1.0 : 0.0;
    zr[r * LDZ + q] = q < a.Q ? a.ZP[(64L * I + r) * a.QB + q] : one;
    zj[r * LDZ + q] = q < a.Q ? a.ZP[(64L * J + r) * a.QB + q] : one;
  }
  // Bbar[row 64 I + 16 rb + 4 lb + lk][col 64 J + 16 w + 4 cq + li] in the layout of the E / T registers: re-read (from the vector L1 / L2: the
  // tile is 32 KB per workgroup) at the start of every phase A instead of being held in 32 VGPRs through phase B
  const double* bbp = a.Bbar + (64L * I + 4 * lb + lk) * a.Mp + 64 * J + 16 * w + li;
  double Gc[NQ], Gr[NQ];   // grad_Z of (column 64 J + 16 w + 4 lb + lk | row 64 I + 16 w + 4 lb + lk), q = 4 qq + li, over this stream's points
#pragma unroll
  for (int qq = 0; qq < NQ; ++qq) { Gc[qq] = 0.0; Gr[qq] = 0.0; }
  double* tx = txb + g * (64 * TXS);
  double* red = redb + (g * 4 + w) * RW;
  double* rr = rrb + wave * 32;
  const int pofs = 16 * (li & 1), pk = 16 * (lk & 1);
  const unsigned aA = lds_byte_addr(zr) + 8u * (unsigned)(lr * LDZ + lk);                       // GEMM1 A: Z1_I[16 rb + lr][4 k4 + lk]
  const unsigned aB = lds_byte_addr(zj) + 8u * (unsigned)((16 * w + li) * LDZ + lk);            // GEMM1 B: Z1_J[16 w + 4 cq + li][4 k4 + lk]
  // column side A: T[4 s + lk][16 w + 4 lb + li] (two bases: the row's bit 4 is flipped for odd columns); B: Z1_I[4 s + lk][4 qq + li]
  const int cb = (16 * w + 4 * lb + li) * TXS + lk;
  const unsigned aC0 = lds_byte_addr(tx) + 8u * (unsigned)(cb + pofs), aC1 = lds_byte_addr(tx) + 8u * (unsigned)(cb - pofs);
  const unsigned aZc = lds_byte_addr(zr) + 8u * (unsigned)(lk * LDZ + li);
  // row side A: T[16 w + 4 lb + li][4 s + lk]; B: Z1_J[4 s + lk][4 qq + li]
  const unsigned aR = lds_byte_addr(tx) + 8u * (unsigned)(lk * TXS + ((16 * w + 4 * lb + li) ^ pk));
  const unsigned aZr = lds_byte_addr(zj) + 8u * (unsigned)(lk * LDZ + li);
  // tile store: reg (rb, cq) is T[16 rb + 4 lb + lk][16 w + 4 cq + li]
  const int wb = (16 * w + li) * TXS + 4 * lb + lk;

  // one side's contraction t[x][q] = sum over the 64 rows (COL) / columns (!COL) of the tile, result (x = 16 w + 4 lb + lk, q = 4 qq + li).
  // Sixteen steps of NQ MFMAs (one A operand, NQ B operands).  Operands are requested RS steps ahead and every B register is re-requested
  // right behind the MFMA that consumed it (the LDS return is at least an order of magnitude later than the MFMA's operand read), so
  // RS * NQ + RS + 1 operand registers cover >= 128 cycles of MFMA issue per LDS round trip for every QT; RS is as deep as the 4-bit
  // lgkmcnt allows.  Issue order: ..., M(s,0), A(s+RS), B(s+RS,0), M(s,1), B(s+RS,1), ...
  auto contract = [&](auto colc, double (&tq)[NQ]) {
    constexpr bool COL = decltype(colc)::value != 0;
    constexpr int RS = 1 + (15 - NQ) / (NQ + 1);
    double ta[RS + 1], bz[RS][NQ];
    auto rdA = [&](auto sc) {
      constexpr int s = decltype(sc)::value, slot = s % (RS + 1);
      if constexpr (COL) ta[slot] = ((s >> 2) & 1) ? ds_read64<(4 * s) * 8>(aC1) : ds_read64<(4 * s) * 8>(aC0);
      else ta[slot] = ds_read64<(4 * s * TXS) * 8>(aR);
    };
    auto rdB = [&](auto sc, auto jc) {
      constexpr int s = decltype(sc)::value, j = decltype(jc)::value;
      bz[s % RS][j] = ds_read64<(4 * s * LDZ + 4 * j) * 8>(COL ? aZc : aZr);
    };
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    static_for<0, RS>([&](auto sc) { rdA(sc); static_for<0, NQ>([&](auto jc) { rdB(sc, jc); }); });
    static_for<0, 16>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      static_for<0, NQ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        // reads requested behind B(s, j): the rest of step s, the steps s + 1 .. s + RS - 1, and what M(s, 0 .. j - 1) re-requested
        constexpr int last = (s + RS - 1) < 15 ? (s + RS - 1) : 15;
        constexpr int newer = (NQ - 1 - j) + (last - s) * (NQ + 1) + ((s + RS <= 15 && j >= 1) ? j + 1 : 0);
        // lgkmcnt is a 4-bit counter: with more than fifteen newer requests (QT = 64: sixteen) the wave cannot have issued them all while the
        // operand of this MFMA was still outstanding, so "at most fifteen outstanding" implies it has landed
#ifndef GPARML_TILE_ABLATE_READS
        lgkm_wait<(newer < 15 ? newer : 15)>();
#else
        if constexpr (s == 0 && j == 0) lgkm_wait<0>();
#endif
        if constexpr (s == 0) mfma444_zero(tq[j], ta[s % (RS + 1)], bz[s % RS][j]);
        else mfma444_acc(tq[j], ta[s % (RS + 1)], bz[s % RS][j]);
#ifndef GPARML_TILE_ABLATE_READS      /* ablation build: stale operands, no LDS reads in the contractions' main loop */
        if constexpr (s + RS <= 15) {
          if constexpr (j == 0) rdA(IC<s + RS>{});
          rdB(IC<s + RS>{}, jc);
        }
#endif
      });
      if constexpr (NQ < 3) asm volatile("s_nop 15");                     // dependent accumulation: the asm MFMAs get no automatic wait states
    });
    mfma_drain(tq[NQ - 1]);
    acc_fence<NQ>(tq);
    __builtin_amdgcn_sched_barrier(0);
  };
  // fold one side's t into its grad_Z accumulator; returns r (broadcast inside the lane quad) and leaves p[qq] = z t if wanted.  The four
  // table reads per q-quad are asm reads one quad ahead (left to the compiler all 4 NQ loads are hoisted: 100 VGPRs in flight)
  auto fold = [&](const double (&tq)[NQ], double (&G)[NQ], unsigned aZo, unsigned aQt, double* p) -> double {
    const double r = quad_xchg<0xFF>(tq[NQ - 1]);                         // the ones column sits at q = QT - 1 (qq = NQ - 1, li = 3)
    constexpr int FD = GPARML_TILE_FOLD_DEPTH;            // q-quads requested ahead (an LDS round trip is ~4 quads of fold arithmetic)
    double tb[FD + 1][4];
    auto rdF = [&](auto qc) {
      constexpr int qq = decltype(qc)::value, sl = qq % (FD + 1);
      tb[sl][0] = ds_read64<(4 * qq) * 8>(aZo); tb[sl][1] = ds_read64<(4 * qq) * 8>(aQt);
      tb[sl][2] = ds_read64<(QT + 4 * qq) * 8>(aQt); tb[sl][3] = ds_read64<(2 * QT + 4 * qq) * 8>(aQt);
    };
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, (FD < NQ ? FD : NQ)>([&](auto qc) { rdF(qc); });
    static_for<0, NQ>([&](auto qc) {
      constexpr int qq = decltype(qc)::value, cur = qq % (FD + 1);
      if constexpr (qq + FD < NQ) rdF(IC<qq + FD>{});
      constexpr int ahead = (NQ - 1 - qq) < FD ? (NQ - 1 - qq) : FD;
      lgkm_wait<4 * ahead>();
      __builtin_amdgcn_sched_barrier(0);
      const double z = tb[cur][0], kap = tb[cur][1], c1 = tb[cur][2], c2 = tb[cur][3];
      G[qq] = fma(tq[qq], kap, fma(r, fma(-z, c1, c2), G[qq]));
      if (p) p[qq] = z * tq[qq];
      __builtin_amdgcn_sched_barrier(0);
    });
    return r;
  };
  // ---- the next point's Q-vectors travel while the current point computes (LEA is requested at the start of phase A: GEMM1 covers it)
  double nq0 = 0.0, nq1 = 0.0, nq2 = 0.0;
  auto load_point = [&](long n) {
    if (ts < a.Q) {
      const double wq = a.WP[n * a.QB + ts];
      nq0 = 2.0 * a.V2P[n * a.QB + ts]; nq1 = a.alphaP[ts] + wq; nq2 = 2.0 * wq * a.MUP[n * a.QB + ts];     // kappa = alpha - w = 2 (-2 V)
    }
  };
  // Schedule: the two streams run half a point apart.  Stream g is in phase A (GEMM1, exp, tile store) of its point i at half-step
  // 2 i + g and in phase B (both contractions, folds, sums) at 2 i + g + 1; every half-step starts with one workgroup barrier, so a
  // SIMD always hosts one wave in each phase.
  const long cnt_g = (nb > na) ? (nb - na + 1 - g) / 2 : 0;                 // points of this stream: na + g, na + g + 2, ...
  const int H = (nb > na) ? (int)(nb - na) + 2 : 0;                         // half-steps 0 .. H - 1 cover both streams' last phase B and its flush
  if (cnt_g > 0) load_point(na + g);
  long n_flush = -1;
  __syncthreads();
  for (int h = 0; h < H; ++h) {
    const int hh = h - g;
    const bool isA = (hh & 1) == 0;
    const long i = hh >> 1;
    const bool act = hh >= 0 && i < cnt_g;
    double* qt = qtb + (g * 2 + (int)(i & 1)) * (3 * QT);
    const long n_cur = na + g + 2 * i;
    if (isA && act) {
      if (ts < QT) { qt[ts] = ts < a.Q ? nq0 : 0.0; qt[QT + ts] = ts < a.Q ? nq1 : 0.0; qt[2 * QT + ts] = ts < a.Q ? nq2 : 0.0; }
      if (i + 1 < cnt_g) load_point(n_cur + 2);
    }
    TSEC(1)
    __syncthreads();
    TSEC(0)
    if (n_flush >= 0 && ts < PW) {
      // the sums of this stream's previous phase B: four waves -> pp[tile][i][n]   (i = 0: s0 = the ones column of s1)
      const double* rp = redb + g * 4 * RW;
      const int src = ts == 0 ? QT - 1 : ((ts - 1) / a.Q) * QT + (ts - 1) % a.Q;
      a.pp[((long)blockIdx.x * PW + ts) * a.CH + (n_flush - a.n0)] = (rp[src] + rp[RW + src]) + (rp[2 * RW + src] + rp[3 * RW + src]);
    }
    n_flush = -1;
    if (!act) continue;
#if defined(GPARML_TILE_ABLATE) && GPARML_TILE_ABLATE == 3          /* ablation: no phase A at all (stale tile) */
    if (isA) continue;
#endif
#if defined(GPARML_TILE_ABLATE) && GPARML_TILE_ABLATE == 4          /* ablation: no phase B at all */
    if (!isA) continue;
#endif
    // the waves in phase B issue first: its contractions then run at the MFMA rate and its latency-bound tail (folds, reduce-scatter, sums)
    // starts earlier, under the other stream's GEMM1 (same-box A/B, profiles/r03_tile_kernel_ab.txt: -2 % at Q = 50, -6 % at Q = 10; phase A first: +2 % / +6 %)
    if (isA) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2);
    if (isA) {
      // ---- GEMM1 (operands of step k4 + 1 are read while the 16 MFMAs of step k4 execute; counted waits: only asm LDS reads in here)
      double T[4][4];
      // the 24 per-lane operands of the exp stage are requested before GEMM1, which covers their latency (L2: ~1.5k cycles)
      double lrow[4], lcol[4], bb[4][4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) lrow[rb] = a.LEA[n_cur * a.Mp + 64 * I + 16 * rb + 4 * lb + lk];
#pragma unroll
      for (int cq = 0; cq < 4; ++cq) lcol[cq] = a.LEA[n_cur * a.Mp + 64 * J + 16 * w + 4 * cq + li];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) bb[rb][cq] = bbp[(long)(16 * rb) * a.Mp + 4 * cq];
      {
        const unsigned aV = lds_byte_addr(qt) + 8u * (unsigned)lk;
        double av[2][4], bv[2][4], vv[2];
        auto rd = [&](auto kc, double (&a_)[4], double (&b_)[4], double& v_) {
          constexpr int k4 = decltype(kc)::value;
          static_for<0, 4>([&](auto rc) { constexpr int rb = decltype(rc)::value; a_[rb] = ds_read64<(16 * rb * LDZ + 4 * k4) * 8>(aA); });
          v_ = ds_read64<4 * k4 * 8>(aV);
          static_for<0, 4>([&](auto cc) { constexpr int cq = decltype(cc)::value; b_[cq] = ds_read64<(4 * cq * LDZ + 4 * k4) * 8>(aB); });
        };
        __builtin_amdgcn_sched_barrier(0);
        rd(IC<0>{}, av[0], bv[0], vv[0]);
        static_for<0, NQ>([&](auto kc) {
          constexpr int k4 = decltype(kc)::value, cur = k4 & 1;
          if constexpr (k4 + 1 < NQ) { rd(IC<k4 + 1>{}, av[cur ^ 1], bv[cur ^ 1], vv[cur ^ 1]); lgkm_wait<9>(); }
          else lgkm_wait<0>();
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) av[cur][rb] = mul_asm(av[cur][rb], vv[cur]);
          asm volatile("s_nop 1");
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) {
              if constexpr (k4 == 0) mfma444_zero(T[rb][cq], av[cur][rb], bv[cur][cq]);
              else mfma444_acc(T[rb][cq], av[cur][rb], bv[cur][cq]);
            }
        });
        mfma_drain(T[3][3]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc_fence<4>(T[rb]);
        __builtin_amdgcn_sched_barrier(0);
      }
      TSEC(2)
      // ---- T = Bbar o exp(E + LEA[n, row] + LEA[n, col]),  E = 1/2 sum_q kappa_q z_mq z_m'q;  tile -> LDS
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
#if defined(GPARML_TILE_ABLATE) && GPARML_TILE_ABLATE == 1      /* ablation: no exponential */
          const double t = bb[rb][cq] * (fma(0.5, T[rb][cq], lrow[rb]) + lcol[cq]);
#else
          const double t = bb[rb][cq] * fexp(fma(0.5, T[rb][cq], lrow[rb]) + lcol[cq]);
#endif
          tx[wb + cq * (4 * TXS) + 16 * rb + ((rb & 1) ? -pofs : pofs)] = t;
        }
        __builtin_amdgcn_sched_barrier(0);       // four exponentials at a time: all sixteen interleaved cost ~100 VGPRs of temporaries
      }
      TSEC(3)
    } else {
      // ---- column side: t[col][q] over the 64 rows of the tile -> grad_Z of slab J's rows, s3 (z^T T z of the mirrored tile is the same number)
      double tq[NQ], p[NQ];
      contract(IC<1>{}, tq);
#if defined(GPARML_TILE_ABLATE) && GPARML_TILE_ABLATE == 2        /* ablation: contractions only (no folds, no sums) */
      if (offd) contract(IC<0>{}, p);
      Gc[0] += tq[0] + p[1];
      n_flush = n_cur;
      continue;
#endif
      TSEC(4)
      const unsigned aQt = lds_byte_addr(qt) + 8u * (unsigned)li;
      const double rc = fold(tq, Gc, lds_byte_addr(zj) + 8u * (unsigned)((16 * w + 4 * lb + lk) * LDZ + li), aQt, p);
      double S3 = reduce16<NQ>(p, lane);
      if (offd) S3 *= 2.0;
      // pin the results HERE (asm statements keep their order): otherwise the whole fold is sunk below the row side's contraction -- its
      // basic block -- and 4 NQ table values + NQ sums stay live across it (the register peak of the kernel)
      acc_fence<NQ>(Gc);
      asm volatile("" : "+v"(S3));
      if (li == 3) rr[4 * lb + lk] = rc;
      TSEC(5)
      if (offd) {
        // ---- row side: t[row][q] over the 64 columns -> grad_Z of slab I's rows
        contract(IC<0>{}, tq);
        const double rw_ = fold(tq, Gr, lds_byte_addr(zr) + 8u * (unsigned)((16 * w + 4 * lb + lk) * LDZ + li), aQt, nullptr);
        acc_fence<NQ>(Gr);
        if (li == 3) rr[16 + 4 * lb + lk] = rw_;
      }
      TSEC(6)
      // ---- s1_q = sum z r, s2_q = sum z^2 r over the wave's 16 columns (and 16 rows): lane q, r broadcast from LDS
      double S1 = 0.0, S2 = 0.0;
      {
        // every lane runs the loop (lanes >= QT read inside the arrays and drop the result): asm reads four entries ahead
        const int lq = lane < QT ? lane : 0;
        const unsigned aRr = lds_byte_addr(rr);
        auto sums = [&](unsigned aZs, auto sidec) {
          constexpr int side = decltype(sidec)::value;
          double zb[4], rb_[4];
          auto rdS = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            zb[c & 3] = ds_read64<(c * LDZ) * 8>(aZs); rb_[c & 3] = ds_read64<(16 * side + c) * 8>(aRr);
          };
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the wave's own r stores have landed
          static_for<0, 3>([&](auto cc) { rdS(cc); });
          static_for<0, 16>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (c + 3 < 16) { rdS(IC<c + 3>{}); lgkm_wait<6>(); }
            else lgkm_wait<2 * (15 - c)>();
            __builtin_amdgcn_sched_barrier(0);
            const double zr_ = zb[c & 3] * rb_[c & 3];
            S1 += zr_; S2 = fma(zb[c & 3], zr_, S2);
            __builtin_amdgcn_sched_barrier(0);
          });
        };
        sums(lds_byte_addr(zj) + 8u * (unsigned)(16 * w * LDZ + lq), IC<0>{});
        if (offd) sums(lds_byte_addr(zr) + 8u * (unsigned)(16 * w * LDZ + lq), IC<1>{});
        if (lane < QT) { red[lane] = S1; red[QT + lane] = S2; red[2 * QT + lane] = S3; }
      }
      n_flush = n_cur;
      TSEC(7)
    }
  }
#ifdef GPARML_TILE_TIMING
  if (a.dbg && lane == 0) for (int k = 0; k < 8; ++k) a.dbg[((long)(blockIdx.y * a.T + blockIdx.x) * 8 + wave) * 8 + k] = tsec[k];
#endif
  __syncthreads();
  // ---- grad_Z partials of the workgroup: the two streams are added through LDS (the tile buffers are free now)
  double* gs = smem;        // [2 streams][2 sides][64][QT]
#pragma unroll
  for (int qq = 0; qq < NQ; ++qq) {
    gs[((g * 2 + 0) * 64 + 16 * w + 4 * lb + lk) * QT + 4 * qq + li] = Gc[qq];
    gs[((g * 2 + 1) * 64 + 16 * w + 4 * lb + lk) * QT + 4 * qq + li] = Gr[qq];
  }
  __syncthreads();
  double* dst = a.Gt + ((long)blockIdx.y * a.T + blockIdx.x) * (2L * 64 * a.Q);
  for (int e = tid; e < 2 * 64 * a.Q; e += 512) {
    const int side = e / (64 * a.Q), rq = e - side * 64 * a.Q, r = rq / a.Q, q = rq - r * a.Q;
    const double v = gs[((0 * 2 + side) * 64 + r) * QT + q] + gs[((1 * 2 + side) * 64 + r) * QT + q];
    dst[e] = (a.accumulate ? dst[e] : 0.0) + v;
  }
}

// pp[0][i][n] += sum over the other tiles (in place: the first tile's slot becomes the total)
__global__ void __launch_bounds__(256) pt2_sum_tiles_kernel(double* __restrict__ pp, int T, int PW, long CH, long cnt) {
  const long total = (long)PW * cnt;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const long i = e / cnt, k = e - i * cnt;
    double s = 0.0;
    for (int t = 0; t < T; ++t) s += pp[((long)t * PW + i) * CH + k];
    pp[i * CH + k] = s;
  }
}

// per-point finish from the summed sums [sr | zr_q | z2r_q | zt_q] (psi2.hip, psi2_points_finish_kernel)
struct PT2Fin {
  const double* pp; const double* Wn; const double* mu; const double* S; const double* alpha;
  double* gmu; double* gS; double* gapart2; long n0, n1, CH; int Q, accumulate;
};
__global__ void __launch_bounds__(256) pt2_points_finish_kernel(PT2Fin a) {
  __shared__ double redq[256];
  for (int q = 0; q < a.Q; ++q) {
    double ga = 0.0;
    for (long n = a.n0 + blockIdx.x * 256L + threadIdx.x; n < a.n1; n += (long)gridDim.x * 256L) {
      const long k = n - a.n0;
      const double sr = a.pp[k], zr = a.pp[(long)(1 + q) * a.CH + k], z2r = a.pp[(long)(1 + a.Q + q) * a.CH + k],
                   zt = a.pp[(long)(1 + 2 * a.Q + q) * a.CH + k];
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
    if (threadIdx.x == 0) { double* d = a.gapart2 + (long)blockIdx.x * a.Q + q; *d = (a.accumulate ? *d : 0.0) + redq[0]; }
    __syncthreads();
  }
}

// grads[m][q] += sum over slices and over the tiles that contain slab(m) (column side when it is the tile's J, row side when its I < J);
// grads[M Q + q] += the per-point kernel's alpha partials.  Fixed order: bit-identical from run to run.
__global__ void __launch_bounds__(256) pt2_gz_reduce_kernel(const double* __restrict__ Gt, const int* __restrict__ tiles, int T, int S, int M, int Q,
                                                            const double* __restrict__ gapart2, int nb2, double* __restrict__ grads) {
  const long MQ = (long)M * Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < MQ + Q; i += (long)gridDim.x * 256L) {
    double s = 0.0;
    if (i < MQ) {
      const int m = (int)(i / Q), q = (int)(i - (long)m * Q), slab = m >> 6, r = m & 63;
      for (int t = 0; t < T; ++t) {
        const int I = tiles[2 * t], J = tiles[2 * t + 1];
        const int side = (J == slab) ? 0 : ((I == slab) ? 1 : -1);     // a diagonal tile contributes through its column side only
        if (side < 0) continue;
        for (int sl = 0; sl < S; ++sl) s += Gt[(((long)sl * T + t) * 2 + side) * (64L * Q) + (long)r * Q + q];
      }
    } else {
      for (int b = 0; b < nb2; ++b) s += gapart2[(long)b * Q + (i - MQ)];
    }
    grads[i] += s;
  }
}

int pt2_width(int Q) { return Q <= 3 ? 4 : Q <= 7 ? 8 : Q <= 11 ? 12 : Q <= 15 ? 16 : Q <= 23 ? 24 : Q <= 31 ? 32 : Q <= 39 ? 40 : Q <= 51 ? 52 : Q <= 63 ? 64 : 0; }

template <int QT>
static size_t pt2_lds_bytes() { return ((size_t)2 * 64 * (QT + 2) + 2 * 64 * TXS + 2 * 2 * 3 * QT + 2 * 4 * 3 * QT + 8 * 32) * sizeof(double); }

template <int QT>
static int launch_tile(gp_ctx* c, const PT2Args& a) {
  const size_t smem = std::max(pt2_lds_bytes<QT>(), (size_t)2 * 2 * 64 * QT * sizeof(double));
  GP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(psi2_tile_kernel<QT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((psi2_tile_kernel<QT>), dim3(a.T, a.S), dim3(512), smem, c->stream, a);
  return GP_OK;
}

// The tile-pair phase 2 applies when a compiled width has room for the column of ones (Q <= 63; r04: the 64-wide instantiation, 230 VGPRs, 156 KB of LDS -- Q = 60, M = 1024, 2e4 points: 89 ms against 161 for
// psi2_cols_mfma_kernel) and is used from Q = 17 on: the per-point
// folds and sums do not shrink with Q, so below that the VALU kernels of psi2.hip are faster (same-box, ms of the phase-2 kernel per 1e5
// points: Q = 4, M = 512: 21.7 (cols) vs 41.8 here; Q = 10, M = 512: 31.7 (psi2_sym) vs 40.5; Q = 16, M = 512: 47.2 vs 56.5;
// Q = 20, M = 256: 23.1 vs 17.0; Q = 24, M = 512: 93.9 vs 69.7; Q = 50, M = 1024, 2e4 points: 96.4 (psi2_cols_mfma) vs 75.8).
// GPARML_B_PHASE2=tiles forces this kernel below Q = 17 as well (tests: every compiled width); =cols keeps the VALU kernels where they exist (Q <= 16).
// Decided once per context (c->b_tile).  (r06: the column kernels' instantiations for Q >= 17 and psi2_cols_mfma_kernel -- reachable only through =cols,
// 28-228 B of scratch per lane -- are gone; Q >= 64 runs on psi2_generic.hip.)
bool pt2_applicable(const gp_ctx* c, bool sym_available) {
  static const int mode = [] { const char* e = getenv("GPARML_B_PHASE2"); return !e ? 0 : (std::string(e) == "cols" ? 1 : (std::string(e) == "tiles" ? 2 : 0)); }();
  (void)sym_available;
  if (pt2_width(c->Q) == 0) return false;
  if (c->Q >= 17) return true;
  return mode == 2;
}

int run_phase2_b_tiles(gp_ctx* c) {
  const int T = c->n_tiles64, Q = c->Q, PW = 3 * Q + 1;
  const long N = c->N;
  // points per launch: the per-tile sums of a launch live in pp [T][PW][CH] (<= 1.5 GB)
  if (!c->ppt || !c->Gt) {
    if (c->ppt) { (void)hipFree(c->ppt); c->ppt = nullptr; }
    if (c->Gt) { (void)hipFree(c->Gt); c->Gt = nullptr; }
    long ch = std::min<long>(N, 8192);
    while (ch > 512 && (double)T * PW * ch * 8.0 > 1.5e9) ch /= 2;
    c->b_ch = ch;
    // slices: fill the 256 CUs (one 512-thread workgroup each) in whole rounds, >= 16 points per slice
    int bestS = 1; double best = -1.0;
    for (int S = 1; S <= 64 && (long)S * 16 <= std::max<long>(16, ch); ++S) {
      const double rounds = std::ceil((double)T * S / 256.0), eff = (double)T * S / (256.0 * rounds);
      if (eff > best + 0.02) { best = eff; bestS = S; }
    }
    c->b_S = bestS;
    // both or neither: a failed second allocation must not leave the first behind (the next call would skip this block and launch with a null Gt).
    // Neither buffer needs zeroing: the first launch of an evaluation (accumulate = 0) writes every pp[t][i][k < count] and all T * S
    // workgroups store their Gt slot unconditionally.
    GP_TRY_RC(dalloc_bytes(c, (void**)&c->ppt, (size_t)T * PW * ch * sizeof(double), DA_RAW));
    if (dalloc_bytes(c, (void**)&c->Gt, (size_t)bestS * T * 2 * 64 * Q * sizeof(double), DA_RAW) != GP_OK) {
      (void)hipFree(c->ppt); c->ppt = nullptr; c->Gt = nullptr;
      return fail(c, GP_ERR_HIP, "regime-B tile phase 2: allocation of the grad_Z partial buffer failed");
    }
  }
  PT2Args a;
  a.ZP = c->ZP; a.Bbar = c->Bbar; a.LEA = c->LET; a.V2P = c->V2P; a.WP = c->WP; a.MUP = c->MUP; a.alphaP = c->alphaP;
  a.tiles = c->tiles64; a.Gt = c->Gt; a.pp = c->ppt; a.CH = c->b_ch; a.Np = c->Np; a.Mp = c->Mp; a.M = c->M; a.Q = Q; a.QB = c->QB; a.T = T; a.S = c->b_S;
  a.dbg = nullptr;
#ifdef GPARML_TILE_TIMING
  static long long* dbg = nullptr;
  const size_t ndbg = (size_t)c->b_S * T * 64;
  if (!dbg) GP_HIP(c, hipMalloc((void**)&dbg, ndbg * sizeof(long long)));
  a.dbg = dbg;
#endif
  PT2Fin f;
  f.pp = c->ppt; f.Wn = c->Wn; f.mu = c->mu; f.S = c->S; f.alpha = c->alpha; f.gmu = c->gXmu; f.gS = c->gXs; f.gapart2 = c->gapart2; f.CH = c->b_ch; f.Q = Q;
  const int fin_blocks = (int)std::min<long>(c->pb_blocks, 256);
  GP_EV(c, 12);
  int k = 0;
  for (long n0 = 0; n0 < N; n0 += c->b_ch, ++k) {
    a.n0 = n0; a.n1 = std::min(N, n0 + c->b_ch); a.accumulate = k > 0 ? 1 : 0;
    int rc = GP_OK;
    switch (pt2_width(Q)) {
      case 4: rc = launch_tile<4>(c, a); break;
      case 8: rc = launch_tile<8>(c, a); break;
      case 12: rc = launch_tile<12>(c, a); break;
      case 16: rc = launch_tile<16>(c, a); break;
      case 24: rc = launch_tile<24>(c, a); break;
      case 32: rc = launch_tile<32>(c, a); break;
      case 40: rc = launch_tile<40>(c, a); break;
      case 52: rc = launch_tile<52>(c, a); break;
      default: rc = launch_tile<64>(c, a); break;
    }
    if (rc != GP_OK) return rc;
    GP_HIP(c, hipGetLastError());
    const long cnt = a.n1 - a.n0;
    hipLaunchKernelGGL(pt2_sum_tiles_kernel, dim3((unsigned)std::min<long>(((long)PW * cnt + 255) / 256, 4096)), dim3(256), 0, c->stream, c->ppt, T, PW,
                       c->b_ch, cnt);
    f.n0 = a.n0; f.n1 = a.n1; f.accumulate = a.accumulate;
    hipLaunchKernelGGL(pt2_points_finish_kernel, dim3(fin_blocks), dim3(256), 0, c->stream, f);
    GP_HIP(c, hipGetLastError());
  }
  GP_EV(c, 13);
#ifdef GPARML_TILE_TIMING
  {
    std::vector<long long> h(ndbg);
    GP_HIP(c, hipMemcpy(h.data(), dbg, ndbg * sizeof(long long), hipMemcpyDeviceToHost));
    // tile (0, 1) of slice 0, waves 0 (stream 0) and 4 (stream 1); s_memtime ticks at 100 MHz
    const long t = 1;       // (0, 1): an off-diagonal tile
    const char* names[8] = {"barrier", "prep", "GEMM1", "exp+store", "col contract", "col fold+s3", "row contract+fold", "s1/s2 sums"};
    for (int wv : {0, 4}) {
      fprintf(stderr, "[tile timing] last launch, tile %ld wave %d (ticks of s_memtime):", t, wv);
      for (int k = 0; k < 8; ++k) fprintf(stderr, " %s=%lld", names[k], h[(size_t)(t * 8 + wv) * 8 + k]);
      fprintf(stderr, "\n");
    }
  }
#endif
  const long MQ = (long)c->M * Q;
  hipLaunchKernelGGL(pt2_gz_reduce_kernel, dim3((unsigned)std::min<long>((MQ + Q + 255) / 256, 2048)), dim3(256), 0, c->stream, (const double*)c->Gt,
                     (const int*)c->tiles64, T, c->b_S, c->M, Q, (const double*)c->gapart2, fin_blocks, c->grads);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

}  // namespace gp
