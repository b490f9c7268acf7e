// Regime B (Bayesian GPLVM), phase 2 on TILE PAIRS of the symmetric per-point matrix T_n = Bbar o psi2_n  (Q <= 51).
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
#include <algorithm>
#include <cstdlib>

namespace gp {

struct PT2Args {
  const double* ZP; const double* Bbar; const double* LEA; const double* V2P; const double* WP; const double* MUP; const double* alphaP;
  const int* tiles;     // [T][2] (I, J), I <= J, 64-row slabs
  double* Gt;           // [S][T][2][64][Q] grad_Z partials per workgroup: side 0 = rows of slab J (column side), 1 = rows of slab I (row side)
  double* pp;           // [T][3Q+1][CH] per-point sums of every tile for the points of this launch
  long n0, n1, CH, Np;
  int Mp, M, Q, QB, T, S, accumulate;
};

template <int MASK>
__device__ __forceinline__ double lane_xor(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  if constexpr (MASK == 32) {
    lo = __shfl_xor(lo, 32); hi = __shfl_xor(hi, 32);
  } else {
    constexpr int pat = 0x1F | (MASK << 10);      // ds_swizzle bit mode: and 0x1f, or 0, xor MASK (inside 32 lanes; no memory access)
    lo = __builtin_amdgcn_ds_swizzle(lo, pat); hi = __builtin_amdgcn_ds_swizzle(hi, pat);
  }
  return __hiloint2double(hi, lo);
}
// one reduce-scatter stage over lane bit BIT: the lane pair (l, l ^ 2^BIT) splits the N values, lane bit 0 keeps the even indices
template <int N, int BIT>
__device__ __forceinline__ void halve(const double (&v)[N], double (&w)[(N + 1) / 2], int lane) {
  const bool sel = (lane >> BIT) & 1;
#pragma unroll
  for (int i = 0; i < N / 2; ++i) {
    const double keep = sel ? v[2 * i + 1] : v[2 * i];
    const double give = sel ? v[2 * i] : v[2 * i + 1];     // selected BEFORE the cross-lane move: every lane executes the move
    w[i] = keep + lane_xor<(1 << BIT)>(give);
  }
  if constexpr (N & 1) w[N / 2] = v[N - 1] + lane_xor<(1 << BIT)>(v[N - 1]);
}
// sum over the 16 lanes that differ in lane bits 2..5; lane l returns the complete sum of v[l >> 2] (if l >> 2 < N), N <= 16
template <int N>
__device__ __forceinline__ double reduce16(const double (&v)[N], int lane) {
  constexpr int N1 = (N + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2;
  static_assert((N3 + 1) / 2 == 1, "reduce16 handles up to 16 values");
  double a[N1], b[N2], c[N3], d[1];
  halve<N, 2>(v, a, lane); halve<N1, 3>(a, b, lane); halve<N2, 4>(b, c, lane); halve<N3, 5>(c, d, lane);
  return d[0];
}
__device__ __forceinline__ double mul_asm(double a, double b) {   // ordered with the asm MFMAs around it (the compiler may not move it)
  double r;
  asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int QT>
__global__ void __launch_bounds__(512, 2) psi2_tile_kernel(PT2Args a) {
  constexpr int NQ = QT / 4, LDZ = QT + 2, TS = 65, RW = 3 * QT, NG = (NQ + 3) / 4;
  extern __shared__ double smem[];
  double* zr = smem;                          // [64][LDZ]  Z1 of slab I (rows of the tile)
  double* zj = zr + 64 * LDZ;                 // [64][LDZ]  Z1 of slab J (columns)
  double* txb = zj + 64 * LDZ;                // [2 streams][64 cols][TS]  T^T of the current point
  double* qtb = txb + 2 * 64 * TS;            // [2 streams][2 parities][3][QT]  v2 | alpha + w | 2 w mu
  double* redb = qtb + 2 * 2 * 3 * QT;        // [2 streams][4 waves][RW]  per-wave s1 | s2 | s3
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, w = wave & 3, ts = tid & 255;
  const int li = lane & 3, lb = (lane >> 2) & 3, lk = lane >> 4, lr = lane & 15;
  const int I = a.tiles[2 * blockIdx.x], J = a.tiles[2 * blockIdx.x + 1];
  const bool offd = I != J;
  const long per = (a.n1 - a.n0 + a.S - 1) / a.S;
  const long na = a.n0 + (long)blockIdx.y * per, nb = min(a.n1, na + per);
  const int PW = 3 * a.Q + 1;
  for (int e = tid; e < 64 * QT; e += 512) {
    const int r = e / QT, q = e - r * QT;
    const double one = (q == QT - 1) ? 1.0 : 0.0;
    zr[r * LDZ + q] = q < a.Q ? a.ZP[(64L * I + r) * a.QB + q] : one;
    zj[r * LDZ + q] = q < a.Q ? a.ZP[(64L * J + r) * a.QB + q] : one;
  }
  double bb[4][4];     // Bbar[row 64 I + 16 rb + 4 lb + lk][col 64 J + 16 w + 4 cq + li]: the layout of the E / T registers
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cq = 0; cq < 4; ++cq) bb[rb][cq] = a.Bbar[(64L * I + 16 * rb + 4 * lb + lk) * a.Mp + 64 * J + 16 * w + 4 * cq + li];
  double Gc[NQ], Gr[NQ];   // grad_Z of (column 64 J + 16 w + 4 lb + lk | row 64 I + 16 w + 4 lb + lk), q = 4 qq + li, over this stream's points
#pragma unroll
  for (int qq = 0; qq < NQ; ++qq) { Gc[qq] = 0.0; Gr[qq] = 0.0; }
  double* tx = txb + g * (64 * TS);
  double* red = redb + (g * 4 + w) * RW;
  const unsigned aA = lds_byte_addr(zr) + 8u * (unsigned)(lr * LDZ + lk);                       // GEMM1 A: Z1_I[16 rb + lr][4 k4 + lk]
  const unsigned aB = lds_byte_addr(zj) + 8u * (unsigned)((16 * w + li) * LDZ + lk);            // GEMM1 B: Z1_J[16 w + 4 cq + li][4 k4 + lk]
  const unsigned aB2 = lds_byte_addr(zr) + 8u * (unsigned)((4 * lb + lk) * LDZ + li);           // column side B: Z1_I[16 rb + 4 lb + lk][4 qq + li]
  const unsigned aT = lds_byte_addr(tx) + 8u * (unsigned)(lk * TS + 16 * w + 4 * lb + li);      // row side A: T[16 w + 4 lb + li][4 cq' + lk]
  const unsigned aZ = lds_byte_addr(zj) + 8u * (unsigned)(lk * LDZ + li);                       // row side B: Z1_J[4 cq' + lk][4 qq + li]
  // ---- the next point's operands travel while the current point computes
  double nlr[4], nlc[4], nq0 = 0.0, nq1 = 0.0, nq2 = 0.0;
  auto load_point = [&](long n) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) nlr[rb] = a.LEA[n * a.Mp + 64 * I + 16 * rb + 4 * lb + lk];
#pragma unroll
    for (int cq = 0; cq < 4; ++cq) nlc[cq] = a.LEA[n * a.Mp + 64 * J + 16 * w + 4 * cq + li];
    if (ts < a.Q) {
      const double wq = a.WP[n * a.QB + ts];
      nq0 = a.V2P[n * a.QB + ts]; nq1 = a.alphaP[ts] + wq; nq2 = 2.0 * wq * a.MUP[n * a.QB + ts];
    }
  };
  long n = na + g;
  bool act = n < nb;
  if (act) load_point(n);
  long n_prev = -1;
  const int iters = (int)((nb - na + 1) / 2);
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    double* qt = qtb + (g * 2 + (it & 1)) * (3 * QT);
    if (act && ts < QT) { qt[ts] = ts < a.Q ? nq0 : 0.0; qt[QT + ts] = ts < a.Q ? nq1 : 0.0; qt[2 * QT + ts] = ts < a.Q ? nq2 : 0.0; }
    double lrow[4], lcol[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { lrow[k] = nlr[k]; lcol[k] = nlc[k]; }
    const long n_cur = n;
    const bool act_cur = act;
    n += 2;
    act = n < nb;
    if (act) load_point(n);
    __syncthreads();                                                                   // barrier A
    if (n_prev >= 0 && ts < PW) {
      // the previous point's sums of this stream: four waves -> pp[tile][i][n]   (i = 0: s0 = the ones column of s1)
      const double* rp = redb + g * 4 * RW;
      const int src = ts == 0 ? QT - 1 : ((ts - 1) / a.Q) * QT + (ts - 1) % a.Q;
      a.pp[((long)blockIdx.x * PW + ts) * a.CH + (n_prev - a.n0)] = (rp[src] + rp[RW + src]) + (rp[2 * RW + src] + rp[3 * RW + src]);
    }
    double S1 = 0.0, S2 = 0.0, S3 = 0.0;
    if (act_cur) {
      // ---- GEMM1 (operands of step k4 + 1 are read while the 16 MFMAs of step k4 execute; counted waits: only asm LDS reads in here)
      double T[4][4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) T[rb][cq] = 0.0;
      // the zeros become asm-defined values HERE: left to the register allocator they are rematerialised (v_mov 0) right in front of the
      // first asm MFMA of each accumulator -- a VALU write the MFMA reads without wait states, on a register the previous MFMA may still read
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc_fence<4>(T[rb]);
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
            for (int cq = 0; cq < 4; ++cq) mfma444_acc(T[rb][cq], av[cur][rb], bv[cur][cq]);
        });
        mfma_drain(T[3][3]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc_fence<4>(T[rb]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- T = Bbar o exp(E + LEA[n, row] + LEA[n, col])
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) T[rb][cq] = bb[rb][cq] * fexp(T[rb][cq] + lrow[rb] + lcol[cq]);
      if (offd) {
        // T^T into LDS for the row side: element (row, col) at tx[col][row]
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int cq = 0; cq < 4; ++cq) tx[(16 * w + 4 * cq + li) * TS + 16 * rb + 4 * lb + lk] = T[rb][cq];
      }
      // ---- column side: t[col][q] over the tile's 64 rows, four q-quads at a time; de-replicated into tq[qq] (col 16 w + 4 lb + lk, q = 4 qq + li)
      double tq[NQ];
      static_for<0, NG>([&](auto gc) {
        constexpr int gi = decltype(gc)::value, nq = NQ / NG + (gi < NQ % NG ? 1 : 0), qq0 = gi * (NQ / NG) + (gi < NQ % NG ? gi : NQ % NG);
        double tc[4][nq];
#pragma unroll
        for (int cq = 0; cq < 4; ++cq)
#pragma unroll
          for (int j = 0; j < nq; ++j) tc[cq][j] = 0.0;
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) acc_fence<nq>(tc[cq]);             // asm-defined zeros (see GEMM1)
        double b2[2][nq];
        auto rd2 = [&](auto rc, double (&b_)[nq]) {
          constexpr int rb = decltype(rc)::value;
          static_for<0, nq>([&](auto jc) { constexpr int j = decltype(jc)::value; b_[j] = ds_read64<(16 * rb * LDZ + 4 * (qq0 + j)) * 8>(aB2); });
        };
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rd2(IC<0>{}, b2[0]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc_fence<4>(T[rb]);
        asm volatile("s_nop 4");                                          // VALU-written T -> MFMA operand
        static_for<0, 4>([&](auto rc) {
          constexpr int rb = decltype(rc)::value, cur = rb & 1;
          if constexpr (rb + 1 < 4) { rd2(IC<rb + 1>{}, b2[cur ^ 1]); lgkm_wait<nq>(); }
          else lgkm_wait<0>();
#pragma unroll
          for (int cq = 0; cq < 4; ++cq)
#pragma unroll
            for (int j = 0; j < nq; ++j) mfma444_acc(tc[cq][j], T[rb][cq], b2[cur][j]);
        });
        mfma_drain(tc[3][nq - 1]);
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) acc_fence<nq>(tc[cq]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < nq; ++j) {
          const double four[4] = {tc[0][j], tc[1][j], tc[2][j], tc[3][j]};
          double two[2], one[1];
          halve<4, 2>(four, two, lane);
          halve<2, 3>(two, one, lane);
          tq[qq0 + j] = one[0];
        }
      });
      {
        // fold: this lane owns column mc = 16 w + 4 lb + lk of slab J and q = 4 qq + li
        const double r = quad_xchg<0xFF>(tq[NQ - 1]);                     // the ones column sits at q = QT - 1 (qq = NQ - 1, li = 3)
        const double* zo = zj + (16 * w + 4 * lb + lk) * LDZ + li;
        double s1p[NQ], s2p[NQ], s3p[NQ];
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) {
          const double z = zo[4 * qq], kap = 2.0 * qt[4 * qq + li], c1 = qt[QT + 4 * qq + li], c2 = qt[2 * QT + 4 * qq + li];
          Gc[qq] = fma(tq[qq], kap, fma(r, fma(-z, c1, c2), Gc[qq]));
          const double zr_ = z * r;
          s1p[qq] = zr_; s2p[qq] = z * zr_; s3p[qq] = z * tq[qq];
        }
        S1 = reduce16<NQ>(s1p, lane); S2 = reduce16<NQ>(s2p, lane); S3 = reduce16<NQ>(s3p, lane);
        if (offd) S3 *= 2.0;                                              // z^T T z of the mirrored tile is the same number
      }
    }
    __syncthreads();                                                                   // barrier B: the T tiles are visible
    if (act_cur) {
      if (offd) {
        // ---- row side: this wave owns rows 16 w .. 16 w + 15 of slab I across all 64 columns; result (row 16 w + 4 lb + lk, q = 4 qq + li)
        double tq[NQ];
        static_for<0, NG>([&](auto gc) {
          // q-quads in NG groups of equal size (13 -> 4, 3, 3, 3): an accumulator is reused every nq MFMAs
          constexpr int gi = decltype(gc)::value, nq = NQ / NG + (gi < NQ % NG ? 1 : 0), qq0 = gi * (NQ / NG) + (gi < NQ % NG ? gi : NQ % NG);
          double tr[nq];
#pragma unroll
          for (int j = 0; j < nq; ++j) tr[j] = 0.0;
          acc_fence<nq>(tr);                                              // asm-defined zeros (see GEMM1)
          double ta[2], bz[2][nq];
          auto rd3 = [&](auto cc, double& a_, double (&b_)[nq]) {
            constexpr int c4 = decltype(cc)::value;
            a_ = ds_read64<(4 * c4 * TS) * 8>(aT);
            static_for<0, nq>([&](auto jc) { constexpr int j = decltype(jc)::value; b_[j] = ds_read64<(4 * c4 * LDZ + 4 * (qq0 + j)) * 8>(aZ); });
          };
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          rd3(IC<0>{}, ta[0], bz[0]);
          static_for<0, 16>([&](auto cc) {
            constexpr int c4 = decltype(cc)::value, cur = c4 & 1;
            if constexpr (c4 + 1 < 16) { rd3(IC<c4 + 1>{}, ta[cur ^ 1], bz[cur ^ 1]); lgkm_wait<nq + 1>(); }
            else lgkm_wait<0>();
#pragma unroll
            for (int j = 0; j < nq; ++j) mfma444_acc(tr[j], ta[cur], bz[cur][j]);
            if constexpr (nq < 3) asm volatile("s_nop 15");               // dependent accumulation: the asm MFMAs get no automatic wait states
          });
          mfma_drain(tr[nq - 1]);
          acc_fence<nq>(tr);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < nq; ++j) tq[qq0 + j] = tr[j];
        });
        const double r = quad_xchg<0xFF>(tq[NQ - 1]);
        const double* zo = zr + (16 * w + 4 * lb + lk) * LDZ + li;
        double s1p[NQ], s2p[NQ];
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) {
          const double z = zo[4 * qq], kap = 2.0 * qt[4 * qq + li], c1 = qt[QT + 4 * qq + li], c2 = qt[2 * QT + 4 * qq + li];
          Gr[qq] = fma(tq[qq], kap, fma(r, fma(-z, c1, c2), Gr[qq]));
          const double zr_ = z * r;
          s1p[qq] = zr_; s2p[qq] = z * zr_;
        }
        S1 += reduce16<NQ>(s1p, lane); S2 += reduce16<NQ>(s2p, lane);
      }
      if (lane < QT) { red[lane] = S1; red[QT + lane] = S2; red[2 * QT + lane] = S3; }
    }
    n_prev = act_cur ? n_cur : -1;
  }
  __syncthreads();
  if (n_prev >= 0 && ts < PW) {
    const double* rp = redb + g * 4 * RW;
    const int src = ts == 0 ? QT - 1 : ((ts - 1) / a.Q) * QT + (ts - 1) % a.Q;
    a.pp[((long)blockIdx.x * PW + ts) * a.CH + (n_prev - a.n0)] = (rp[src] + rp[RW + src]) + (rp[2 * RW + src] + rp[3 * RW + src]);
  }
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

int pt2_width(int Q) { return Q <= 3 ? 4 : Q <= 7 ? 8 : Q <= 11 ? 12 : Q <= 15 ? 16 : Q <= 23 ? 24 : Q <= 31 ? 32 : Q <= 51 ? 52 : 0; }

template <int QT>
static size_t pt2_lds_bytes() { return ((size_t)2 * 64 * (QT + 2) + 2 * 64 * 65 + 2 * 2 * 3 * QT + 2 * 4 * 3 * QT) * sizeof(double); }

template <int QT>
static int launch_tile(gp_ctx* c, const PT2Args& a) {
  const size_t smem = std::max(pt2_lds_bytes<QT>(), (size_t)2 * 2 * 64 * QT * sizeof(double));
  GP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(psi2_tile_kernel<QT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((psi2_tile_kernel<QT>), dim3(a.T, a.S), dim3(512), smem, c->stream, a);
  return GP_OK;
}

// the tile-pair phase 2 applies when a compiled width has room for the column of ones (Q <= 51); GPARML_B_PHASE2=cols keeps the older kernels
bool pt2_applicable(const gp_ctx* c) {
  static const bool off = [] { const char* e = getenv("GPARML_B_PHASE2"); return e && std::string(e) == "cols"; }();
  return !off && pt2_width(c->Q) > 0;
}

int run_phase2_b_tiles(gp_ctx* c) {
  const int T = c->n_tiles64, Q = c->Q, PW = 3 * Q + 1;
  const long N = c->N;
  // points per launch: the per-tile sums of a launch live in pp [T][PW][CH] (<= 1.5 GB)
  if (!c->ppt) {
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
    GP_HIP(c, hipMalloc((void**)&c->ppt, (size_t)T * PW * ch * sizeof(double)));
    GP_HIP(c, hipMalloc((void**)&c->Gt, (size_t)bestS * T * 2 * 64 * Q * sizeof(double)));
  }
  PT2Args a;
  a.ZP = c->ZP; a.Bbar = c->Bbar; a.LEA = c->LET; a.V2P = c->V2P; a.WP = c->WP; a.MUP = c->MUP; a.alphaP = c->alphaP;
  a.tiles = c->tiles64; a.Gt = c->Gt; a.pp = c->ppt; a.CH = c->b_ch; a.Np = c->Np; a.Mp = c->Mp; a.M = c->M; a.Q = Q; a.QB = c->QB; a.T = T; a.S = c->b_S;
  PT2Fin f;
  f.pp = c->ppt; f.Wn = c->Wn; f.mu = c->mu; f.S = c->S; f.alpha = c->alpha; f.gmu = c->gXmu; f.gS = c->gXs; f.gapart2 = c->gapart2; f.CH = c->b_ch; f.Q = Q;
  const int fin_blocks = (int)std::min<long>(c->pb_blocks, 256);
  (void)hipEventRecord(c->ev[12], c->stream);
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
      default: rc = launch_tile<52>(c, a); break;
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
  (void)hipEventRecord(c->ev[13], c->stream);
  const long MQ = (long)c->M * Q;
  hipLaunchKernelGGL(pt2_gz_reduce_kernel, dim3((unsigned)std::min<long>((MQ + Q + 255) / 256, 2048)), dim3(256), 0, c->stream, (const double*)c->Gt,
                     (const int*)c->tiles64, T, c->b_S, c->M, Q, (const double*)c->gapart2, fin_blocks, c->grads);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

}  // namespace gp
