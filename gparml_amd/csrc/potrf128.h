// Cholesky factor and inverse factor of a 128x128 diagonal block, LDS-resident (used by linalg.hip; tools/ubench/potrf_ubench.hip
// times its phases through POTRF_STAMP).
#pragma once
#include <hip/hip_runtime.h>
#include "mma_f64.h"

#ifndef POTRF_STAMP
#define POTRF_STAMP(i)
#endif

namespace gp {

constexpr int NB = 128;  // panel width = GEMM tile

// Cholesky factor AND inverse factor of the 128x128 diagonal block j of each matrix in the batch, in one launch with the block
// resident in LDS (row stride 130: the 16 rows x {k, k+1} of an MFMA operand read fall on 32 distinct 8-byte slots).
// The block is processed as 4 x 4 sub-blocks of 32:
//   * wave 0 factors the 32x32 diagonal sub-block with one ROW per lane in registers (pivots and L_kj by v_readlane, 1/sqrt from
//     v_rsq_f64 + a third-order correction: no LDS traffic, no divergence, no workgroup barrier on the serial chain) and inverts
//     it in place (two 16x16 halves by forward substitution with the column in registers and the rows of L as prefetched
//     broadcast reads, merged by two 16x16x16 MFMA tiles);
//   * all eight waves solve the sub-blocks below it (L_is = A_is W_s^T) and apply the rank-32 update of the trailing
//     sub-blocks as 16x16 tiles of 4x4x4 FP64 MFMAs fed from the LDS image;
//   * the off-diagonal part of the inverse, X21 = -X22 (L21 X11) at sizes 32 and 64, is two more rounds of such tiles, in place
//     (L21 is final in global memory by then, its LDS slot receives L21 X11 and then X21).
// Every LDS operand group is read ahead of its use behind a scheduling barrier: hipcc otherwise sinks each ds_read to its
// first use and waits for it there (one LDS round trip per two FMAs: the substitution alone took 19.6 k cycles per sub-block).
// The earlier pair (register-blocked factorisation with two workgroup barriers per 4-column step + a VALU inverse over LDS
// dot products) took 59 + 36 us per block.
constexpr int PLD = NB + 2;

// LDS-only synchronisation: wait for this wave's LDS operations, not for its global stores (a workgroup-scope fence or
// __syncthreads() also waits for vmcnt(0), i.e. 1-2 us behind every batch of stores of L to global memory)
__device__ __forceinline__ void wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void block_sync_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(x) for a normal positive x: v_rsq_f64 seed + one third-order correction y (1 + e/2 + 3/8 e^2), e = 1 - x y^2
// (the library call costs ~150 cycles of dependent instructions on the factorisation's serial chain, this one ~70)
__device__ __forceinline__ double rsqrt_chain(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y), y, 1.0);
  return fma(y * e, fma(e, 0.375, 0.5), y);
}

// one wave: acc (16x16 tile, MFMA result layout) += sum_{kb <= k < ke} A(i,k) B(k,j) (kb, ke multiples of 16), operands in the LDS panel
//   A(i,k) = S[(ar + i) * PLD + ac + k];  B(k,j) = BT ? S[(br + j) * PLD + bc + k] : S[(br + k) * PLD + bc + j]
// 16-deep chunks, the next chunk's twenty operand reads in flight during the sixteen MFMAs of the current one.
template <bool BT>
__device__ __forceinline__ void lds_mma16(const double* S, int ar, int ac, int br, int bc, int kb, int ke, int lane, double (&acc)[4]) {
  const int lr = lane & 15, lk = lane >> 4, lj = lane & 3;
  const double* pa = S + (ar + lr) * PLD + ac + lk;
  const double* pb = BT ? S + (br + lj) * PLD + bc + lk : S + (br + lk) * PLD + bc + lj;
  auto load = [&](int k0, double (&a)[4], double (&b)[4][4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = k0 + 4 * s;
      a[s] = pa[k];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[s][j] = BT ? pb[4 * j * PLD + k] : pb[k * PLD + 4 * j];
    }
  };
  auto mma = [&](const double (&a)[4], const double (&b)[4][4]) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[s], b[s][j], acc[j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  double a0[4], b0[4][4], a1[4], b1[4][4];
  if (kb < ke) load(kb, a0, b0);
  for (int k = kb; k < ke; k += 32) {
    const bool more1 = k + 16 < ke, more2 = k + 32 < ke;
    if (more1) load(k + 16, a1, b1);
    mma(a0, b0);
    if (more1) {
      if (more2) load(k + 32, a0, b0);
      mma(a1, b1);
    }
  }
}
// element (row, col) of the 16x16 tile that acc[j] holds for this lane
__device__ __forceinline__ int t16_row(int lane) { return 4 * ((lane >> 2) & 3) + (lane >> 4); }
__device__ __forceinline__ int t16_col(int j, int lane) { return 4 * j + (lane & 3); }

constexpr int LSLD = 34;   // row stride of the spare copy of L_ss
constexpr int POTRF_LDS_DOUBLES = NB * PLD + 32 + 32 * LSLD + NB;

__global__ void __launch_bounds__(512) potrf_trinv128_kernel(double* A, long ld, long bstride, int j, double* Linv, double* fail,
                                                             double* logdet2) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* S = sm;                    // [128][PLD] the block: A -> L -> L^-1
  double* Tw = S + NB * PLD;         // [32] reciprocal diagonal of the current sub-block
  double* Lsp = Tw + 32;             // [32][LSLD] L_ss on its way to global memory (its slot in S is inverted in place)
  double* Dg = Lsp + 32 * LSLD;      // [128] diagonal of L (log-determinant, summed at the end)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long off = (long)blockIdx.x * bstride + ((long)j * NB) * ld + (long)j * NB;
  double* blk = A + off;
  double* xblk = Linv + off;
  {
    // 16 x 16 bytes per thread, all in flight together; a wave reads whole 1 KB rows
    double2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int e = tid + 512 * i; v[i] = *reinterpret_cast<const double2*>(blk + (long)(e >> 6) * ld + 2 * (e & 63)); }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + 512 * i, r = e >> 6, c = 2 * (e & 63);
      *reinterpret_cast<double2*>(S + r * PLD + c) = make_double2(c <= r ? v[i].x : 0.0, c + 1 <= r ? v[i].y : 0.0);
    }
  }
  __syncthreads();
  POTRF_STAMP(0);
  double bad = 0.0;   // 1: negative (or NaN) pivot -- indefinite; 2: pivot exactly zero -- singular (numpy.linalg.inv raises for that one)
  for (int s = 0; s < 4; ++s) {
    const int o = 32 * s;
    if (wave == 0) {
      // ---- factor the 32x32 diagonal sub-block: lane i (and its idle twin i + 32) owns row i in registers.  Column j: the
      // pivot comes from lane j by v_readlane, every lane forms 1/sqrt (uniform), scales its entry, and the rank-1 update takes
      // L_kj from lane k the same way: 496 FMAs + 1056 readlanes per sub-block, no LDS traffic and no divergence on the chain.
      // (The 8x8-lanes-of-4x4 form ran ~300 FP64 instructions per 4-column step -- 4x4 factor, solve, 64 update FMAs, each a
      // 4-cycle issue whatever the number of active lanes -- 15 k cycles per sub-block against 6 k.)
      const int i = lane & 31;
      double a[32];
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const double2 v = *reinterpret_cast<const double2*>(S + (o + i) * PLD + o + 2 * p);
        a[2 * p] = v.x; a[2 * p + 1] = v.y;
      }
      double myinv = 1.0, mydiag = 1.0;
      static_for<0, 32>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const double d = readlane_f64(a[j], j);
        if (!(d > 0.0)) bad = fmax(bad, d == 0.0 ? 2.0 : 1.0);     // no repair: the flagged factor is never used (linalg.hip: check_global)
        const double inv = rsqrt_chain(d);
        const double l = a[j] * inv;
        a[j] = l;
        if (i == j) { myinv = inv; mydiag = l; }
        static_for<j + 1, 32>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          a[k] = fma(-l, readlane_f64(l, k), a[k]);
        });
      });
      POTRF_STAMP(1 + 5 * s);
      // L_ss to the LDS image (strict upper part zero) and to the spare copy that wave 7 writes out behind the next barrier
      if (lane < 32) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          const double2 v = make_double2(2 * p <= i ? a[2 * p] : 0.0, 2 * p + 1 <= i ? a[2 * p + 1] : 0.0);
          *reinterpret_cast<double2*>(S + (o + i) * PLD + o + 2 * p) = v;
          *reinterpret_cast<double2*>(Lsp + i * LSLD + 2 * p) = v;
        }
        Tw[i] = myinv;
        Dg[o + i] = mydiag;
      }
      wave_sync();
      POTRF_STAMP(2 + 5 * s);
      // W_s = L_ss^-1 in place.  The two 16x16 diagonal halves by forward substitution, side by side: lane c (< 32) owns column
      // c & 15 of half c >> 4, w_i = (delta_ic - sum_{k<i} L_ik w_k) / L_ii with the column in registers and the rows of L as
      // prefetched reads (one address per half); then W21 = -W22 (L21 W11) as two pairs of 16x16x16 MFMA tiles.
      if (lane < 32) {
        const int h = lane >> 4, cl = lane & 15;
        double w[16];
        double2 row[2][8];
        double dinv[2];
        const double* Ls = S + (o + 16 * h) * PLD + o + 16 * h;
        const double* Tl = Tw + 16 * h;
        auto load_row = [&](auto ic, double2 (&b)[8], double& d) {
          constexpr int i = decltype(ic)::value;
#pragma unroll
          for (int p = 0; p < (i + 1) / 2; ++p) b[p] = *reinterpret_cast<const double2*>(Ls + i * PLD + 2 * p);
          d = Tl[i];
        };
        load_row(IC<0>{}, row[0], dinv[0]);
        static_for<0, 16>([&](auto ic) {
          constexpr int i = decltype(ic)::value, cur = i & 1;
          if constexpr (i + 1 < 16) load_row(IC<i + 1>{}, row[cur ^ 1], dinv[cur ^ 1]);
          __builtin_amdgcn_sched_barrier(0);
          double s0 = (cl == i) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
          for (int p = 0; p < i / 2; ++p) {
            s0 = fma(-row[cur][p].x, w[2 * p], s0);
            s1 = fma(-row[cur][p].y, w[2 * p + 1], s1);
          }
          if constexpr (i & 1) s0 = fma(-row[cur][i / 2].x, w[i - 1], s0);
          w[i] = (s0 + s1) * dinv[cur];
          __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int i = 0; i < 16; ++i) S[(o + 16 * h + i) * PLD + o + 16 * h + cl] = w[i];   // rows above the column's diagonal get their zeros
      }
      wave_sync();
      {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        lds_mma16<false>(S, o + 16, o, o, o, 0, 16, lane, acc);               // T = L21 W11
        wave_sync();
        const int R = o + 16 + t16_row(lane);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { S[R * PLD + o + t16_col(jj, lane)] = acc[jj]; acc[jj] = 0.0; }
        wave_sync();
        lds_mma16<false>(S, o + 16, o + 16, o + 16, o, 0, 16, lane, acc);     // W22 T
        wave_sync();
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) S[R * PLD + o + t16_col(jj, lane)] = -acc[jj];
      }
      wave_sync();
    }
    block_sync_lds();
    POTRF_STAMP(3 + 5 * s);
    if (wave == 7) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int e = lane + 64 * it, r = e >> 4, c = 2 * (e & 15);
        *reinterpret_cast<double2*>(blk + (long)(o + r) * ld + o + c) = *reinterpret_cast<const double2*>(Lsp + r * LSLD + c);
      }
    }
    if (s == 3) break;
    // ---- the sub-blocks below: L_is = A_is W_s^T, 16x16 tiles (at most 12: two per wave), in place behind a barrier
    const int nrb = (NB - o - 32) / 16;
    {
      double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = wave + 8 * u;
        if (t < 2 * nrb) lds_mma16<true>(S, o + 32 + 16 * (t >> 1), o, o + 16 * (t & 1), o, 0, 16 * (t & 1) + 16, lane, acc[u]);   // W_s lower: k <= j
      }
      block_sync_lds();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = wave + 8 * u;
        if (t < 2 * nrb) {
          const int R = o + 32 + 16 * (t >> 1) + t16_row(lane);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int C = o + 16 * (t & 1) + t16_col(jj, lane);
            S[R * PLD + C] = acc[u][jj];
            blk[(long)R * ld + C] = acc[u][jj];
          }
        }
      }
      block_sync_lds();
    }
    POTRF_STAMP(4 + 5 * s);
    // ---- rank-32 update of the trailing lower 16x16 tiles.  Look-ahead (r05): the first pass (tiles 0..7, one per wave) contains the three tiles
    // of the NEXT diagonal sub-block (0, 1, 2 in the row-wise enumeration of the lower triangle); behind the barrier that follows it wave 0 goes
    // straight to that sub-block's factorisation while waves 1..7 finish the remaining tiles (rows from o + 64 on: disjoint from what wave 0 reads and
    // writes) and meet it at the barrier behind the factorisation -- the next solve reads those tiles only after that barrier.  The serial
    // factorisation chain (38 % of the kernel) used to wait for all of the update.
    auto update_tile = [&](int t) {
      int r16 = 0, rem = t;
      while (rem > r16) { rem -= r16 + 1; ++r16; }
      const int c16 = rem;
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      lds_mma16<true>(S, o + 32 + 16 * r16, o, o + 32 + 16 * c16, o, 0, 32, lane, acc);
      const int R = o + 32 + 16 * r16 + t16_row(lane);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) S[R * PLD + o + 32 + 16 * c16 + t16_col(jj, lane)] -= acc[jj];
    };
    const int ntile = nrb * (nrb + 1) / 2;
    if (wave < ntile) update_tile(wave);
    block_sync_lds();
    if (wave > 0)
      for (int t = 8 + (wave - 1); t < ntile; t += 7) update_tile(t);
    POTRF_STAMP(5 + 5 * s);
  }
  // ---- off-diagonal part of the inverse: X21 = -X22 (L21 X11) for block size 32 (two pairs) and 64
  auto level = [&](auto bsc) {
    constexpr int bs = decltype(bsc)::value, tpp = (bs / 16) * (bs / 16), ntask = (NB / (2 * bs)) * tpp;   // 8, then 16 tiles: up to two per wave
    double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    int R0[2], C0[2], rb[2], cb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = wave + 8 * u, p = t / tpp, tt = t % tpp;
      R0[u] = (2 * p + 1) * bs; C0[u] = 2 * p * bs; rb[u] = tt / (bs / 16); cb[u] = tt % (bs / 16);
      if (t < ntask) lds_mma16<false>(S, R0[u] + 16 * rb[u], C0[u], C0[u], C0[u] + 16 * cb[u], 16 * cb[u], bs, lane, acc[u]);   // T = L21 X11 (X11 lower: k >= j)
    }
    block_sync_lds();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (wave + 8 * u < ntask) {
        const int R = R0[u] + 16 * rb[u] + t16_row(lane);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { S[R * PLD + C0[u] + 16 * cb[u] + t16_col(jj, lane)] = acc[u][jj]; acc[u][jj] = 0.0; }
      }
    }
    block_sync_lds();
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (wave + 8 * u < ntask) lds_mma16<false>(S, R0[u] + 16 * rb[u], R0[u], R0[u], C0[u] + 16 * cb[u], 0, 16 * rb[u] + 16, lane, acc[u]);  // X22 T (X22 lower: k <= i)
    block_sync_lds();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (wave + 8 * u < ntask) {
        const int R = R0[u] + 16 * rb[u] + t16_row(lane);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) S[R * PLD + C0[u] + 16 * cb[u] + t16_col(jj, lane)] = -acc[u][jj];
      }
    }
    block_sync_lds();
  };
  level(IC<32>{});
  level(IC<64>{});
  POTRF_STAMP(20);
  {
    // addresses from an opaque copy of the thread id: computed at kernel entry they were spilled, and every reload here waited
    // (vmcnt) for all the stores before it
    int t2 = tid;
    asm volatile("" : "+v"(t2));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = t2 + 512 * i, r = e >> 6, c = 2 * (e & 63);
      const double2 x = *reinterpret_cast<const double2*>(S + r * PLD + c);
      *reinterpret_cast<double2*>(xblk + (long)r * ld + c) = make_double2(c <= r ? x.x : 0.0, c + 1 <= r ? x.y : 0.0);
      if (c > r) *reinterpret_cast<double2*>(blk + (long)r * ld + c) = make_double2(0.0, 0.0);
      else if (c == r) blk[(long)r * ld + c + 1] = 0.0;
    }
  }
  POTRF_STAMP(21);
  if (wave == 0) {
    double lg = log(Dg[lane]) + log(Dg[lane + 64]);
    for (int sh = 32; sh > 0; sh >>= 1) { lg += __shfl_xor(lg, sh); bad = fmax(bad, __shfl_xor(bad, sh)); }
    if (lane == 0) {
      if (bad != 0.0) fail[blockIdx.x] = fmax(fail[blockIdx.x], bad);
      logdet2[blockIdx.x] += 2.0 * lg;                                // panels run one after the other on the stream: fixed order
    }
  }
}

}  // namespace gp
