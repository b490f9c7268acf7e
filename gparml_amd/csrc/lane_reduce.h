// Sums across the lanes of a wave as reduce-scatters: a stage over lane bit b exchanges HALF of the values with lane ^ 2^b and adds, so N values cost about N
// cross-lane moves and N adds in total instead of 6 N of each for N butterfly sums (psi2_tile.hip since r04; psi2_sym_kernel's per-point sums since r06).
#pragma once
#include <hip/hip_runtime.h>
#include "quad_mma.h"

namespace gp {

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

// The same stage with an odd count padded by a zero (the index map stays a plain bit field: after stages over bits 0 .. k - 1 lane l holds index l mod 2^k).
template <int N, int BIT>
__device__ __forceinline__ void halve_z(const double (&v)[N], double (&w)[(N + 1) / 2], int lane) {
  const bool sel = (lane >> BIT) & 1;
#pragma unroll
  for (int i = 0; i < (N + 1) / 2; ++i) {
    const double odd = (2 * i + 1 < N) ? v[2 * i + 1 < N ? 2 * i + 1 : 0] : 0.0;
    const double keep = sel ? odd : v[2 * i];
    const double give = sel ? v[2 * i] : odd;               // selected BEFORE the cross-lane move: every lane executes the move
    double got;
    if constexpr (BIT == 0) got = quad_xchg<0xB1>(give);    // lanes ^ 1 and ^ 2 are DPP quad permutations: VALU moves, no LDS crossbar
    else if constexpr (BIT == 1) got = quad_xchg<0x4E>(give);
    else got = lane_xor<(1 << BIT)>(give);
    w[i] = keep + got;
  }
}
template <int N, int BIT>
__device__ __forceinline__ double reduce_scatter_from(const double (&v)[N], int lane) {
  if constexpr (BIT == 6) {
    static_assert(N == 1, "at most 64 values");
    return v[0];
  } else {
    double w[(N + 1) / 2];
    halve_z<N, BIT>(v, w, lane);
    return reduce_scatter_from<(N + 1) / 2, BIT + 1>(w, lane);
  }
}
// Sum of every v[i] over the 64 lanes of the wave (all lanes active); lane i (< N <= 64) returns the total of v[i], the other lanes return zeros or copies.
template <int N>
__device__ __forceinline__ double wave_reduce_scatter(const double (&v)[N], int lane) { return reduce_scatter_from<N, 0>(v, lane); }

}  // namespace gp
