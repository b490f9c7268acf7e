// Row sums across a wave on the FP64 matrix core (used by psi2_sym_kernel; tools/ubench/quad_mma_check.hip verifies the lane maps).
#pragma once
#include <hip/hip_runtime.h>

namespace gp {

// a double moved across lanes by a DPP control (quad_perm 0..0xFF, row_ror:n = 0x120 + n): two 32-bit DPP moves
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int QP> __device__ __forceinline__ double quad_xchg(double v) { return dpp_move<QP>(v); }
template <int N> __device__ __forceinline__ double row_ror(double v) { return dpp_move<0x120 + N>(v); }

// T[i] (i < 4): lane c holds element (row i, column c) of a 4 x 64 matrix.  ZB[v][qq]: lane l = 16 k + 4 b + j holds
// F[column 16 k + 4 b + v][feature 4 qq + j] of a 64 x 4 NQ matrix.  Result: acc[qq] in lane 16 i + 4 b + j = sum_c T[i][c] F[c][4 qq + j]
// (identical in the four blocks b).
//   1. 4x4 transpose inside each lane quad (quad_perm moves): At[v] at quad position i = T[i] at quad position v, which is the A
//      operand of v_mfma_f64_4x4x4_4b for the four columns {16 k + 4 b + v}: A_b[i][k] = T[i][16 k + 4 b + v];
//   2. 4 NQ MFMAs: D_b[i][j] += sum_k A_b[i][k] B_b[k][j] for v = 0..3 -- block b's partial over its sixteen columns;
//   3. the four blocks are added with rotations by 8 and 4 lanes inside each row of 16 lanes.
template <int NQ>
__device__ __forceinline__ void wave_rows_times_features(const double (&T)[4], const double (&ZB)[4][NQ], double (&acc)[NQ]) {
  const int lane = threadIdx.x & 63;
  const bool p0 = (lane & 1) == 0, p1 = (lane & 2) == 0;
  // the cross-lane moves are executed by ALL lanes before the selects: inside `cond ? a : move(b)` the move would run under
  // the condition's exec mask and read its source lanes -- exactly the disabled ones -- as zero
  const double x0 = quad_xchg<0xB1>(T[0]), x1 = quad_xchg<0xB1>(T[1]), x2 = quad_xchg<0xB1>(T[2]), x3 = quad_xchg<0xB1>(T[3]);
  const double A0 = p0 ? T[0] : x1, A1 = p0 ? x0 : T[1], A2 = p0 ? T[2] : x3, A3 = p0 ? x2 : T[3];
  const double y0 = quad_xchg<0x4E>(A0), y1 = quad_xchg<0x4E>(A1), y2 = quad_xchg<0x4E>(A2), y3 = quad_xchg<0x4E>(A3);
  double At[4];
  At[0] = p1 ? A0 : y2;
  At[1] = p1 ? A1 : y3;
  At[2] = p1 ? y0 : A2;
  At[3] = p1 ? y1 : A3;
#pragma unroll
  for (int qq = 0; qq < NQ; ++qq) acc[qq] = 0.0;
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) acc[qq] = __builtin_amdgcn_mfma_f64_4x4x4f64(At[v], ZB[v][qq], acc[qq], 0, 0, 0);
#pragma unroll
  for (int qq = 0; qq < NQ; ++qq) {
    acc[qq] += row_ror<8>(acc[qq]);
    acc[qq] += row_ror<4>(acc[qq]);
  }
}

}  // namespace gp
