// Row sums across a wave on the FP64 matrix core (used by psi2_sym_kernel; tools/ubench/quad_mma_check.hip verifies the lane maps).
#pragma once
#include <hip/hip_runtime.h>

namespace gp {

// a double moved across lanes by a DPP control (quad_perm 0..0xFF, row_ror:n = 0x120 + n): two 32-bit DPP moves
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  // mov_dpp (no "old" operand): with update_dpp(0, ...) every move was preceded by a v_mov_b32 0 of its destination
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
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
  // Each of the two butterfly stages is a select between a lane's own value and its quad neighbour's: v_cndmask_b32 takes a DPP source, so
  // move + select are ONE instruction per 32-bit half (16 for the whole transpose; as separate v_mov_b32_dpp + v_cndmask_b32 hipcc
  // emitted 32, and another 28 v_mov_b32 0 for update_dpp's "old" operand).  D = vcc ? src1 : dpp(src0); vcc = lane parity masks.
  // All 64 lanes must be active.  The leading s_nop covers the VALU-write -> DPP-read hazard the compiler cannot see into.
  int t[4][2], a[4][2], o[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { t[i][0] = __double2loint(T[i]); t[i][1] = __double2hiint(T[i]); }
  asm volatile(
      "s_nop 1\n\t"
      "s_mov_b64 vcc, %[m0]\n\t"
      "v_cndmask_b32_dpp %[a0l], %[t1l], %[t0l], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[a0h], %[t1h], %[t0h], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[a2l], %[t3l], %[t2l], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[a2h], %[t3h], %[t2h], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_not_b64 vcc, vcc\n\t"
      "v_cndmask_b32_dpp %[a1l], %[t0l], %[t1l], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[a1h], %[t0h], %[t1h], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[a3l], %[t2l], %[t3l], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[a3h], %[t2h], %[t3h], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_mov_b64 vcc, %[m1]\n\t"
      "v_cndmask_b32_dpp %[o0l], %[a2l], %[a0l], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[o0h], %[a2h], %[a0h], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[o1l], %[a3l], %[a1l], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[o1h], %[a3h], %[a1h], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_not_b64 vcc, vcc\n\t"
      "v_cndmask_b32_dpp %[o2l], %[a0l], %[a2l], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[o2h], %[a0h], %[a2h], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[o3l], %[a1l], %[a3l], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[o3h], %[a1h], %[a3h], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : [a0l] "=&v"(a[0][0]), [a0h] "=&v"(a[0][1]), [a1l] "=&v"(a[1][0]), [a1h] "=&v"(a[1][1]), [a2l] "=&v"(a[2][0]), [a2h] "=&v"(a[2][1]),
        [a3l] "=&v"(a[3][0]), [a3h] "=&v"(a[3][1]), [o0l] "=&v"(o[0][0]), [o0h] "=&v"(o[0][1]), [o1l] "=&v"(o[1][0]), [o1h] "=&v"(o[1][1]),
        [o2l] "=&v"(o[2][0]), [o2h] "=&v"(o[2][1]), [o3l] "=&v"(o[3][0]), [o3h] "=&v"(o[3][1])
      : [t0l] "v"(t[0][0]), [t0h] "v"(t[0][1]), [t1l] "v"(t[1][0]), [t1h] "v"(t[1][1]), [t2l] "v"(t[2][0]), [t2h] "v"(t[2][1]),
        [t3l] "v"(t[3][0]), [t3h] "v"(t[3][1]), [m0] "s"(0x5555555555555555ull), [m1] "s"(0x3333333333333333ull)
      : "vcc", "scc");
  double At[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) At[i] = __hiloint2double(o[i][1], o[i][0]);
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
