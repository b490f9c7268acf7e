// FP64 matrix-core building blocks for gfx950 (MI355X).
//
// Measured on MI355X (tools/ubench, profiles/ubench_f64.txt):
//   v_mfma_f64_16x16x4_f64      ~36 TF  (128+ cycles / instruction: HALF the FP64 rate)
//   v_mfma_f64_4x4x4_4b_f64     ~74 TF  (16 cycles / instruction, = the FP64 VALU FMA rate, same pipe)
//   v_fma_f64 (VALU)            ~74 TF
// so the dense FP64 contractions use the 4x4x4 (4-block) MFMA: full rate with 1/4 of the issue slots and
// 1/4 of the operand traffic of VALU FMAs.
//
// Lane map of __builtin_amdgcn_mfma_f64_4x4x4f64 (verified by tools/ubench/f64_ubench2.hip): lane l belongs to
// block b = (l>>2)&3;  A operand: A_b[i = l&3][k = l>>4];  B operand: B_b[k = l>>4][j = l&3];
// result: D_b[i = l>>4][j = l&3].  We give the four blocks four consecutive row-quads and the SAME 4 columns,
// so one instruction is a 16x4 (rows x cols) output tile with k = 4:
//   A register: lane l holds A[row = l&15][k = l>>4]                     (64 distinct values)
//   B register: lane l holds B[k = l>>4][col = l&3]                      (16 distinct values, replicated over (l>>2)&3)
//   D register: lane l holds D[row = 4*((l>>2)&3) + (l>>4)][col = l&3]
// A wave owns a 64x64 output tile: 4 A registers x 16 B registers = 64 accumulators (128 VGPRs).
#pragma once
#include <hip/hip_runtime.h>

namespace gp {

// compile-time loop: f(IC<B>{}), ..., f(IC<E-1>{}) -- the index is a constant expression inside f (immediate offsets, register arrays)
template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
}

constexpr int TILE = 128;          // workgroup output tile (rows and cols), 4 waves as 2x2 of 64x64
constexpr int WT = 64;             // wave tile
constexpr int KC = 16;             // k-chunk staged through LDS per iteration
constexpr int LDS_RC = TILE + 16;  // LDS row stride (doubles) of a [KC][TILE] tile (operand contiguous along its free index):
                                   // +16 doubles puts the two k-rows a 32-lane read group touches on different bank halves
                                   // (unpadded, every one of the 20 operand reads per k-step is a 2-way conflict)
constexpr int TILE_LDS_DOUBLES = KC * LDS_RC;  // 2304 doubles = 18 KB per operand tile (a [TILE][KC] tile uses 2048 of them)

// how an operand tile is stored (in global memory and, identically, in LDS)
enum Layout : int {
  FREE_CONTIG = 0,   // stored [k][r]: rows indexed by k, free index r contiguous
  K_CONTIG = 1       // stored [r][k]: rows indexed by r, k contiguous
};

// Inline asm keeps the accumulation in place (dst == srcC): with the builtin hipcc renames accumulators to make
// room for ds_read2 destinations and then pays ~130 v_mov_b64 per k-chunk to rotate them back.  The operands
// come straight from ds_read (the compiler's s_waitcnt covers them).  hipcc does not model the MFMA inside the
// string, so anything that READS an accumulator with a non-MFMA instruction must first pass mfma_drain().
__device__ __forceinline__ void mfma444_acc(double& c, double a, double b) {
  asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// first MFMA of an accumulation: C = 0 as an inline constant (no zeroing moves; "&": the result may not share a register with an operand)
__device__ __forceinline__ void mfma444_zero(double& c, double a, double b) {
  asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
// wait states between the last in-flight MFMA and a VALU/VMEM read of an accumulator (4-pass DGEMM result).  The s_nop is
// tied to one accumulator only; every OTHER accumulator that is read afterwards must pass acc_fence (an empty asm with the
// register as an in/out operand, placed after the s_nop: volatile asms keep their order), otherwise the scheduler may
// legally hoist such a read to just behind that accumulator's last MFMA -- inside the hazard window.
__device__ __forceinline__ void mfma_drain(double& last_acc) { asm volatile("s_nop 15" : "+v"(last_acc)); }
__device__ __forceinline__ void acc_fence8(double (&x)[8]) {
  asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}
__device__ __forceinline__ void acc_fence16(double (&x)[16]) {
  asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
  asm volatile("" : "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
}
template <int N>
__device__ __forceinline__ void acc_fence(double (&x)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(x[i]));
}
__device__ __forceinline__ double mfma444(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// ---- LDS operand reads as explicit ds_read_b64 -----------------------------------------------------------------------
// hipcc merges two 8-byte LDS loads off one address register into ds_read2_b64 / ds_read2st64_b64.  Those are serviced in
// 16-lane groups over 32 banks (half the rate of two ds_read_b64, MI355X_MICROARCH.md LDS table), and the K_CONTIG image above --
// laid out for the 32-lane groups of ds_read_b64 -- becomes a 2-way bank conflict on every A-operand read (PMC: 6.2e8 conflict
// cycles in the phase-2 kernel, 0 in phase 1 whose operands are both FREE_CONTIG).  The hot loops therefore issue their reads
// through inline asm; the compiler does not track those, so the loops carry their own s_waitcnt lgkmcnt(n): LDS returns in
// order, and lgkmcnt(n) bounds ALL outstanding LGKM operations, so at most n of the wave's reads are still in flight.
typedef __attribute__((address_space(3))) const char lds_cchar;
__device__ __forceinline__ unsigned lds_byte_addr(const double* p) { return (unsigned)(size_t)(lds_cchar*)p; }
template <int OFF>
__device__ __forceinline__ double ds_read64(unsigned byte_addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b64 immediate offset is 16 bits");
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

struct Acc {
  double v[4][16];
  // drain the matrix pipe and order every later read of this tile's accumulators behind it
  __device__ __forceinline__ void drain() {
    asm volatile("s_nop 15" : "+v"(v[3][15]));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("" : "+v"(v[i][0]), "+v"(v[i][1]), "+v"(v[i][2]), "+v"(v[i][3]), "+v"(v[i][4]), "+v"(v[i][5]), "+v"(v[i][6]), "+v"(v[i][7]));
      asm volatile("" : "+v"(v[i][8]), "+v"(v[i][9]), "+v"(v[i][10]), "+v"(v[i][11]), "+v"(v[i][12]), "+v"(v[i][13]), "+v"(v[i][14]), "+v"(v[i][15]));
    }
  }
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) v[i][j] = 0.0;
  }
};

// element (row, col) inside the wave's 64x64 tile that accumulator [ar][bc] holds for this lane
__device__ __forceinline__ int acc_row(int ar, int lane) { return 16 * ar + 4 * ((lane >> 2) & 3) + (lane >> 4); }
__device__ __forceinline__ int acc_col(int bc, int lane) { return 4 * bc + (lane & 3); }

// ---- global -> LDS staging of one operand tile by LDS-DMA (global_load_lds_dwordx4), 256 threads ----------------
// The DMA writes wave-uniform base + lane*16 B, so each wave-instruction fills 1 KB of contiguous LDS.
//  FREE_CONTIG tile [KC][TILE]: one instruction = one k-row (128 doubles); rows are padded to LDS_RC doubles
//    (padding sits BETWEEN instructions, which is allowed) so that the two k-rows a 32-lane group reads land on
//    different bank halves.
//  K_CONTIG tile [TILE][KC]: one instruction = 8 rows x 128 B, unpadded.  To read 16 rows x one k conflict-free
//    the image is permuted through the SOURCE address (the destination is fixed by the hardware):
//      slot(row)   = row with bits 0 and 3 swapped          (so the slot parity = bit 3 of the row)
//      pairpos(kp) = kp ^ (row & 7)                         (16-byte pair index inside the 128-B row)
//    A 32-lane read group (16 rows, k and k+1) then covers all 32 8-byte bank positions exactly once.
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16(const double* gsrc, double* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int swap03(int r) { return (r & ~9) | ((r & 1) << 3) | ((r >> 3) & 1); }

// index (in doubles) of element (r, k) inside a K_CONTIG LDS tile
__device__ __forceinline__ int kc_index(int r, int k) { return swap03(r) * KC + 2 * ((k >> 1) ^ (r & 7)) + (k & 1); }

// src -> element (k0, r0) [FREE_CONTIG] or (r0, k0) [K_CONTIG]; ld = row stride in doubles (even, 16-B aligned rows)
template <Layout L>
__device__ __forceinline__ void tile_dma(double* lds_tile, const double* __restrict__ src, long ld, int wave, int lane) {
  if (L == FREE_CONTIG) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 4 + i;
      glds16(src + (long)row * ld + 2 * lane, lds_tile + row * LDS_RC);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int slot = (wave * 4 + i) * 8 + (lane >> 3);
      const int row = swap03(slot);
      const int kp = (lane & 7) ^ (row & 7);
      glds16(src + (long)row * ld + 2 * kp, lds_tile + (wave * 4 + i) * 8 * KC);
    }
  }
}

// all of this wave's DMAs have landed
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- one KC-deep update of the wave's 64x64 accumulator from LDS tiles ---------------------------------------
// Per-lane LDS offsets (in doubles) are split into a lane-dependent base, computed once per kernel, plus
// compile-time constants that fold into the ds_read immediate offset.
//   FREE_CONTIG:  idx(k, x) = k*LDS_RC + x                                   base = lk*LDS_RC + x_lane
//   K_CONTIG A :  row = wrow0 + 16*ar + lr; idx = kc_index(row, 4*k4 + lk)   base[k4] (4 values) + 256*ar
//   K_CONTIG B :  col = wcol0 + 4*bc + lj;  idx = kc_index(col, 4*k4 + lk)   base[k4 & 1] (2 values) + const(bc, k4)
struct LaneOfs {
  int a[4];  // A operand base per k4 (FREE_CONTIG uses a[0] only)
  int b[2];  // B operand base per k4 parity (FREE_CONTIG uses b[0] only)
};

template <Layout LA, Layout LB>
__device__ __forceinline__ LaneOfs lane_offsets(int wrow0, int wcol0, int lane) {
  const int lr = lane & 15, lk = lane >> 4, lj = lane & 3;
  LaneOfs o;
  if (LA == FREE_CONTIG) {
    o.a[0] = o.a[1] = o.a[2] = o.a[3] = lk * LDS_RC + wrow0 + lr;
  } else {
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
      o.a[k4] = (wrow0 + swap03(lr)) * KC + 2 * (((k4 ^ ((lr >> 1) & 3)) << 1) | ((lk >> 1) ^ (lr & 1))) + (lk & 1);
  }
  if (LB == FREE_CONTIG) {
    o.b[0] = o.b[1] = lk * LDS_RC + wcol0 + lj;
  } else {
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int P = ((par ^ ((lj >> 1) & 1)) << 1) | ((lk >> 1) ^ (lj & 1));
      o.b[par] = wcol0 * KC + (lj & 1) * 128 + (lj & 2) * 16 + 2 * P + (lk & 1);
    }
  }
  return o;
}

template <Layout LA, Layout LB>
__device__ __forceinline__ void load_operands(const double* sA, const double* sB, const LaneOfs& o, int k4, double (&a)[4], double (&b)[16]) {
#pragma unroll
  for (int ar = 0; ar < 4; ++ar)
    a[ar] = (LA == FREE_CONTIG) ? sA[o.a[0] + 4 * k4 * LDS_RC + 16 * ar] : sA[o.a[k4] + 256 * ar];
#pragma unroll
  for (int bc = 0; bc < 16; ++bc)
    b[bc] = (LB == FREE_CONTIG)
                ? sB[o.b[0] + 4 * k4 * LDS_RC + 4 * bc]
                : sB[o.b[k4 & 1] + (bc & 1) * 64 + ((bc >> 1) & 1) * 16 + (bc >> 2) * 256 + (((k4 >> 1) ^ (bc & 1)) << 3)];
}

__device__ __forceinline__ void mma_step(Acc& acc, const double (&a)[4], const double (&b)[16]) {
#pragma unroll
  for (int ar = 0; ar < 4; ++ar)
#pragma unroll
    for (int bc = 0; bc < 16; ++bc) mfma444_acc(acc.v[ar][bc], a[ar], b[bc]);
}

// Operand registers are double-buffered: the ds_reads of step k4+1 are issued before the 64 MFMAs of step k4,
// so one wave alone keeps the matrix pipe busy inside a chunk.
template <Layout LA, Layout LB>
__device__ __forceinline__ void mma_chunk(const double* sA, const double* sB, Acc& acc, const LaneOfs& o) {
  double a0[4], b0[16], a1[4], b1[16];
  load_operands<LA, LB>(sA, sB, o, 0, a0, b0);
  load_operands<LA, LB>(sA, sB, o, 1, a1, b1);
  __builtin_amdgcn_sched_barrier(0);
  mma_step(acc, a0, b0);
  __builtin_amdgcn_sched_barrier(0);
  load_operands<LA, LB>(sA, sB, o, 2, a0, b0);
  __builtin_amdgcn_sched_barrier(0);
  mma_step(acc, a1, b1);
  __builtin_amdgcn_sched_barrier(0);
  load_operands<LA, LB>(sA, sB, o, 3, a1, b1);
  __builtin_amdgcn_sched_barrier(0);
  mma_step(acc, a0, b0);
  __builtin_amdgcn_sched_barrier(0);
  mma_step(acc, a1, b1);
}

// single-buffered variant (40 fewer VGPRs): for kernels that keep other state live across the k-loop
template <Layout LA, Layout LB>
__device__ __forceinline__ void mma_chunk_sb(const double* sA, const double* sB, Acc& acc, const LaneOfs& o) {
#pragma unroll
  for (int k4 = 0; k4 < KC / 4; ++k4) {
    double a[4], b[16];
    load_operands<LA, LB>(sA, sB, o, k4, a, b);
    mma_step(acc, a, b);
  }
}

// lowest-register variant: B operands in two groups of 8 (16 fewer VGPRs than mma_chunk_sb)
template <Layout LA, Layout LB>
__device__ __forceinline__ void mma_chunk_lo(const double* sA, const double* sB, Acc& acc, const LaneOfs& o) {
#pragma unroll
  for (int k4 = 0; k4 < KC / 4; ++k4) {
    double a[4];
#pragma unroll
    for (int ar = 0; ar < 4; ++ar)
      a[ar] = (LA == FREE_CONTIG) ? sA[o.a[0] + 4 * k4 * LDS_RC + 16 * ar] : sA[o.a[k4] + 256 * ar];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double b[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int bc = 8 * h + j;
        b[j] = (LB == FREE_CONTIG)
                   ? sB[o.b[0] + 4 * k4 * LDS_RC + 4 * bc]
                   : sB[o.b[k4 & 1] + (bc & 1) * 64 + ((bc >> 1) & 1) * 16 + (bc >> 2) * 256 + (((k4 >> 1) ^ (bc & 1)) << 3)];
      }
#pragma unroll
      for (int ar = 0; ar < 4; ++ar)
#pragma unroll
        for (int j = 0; j < 8; ++j) mfma444_acc(acc.v[ar][8 * h + j], a[ar], b[j]);
    }
  }
}

}  // namespace gp
