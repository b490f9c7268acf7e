// The two double-double products of the global step on the INT8 matrix core (r05), for M >= 1024:
//     G = K_mm^-1 Psi2   (M x M x M)         and         R = C - (K_mm + beta Psi2) E   (M x M x D, the residual of the refinement step)
// (partial_terms.py:102-131 are the formulas they feed; DESIGN.md section 5 "global step").  On the FP64 pipe an error-free multiply-add is ten
// instructions (ddacc_block: 0.36 ms per product at M = 1024, 2.9 ms at M = 2048 -- the bound of the step at those sizes, and six exact
// 21-bit-slice products on the FP64 matrix core would cost as much: FP64 MFMA = FP64 VALU rate).  Here both operands are cut into TEN signed 7-bit
// digits below a power-of-two scale per column,
//     x = scale * sum_{d=1..10} q_d 128^-d,   q_d in [-64, 64]  (round to nearest at every digit: unbiased),   70 bits below the column's largest entry,
// the 55 digit products with a + b <= 11 are EXACT integer matrix products on v_mfma_i32_32x32x32_i8 (products of equal order share an int32
// accumulator: |q q'| <= 4096, at most 10 pairs per order, K <= 2048 rows: 8.4e7 < 2^31), and the ten order sums -- exact doubles once multiplied by
// their power-of-two weights and scales -- are added in double-double (two-sum, small terms first).  What is lost: operand bits below 2^-70 of the
// column's scale and products below 2^-84 of scale x scale; against the float64 product the device used before round 4 (2^-53 relative to every
// PARTIAL sum) that is 1e5 times closer, which is what the ten truths need (tests/test_hp_truth_large.py run through this path at M >= 1024 only by
// option; the M = 512 headline keeps ddacc_block).
//
// Both operands are columns of one digit array W = [A^T-side columns | B-side columns], laid out as the matrix core reads it:
//     planes[digit][k / 16][column][16 bytes]   -- the 16 consecutive k a lane feeds as ONE operand register quad
// (A = K_mm^-1 and A = K_mm + beta Psi2 are symmetric, so "column i of W" is row i of A).  One wave owns a 32 x 32 output tile and loads its operands
// straight from global memory in operand order (512-byte runs; the whole problem sits in L2), the next k-step's 20 loads in flight under the 55 MFMAs of
// the current one; four waves per workgroup, one workgroup per CU (160 accumulator + 160 operand registers: AccVGPRs at one wave per SIMD).  What bounds it at
// M = 1024 is L2 bandwidth: 1024 tiles x 32 k-steps x 20 KB = 655 MB of operands per product in 58 us = 11 TB/s (a second k-step of look-ahead spilled and
// could not have helped); sharing operands between the four waves of a workgroup through LDS would halve that -- not built, the product is 4 % of the step.
#include "gp_common.h"
#include <atomic>
#include <cstdlib>

namespace gp {

constexpr int GS_S = 10;                   // digits per operand
typedef int gs_v4i __attribute__((ext_vector_type(4)));
typedef int gs_v16i __attribute__((ext_vector_type(16)));

std::atomic<int> g_opt_gs_i8{[] { const char* e = getenv("GPARML_GS_I8"); return (e && e[0] == '0') ? 0 : 1; }()};

struct GsMat { const double* X; long ld; int ncols; int c0; };      // a block of W's columns: X [K rows][ncols] row-major, columns c0 .. c0 + ncols of W

// scale[c] = the power of two >= 2 max_k |X[k][c]| (1 for a zero column): |x| / scale <= 1/2, so the first digit is in [-64, 64]
__global__ void __launch_bounds__(256) gsi8_colscale_kernel(GsMat m0, GsMat m1, int K, double* __restrict__ scale) {
  // workgroup = 16 columns x 16 row groups (128-byte runs per row; the first form -- 64 columns x 4 row groups, 16 workgroups in all at M = 512 -- took
  // 40 us at M = 512 and 100 us at M = 1024, more than the product it served)
  __shared__ double red[16][17];
  const GsMat m = blockIdx.y ? m1 : m0;
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4, col = blockIdx.x * 16 + cl;
  double mx = 0.0;
  if (col < m.ncols) {
#pragma unroll 8
    for (int k = rg; k < K; k += 16) mx = fmax(mx, fabs(m.X[(long)k * m.ld + col]));
  }
  red[rg][cl] = mx;
  __syncthreads();
  if (rg == 0 && col < m.ncols) {
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmax(mx, red[r][cl]);
    int e = 0;
    if (mx > 0.0) (void)frexp(mx, &e);           // mx = f 2^e, f in [1/2, 1)
    scale[m.c0 + col] = mx > 0.0 ? ldexp(1.0, e + 1) : 1.0;
  }
}

// thread = (16 consecutive k, one column): ten digits per element by t <- 128 t, q = rint(t), t <- t - q (every step exact in float64), the 16 bytes
// of a digit stored as one 16-byte word; consecutive threads = consecutive columns = consecutive words
__global__ void __launch_bounds__(256) gsi8_digits_kernel(GsMat m0, GsMat m1, const double* __restrict__ scale, int8_t* __restrict__ planes,
                                                          long plane_stride, int wcols) {
  const GsMat m = blockIdx.z ? m1 : m0;
  const int col = blockIdx.x * 256 + threadIdx.x, kb = blockIdx.y;
  if (col >= m.ncols) return;
  const double inv = 1.0 / scale[m.c0 + col];     // a power of two: exact
  unsigned w[GS_S][4];
#pragma unroll
  for (int d = 0; d < GS_S; ++d)
#pragma unroll
    for (int j = 0; j < 4; ++j) w[d][j] = 0u;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    double t = m.X[(long)(16 * kb + r) * m.ld + col] * inv;
#pragma unroll
    for (int d = 0; d < GS_S; ++d) {
      t *= 128.0;
      const double q = rint(t);
      t -= q;
      w[d][r >> 2] |= ((unsigned)(int)q & 0xffu) << (8 * (r & 3));
    }
  }
#pragma unroll
  for (int d = 0; d < GS_S; ++d) {
    gs_v4i v = {(int)w[d][0], (int)w[d][1], (int)w[d][2], (int)w[d][3]};
    *reinterpret_cast<gs_v4i*>(planes + (long)d * plane_stride + ((long)kb * wcols + m.c0 + col) * 16) = v;
  }
}

struct GsI8Args {
  const int8_t* planes; long plane_stride; int wcols;
  int ca0, cb0;                  // first column in W of the A-side block (output rows) and of the B-side block (output columns)
  int K;                         // contraction length (rows of W), a multiple of 32
  const double* scale;
  double* out; long ldo;
  const double* Csub;            // nullptr: out = sum; else out = (Csub - hi) - lo (the refinement residual, solve_residual_kernel's rounding)
};

__global__ void __launch_bounds__(256, 1) gsi8_gemm_kernel(GsI8Args a) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ti = blockIdx.y * 2 + (wave >> 1), tj = blockIdx.x * 2 + (wave & 1);
  const int r32 = lane & 31, kg = lane >> 5;
  const long kstep = 2L * a.wcols * 16;
  const int8_t* pa = a.planes + ((long)kg * a.wcols + a.ca0 + 32 * ti + r32) * 16;
  const int8_t* pb = a.planes + ((long)kg * a.wcols + a.cb0 + 32 * tj + r32) * 16;
  gs_v16i acc[GS_S];
#pragma unroll
  for (int o = 0; o < GS_S; ++o)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[o][i] = 0;
  gs_v4i av[GS_S], bv[GS_S], an[GS_S], bn[GS_S];
#pragma unroll
  for (int d = 0; d < GS_S; ++d) {
    av[d] = *reinterpret_cast<const gs_v4i*>(pa + (long)d * a.plane_stride);
    bv[d] = *reinterpret_cast<const gs_v4i*>(pb + (long)d * a.plane_stride);
  }
  const int nks = a.K / 32;
  for (int ks = 0; ks < nks; ++ks) {
    if (ks + 1 < nks) {
#pragma unroll
      for (int d = 0; d < GS_S; ++d) {
        an[d] = *reinterpret_cast<const gs_v4i*>(pa + (long)d * a.plane_stride + (long)(ks + 1) * kstep);
        bn[d] = *reinterpret_cast<const gs_v4i*>(pb + (long)d * a.plane_stride + (long)(ks + 1) * kstep);
      }
    }
#pragma unroll
    for (int da = 0; da < GS_S; ++da)
#pragma unroll
      for (int db = 0; db < GS_S; ++db)
        if (da + db < GS_S) acc[da + db] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[da], bv[db], acc[da + db], 0, 0, 0);
#pragma unroll
    for (int d = 0; d < GS_S; ++d) { av[d] = an[d]; bv[d] = bn[d]; }
  }
  // C / D map of the 32 x 32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); order o = da + db has weight 128^-(o + 2)
  const double sb = a.scale[a.cb0 + 32 * tj + r32];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
#pragma clang fp contract(off)   // the two-sum below must not be fused
    const int gi = 32 * ti + (i & 3) + 8 * (i >> 2) + 4 * kg, gj = 32 * tj + r32;
    const double sc = a.scale[a.ca0 + gi] * sb;                    // powers of two: exact
    double hi = 0.0, lo = 0.0, w = 0x1p-77;                        // 128^-11, the weight of the last order kept
#pragma unroll
    for (int o = GS_S - 1; o >= 0; --o) {
      const double x = (double)acc[o][i] * w * sc;                 // exact: an int32 times powers of two
      const double t = hi + x, bb = t - hi;                        // two-sum
      lo += (hi - (t - bb)) + (x - bb);
      hi = t;
      w *= 128.0;
    }
    const long idx = (long)gi * a.ldo + gj;
    a.out[idx] = a.Csub ? (a.Csub[idx] - hi) - lo : hi + lo;
  }
}

// GPARML_GS_I8_MIN_M (default 1024): the smallest padded M the int8 products replace the double-double kernels at (512: measured)
static const int g_gs_i8_min_m = [] { const char* e = getenv("GPARML_GS_I8_MIN_M"); const int v = e ? atoi(e) : 1024; return v >= 512 ? v : 512; }();
bool gs_i8_wanted(const gp_ctx* c) { return g_opt_gs_i8.load() && c->Mp >= g_gs_i8_min_m && c->Mp <= 2048 && c->gsd != nullptr; }

// out [nA][nB] = (A-side)^T-columns x B-side columns over K rows: out[i][j] = sum_k A[k][i] B[k][j]   (A symmetric in both uses: = sum_k A[i][k] B[k][j])
int run_gs_i8_product(gp_ctx* c, hipStream_t st, const double* A, long lda, int nA, const double* B, long ldb, int nB, int K, double* out, long ldo,
                      const double* Csub) {
  const int wcols = nA + nB;
  const long plane = (long)(K / 16) * wcols * 16;
  if ((size_t)GS_S * plane > c->gsd_bytes || (size_t)wcols > c->gss_count) return fail(c, GP_ERR_STATE, "int8 global-step product: workspace too small");
  GsMat m0{A, lda, nA, 0}, m1{B, ldb, nB, nA};
  hipLaunchKernelGGL(gsi8_colscale_kernel, dim3((std::max(nA, nB) + 15) / 16, 2), dim3(256), 0, st, m0, m1, K, c->gss);
  hipLaunchKernelGGL(gsi8_digits_kernel, dim3((std::max(nA, nB) + 255) / 256, K / 16, 2), dim3(256), 0, st, m0, m1, (const double*)c->gss, c->gsd, plane, wcols);
  GsI8Args g{c->gsd, plane, wcols, 0, nA, K, c->gss, out, ldo, Csub};
  hipLaunchKernelGGL(gsi8_gemm_kernel, dim3(nB / 64, nA / 64), dim3(256), 0, st, g);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

}  // namespace gp
