// Table-free FP64 exp for the statistics kernels (shared by psi.hip and psi2.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace gp {

constexpr double kPadLog = -1.0e5;   // log-density of padded rows / columns / points: exp(kPadLog + anything) == 0

// exp for the Psi-statistics kernels: 17 FP64 instructions, no table, no special cases.  x = k ln2 + r, |r| <= ln2/2; degree-11
// near-minimax polynomial (Chebyshev-node fit in extended precision: approximation error 1.6e-17, error of the double
// evaluation 1.7e-16 relative -- far inside the 1e-6 / 1e-5 parity budget).  Arguments are finite log-densities or sums
// of the padding marker kPadLog; anything below about -745 returns exactly 0 through ldexp's underflow.  (Not valid for
// |x| > 1e290 -- that is why the padding marker is -1e5 and not -1e300.)
__device__ __forceinline__ double fexp(double x) {
  const double k = rint(x * 1.4426950408889634074);
  double r = fma(k, -6.93147180369123816490e-01, x);
  r = fma(k, -1.90821492927058770002e-10, r);
  double p = 0x1.af633307a1519p-26;
  p = fma(p, r, 0x1.28b409b390145p-22);
  p = fma(p, r, 0x1.71ddf56b7e3cbp-19);
  p = fma(p, r, 0x1.a01991a5ecd16p-16);
  p = fma(p, r, 0x1.a01a01b143788p-13);
  p = fma(p, r, 0x1.6c16c187ffce5p-10);
  p = fma(p, r, 0x1.111111110f247p-7);
  p = fma(p, r, 0x1.555555554f0aep-5);
  p = fma(p, r, 0x1.555555555555ap-3);
  p = fma(p, r, 0x1.0000000000011p-1);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)k);
}

// ---- the same with a 64-entry table of 2^(j/64): x = (64 e + j) ln2/64 + r, |r| <= ln2/128, exp(x) = 2^e * 2^(j/64) * exp(r) with a degree-5
// polynomial -- 12 FP64 instructions instead of 17.  The table lives in REGISTERS, one entry per lane (exp_tab_lane()), and is looked up
// with two ds_bpermute_b32 (the LDS crossbar, no memory, no bank conflicts; not FP64-pipe instructions).  The lookup reads other lanes'
// registers: call it with ALL lanes of the wave active (compute unconditionally, select afterwards).  Relative error 3.3e-16 (measured over
// [-700, 5]); arguments below about -745 (the padding marker kPadLog) return exactly 0 through ldexp's underflow, as fexp does.
// Used where the loop has no scalar loads (psi1_kernel 1.01 -> 0.91 ms, psi2_pairs_kernel 144 -> 134 ms): ds_bpermute shares lgkmcnt with
// s_load, and in the kernels that feed their row operands through scalar loads (psi2_sym/cols) or keep LDS operand reads in flight (the Q >= 25
// MFMA kernels) the extra waits cancel the saved instructions (measured: no gain or slightly slower).
static __device__ const double kExp2Tab64[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};
struct ExpTab { int lo, hi; };
__device__ __forceinline__ ExpTab exp_tab_lane() {
  const double t = kExp2Tab64[threadIdx.x & 63];
  return ExpTab{__double2loint(t), __double2hiint(t)};
}
__device__ __forceinline__ double fexp_t(double x, const ExpTab& tb) {
  // (r05: rint + convert replaced by the magic-number add x c + 1.5 2^52 -- one instruction fewer -- changed nothing: psi2_pairs_kernel 13.18 vs 13.16 ms per
  // 1e5 points, psi1_kernel 0.945 vs 0.923: neither loop is bound by its VALU instruction count; v_rndne_f64, v_cvt_i32_f64 and v_ldexp_f64 issue at
  // the rate of v_fma_f64, tools/ubench/valu_ubench.hip)
  const double n = rint(x * 9.23324826168936568e+01);           // 64 / ln 2
  double r = fma(n, -0x1.62e42ff000000p-7, x);                   // ln2/64, upper 30 bits (n * hi is exact for |n| < 2^22)
  r = fma(n, 6.56392980106419468e-13, r);                       // -(ln2/64 - hi)
  const int ni = (int)n;
  const int sel = ni << 2;                                       // ds_bpermute takes the lane as (byte address / 4) mod 64: no mask needed
  const int lo = __builtin_amdgcn_ds_bpermute(sel, tb.lo), hi = __builtin_amdgcn_ds_bpermute(sel, tb.hi);
  double p = 1.0 / 120.0;
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(__hiloint2double(hi, lo) * p, ni >> 6);
}

// (r06: the table lookup as a global load -- vmcnt instead of lgkmcnt, which the ds_bpermute share with the scalar row loads of psi2_pairs_kernel /
// psi2_sym_kernel -- was measured: the 64-lane gather into a 512-byte table costs more than the waits it removes, psi2_pairs_kernel 13.1 -> 18.6 ms and
// psi2_sym_kernel 29.9 -> 33.3 ms per 1e5 points; profiles/r06_gplvm_experiments.txt.)

}  // namespace gp
