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


}  // namespace gp
