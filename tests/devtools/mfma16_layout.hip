// Dev probe: register layout of v_mfma_f64_16x16x4_f64 (which D element each lane/register holds).
// hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma16_layout tests/devtools/mfma16_layout.hip && /tmp/mfma16_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(double* out) {
  const int l = threadIdx.x;
  // assume A lane (row l & 15, k l >> 4), B lane (k l >> 4, col l & 15); A[i][k] = 1 + i + 100 k, B[k][j] = (k == 0) ? 1000 * (j + 1) : 0  ->  D[i][j] = (1 + i) * 1000 (j + 1)
  const double a = 1.0 + (l & 15) + 100.0 * (l >> 4);
  const double b = ((l >> 4) == 0) ? 1000.0 * ((l & 15) + 1) : 0.0;
  v4d d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}
int main() {
  double* d; hipMalloc(&d, 256 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) {   // measured on gfx950: register r of lane l holds D[4 r + (l >> 4)][l & 15]
    printf("lane %2d:", l);
    for (int r = 0; r < 4; ++r) { long v = (long)h[l * 4 + r]; printf("  r%d -> D[%ld][%ld]", r, (v / 1000) % 1000 ? (v / ((v / 1000 % 1000 == 0) ? 1 : 1)) : 0, 0L); }
    printf("  raw %g %g %g %g\n", h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
