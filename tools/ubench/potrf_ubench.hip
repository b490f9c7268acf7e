// Phase timing of potrf_trinv128_kernel (s_memtime stamps of wave 0, lane 0).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/potrf_ubench.hip -o tools/ubench/potrf_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
__device__ unsigned long long g_stamps[32];
#define POTRF_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_stamps[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#include "../../gparml_amd/csrc/potrf128.h"
int main() {
  const int n = 128;
  std::vector<double> h(2 * n * n);
  for (int b = 0; b < 2; ++b)
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < n; ++k) h[b * n * n + i * n + k] = std::exp(-0.05 * (i - k) * (i - k) / (1.0 + b)) + (i == k ? 0.5 : 0.0);
  double *dA, *dX, *dS;
  hipMalloc(&dA, h.size() * 8); hipMalloc(&dX, h.size() * 8); hipMalloc(&dS, 64);
  hipFuncSetAttribute((const void*)gp::potrf_trinv128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, gp::POTRF_LDS_DOUBLES * 8);
  for (int it = 0; it < 3; ++it) {
    hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemset(dS, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(gp::potrf_trinv128_kernel, dim3(2), dim3(512), gp::POTRF_LDS_DOUBLES * 8, 0, dA, (long)n, (long)n * n, 0, dX, dS + 2, dS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[32];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
    printf("run %d: %.1f us (event), %llu ticks load..store\n", it, ms * 1e3, st[21] - st[0]);
    const char* names[22] = {"load", "s0 factor", "s0 writeback", "s0 inverse", "s0 trsm", "s0 update", "s1 factor", "s1 writeback", "s1 inverse", "s1 trsm",
                             "s1 update", "s2 factor", "s2 writeback", "s2 inverse", "s2 trsm", "s2 update", "s3 factor", "s3 writeback", "s3 inverse", "-", "levels", "store"};
    unsigned long long prev = st[0];
    for (int i = 1; i < 22; ++i) { if (i == 19) continue; printf("  %-14s %8llu ticks\n", names[i], st[i] - prev); prev = st[i]; }
  }
  std::vector<double> X(h.size()), L(h.size());
  hipMemcpy(X.data(), dX, h.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(L.data(), dA, h.size() * 8, hipMemcpyDeviceToHost);
  // check L X = I for batch 0
  double err = 0;
  for (int i = 0; i < n; ++i) for (int k = 0; k < n; ++k) { double s = 0; for (int m = 0; m < n; ++m) s += L[i * n + m] * X[m * n + k]; err = std::fmax(err, std::fabs(s - (i == k))); }
  printf("max |L X - I| = %.3e\n", err);
  return 0;
}
