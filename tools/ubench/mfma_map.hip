// Lane map of v_mfma_f64_4x4x4_4b_f64 by one-hot probing: for every (la, lb) which output lanes receive a[la] * b[lb]?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_map.hip -o tools/ubench/mfma_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned long long* out) {
  const int lane = threadIdx.x, la = blockIdx.x, lb = blockIdx.y;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  const unsigned long long m = __ballot(d != 0.0);
  if (lane == 0) out[la * 64 + lb] = m;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 4096 * 8);
  hipLaunchKernelGGL(probe, dim3(64, 64), dim3(64), 0, 0, d);
  std::vector<unsigned long long> h(4096);
  hipMemcpy(h.data(), d, 4096 * 8, hipMemcpyDeviceToHost);
  int shown = 0;
  for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb]) {
    if (la < 6 || (la % 16 == 0) || (la == 21)) { printf("a lane %2d (k?=%d b=%d i=%d)  b lane %2d (hi=%d b=%d lo=%d) -> out lanes:", la, la >> 4, (la >> 2) & 3, la & 3, lb, lb >> 4, (lb >> 2) & 3, lb & 3);
      for (int l = 0; l < 64; ++l) if (h[la * 64 + lb] >> l & 1) printf(" %d", l); printf("\n"); ++shown; }
  }
  return 0;
}
