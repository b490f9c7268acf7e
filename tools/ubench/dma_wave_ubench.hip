// Who pays for an LDS-DMA instruction?  p2_fast8_kernel's ablation (profiles/r03_p2_dma_ablation.txt) charges ~63 cycles of MFMA issue per
// global_load_lds_dwordx4 to the SIMD that issues it.  If that cost is a stall of the ISSUING WAVE only, a dedicated staging wave per SIMD would
// take it off the MFMA waves; if the instruction blocks the SIMD's issue (or the MFMA pipe), it would not.
// Workgroup = 8 waves: waves 0-3 (one per SIMD) run a chain of v_mfma_f64_4x4x4_4b with 8 independent accumulators and time themselves with
// s_memtime; waves 4-7 (the second wave of each SIMD) issue `dma_per_iter` global_load_lds_dwordx4 per loop trip from an L2-resident buffer.
// Variants: no DMA at all; DMA by the partner waves; the same number of DMA instructions issued by the MFMA waves themselves.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/dma_wave_ubench.hip -o tools/ubench/dma_wave_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
constexpr int ITERS = 2000, MF = 32;          // MFMAs per trip (p2_fast8: 128 per chunk and wave, 4 DMA instructions)

__global__ void __launch_bounds__(512) k(const double* __restrict__ src, int mode, int ndma, long long* cyc, double* sink) {
  __shared__ double lds[8 * 2048];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const double a = 1.0 + lane, b = 0.5;
  const double* p = src + ((blockIdx.x * 8 + wave) * 64 + lane) * 2;
  double* dst = lds + wave * 2048;
  if (wave < 4) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
      if (mode == 2) for (int d = 0; d < ndma; ++d) __builtin_amdgcn_global_load_lds((gbl_void*)(p + d * 1024), (lds_void*)(dst + d * 128), 16, 0, 0);
#pragma unroll
      for (int m = 0; m < MF; ++m) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  } else if (mode == 1) {
    for (int it = 0; it < ITERS; ++it) {
      for (int d = 0; d < ndma; ++d) __builtin_amdgcn_global_load_lds((gbl_void*)(p + d * 1024), (lds_void*)(dst + d * 128), 16, 0, 0);
      if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_sleep(8);             // pace the staging wave: ~ndma instructions per MF MFMAs of its partner
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  if (s == 1.2345) sink[0] = s + lds[lane];
}
int main() {
  const int blocks = 256;
  double* src; hipMalloc(&src, blocks * 8 * 64 * 2 * 8 + 16 * 1024 * 8); hipMemset(src, 0, blocks * 8 * 64 * 2 * 8 + 16 * 1024 * 8);
  long long* cyc; hipMalloc(&cyc, blocks * 4 * 8); double* sink; hipMalloc(&sink, 8);
  std::vector<long long> h(blocks * 4);
  for (int ndma : {4, 8}) for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, src, mode, ndma, cyc, sink);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, src, mode, ndma, cyc, sink);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= h.size();
    // s_memtime counts at 100 MHz on gfx950: report relative numbers and per-trip time in ns
    printf("%d DMA instructions per %d MFMAs, %-38s %.1f memtime ticks per trip\n", ndma, MF,
           mode == 0 ? "no DMA:" : mode == 1 ? "DMA issued by the partner wave:" : "DMA issued by the MFMA wave itself:", m / ITERS);
  }
  return 0;
}
