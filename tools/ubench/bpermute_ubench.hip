// What do the LDS counters report for ds_bpermute_b32 (the crossbar lookup of fexp_t, fexp.h)?  VERDICT r03 weak #5: psi2_pairs_kernel<10> shows
// SQ_LDS_BANK_CONFLICT = 2 x SQ_INSTS_LDS and SQ_LDS_ADDR_CONFLICT = SQ_INSTS_LDS although the kernel touches no LDS memory.
// Three kernels with the same instruction count: (a) ds_bpermute_b32 with a lane permutation, (b) ds_read_b32 of a conflict-free LDS array,
// (c) ds_read_b32 with a 2-way bank conflict.  Run under rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT
// SQ_LDS_IDX_ACTIVE; the program prints its own timing (cycles per instruction per wave).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/bpermute_ubench.hip -o tools/ubench/bpermute_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 4096, UNROLL = 8;
__global__ void __launch_bounds__(256) k_bpermute(int* out) {
  int v = threadIdx.x, idx = ((threadIdx.x * 7 + 3) & 63) << 2;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v = __builtin_amdgcn_ds_bpermute(idx, v) + u;
  }
  if (v == 0x7fffffff) out[0] = v;
}
__global__ void __launch_bounds__(256) k_read_clean(int* out) {
  __shared__ int s[256];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  int v = threadIdx.x;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v = s[(v + u) & 255];          // lanes read distinct consecutive words: conflict-free
  }
  if (v == 0x7fffffff) out[0] = v;
}
__global__ void __launch_bounds__(256) k_read_conflict(int* out) {
  __shared__ int s[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) s[i] = (i / 128) & 63;
  __syncthreads();
  int v = threadIdx.x & 63;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v = s[((v & 63) * 128 + u) & 8191];   // stride 128 words: every lane on the same bank
  }
  if (v == 0x7fffffff) out[0] = v;
}
int main() {
  int* d; hipMalloc(&d, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 1024;
  auto run = [&](const char* name, void (*k)(int*)) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 4 waves per block, 1024 blocks over 1024 SIMDs -> 4 waves per SIMD; dependent chain per wave
    printf("%-16s %.3f ms: %.1f cycles per instruction per wave (dependent chain, 4 waves per SIMD) at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / (ITERS * UNROLL));
  };
  run("ds_bpermute_b32", k_bpermute);
  run("ds_read clean", k_read_clean);
  run("ds_read 64-way", k_read_conflict);
  return 0;
}
