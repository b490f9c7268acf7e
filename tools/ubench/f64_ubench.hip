// Micro-benchmarks that pin the FP64 numbers the design depends on (gfx950):
//  1. v_mfma_f64_16x16x4_f64 lane map (A=asymmetric, B=asymmetric vs host).
//  2. MFMA-only issue rate, VALU-f64-FMA-only rate, and the two interleaved in one wave
//     (k VALU FMAs per MFMA) to see whether the pipes overlap.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 f64_ubench.hip -o f64_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void layout_kernel(const double* A, const double* B, double* D) {
  // A is 16x4 row-major, B is 4x16 row-major, D 16x16 row-major. One wave.
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

template <int NACC, int KV>
__global__ void __launch_bounds__(256) mix_kernel(double* out, int iters, double seed) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + threadIdx.x * 1e-4;
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
  double m1 = 0.999999, m2 = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < KV; ++k) v[k & 7] = __builtin_fma(v[k & 7], m1, m2);
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KV>
__global__ void __launch_bounds__(256) valu_kernel(double* out, int iters, double seed) {
  double v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x;
  double m1 = 0.999999, m2 = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < KV; ++k) v[k & 15] = __builtin_fma(v[k & 15], m1, m2);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) exp_kernel(double* out, int iters, double seed) {
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = -seed * (i + 1) - threadIdx.x * 1e-3;
  double s = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { s += exp(v[k]); v[k] -= 1e-6; }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_ms(F f, int reps = 5) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  return best;
}

template <int NACC, int KV>
void run_mix(double* d_out, int blocks, int threads, int iters) {
  double ms = time_ms([&] { hipLaunchKernelGGL((mix_kernel<NACC, KV>), dim3(blocks), dim3(threads), 0, 0, d_out, iters, 1.0); });
  double waves = (double)blocks * threads / 64;
  double mfma = waves * iters * NACC;
  double tf = mfma * 2.0 * 16 * 16 * 4 / (ms * 1e-3) / 1e12;
  double vtf = waves * iters * NACC * KV * 64 * 2.0 / (ms * 1e-3) / 1e12;
  // cycles per MFMA per SIMD assuming 2.4 GHz and waves spread evenly over 1024 SIMDs
  double wps = waves / 1024.0;
  double cyc = ms * 1e-3 * 2.4e9 / (iters * NACC * wps);
  printf("mix NACC=%2d KV=%2d blocks=%d thr=%d: %.3f ms  MFMA %.1f TF  VALU %.1f TF  sum %.1f TF  ~%.1f cyc/MFMA/SIMD@2.4GHz\n",
         NACC, KV, blocks, threads, ms, tf, vtf, tf + vtf, cyc);
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  // 1. layout
  {
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = 1 + i * 0.37 + (i % 5); B[i] = 2 - i * 0.11 + (i % 7) * 3; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
    double me = 0; for (int i = 0; i < 256; ++i) me = fmax(me, fabs(D[i] - R[i]));
    printf("layout check max abs err %.3e (%s)\n", me, me < 1e-9 ? "OK" : "WRONG");
  }
  double* d_out; CK(hipMalloc(&d_out, sizeof(double) * 4096 * 512));
  int iters = 2000;
  // 1 wave / SIMD
  run_mix<4, 0>(d_out, 256, 256, iters);
  run_mix<8, 0>(d_out, 256, 256, iters);
  run_mix<16, 0>(d_out, 256, 256, iters);
  run_mix<16, 0>(d_out, 512, 256, iters);   // 2 waves / SIMD
  run_mix<16, 0>(d_out, 1024, 256, iters);  // 4 waves / SIMD
  run_mix<16, 2>(d_out, 256, 256, iters);
  run_mix<16, 4>(d_out, 256, 256, iters);
  run_mix<16, 8>(d_out, 256, 256, iters);
  run_mix<16, 12>(d_out, 256, 256, iters);
  run_mix<16, 16>(d_out, 256, 256, iters);
  run_mix<16, 24>(d_out, 256, 256, iters);
  run_mix<16, 8>(d_out, 512, 256, iters);
  run_mix<16, 16>(d_out, 512, 256, iters);
  run_mix<16, 24>(d_out, 512, 256, iters);
  // VALU only
  for (int blocks : {256, 512, 1024, 2048}) {
    double ms = time_ms([&] { hipLaunchKernelGGL((valu_kernel<64>), dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0); });
    double fl = (double)blocks * 256 * iters * 64 * 2.0;
    double wps = blocks * 4 / 1024.0;
    printf("valu f64 fma blocks=%d: %.3f ms %.1f TF  ~%.2f cyc/instr/SIMD@2.4GHz\n", blocks, ms, fl / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / (iters * 64.0 * wps));
  }
  for (int blocks : {256, 1024, 2048}) {
    int it2 = 500;
    double ms = time_ms([&] { hipLaunchKernelGGL(exp_kernel, dim3(blocks), dim3(256), 0, 0, d_out, it2, 1.0); });
    double n = (double)blocks * 256 * it2 * 8;
    double wps = blocks * 4 / 1024.0;
    printf("libm exp f64 blocks=%d: %.3f ms %.2f Gexp/s  ~%.1f cyc/wave-exp/SIMD@2.4GHz\n", blocks, ms, n / (ms * 1e-3) / 1e9,
           ms * 1e-3 * 2.4e9 / (it2 * 8.0 * wps));
  }
  return 0;
}
