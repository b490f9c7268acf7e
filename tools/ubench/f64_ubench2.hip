// ubench 2: v_mfma_f64_4x4x4_4b_f64 rate + lane map, and MFMA 4x4x4 + VALU overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void layout444(const double* A, const double* B, double* D) {
  int l = threadIdx.x;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
  D[l] = d;
}

template <int NACC, int KV>
__global__ void __launch_bounds__(256) mix444(double* out, int iters, double seed) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + threadIdx.x * 1e-4;
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
  double m1 = 0.999999, m2 = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < KV; ++k) v[k & 7] = __builtin_fma(v[k & 7], m1, m2);
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  for (int i = 0; i < 8; ++i) s += v[i];
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
void run(double* d_out, int blocks, int iters) {
  double ms = time_ms([&] { hipLaunchKernelGGL((mix444<NACC, KV>), dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0); });
  double waves = blocks * 4.0;
  double tf = waves * iters * NACC * 2.0 * 256 / (ms * 1e-3) / 1e12;
  double vtf = waves * iters * NACC * KV * 64 * 2.0 / (ms * 1e-3) / 1e12;
  double cyc = ms * 1e-3 * 2.4e9 / (iters * (double)NACC * (waves / 1024.0));
  printf("mfma444 NACC=%2d KV=%2d blocks=%d: %.3f ms MFMA %.1f TF VALU %.1f TF sum %.1f  ~%.1f cyc/MFMA/SIMD\n", NACC, KV, blocks, ms, tf, vtf, tf + vtf, cyc);
}

int main() {
  // layout probe: find which (block,i,k)/(block,k,j)/(block,i,j) mapping reproduces host result
  std::vector<double> A(64), B(64), D(64);
  for (int i = 0; i < 64; ++i) { A[i] = 1 + 0.37 * i + (i % 5); B[i] = 2 - 0.11 * i + 3 * (i % 7); }
  double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 512));
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(layout444, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  CK(hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost));
  // hypothesis H1: lane l: block b=l>>4? or b = l&3?  try several
  for (int hyp = 0; hyp < 4; ++hyp) {
    double me = 0;
    for (int l = 0; l < 64; ++l) {
      // output lane l holds D_b[i][j]
      int b, i, j;
      if (hyp == 0) { b = l >> 4; i = (l >> 2) & 3; j = l & 3; }
      else if (hyp == 1) { b = l >> 4; j = (l >> 2) & 3; i = l & 3; }
      else if (hyp == 2) { b = l & 3; i = (l >> 4); j = (l >> 2) & 3; }
      else { b = (l >> 2) & 3; i = l >> 4; j = l & 3; }
      // operand lanes: A_b[i][k] at lane la, B_b[k][j] at lane lb under same hypothesis family
      double r = 0;
      for (int k = 0; k < 4; ++k) {
        int la, lb;
        if (hyp == 0) { la = b * 16 + k * 4 + i; lb = b * 16 + k * 4 + j; }
        else if (hyp == 1) { la = b * 16 + k * 4 + i; lb = b * 16 + k * 4 + j; }
        else if (hyp == 2) { la = k * 16 + i * 4 + b; lb = k * 16 + j * 4 + b; }
        else { la = k * 16 + b * 4 + i; lb = k * 16 + b * 4 + j; }
        r += A[la] * B[lb];
      }
      me = fmax(me, fabs(r - D[l]));
    }
    printf("4x4x4 layout hypothesis %d: max err %.3e\n", hyp, me);
  }
  double* d_out; CK(hipMalloc(&d_out, sizeof(double) * 4096 * 512));
  int iters = 4000;
  run<8, 0>(d_out, 256, iters);
  run<16, 0>(d_out, 256, iters);
  run<16, 0>(d_out, 512, iters);
  run<16, 0>(d_out, 1024, iters);
  run<16, 2>(d_out, 512, iters);
  run<16, 4>(d_out, 512, iters);
  run<16, 8>(d_out, 512, iters);
  return 0;
}
