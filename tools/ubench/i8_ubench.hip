// Ozaki gate, step 2: what the int8 matrix core and the f64 <-> int8-slice conversions cost on gfx950.
//   (1) lane map check of v_mfma_i32_16x16x64_i8: a 16x64 by 64x16 product with random int8 operands against the host, operands loaded with
//       the assumption "lane l holds row/col l & 15 and the 16 consecutive k bytes 16 (l >> 4) ..", C/D col = l & 15, row = 4 (l >> 4) + reg;
//   (2) sustained rate of the instruction (register operands, 8 independent accumulators per wave, 1 .. 4 waves per SIMD);
//   (3) slicing: f64 -> s signed 7-bit digits (int8), values per second; (4) recombination: int32 accumulator tiles -> f64 (scale + add).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/i8_ubench.hip -o tools/ubench/i8_ubench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void map_check(const int8_t* A /*[16][64]*/, const int8_t* B /*[64][16] stored [col][k]*/, int* D /*[16][16]*/) {
  const int l = threadIdx.x, r = l & 15, kg = l >> 4;
  v4i a = *(const v4i*)(A + r * 64 + 16 * kg);
  v4i b = *(const v4i*)(B + r * 64 + 16 * kg);
  v4i c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(4 * kg + i) * 16 + r] = c[i];
}

typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void map_check32(const int8_t* A /*[32][32]*/, const int8_t* B /*[32 cols][32 k]*/, int* D /*[32][32]*/) {
  const int l = threadIdx.x, r = l & 31, kg = l >> 5;
  v4i a = *(const v4i*)(A + r * 32 + 16 * kg);
  v4i b = *(const v4i*)(B + r * 32 + 16 * kg);
  v16i c;
  for (int i = 0; i < 16; ++i) c[i] = 0;
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * kg) * 32 + r] = c[i];
}

template <int NACC>
__global__ void __launch_bounds__(256) rate32_kernel(int iters, int* out) {
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
  v16i c[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[i], 0, 0, 0);
  }
  int s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += c[i][j];
  if (s == 0x7fffffff) out[0] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) rate_kernel(int iters, int* out) {
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
  v4i c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = v4i{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
  }
  int s = 0;
  for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  if (s == 0x7fffffff) out[0] = s;
}

// f64 in [-0.5, 0.5] * 2^e -> S signed digits d_j in [-64, 64], x = 2^e sum_j d_j 128^-j; one lane = one column, 16 consecutive rows packed per store
template <int S>
__global__ void __launch_bounds__(256) slice_kernel(const double* __restrict__ X, long rows, int cols, double scale, int8_t* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  for (long r0 = blockIdx.y * 16L; r0 < rows; r0 += gridDim.y * 16L) {
    unsigned pk[S][4];
#pragma unroll
    for (int j = 0; j < S; ++j) { pk[j][0] = pk[j][1] = pk[j][2] = pk[j][3] = 0u; }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      double t = X[(r0 + i) * cols + c] * scale;
#pragma unroll
      for (int j = 0; j < S; ++j) {
        t *= 128.0;
        const double d = __builtin_rint(t);
        t -= d;
        pk[j][i >> 2] |= ((unsigned)(int)d & 0xffu) << (8 * (i & 3));
      }
    }
#pragma unroll
    for (int j = 0; j < S; ++j) {
      uint4 v = {pk[j][0], pk[j][1], pk[j][2], pk[j][3]};
      *(uint4*)(out + (((long)j * (rows / 16) + r0 / 16) * cols + c) * 16) = v;
    }
  }
}

int main(int argc, char** argv) {
  // (1)
  std::vector<int8_t> hA(16 * 64), hB(16 * 64);
  srand(3);
  for (auto& v : hA) v = (int8_t)(rand() % 129 - 64);
  for (auto& v : hB) v = (int8_t)(rand() % 129 - 64);
  int8_t *dA, *dB; int* dD;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
  hipMemcpy(dA, hA.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(map_check, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  std::vector<int> hD(256);
  hipMemcpy(hD.data(), dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0, badT = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    int s = 0;
    for (int k = 0; k < 64; ++k) s += (int)hA[i * 64 + k] * (int)hB[j * 64 + k];
    if (hD[i * 16 + j] != s) ++bad;
    if (hD[j * 16 + i] != s) ++badT;
  }
  printf("lane map: D[row = 4 (l >> 4) + reg][col = l & 15] = sum_k A[row][k] B[k][col]: %d mismatches (transposed reading: %d)\n", bad, badT);
  {
    std::vector<int8_t> gA(32 * 32), gB(32 * 32);
    for (auto& v : gA) v = (int8_t)(rand() % 129 - 64);
    for (auto& v : gB) v = (int8_t)(rand() % 129 - 64);
    int* dD2; hipMalloc(&dD2, 4096);
    hipMemcpy(dA, gA.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, gB.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(map_check32, dim3(1), dim3(64), 0, 0, dA, dB, dD2);
    std::vector<int> gD(1024);
    hipMemcpy(gD.data(), dD2, 4096, hipMemcpyDeviceToHost);
    int bad2 = 0, bad2T = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      int s = 0;
      for (int k = 0; k < 32; ++k) s += (int)gA[i * 32 + k] * (int)gB[j * 32 + k];
      if (gD[i * 32 + j] != s) ++bad2;
      if (gD[j * 32 + i] != s) ++bad2T;
    }
    printf("32x32x32 lane map: D[row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)][col = l & 31]: %d mismatches (transposed reading: %d)\n", bad2, bad2T);
  }
  // (2)
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int* dout; hipMalloc(&dout, 64);
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int iters = 20000, blocks = 256 * wps;
    hipLaunchKernelGGL((rate_kernel<8>), dim3(blocks), dim3(256), 0, 0, 100, dout);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<8>), dim3(blocks), dim3(256), 0, 0, iters, dout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = 2.0 * 16 * 16 * 64 * 8.0 * iters * blocks * 4;
    printf("v_mfma_i32_16x16x64_i8, %d wave(s) per SIMD, 8 accumulators: %.0f TOPS (%.1f cycles per instruction per SIMD at 2.4 GHz)\n", wps, ops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (8.0 * iters * wps));
  }
  for (int wps = 1; wps <= 2; wps *= 2) {
    const int iters = 20000, blocks = 256 * wps;
    hipLaunchKernelGGL((rate32_kernel<4>), dim3(blocks), dim3(256), 0, 0, 100, dout);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate32_kernel<4>), dim3(blocks), dim3(256), 0, 0, iters, dout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = 2.0 * 32 * 32 * 32 * 4.0 * iters * blocks * 4;
    printf("v_mfma_i32_32x32x32_i8, %d wave(s) per SIMD, 4 accumulators: %.0f TOPS (%.1f cycles per instruction per SIMD at 2.4 GHz)\n", wps, ops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (4.0 * iters * wps));
  }
  // (2b) r05: the issue rate as a function of the distance between two MFMAs on the SAME accumulator (NACC accumulators round robin: distance NACC).
  // The int8 phase 2 (csrc/p2i8.hip) adds seven digit products into its highest-order accumulator per k-step: how far apart must they be?
  {
    auto run = [&](auto kern, int nacc, int wps) {
      const int iters = 20000, blocks = 256 * wps;
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, 100, dout);
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, dout);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("v_mfma_i32_32x32x32_i8, %d wave(s) per SIMD, same-accumulator distance %d: %.1f cycles per instruction per SIMD at 2.4 GHz\n", wps, nacc,
             ms * 1e-3 * 2.4e9 / ((double)nacc * iters * wps));
    };
    // sustained: the same kernel for ~2, 10 and 50 ms -- does the rate hold (power / clock management)?
    for (int iters : {40000, 200000, 1000000}) {
      hipLaunchKernelGGL(rate32_kernel<7>, dim3(512), dim3(256), 0, 0, 100, dout);
      hipEventRecord(e0);
      hipLaunchKernelGGL(rate32_kernel<7>, dim3(512), dim3(256), 0, 0, iters, dout);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("v_mfma_i32_32x32x32_i8 sustained, 2 waves per SIMD, 7 accumulators, %.1f ms: %.0f TOPS (%.1f cycles per instruction per SIMD at 2.4 GHz)\n", ms,
             2.0 * 32 * 32 * 32 * 7.0 * iters * 512 * 4 / ms / 1e9, ms * 1e-3 * 2.4e9 / (7.0 * iters * 2));
    }
    for (int wps = 1; wps <= 2; ++wps) {
      run(rate32_kernel<1>, 1, wps); run(rate32_kernel<2>, 2, wps); run(rate32_kernel<3>, 3, wps); run(rate32_kernel<4>, 4, wps); run(rate32_kernel<7>, 7, wps);
    }
  }
  // (3)
  const long rows = 1 << 20; const int cols = 512;
  double* dX; hipMalloc(&dX, rows * cols * 8);
  std::vector<double> hX((size_t)rows * cols);
  for (auto& v : hX) v = (rand() / (double)RAND_MAX - 0.5);
  hipMemcpy(dX, hX.data(), hX.size() * 8, hipMemcpyHostToDevice);
  int8_t* dS; hipMalloc(&dS, rows * cols * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((slice_kernel<5>), dim3(cols / 256, 2048), dim3(256), 0, 0, dX, rows, cols, 1.0, dS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("slicing f64 -> 5 x int8 (2^20 x 512 values, 4.3 GB in, 2.7 GB out): %.3f ms = %.2f TB/s of input\n", ms, rows * cols * 8.0 / ms / 1e9);
    hipEventRecord(e0);
    hipLaunchKernelGGL((slice_kernel<7>), dim3(cols / 256, 2048), dim3(256), 0, 0, dX, rows, cols, 1.0, dS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("slicing f64 -> 7 x int8: %.3f ms\n", ms);
  }
  // check the digits of a few values
  std::vector<int8_t> hS((size_t)rows * cols * 7);
  hipMemcpy(hS.data(), dS, hS.size(), hipMemcpyDeviceToHost);
  double worst = 0;
  for (long r = 0; r < 64; ++r) for (int c = 0; c < 512; c += 37) {
    double x = 0, p = 1;
    for (int j = 0; j < 7; ++j) { p /= 128.0; x += p * hS[(((long)j * (rows / 16) + r / 16) * cols + c) * 16 + (r & 15)]; }
    const double e = fabs(x - hX[r * cols + c]); if (e > worst) worst = e;
  }
  printf("7-slice reconstruction error (max over samples): %.2e (2^-50 = %.2e)\n", worst, 1.0 / (1LL << 50));
  return 0;
}
