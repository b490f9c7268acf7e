// ubench 3: HBM store patterns for a [N][640] row-major buffer of doubles in which 512 columns per row are written
// (the Psi1 block of Kaug).  Which workgroup / wave / lane mapping reaches the fill rate (~5-6 TB/s)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr long LD = 640;

// V0: the psi1 mapping: workgroup = 16 rows x 512 cols, wave w = cols 128w.., lane = 2 adjacent cols, loop over rows
__global__ void __launch_bounds__(256) v0(double* K, double val) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row0 = blockIdx.x * 16L;
  for (int r = 0; r < 16; ++r) { double2 v; v.x = val + r; v.y = val; *reinterpret_cast<double2*>(&K[(row0 + r) * LD + wave * 128 + 2 * lane]) = v; }
}
// V1: wave = one row at a time, 4 stores of 16 B per lane cover the row's 4 KB; workgroup = 4 waves x RPW rows
template <int RPW>
__global__ void __launch_bounds__(256) v1(double* K, double val) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row0 = (blockIdx.x * 4L + wave) * RPW;
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) { double2 v; v.x = val + r; v.y = val + j; *reinterpret_cast<double2*>(&K[(row0 + r) * LD + 128 * j + 2 * lane]) = v; }
}
// V2: thread = 2 columns of one row, workgroup = 1 row (256 threads x 2 cols = 512), grid = N rows  (fill-like)
__global__ void __launch_bounds__(256) v2(double* K, double val) {
  double2 v; v.x = val; v.y = val + 1;
  *reinterpret_cast<double2*>(&K[blockIdx.x * LD + 2 * threadIdx.x]) = v;
}
// V3: as V0 but 64 rows per workgroup (fewer, longer workgroups)
__global__ void __launch_bounds__(256) v3(double* K, double val) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row0 = blockIdx.x * 64L;
  for (int r = 0; r < 64; ++r) { double2 v; v.x = val + r; v.y = val; *reinterpret_cast<double2*>(&K[(row0 + r) * LD + wave * 128 + 2 * lane]) = v; }
}
// V4: V0 with 8-byte stores (one column per lane, wave = 64 columns, 8 waves per workgroup)
__global__ void __launch_bounds__(512) v4(double* K, double val) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row0 = blockIdx.x * 16L;
  for (int r = 0; r < 16; ++r) K[(row0 + r) * LD + wave * 64 + lane] = val + r;
}
// V5: V0 + the per-row record (22 doubles, wave-uniform) read by scalar loads and folded into the stored value
template <int UNROLL>
__global__ void __launch_bounds__(256) v5(double* K, const double* __restrict__ PU, double val) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long row0 = blockIdx.x * 16L;
#pragma unroll UNROLL
  for (int r = 0; r < 16; ++r) {
    const double* rec = PU + (row0 + r) * 22;
    double e = val;
#pragma unroll
    for (int q = 0; q < 22; ++q) e = fma(rec[q], (double)(lane + q), e);
    double2 v; v.x = e; v.y = e + 1.0;
    *reinterpret_cast<double2*>(&K[(row0 + r) * LD + wave * 128 + 2 * lane]) = v;
  }
}
// V6: the records of the workgroup staged in LDS by one coalesced load, read back as broadcast VGPR operands
__global__ void __launch_bounds__(256) v6(double* K, const double* __restrict__ PU, double val) {
  __shared__ double rec_s[16 * 22];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row0 = blockIdx.x * 16L;
  for (int i = threadIdx.x; i < 16 * 22; i += 256) rec_s[i] = PU[row0 * 22 + i];
  __syncthreads();
  for (int r = 0; r < 16; ++r) {
    double e = val;
#pragma unroll
    for (int q = 0; q < 22; ++q) e = fma(rec_s[r * 22 + q], (double)(lane + q), e);
    double2 v; v.x = e; v.y = e + 1.0;
    *reinterpret_cast<double2*>(&K[(row0 + r) * LD + wave * 128 + 2 * lane]) = v;
  }
}
template <typename F> float time_ms(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 5;
}
int main() {
  const long N = 1000000 / 64 * 64;
  double* K; CK(hipMalloc(&K, N * LD * 8));
  const double gb = N * 512 * 8 / 1e9;
  float t;
  t = time_ms([&] { hipLaunchKernelGGL(v0, dim3(N / 16), dim3(256), 0, 0, K, 1.0); }); printf("V0 psi1 mapping (16 rows/WG)      %.3f ms %.2f TB/s\n", t, gb / t);
  t = time_ms([&] { hipLaunchKernelGGL(v1<4>, dim3(N / 16), dim3(256), 0, 0, K, 1.0); }); printf("V1 wave=row, 4 rows/wave          %.3f ms %.2f TB/s\n", t, gb / t);
  t = time_ms([&] { hipLaunchKernelGGL(v1<16>, dim3(N / 64), dim3(256), 0, 0, K, 1.0); }); printf("V1 wave=row, 16 rows/wave         %.3f ms %.2f TB/s\n", t, gb / t);
  t = time_ms([&] { hipLaunchKernelGGL(v2, dim3(N), dim3(256), 0, 0, K, 1.0); }); printf("V2 WG = 1 row                     %.3f ms %.2f TB/s\n", t, gb / t);
  t = time_ms([&] { hipLaunchKernelGGL(v3, dim3(N / 64), dim3(256), 0, 0, K, 1.0); }); printf("V3 psi1 mapping, 64 rows/WG       %.3f ms %.2f TB/s\n", t, gb / t);
  t = time_ms([&] { hipLaunchKernelGGL(v4, dim3(N / 16), dim3(512), 0, 0, K, 1.0); }); printf("V4 8-byte stores                  %.3f ms %.2f TB/s\n", t, gb / t);
  double* PU; CK(hipMalloc(&PU, N * 22 * 8)); CK(hipMemset(PU, 0, N * 22 * 8));
  t = time_ms([&] { hipLaunchKernelGGL(v5<1>, dim3(N / 16), dim3(256), 0, 0, K, PU, 1.0); }); printf("V5 + scalar record, unroll 1      %.3f ms %.2f TB/s\n", t, gb / t);
  t = time_ms([&] { hipLaunchKernelGGL(v5<2>, dim3(N / 16), dim3(256), 0, 0, K, PU, 1.0); }); printf("V5 + scalar record, unroll 2      %.3f ms %.2f TB/s\n", t, gb / t);
  t = time_ms([&] { hipLaunchKernelGGL(v5<4>, dim3(N / 16), dim3(256), 0, 0, K, PU, 1.0); }); printf("V5 + scalar record, unroll 4      %.3f ms %.2f TB/s\n", t, gb / t);
  t = time_ms([&] { hipLaunchKernelGGL(v6, dim3(N / 16), dim3(256), 0, 0, K, PU, 1.0); }); printf("V6 records through LDS            %.3f ms %.2f TB/s\n", t, gb / t);
  CK(hipFree(K));
  return 0;
}
