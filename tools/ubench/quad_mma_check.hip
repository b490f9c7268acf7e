// Lane-map check of wave_rows_times_features (quad_mma.h) against a host sum.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/quad_mma_check.hip -o tools/ubench/quad_mma_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../gparml_amd/csrc/quad_mma.h"
constexpr int NQ = 3;
__global__ void k(const double* T, const double* F, double* out, double* outT) {
  const int lane = threadIdx.x;
  double t[4], zb[4][NQ], acc[NQ];
  for (int i = 0; i < 4; ++i) t[i] = T[i * 64 + lane];
  const int lq = lane & 3, lb = (lane >> 2) & 3, lk = lane >> 4;
  for (int v = 0; v < 4; ++v) for (int qq = 0; qq < NQ; ++qq) zb[v][qq] = F[(16 * lk + 4 * lb + v) * (4 * NQ) + 4 * qq + lq];
  gp::wave_rows_times_features<NQ>(t, zb, acc);
  for (int qq = 0; qq < NQ; ++qq) out[lane * NQ + qq] = acc[qq];
  // also expose the transposed registers
  outT[lane] = gp::quad_xchg<0xB1>(t[0]);
  outT[64 + lane] = gp::quad_xchg<0x4E>(t[0]);
  outT[128 + lane] = gp::row_ror<4>(t[0]);
  outT[192 + lane] = gp::row_ror<8>(t[0]);
}
int main() {
  std::vector<double> T(256), F(64 * 4 * NQ), out(64 * NQ), outT(384);
  for (int i = 0; i < 256; ++i) T[i] = std::sin(0.37 * i) + 0.01 * i;
  for (size_t i = 0; i < F.size(); ++i) F[i] = std::cos(0.11 * i);
  double *dT, *dF, *dO, *dOT;
  hipMalloc(&dT, 256 * 8); hipMalloc(&dF, F.size() * 8); hipMalloc(&dO, out.size() * 8); hipMalloc(&dOT, 384 * 8);
  hipMemcpy(dT, T.data(), 256 * 8, hipMemcpyHostToDevice); hipMemcpy(dF, F.data(), F.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dT, dF, dO, dOT);
  hipMemcpy(out.data(), dO, out.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(outT.data(), dOT, 384 * 8, hipMemcpyDeviceToHost);
  double err = 0;
  for (int lane = 0; lane < 64; ++lane) for (int qq = 0; qq < NQ; ++qq) {
    const int i = lane >> 4, j = lane & 3;
    double s = 0; for (int c = 0; c < 64; ++c) s += T[i * 64 + c] * F[c * 4 * NQ + 4 * qq + j];
    err = std::fmax(err, std::fabs(s - out[lane * NQ + qq]));
    if (lane < 20 && qq == 0) printf("lane %2d: got % .6f want % .6f\n", lane, out[lane * NQ + qq], s);
  }
  printf("max err %.3e\n", err);
  printf("quad_perm B1 lanes 0..7 (expect source lanes 1 0 3 2 5 4 7 6):"); for (int l = 0; l < 8; ++l) for (int c = 0; c < 64; ++c) if (outT[l] == T[c]) printf(" %d", c); printf("\n");
  printf("quad_perm 4E lanes 0..7 (expect 2 3 0 1 6 7 4 5):"); for (int l = 0; l < 8; ++l) for (int c = 0; c < 64; ++c) if (outT[64 + l] == T[c]) printf(" %d", c); printf("\n");
  printf("row_ror 4 lanes 0..19:"); for (int l = 0; l < 20; ++l) for (int c = 0; c < 64; ++c) if (outT[128 + l] == T[c]) printf(" %d", c); printf("\n");
  printf("row_ror 8 lanes 0..19:"); for (int l = 0; l < 20; ++l) for (int c = 0; c < 64; ++c) if (outT[192 + l] == T[c]) printf(" %d", c); printf("\n");
  return 0;
}
