// Compatibility outputs: the reference's per-statistic derivative tensors (the 12-key dictionary of
// local_MapReduce.py:227-240) and the contraction methods that consume them (partial_terms.py:207-240, 286-299,
// 322-333).  They exist so that an unmodified parallel_GPLVM.calculate_global_derivatives (:336-369) can run against
// this backend; the fast path never materialises them (gp_phase2 contracts on the device instead).  Plain one-thread-
// per-output kernels: the shapes are the reference's (M,Q,M)/(Q,M,M)/(M,Q,D)/(Q,M,D) layouts.
#include "gp_common.h"
#include <algorithm>

namespace gp {

// psi2 of every point: out[n][j][m]   (partial_terms.py:45-48, kernel_exp.py:143-146)
__global__ void __launch_bounds__(256) psi2_points_kernel(const double* __restrict__ Kaug, long ld, const double* __restrict__ LE, int Mp, bool le_il,
                                                           const double* __restrict__ Vn, const double* __restrict__ DZ2, long N, int M, int Q,
                                                           int regimeA, double* __restrict__ out) {
  const long total = N * (long)M * M;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int m2 = (int)(i % M);
    const long r = i / M;
    const int m1 = (int)(r % M);
    const long n = r / M;
    double v;
    if (regimeA) {
      v = Kaug[n * ld + m1] * Kaug[n * ld + m2];
    } else {
      double e = LE[le_index(le_il, n, m1, Mp)] + LE[le_index(le_il, n, m2, Mp)];      // (the layout of LE: csrc/psi2.hip)
      for (int q = 0; q < Q; ++q) e = fma(Vn[n * Q + q], DZ2[((long)m1 * M + m2) * Q + q], e);
      v = exp(e);
    }
    out[i] = v;
  }
}

// which: 0 dKmm_dZ (M,Q,M)  1 dKmm_dalpha (Q,M,M)                                   partial_terms.py:146-160, 247-254
__global__ void __launch_bounds__(256) dkmm_kernel(const double* __restrict__ Kmm, int Mp, const double* __restrict__ Z,
                                                    const double* __restrict__ alpha, int M, int Q, int which, double* __restrict__ out) {
  const long total = (long)M * Q * M;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    if (which == 0) {
      const int m2 = (int)(i % M); const int k = (int)((i / M) % Q); const int j = (int)(i / ((long)M * Q));
      out[i] = Kmm[(long)j * Mp + m2] * (-alpha[k]) * (Z[(long)j * Q + k] - Z[(long)m2 * Q + k]);
    } else {
      const int m2 = (int)(i % M); const int m1 = (int)((i / M) % M); const int q = (int)(i / ((long)M * M));
      const double d = Z[(long)m1 * Q + q] - Z[(long)m2 * Q + q];
      out[i] = -0.5 * Kmm[(long)m1 * Mp + m2] * d * d;
    }
  }
}

// which: 0 dexp_K_miY_dZ (M,Q,D)  1 dexp_K_miY_dalpha (Q,M,D)                       partial_terms.py:162-188, 256-271
__global__ void __launch_bounds__(256) dpsi1y_kernel(const double* __restrict__ Kaug, long ld, int Mp, const double* __restrict__ mu,
                                                      const double* __restrict__ S, const double* __restrict__ Z,
                                                      const double* __restrict__ alpha, long N, int M, int Q, int D, int which,
                                                      double* __restrict__ out) {
  const long total = (long)M * Q * D;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    int j, q, d;
    if (which == 0) { d = (int)(i % D); q = (int)((i / D) % Q); j = (int)(i / ((long)D * Q)); }
    else { d = (int)(i % D); j = (int)((i / D) % M); q = (int)(i / ((long)D * M)); }
    const double a = alpha[q], z = Z[(long)j * Q + q];
    double acc = 0.0;
    for (long n = 0; n < N; ++n) {
      const double s = S[n * Q + q], m = mu[n * Q + q];
      const double d1 = a * s + 1.0;
      const double p = Kaug[n * ld + j] * Kaug[n * ld + Mp + d];
      if (which == 0) acc += p * a * (m - z) / d1;
      else { const double t = (m - z) / d1; acc += -0.5 * p * (t * t + s / d1); }
    }
    out[i] = acc;
  }
}

// which: 0 dexp_K_mi_K_im_dZ (M,Q,M)  1 dexp_K_mi_K_im_dalpha (Q,M,M)               partial_terms.py:190-205, 273-284
__global__ void __launch_bounds__(256) dpsi2_kernel(const double* __restrict__ P2, const double* __restrict__ mu, const double* __restrict__ S,
                                                     const double* __restrict__ Z, const double* __restrict__ alpha, long N, int M, int Q,
                                                     int which, double* __restrict__ out) {
  const long total = (long)M * Q * M;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    int j, q, m2;
    if (which == 0) { m2 = (int)(i % M); q = (int)((i / M) % Q); j = (int)(i / ((long)M * Q)); }
    else { m2 = (int)(i % M); j = (int)((i / M) % M); q = (int)(i / ((long)M * M)); }
    const double a = alpha[q], zj = Z[(long)j * Q + q], zm = Z[(long)m2 * Q + q];
    double acc = 0.0;
    for (long n = 0; n < N; ++n) {
      const double s = S[n * Q + q], m = mu[n * Q + q];
      const double d2 = 2.0 * a * s + 1.0;
      const double p = P2[(n * M + j) * M + m2];
      if (which == 0) acc += p * (-0.5 * a * (zj - zm) + 0.5 * a * (2.0 * m - zj - zm) / d2);
      else { const double t = (2.0 * m - zj - zm) / d2; acc += p * (-0.25 * (zj - zm) * (zj - zm) - 0.25 * t * t - s / d2); }
    }
    out[i] = acc;
  }
}

// grad_Z from its parts (partial_terms.py:207-240): out[j,k] = sum_m (A+A^T)[j,m] a3[j,k,m] + sum_d B[j,d] b3[j,k,d] + 2 sum_m C[j,m] c3[j,k,m]
__global__ void __launch_bounds__(256) gradz_parts_kernel(const double* A, const double* a3, const double* B, const double* b3, const double* C,
                                                           const double* c3, int M, int Q, int D, double* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M * Q) return;
  const int j = i / Q, k = i % Q;
  double s = 0.0;
  for (int m = 0; m < M; ++m) s += (A[(long)j * M + m] + A[(long)m * M + j]) * a3[((long)j * Q + k) * M + m] * ((m == j) ? 0.5 : 1.0);
  for (int d = 0; d < D; ++d) s += B[(long)j * D + d] * b3[((long)j * Q + k) * D + d];
  for (int m = 0; m < M; ++m) s += 2.0 * C[(long)j * M + m] * c3[((long)j * Q + k) * M + m];
  out[i] = s;
}

// grad_alpha from its parts (partial_terms.py:286-299): out[q] = <A, a3[q]> + <B, b3[q]> + <C, c3[q]>
__global__ void __launch_bounds__(256) gradalpha_parts_kernel(const double* A, const double* a3, const double* B, const double* b3, const double* C,
                                                               const double* c3, int M, int Q, int D, double* out) {
  __shared__ double red[256];
  const int q = blockIdx.x;
  double s = 0.0;
  for (long i = threadIdx.x; i < (long)M * M; i += 256) s += A[i] * a3[(long)q * M * M + i] + C[i] * c3[(long)q * M * M + i];
  for (long i = threadIdx.x; i < (long)M * D; i += 256) s += B[i] * b3[(long)q * M * D + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) out[q] = red[0];
}

static int grid_for(long n) { return (int)std::max<long>(1, std::min<long>((n + 255) / 256, 16384)); }

// fills a dense device buffer with the requested reference-shaped array; caller frees *out
int compat_build(gp_ctx* c, int which, double** out, long* count) {
  const long N = c->N, M = c->M, Q = c->Q, D = c->D;
  long n = 0;
  switch (which) {
    case GP_ARR_PSI2_POINTS: n = N * M * M; break;
    case GP_ARR_DKMM_DZ: case GP_ARR_DPSI2_DZ: case GP_ARR_DKMM_DALPHA: case GP_ARR_DPSI2_DALPHA: n = M * Q * M; break;
    case GP_ARR_DPSI1TY_DZ: case GP_ARR_DPSI1TY_DALPHA: n = M * Q * D; break;
    default: return fail(c, GP_ERR_BAD_ARG, "compat_build: unknown array %d", which);
  }
  const bool needs_data = !(which == GP_ARR_DKMM_DZ || which == GP_ARR_DKMM_DALPHA);
  if (needs_data && (!c->have_data || c->state < 1)) return fail(c, GP_ERR_STATE, "array %d needs set_data / phase 1 first", which);
  if (!needs_data && c->state < 2) return fail(c, GP_ERR_STATE, "array %d needs the global step (Kmm) first", which);
  const bool needs_p2 = (which == GP_ARR_PSI2_POINTS || which == GP_ARR_DPSI2_DZ || which == GP_ARR_DPSI2_DALPHA);
  if (needs_p2 && N * M * M > (1L << 28)) return fail(c, GP_ERR_UNSUPPORTED, "per-point psi2 tensor (N,M,M) too large for compat mode (%ld doubles)", N * M * M);
  double* buf = nullptr;
  GP_TRY_RC(dalloc_bytes(c, (void**)&buf, (size_t)std::max<long>(n, 1) * 8, DA_RAW));
  double* p2 = nullptr;
  if (needs_p2) {
    if (which == GP_ARR_PSI2_POINTS) p2 = buf; else GP_TRY_RC(dalloc_bytes(c, (void**)&p2, (size_t)(N * M * M) * 8, DA_RAW));
    if (!c->regime_A) { const int rc = run_dz2(c); if (rc != GP_OK) return rc; }
    hipLaunchKernelGGL(psi2_points_kernel, dim3(grid_for(N * M * M)), dim3(256), 0, c->stream, c->Kaug, (long)c->LDK, c->LE, c->Mp, !b_generic(c) && le_interleaved(c->QB), c->Vn, c->DZ2,
                       N, (int)M, (int)Q, c->regime_A ? 1 : 0, p2);
  }
  switch (which) {
    case GP_ARR_PSI2_POINTS: break;
    case GP_ARR_DKMM_DZ: case GP_ARR_DKMM_DALPHA:
      hipLaunchKernelGGL(dkmm_kernel, dim3(grid_for(n)), dim3(256), 0, c->stream, c->KmmKeep, c->Mp, c->Z, c->alpha, (int)M, (int)Q,
                         which == GP_ARR_DKMM_DZ ? 0 : 1, buf);
      break;
    case GP_ARR_DPSI1TY_DZ: case GP_ARR_DPSI1TY_DALPHA:
      hipLaunchKernelGGL(dpsi1y_kernel, dim3(grid_for(n)), dim3(256), 0, c->stream, c->Kaug, (long)c->LDK, c->Mp, c->mu, c->S, c->Z, c->alpha, N,
                         (int)M, (int)Q, (int)D, which == GP_ARR_DPSI1TY_DZ ? 0 : 1, buf);
      break;
    case GP_ARR_DPSI2_DZ: case GP_ARR_DPSI2_DALPHA:
      hipLaunchKernelGGL(dpsi2_kernel, dim3(grid_for(n)), dim3(256), 0, c->stream, p2, c->mu, c->S, c->Z, c->alpha, N, (int)M, (int)Q,
                         which == GP_ARR_DPSI2_DZ ? 0 : 1, buf);
      break;
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (p2 && p2 != buf) (void)hipFree(p2);
  if (e != hipSuccess) { (void)hipFree(buf); return fail(c, GP_ERR_HIP, "compat kernel failed: %s", hipGetErrorString(e)); }
  *out = buf;
  *count = n;
  return GP_OK;
}

}  // namespace gp

using namespace gp;

static int up(gp_ctx* c, const double* h, long n, double** d) {
  GP_TRY_RC(dalloc_bytes(c, (void**)d, (size_t)std::max<long>(n, 1) * 8, DA_RAW));
  GP_HIP(c, hipMemcpyAsync(*d, h, n * 8, hipMemcpyHostToDevice, c->stream));
  return GP_OK;
}

extern "C" int gp_grad_from_parts(gp_ctx* c, int which, const double* dF_dKmm, const double* dKmm_dX, const double* dF_dC, const double* dC_dX,
                                  const double* dF_dPsi2, const double* dPsi2_dX, double* out) {
  if (!c || !dF_dKmm || !dKmm_dX || !dF_dC || !dC_dX || !dF_dPsi2 || !dPsi2_dX || !out) return GP_ERR_BAD_ARG;
  if (which != 0 && which != 1) return fail(c, GP_ERR_BAD_ARG, "gp_grad_from_parts: which must be 0 (Z) or 1 (alpha)");
  GP_HIP(c, hipSetDevice(c->device));
  const long M = c->M, Q = c->Q, D = c->D;
  double *A = nullptr, *a3 = nullptr, *B = nullptr, *b3 = nullptr, *C = nullptr, *c3 = nullptr, *o = nullptr;
  int rc = up(c, dF_dKmm, M * M, &A);
  if (rc == GP_OK) rc = up(c, dKmm_dX, M * Q * M, &a3);
  if (rc == GP_OK) rc = up(c, dF_dC, M * D, &B);
  if (rc == GP_OK) rc = up(c, dC_dX, M * Q * D, &b3);
  if (rc == GP_OK) rc = up(c, dF_dPsi2, M * M, &C);
  if (rc == GP_OK) rc = up(c, dPsi2_dX, M * Q * M, &c3);
  const long no = which == 0 ? M * Q : Q;
  if (rc == GP_OK) rc = dalloc_bytes(c, (void**)&o, (size_t)no * 8, DA_RAW);
  if (rc == GP_OK) {
    if (which == 0) hipLaunchKernelGGL(gradz_parts_kernel, dim3((unsigned)((M * Q + 255) / 256)), dim3(256), 0, c->stream, A, a3, B, b3, C, c3, (int)M, (int)Q, (int)D, o);
    else hipLaunchKernelGGL(gradalpha_parts_kernel, dim3((unsigned)Q), dim3(256), 0, c->stream, A, a3, B, b3, C, c3, (int)M, (int)Q, (int)D, o);
    hipError_t e = hipMemcpyAsync(out, o, no * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(c, GP_ERR_HIP, "gp_grad_from_parts: %s", hipGetErrorString(e));
  }
  for (double* p : {A, a3, B, b3, C, c3, o}) if (p) (void)hipFree(p);
  return rc;
}
