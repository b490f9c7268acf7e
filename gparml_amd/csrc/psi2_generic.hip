// Regime B (free embeddings) for latent widths beyond the compiled tables (Q >= 64): plain kernels, any Q.
//   reference: kernel_exp.py:126-148 (psi2_n has no limit on Q), partial_terms.py:190-205, 273-284, 388-394, 421-427.
// The tuned kernels of psi2.hip / psi2_tile.hip keep Q-long vectors in registers or in compile-time LDS images; this fallback materialises
//   psi2_n[m, m'] = exp(LEA[n, m] + LEA[n, m'] + sum_q (-2 V_nq) z_mq z_m'q)                 (the factorised form of psi2.hip's header)
// for a chunk of P points in HBM ([P][M][M] doubles, <= 64 MB) and runs both phases on it with one thread per output element:
//   phase 1   Psi2 += sum_p psi2_p
//   phase 2   T_p = Bbar o psi2_p;  r_p = T_p 1,  t_p = T_p Z;  per-point sums [sr, zr_q, z2r_q, zt_q] -> pp (finished by psi2_points_finish_kernel);
//             grad_Z[m, q] += sum_p -alpha_q (z_mq r - t) + w_pq (2 mu_pq r - z_mq r - t)
// No symmetry is used and every operand comes from memory: correct for every Q, M and N, at a fraction of the tuned kernels' rate (a latent space
// that wide is outside BASELINE.json's configurations; the reference itself needs an (N, M, M, Q) tensor for it, partial_terms.py:273).
#include "gp_common.h"
#include "fexp.h"
#include <algorithm>

namespace gp {

// LE and LEA [Np][Mp] as b_le_kernel writes them (padded entries = kPadLog), Q at run time
__global__ void __launch_bounds__(256) b_le_generic_kernel(const double* __restrict__ MUP, const double* __restrict__ WP, const double* __restrict__ V2P,
                                                            const double* __restrict__ lnc2h, const double* __restrict__ ZP, long N, long Np, int M,
                                                            int Mp, int Q, double* __restrict__ LE, double* __restrict__ LEA) {
  const long total = Np * Mp;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const long n = e / Mp;
    const int m = (int)(e - n * Mp);
    double le = kPadLog, lea = kPadLog;
    if (n < N && m < M) {
      double s = 0.0, t = 0.0;
      for (int q = 0; q < Q; ++q) {
        const double z = ZP[(long)m * Q + q], d = MUP[n * Q + q] - z;
        s = fma(WP[n * Q + q] * d, d, s);
        t = fma(V2P[n * Q + q] * z, z, t);
      }
      le = lnc2h[n] - 0.5 * s;
      lea = le - 0.5 * t;                        // V = -V2P / 2
    }
    LE[e] = le;
    LEA[e] = lea;
  }
}

// T[p][m][m'] = (Bbar ? Bbar[m][m'] : 1) * psi2_(n0 + p)[m][m']
__global__ void __launch_bounds__(256) psi2n_generic_kernel(const double* __restrict__ LEA, const double* __restrict__ V2P, const double* __restrict__ ZP,
                                                             const double* __restrict__ Bbar, long n0, long cnt, int M, int Mp, int Q,
                                                             double* __restrict__ T) {
  const long mm = (long)M * M, total = cnt * mm;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const long p = e / mm, r = e - p * mm, n = n0 + p;
    const int m = (int)(r / M), m2 = (int)(r - (long)m * M);
    const double* v = V2P + n * Q;
    const double* za = ZP + (long)m * Q;
    const double* zb = ZP + (long)m2 * Q;
    double s = LEA[n * Mp + m] + LEA[n * Mp + m2];
    for (int q = 0; q < Q; ++q) s = fma(v[q] * za[q], zb[q], s);
    const double x = fexp(s);
    T[e] = Bbar ? Bbar[(long)m * Mp + m2] * x : x;
  }
}

__global__ void __launch_bounds__(256) psi2_sum_generic_kernel(const double* __restrict__ T, long cnt, int M, int Mp, int first, double* __restrict__ Psi2) {
  const long mm = (long)M * M;
  for (long r = blockIdx.x * 256L + threadIdx.x; r < mm; r += (long)gridDim.x * 256L) {
    const int m = (int)(r / M), m2 = (int)(r - (long)m * M);
    double s = 0.0;
    for (long p = 0; p < cnt; ++p) s += T[p * mm + r];
    double* dst = Psi2 + (long)m * Mp + m2;
    *dst = (first ? 0.0 : *dst) + s;
  }
}

// rt[p][m][q] = sum_m' T[p][m][m'] z_m'q (q < Q),  rt[p][m][Q] = sum_m' T[p][m][m']
__global__ void __launch_bounds__(256) psi2_rt_generic_kernel(const double* __restrict__ T, const double* __restrict__ ZP, long cnt, int M, int Q,
                                                               double* __restrict__ rt) {
  const long total = cnt * M * (Q + 1);
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const long pm = e / (Q + 1);
    const int q = (int)(e - pm * (Q + 1));
    const double* row = T + pm * M;
    double s = 0.0;
    if (q < Q) for (int m2 = 0; m2 < M; ++m2) s = fma(row[m2], ZP[(long)m2 * Q + q], s);
    else for (int m2 = 0; m2 < M; ++m2) s += row[m2];
    rt[e] = s;
  }
}

// pp[i][n0 + p]: i = 0 sum_m r | 1 + q sum_m z_mq r_m | 1 + Q + q sum_m z_mq^2 r_m | 1 + 2 Q + q sum_m z_mq t_mq
__global__ void __launch_bounds__(256) psi2_pp_generic_kernel(const double* __restrict__ rt, const double* __restrict__ ZP, long n0, long cnt, int M, int Q,
                                                               long Np, double* __restrict__ pp) {
  const int PW = 3 * Q + 1;
  const long total = cnt * PW;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const long p = e / PW;
    const int i = (int)(e - p * PW);
    const double* base = rt + p * M * (Q + 1);
    double s = 0.0;
    if (i == 0) {
      for (int m = 0; m < M; ++m) s += base[(long)m * (Q + 1) + Q];
    } else {
      const int kind = (i - 1) / Q, q = (i - 1) - kind * Q;
      for (int m = 0; m < M; ++m) {
        const double z = ZP[(long)m * Q + q], r = base[(long)m * (Q + 1) + Q], t = base[(long)m * (Q + 1) + q];
        s += kind == 0 ? z * r : (kind == 1 ? z * z * r : z * t);
      }
    }
    pp[(long)i * Np + n0 + p] = s;
  }
}

// grads[m][q] += sum_p -alpha_q (z_mq r - t) + w_pq (2 mu_pq r - z_mq r - t): one thread per element, chunks in order -> bit-identical from run to run
__global__ void __launch_bounds__(256) psi2_gz_generic_kernel(const double* __restrict__ rt, const double* __restrict__ ZP, const double* __restrict__ WP,
                                                               const double* __restrict__ MUP, const double* __restrict__ alpha, long n0, long cnt,
                                                               int M, int Q, double* __restrict__ grads) {
  const long total = (long)M * Q;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int m = (int)(e / Q), q = (int)(e - (long)m * Q);
    const double z = ZP[e], al = alpha[q];
    double g = 0.0;
    for (long p = 0; p < cnt; ++p) {
      const double* b = rt + (p * M + m) * (Q + 1);
      const double r = b[Q], t = b[q], w = WP[(n0 + p) * Q + q], mu = MUP[(n0 + p) * Q + q];
      g += -al * (z * r - t) + w * (2.0 * mu * r - z * r - t);
    }
    grads[e] += g;
  }
}

static unsigned grid_of(long n) { return (unsigned)std::max<long>(1, std::min<long>((n + 255) / 256, 16384)); }

bool b_generic(const gp_ctx* c) { return c->Q >= 64; }

// the chunk buffers: T [P][M][M] (<= 64 MB, at least one point) and rt [P][M][Q + 1]
static int ensure_generic(gp_ctx* c) {
  if (c->gen_T) return GP_OK;
  const long mm = (long)c->M * c->M;
  c->gen_P = std::max<long>(1, std::min<long>(std::min<long>(c->N, 4096), (8L << 20) / std::max<long>(mm, 1)));
  GP_TRY_RC(dalloc_bytes(c, (void**)&c->gen_T, (size_t)c->gen_P * mm * sizeof(double), DA_RAW));
  GP_TRY_RC(dalloc_bytes(c, (void**)&c->gen_rt, (size_t)c->gen_P * c->M * (c->Q + 1) * sizeof(double), DA_RAW));
  return GP_OK;
}

int run_le_generic(gp_ctx* c) {
  hipLaunchKernelGGL(b_le_generic_kernel, dim3(grid_of(c->Np * c->Mp)), dim3(256), 0, c->stream, (const double*)c->MUP, (const double*)c->WP,
                     (const double*)c->V2P, (const double*)c->lnc2h, (const double*)c->ZP, (long)c->N, (long)c->Np, c->M, c->Mp, c->Q, c->LE, c->LET);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

int run_phase1_b_generic(gp_ctx* c) {
  GP_TRY_RC(ensure_generic(c));
  const long mm = (long)c->M * c->M;
  GP_EV(c, 10);
  for (long n0 = 0; n0 < c->N; n0 += c->gen_P) {
    const long cnt = std::min<long>(c->gen_P, c->N - n0);
    hipLaunchKernelGGL(psi2n_generic_kernel, dim3(grid_of(cnt * mm)), dim3(256), 0, c->stream, (const double*)c->LET, (const double*)c->V2P, (const double*)c->ZP,
                       (const double*)nullptr, n0, cnt, c->M, c->Mp, c->Q, c->gen_T);
    hipLaunchKernelGGL(psi2_sum_generic_kernel, dim3(grid_of(mm)), dim3(256), 0, c->stream, (const double*)c->gen_T, cnt, c->M, c->Mp, n0 == 0 ? 1 : 0, c->stats);
  }
  GP_EV(c, 11);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

// fills pp (one group: ngrp = 1, row width 3 QB + 1 with QB = Q) and adds the psi2 part of grad_Z to c->grads; the caller finishes the points
int run_phase2_b_generic(gp_ctx* c) {
  GP_TRY_RC(ensure_generic(c));
  const long mm = (long)c->M * c->M;
  const int Q = c->Q, M = c->M;
  for (long n0 = 0; n0 < c->N; n0 += c->gen_P) {
    const long cnt = std::min<long>(c->gen_P, c->N - n0);
    hipLaunchKernelGGL(psi2n_generic_kernel, dim3(grid_of(cnt * mm)), dim3(256), 0, c->stream, (const double*)c->LET, (const double*)c->V2P, (const double*)c->ZP,
                       (const double*)c->Bbar, n0, cnt, M, c->Mp, Q, c->gen_T);
    hipLaunchKernelGGL(psi2_rt_generic_kernel, dim3(grid_of(cnt * M * (Q + 1))), dim3(256), 0, c->stream, (const double*)c->gen_T, (const double*)c->ZP, cnt, M, Q,
                       c->gen_rt);
    hipLaunchKernelGGL(psi2_pp_generic_kernel, dim3(grid_of(cnt * (3 * Q + 1))), dim3(256), 0, c->stream, (const double*)c->gen_rt, (const double*)c->ZP, n0, cnt, M, Q,
                       (long)c->Np, c->pp);
    hipLaunchKernelGGL(psi2_gz_generic_kernel, dim3(grid_of((long)M * Q)), dim3(256), 0, c->stream, (const double*)c->gen_rt, (const double*)c->ZP,
                       (const double*)c->WP, (const double*)c->MUP, (const double*)c->alphaP, n0, cnt, M, Q, c->grads);
  }
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

}  // namespace gp
