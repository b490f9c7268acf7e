// Regime B (Bayesian GPLVM, variances > 0): the pairwise psi2 statistics.
//   psi2_n[m,m'] = exp( LE[n,m] + LE[n,m'] + sum_q V[n,q] * dz2[m,m',q] )            (kernel_exp.py:143-146, factorised)
//     LE[n,m]    = 1/2 ln c2_n - 1/2 sum_q w_nq (mu_nq - z_mq)^2,  w = alpha/(2 alpha S + 1),  c2 = sf2^2 prod (2 alpha S + 1)^-1/2
//     V[n,q]     = -1/4 (alpha_q - w_nq),   dz2[m,m',q] = (z_mq - z_m'q)^2
// Phase 1: Psi2 = sum_n psi2_n with one THREAD per (m,m') pair (upper triangle), the point index is wave-uniform.
// Phase 2: T_n = Bbar o psi2_n, r_n = T_n 1, t_n = T_n Z with one LANE per point, the pair index is wave-uniform
//          (every pair quantity is a scalar operand), giving the psi2 parts of grad_Z / grad_alpha / grad_X_mu / grad_X_S
//          (partial_terms.py:190-205, 273-284, 388-394, 421-427).
#include "gp_common.h"
#include <algorithm>

namespace gp {

// ---------------------------------------------------------------------------------------------- tables
// per-point tables and LE in both layouts; thread = point for LET (coalesced along n), thread = column for LE
__global__ void __launch_bounds__(256) b_tables_kernel(const double* __restrict__ mu, const double* __restrict__ S,
                                                        const double* __restrict__ alpha, long N, long Np, int Q, double sf2,
                                                        double* __restrict__ Vn, double* __restrict__ Wn, double* __restrict__ lnc2h) {
  for (long n = blockIdx.x * 256L + threadIdx.x; n < Np; n += (long)gridDim.x * 256L) {
    double l = log(sf2);   // half of ln c2 = ln sf2 - 1/4 sum ln(2 a S + 1)
    for (int q = 0; q < Q; ++q) {
      const double a = alpha[q], s = S[n * Q + q];
      const double d2 = 2.0 * a * s + 1.0, w = a / d2;
      Wn[n * Q + q] = w;
      Vn[n * Q + q] = -0.25 * (a - w);
      l -= 0.25 * log(d2);
    }
    lnc2h[n] = l;
  }
}

__global__ void __launch_bounds__(256) b_le_kernel(const double* __restrict__ mu, const double* __restrict__ Wn, const double* __restrict__ Vn,
                                                    const double* __restrict__ lnc2h, const double* __restrict__ Z, long N, long Np,
                                                    int M, int Mp, int Q, double* __restrict__ LE, double* __restrict__ LET) {
  // block: 16 rows (n) x 16 cols (m) tile computed once, written in both layouts through LDS
  __shared__ double tile[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const long n = blockIdx.y * 16L + ty;
  const int m = blockIdx.x * 16 + tx;
  double e = 0.0;
  if (n < N && m < M) {
    for (int q = 0; q < Q; ++q) {
      const double d = mu[n * Q + q] - Z[(long)m * Q + q];
      e = fma(Wn[n * Q + q] * d, d, e);
    }
    e = lnc2h[n] - 0.5 * e;
  } else {
    e = -1e300;   // exp() of a padded entry is exactly 0
  }
  LE[n * Mp + m] = e;
  // LEA = LE + sum_q V_nq z_mq^2: with it the pair exponent is LEA_nm + LEA_nm' - 2 sum_q V_nq z_mq z_m'q
  double ea = e;
  if (n < N && m < M) {
    for (int q = 0; q < Q; ++q) { const double z = Z[(long)m * Q + q]; ea = fma(Vn[n * Q + q] * z, z, ea); }
  }
  tile[ty][tx] = ea;
  __syncthreads();
  const long n2 = blockIdx.y * 16L + tx;
  const int m2 = blockIdx.x * 16 + ty;
  LET[(long)m2 * Np + n2] = tile[tx][ty];
}

__global__ void __launch_bounds__(256) dz2_kernel(const double* __restrict__ Z, int M, int Q, double* __restrict__ DZ2) {
  const long total = (long)M * M * Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int q = (int)(i % Q);
    const long mm = i / Q;
    const int m2 = (int)(mm % M), m1 = (int)(mm / M);
    const double d = Z[(long)m1 * Q + q] - Z[(long)m2 * Q + q];
    DZ2[i] = d * d;
  }
}

// ---------------------------------------------------------------------------------------------- phase 1
// grid (pair tiles, n slices); thread (i,j) of a 16x16 tile owns the pair (m = I*16+i, m' = J*16+j), J >= I.
template <int QT>
__global__ void __launch_bounds__(256) psi2_pairs_kernel(const double* __restrict__ LE, const double* __restrict__ Vn,
                                                          const double* __restrict__ DZ2, const int* __restrict__ ptiles, long N, int M,
                                                          int Mp, int Q, int S, double* __restrict__ part, int T) {
  const int tile = blockIdx.x, slice = blockIdx.y;
  const int I = ptiles[2 * tile], J = ptiles[2 * tile + 1];
  const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
  const int m1 = I * 16 + i, m2 = J * 16 + j;
  const bool valid = (m1 < M) && (m2 < M);
  double dz[QT > 0 ? QT : 1];
#pragma unroll
  for (int q = 0; q < QT; ++q) dz[q] = (valid && q < Q) ? DZ2[((long)m1 * M + m2) * Q + q] : 0.0;
  const long per = (N + S - 1) / S;
  const long n0 = slice * per, n1 = min(N, n0 + per);
  double acc = 0.0;
  if (valid) {
    for (long n = n0; n < n1; ++n) {
      double e = LE[n * Mp + m1] + LE[n * Mp + m2];
      const double* v = Vn + n * Q;       // wave-uniform: scalar loads
      if (QT > 0) {
#pragma unroll
        for (int q = 0; q < QT; ++q) if (q < Q) e = fma(v[q], dz[q], e);
      } else {
        const double* dzp = DZ2 + ((long)m1 * M + m2) * Q;
        for (int q = 0; q < Q; ++q) e = fma(v[q], dzp[q], e);
      }
      acc += exp(e);
    }
  }
  part[((long)slice * T + tile) * 256 + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256) psi2_reduce_kernel(const double* __restrict__ part, const int* __restrict__ ptiles, int T, int S,
                                                          int M, int Mp, double* __restrict__ Psi2) {
  const int tile = blockIdx.x;
  const int I = ptiles[2 * tile], J = ptiles[2 * tile + 1];
  const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
  const int m1 = I * 16 + i, m2 = J * 16 + j;
  double s = 0.0;
  for (int sl = 0; sl < S; ++sl) s += part[((long)sl * T + tile) * 256 + threadIdx.x];
  if (m1 < M && m2 < M) {
    if (I != J || m2 >= m1) {
      Psi2[(long)m1 * Mp + m2] = s;
      Psi2[(long)m2 * Mp + m1] = s;
    }
  }
}

// zero the M x M block (and pads) of Psi2 before the pair reduce writes it
__global__ void fill_kernel(double* x, long n, double v) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) x[i] = v;
}

// ---------------------------------------------------------------------------------------------- phase 2
// lane = point.  For every m: r = sum_m' T[m,m'], t_q = sum_m' T[m,m'] z_m'q with T = Bbar[m,m'] psi2_n[m,m'].
//   grad_Z psi2 part  G[m,k] += -a_k z_mk r + a_k t_k + w_k (2 mu_k r - z_mk r - t_k)          (partial_terms.py:190-205, x2 at :238)
//   per point: sr, zr_q, z2r_q, zt_q -> quad = 4 mu^2 sr - 8 mu zr + 2 z2r + 2 zt
//   grad_alpha += -1/4 quad/d2^2 - (S/d2) sr ; grad_X_mu += -w (2 mu sr - 2 zr) ; grad_X_S += 1/2 w^2 quad - w sr
struct PB2Args {
  const double* LET; const double* Vn; const double* Wn; const double* mu; const double* S; const double* DZ2; const double* Z;
  const double* Bbar; const double* alpha; double* Gpart; double* gapart2; double* gmu; double* gS; double* pp;
  long N, Np; int M, Mp, Q, groups_per_block;
};

// MC inducing rows per pass: one LEA[m'][n] load and one uniform z_m' vector serve MC pair terms (MC = 4 for Q <= 16; the
// register arrays p[MC][Q], t[MC][Q] force MC = 1 for larger Q)
template <int QT, int MC>
__global__ void __launch_bounds__(256, 2) psi2_rows_kernel(PB2Args a) {
  // LET holds LEA (m-major): exponent(n; m, m') = LEA[m][n] + LEA[m'][n] + sum_q p_mq z_m'q with p_mq = -2 V_nq z_mq.
  // Per point the running sums sr, zr_q, z2r_q, zt_q live in a.pp (global, touched once per MC rows), so the inner loop
  // keeps only p[MC][Q], t[MC][Q] and r[MC] in registers.
  __shared__ double red[4][MC][QT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* G = a.Gpart + (long)blockIdx.x * a.M * a.Q;
  const int PW = 3 * a.Q + 1;
  for (int grp = 0; grp < a.groups_per_block; ++grp) {
    const long n = ((long)blockIdx.x * a.groups_per_block + grp) * 256 + tid;
    const bool live = n < a.N;
    const long nn = live ? n : 0;
    double* ppn = a.pp + nn * PW;
    if (live) for (int k = 0; k < PW; ++k) ppn[k] = 0.0;
    const double* lcol = a.LET + nn;
    for (int m0 = 0; m0 < a.M; m0 += MC) {
      double p[MC][QT], t[MC][QT], r[MC], lem[MC];
#pragma unroll
      for (int k = 0; k < MC; ++k) {
        const double* zm = a.Z + (long)(m0 + k) * a.Q;               // wave-uniform (rows >= M are zero)
        lem[k] = (live && m0 + k < a.M) ? lcol[(long)(m0 + k) * a.Np] : -1e300;
        r[k] = 0.0;
#pragma unroll
        for (int q = 0; q < QT; ++q) { p[k][q] = (q < a.Q) ? -2.0 * a.Vn[nn * a.Q + q] * zm[q] : 0.0; t[k][q] = 0.0; }
      }
      const double* brow = a.Bbar + (long)m0 * a.Mp;                 // wave-uniform, MC consecutive rows
#pragma unroll 2
      for (int m2 = 0; m2 < a.M; ++m2) {
        const double* z2 = a.Z + (long)m2 * a.Q;                     // wave-uniform
        const double l2 = lcol[(long)m2 * a.Np];
        double zz[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) zz[q] = (q < a.Q) ? z2[q] : 0.0;
#pragma unroll
        for (int k = 0; k < MC; ++k) {
          double e = lem[k] + l2;
#pragma unroll
          for (int q = 0; q < QT; ++q) e = fma(p[k][q], zz[q], e);
          const double T = brow[(long)k * a.Mp + m2] * exp(e);
          r[k] += T;
#pragma unroll
          for (int q = 0; q < QT; ++q) t[k][q] = fma(T, zz[q], t[k][q]);
        }
      }
      // fold the MC rows into the per-point sums and the block-level grad_Z contribution
      double dsr = 0.0;
#pragma unroll
      for (int k = 0; k < MC; ++k) {
        const double* zm = a.Z + (long)(m0 + k) * a.Q;
        const double rk = live ? r[k] : 0.0;
        dsr += rk;
#pragma unroll
        for (int q = 0; q < QT; ++q) {
          if (q < a.Q) {
            const double z = zm[q];
            const double tq = live ? t[k][q] : 0.0;
            if (live) {
              ppn[1 + q] = fma(z, rk, ppn[1 + q]);
              ppn[1 + a.Q + q] = fma(z * z, rk, ppn[1 + a.Q + q]);
              ppn[1 + 2 * a.Q + q] = fma(z, tq, ppn[1 + 2 * a.Q + q]);
            }
            const double w = a.Wn[nn * a.Q + q], mu = a.mu[nn * a.Q + q];
            double g = -a.alpha[q] * (z * rk - tq) + w * (2.0 * mu * rk - z * rk - tq);
            for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o);
            if (lane == 0) red[wave][k][q] = g;
          }
        }
      }
      if (live) ppn[0] += dsr;
      __syncthreads();
      if (tid < MC * a.Q) {
        const int k = tid / a.Q, q = tid - k * a.Q;
        if (m0 + k < a.M) {
          const double s = red[0][k][q] + red[1][k][q] + red[2][k][q] + red[3][k][q];
          double* dst = G + (long)(m0 + k) * a.Q + q;
          *dst = ((grp == 0) ? 0.0 : *dst) + s;
        }
      }
      __syncthreads();
    }
  }
}

// per-point finish of the psi2 part from the running sums pp[n] = [sr, zr_q, z2r_q, zt_q]
__global__ void __launch_bounds__(256) psi2_points_finish_kernel(PB2Args a) {
  __shared__ double redq[256];
  const int PW = 3 * a.Q + 1;
  for (int q = 0; q < a.Q; ++q) {
    double ga = 0.0;
    for (long n = blockIdx.x * 256L + threadIdx.x; n < a.N; n += (long)gridDim.x * 256L) {
      const double* ppn = a.pp + n * PW;
      const double sr = ppn[0], zr = ppn[1 + q], z2r = ppn[1 + a.Q + q], zt = ppn[1 + 2 * a.Q + q];
      const double s = a.S[n * a.Q + q], al = a.alpha[q], w = a.Wn[n * a.Q + q], mu = a.mu[n * a.Q + q];
      const double d2 = 2.0 * al * s + 1.0;
      const double quad = 4.0 * mu * mu * sr - 8.0 * mu * zr + 2.0 * z2r + 2.0 * zt;
      ga += -0.25 * quad / (d2 * d2) - (s / d2) * sr;
      a.gmu[n * a.Q + q] += -w * (2.0 * mu * sr - 2.0 * zr);
      a.gS[n * a.Q + q] += 0.5 * w * w * quad - w * sr;
    }
    redq[threadIdx.x] = ga;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) redq[threadIdx.x] += redq[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) a.gapart2[(long)blockIdx.x * a.Q + q] = redq[0];
    __syncthreads();
  }
}

// grads[0:M*Q] += sum_blocks Gpart ; grads[M*Q + q] += sum_blocks gapart2
__global__ void __launch_bounds__(256) pb2_reduce_kernel(const double* __restrict__ Gpart, const double* __restrict__ gapart2, int nb, int nb2,
                                                         long MQ, int Q, double* __restrict__ grads) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < MQ + Q; i += (long)gridDim.x * 256L) {
    double s = 0.0;
    if (i < MQ) for (int b = 0; b < nb; ++b) s += Gpart[(long)b * MQ + i];
    else for (int b = 0; b < nb2; ++b) s += gapart2[(long)b * Q + (i - MQ)];
    grads[i] += s;
  }
}

// ---------------------------------------------------------------------------------------------- host side
template <typename T>
static int balloc(gp_ctx* c, T** p, size_t count) {
  GP_HIP(c, hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)));
  GP_HIP(c, hipMemsetAsync(*p, 0, std::max<size_t>(count, 1) * sizeof(T), c->stream));
  return GP_OK;
}

int ensure_regime_b_buffers(gp_ctx* c) {
  if (c->b_alloc) return GP_OK;
  const long Np = c->Np, Mp = c->Mp, M = c->M, Q = c->Q;
  int rc = GP_OK;
  auto A = [&](auto** p, size_t n) { if (rc == GP_OK) rc = balloc(c, p, n); };
  A(&c->LE, (size_t)Np * Mp); A(&c->LET, (size_t)Mp * Np); A(&c->Vn, (size_t)Np * Q); A(&c->Wn, (size_t)Np * Q);
  A(&c->DZ2, (size_t)M * M * Q); A(&c->lnc2h, (size_t)Np);
  const long groups = (c->N + 255) / 256;
  c->pb_blocks = (int)std::min<long>(groups, 2048);
  A(&c->Gpart, (size_t)c->pb_blocks * M * Q); A(&c->gapart2, (size_t)c->pb_blocks * Q); A(&c->pp, (size_t)Np * (3 * Q + 1));
  std::vector<int> t;
  const int Mt = (int)((M + 15) / 16);
  for (int i = 0; i < Mt; ++i) for (int j = i; j < Mt; ++j) { t.push_back(i); t.push_back(j); }
  c->n_ptiles = (int)t.size() / 2;
  A(&c->ptiles, t.size());
  if (rc != GP_OK) return rc;
  GP_HIP(c, hipMemcpyAsync(c->ptiles, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  GP_HIP(c, hipStreamSynchronize(c->stream));
  // the pair kernel's split-n partials live in c->part: make sure it is large enough
  const size_t need = (size_t)c->n_ptiles * 256 * 64;
  if (need > c->part_doubles) {
    (void)hipFree(c->part);
    c->part = nullptr;
    GP_HIP(c, hipMalloc((void**)&c->part, need * 8));
    c->part_doubles = need;
  }
  c->b_alloc = true;
  return GP_OK;
}

int run_dz2(gp_ctx* c) {
  const long total = (long)c->M * c->M * c->Q;
  hipLaunchKernelGGL(dz2_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, c->stream, c->Z, c->M, c->Q, c->DZ2);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

int run_generate_b(gp_ctx* c) {
  int rc = ensure_regime_b_buffers(c);
  if (rc != GP_OK) return rc;
  rc = run_dz2(c);
  if (rc != GP_OK) return rc;
  hipLaunchKernelGGL(b_tables_kernel, dim3(c->kl_blocks), dim3(256), 0, c->stream, c->mu, c->S, c->alpha, (long)c->N, (long)c->Np, c->Q,
                     c->sf2, c->Vn, c->Wn, c->lnc2h);
  dim3 grid(c->Mp / 16, (unsigned)(c->Np / 16));
  hipLaunchKernelGGL(b_le_kernel, grid, dim3(256), 0, c->stream, c->mu, c->Wn, c->Vn, c->lnc2h, c->Z, (long)c->N, (long)c->Np, c->M, c->Mp, c->Q,
                     c->LE, c->LET);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

template <int QT>
static void launch_pairs(gp_ctx* c, int S) {
  hipLaunchKernelGGL((psi2_pairs_kernel<QT>), dim3(c->n_ptiles, S), dim3(256), 0, c->stream, c->LE, c->Vn, c->DZ2, c->ptiles, (long)c->N,
                     c->M, c->Mp, c->Q, S, c->part, c->n_ptiles);
}

int run_phase1_b(gp_ctx* c) {
  int S = (int)std::max<long>(1, std::min<long>(64, std::min<long>(c->N, (4096 + c->n_ptiles - 1) / c->n_ptiles)));
  if (c->Q <= 4) launch_pairs<4>(c, S);
  else if (c->Q <= 10) launch_pairs<10>(c, S);
  else if (c->Q <= 16) launch_pairs<16>(c, S);
  else if (c->Q <= 32) launch_pairs<32>(c, S);
  else launch_pairs<0>(c, S);
  GP_HIP(c, hipGetLastError());
  hipLaunchKernelGGL(psi2_reduce_kernel, dim3(c->n_ptiles), dim3(256), 0, c->stream, c->part, c->ptiles, c->n_ptiles, S, c->M, c->Mp, c->stats);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

template <int QT, int MC>
static void launch_rows(gp_ctx* c, const PB2Args& a, int blocks) {
  hipLaunchKernelGGL((psi2_rows_kernel<QT, MC>), dim3(blocks), dim3(256), 0, c->stream, a);
}

int run_phase2_b(gp_ctx* c) {
  if (c->Q > 64) return fail(c, GP_ERR_UNSUPPORTED, "regime B supports Q <= 64 (got %d)", c->Q);
  PB2Args a;
  a.LET = c->LET; a.Vn = c->Vn; a.Wn = c->Wn; a.mu = c->mu; a.S = c->S; a.DZ2 = c->DZ2; a.Z = c->Z; a.Bbar = c->Bbar; a.alpha = c->alpha;
  a.Gpart = c->Gpart; a.gapart2 = c->gapart2; a.gmu = c->gXmu; a.gS = c->gXs; a.pp = c->pp;
  a.N = c->N; a.Np = c->Np; a.M = c->M; a.Mp = c->Mp; a.Q = c->Q;
  const long groups = (c->N + 255) / 256;
  a.groups_per_block = (int)((groups + c->pb_blocks - 1) / c->pb_blocks);
  const int blocks = (int)((groups + a.groups_per_block - 1) / a.groups_per_block);
  if (c->Q <= 4) launch_rows<4, 4>(c, a, blocks);
  else if (c->Q <= 10) launch_rows<10, 4>(c, a, blocks);
  else if (c->Q <= 16) launch_rows<16, 2>(c, a, blocks);
  else if (c->Q <= 32) launch_rows<32, 1>(c, a, blocks);
  else launch_rows<64, 1>(c, a, blocks);
  GP_HIP(c, hipGetLastError());
  hipLaunchKernelGGL(psi2_points_finish_kernel, dim3(c->pb_blocks), dim3(256), 0, c->stream, a);
  GP_HIP(c, hipGetLastError());
  const long MQ = (long)c->M * c->Q;
  hipLaunchKernelGGL(pb2_reduce_kernel, dim3((unsigned)std::min<long>((MQ + c->Q + 255) / 256, 1024)), dim3(256), 0, c->stream, c->Gpart,
                     c->gapart2, blocks, c->pb_blocks, MQ, c->Q, c->grads);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

}  // namespace gp
