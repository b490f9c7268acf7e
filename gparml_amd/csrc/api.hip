// extern "C" entry points of libgparml_hip.so (see include/gparml_hip.h).
#include "gp_common.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>

namespace gp {
thread_local std::string g_create_error;

int fail(gp_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf; else g_create_error = buf;
  return code;
}

std::atomic<int> g_opt_poison{[] { const char* e = getenv("GPARML_POISON"); return (e && e[0] == '1') ? 1 : 0; }()};
int dalloc_bytes(gp_ctx* c, void** p, size_t bytes, int mode) {
  bytes = std::max<size_t>(bytes, 8);
  GP_HIP(c, hipMalloc(p, bytes));
  const bool poison = g_opt_poison.load() && mode != DA_ZERO;
  if (poison || mode != DA_RAW) GP_HIP(c, hipMemsetAsync(*p, poison ? 0xFF : 0, bytes, c->stream));
  return GP_OK;
}
template <typename T>
static int dalloc(gp_ctx* c, T** p, size_t count, int mode = DA_INIT) {
  return dalloc_bytes(c, (void**)p, count * sizeof(T), mode);
}
#define GP_TRY(x) do { int rc__ = (x); if (rc__ != GP_OK) return rc__; } while (0)

__global__ void sumsq_kernel(const double* __restrict__ x, long n, double* part) {
  __shared__ double red[256];
  double s = 0.0;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) s += x[i] * x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// gather a padded device matrix [rows_p][ld] into a dense host-shaped [rows][cols] staging buffer
__global__ void gather2d_kernel(const double* __restrict__ src, long ld, long rows, long cols, double* __restrict__ dst) {
  const long total = rows * cols;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const long r = i / cols, c = i - r * cols;
    dst[i] = src[r * ld + c];
  }
}
__global__ void scatter2d_kernel(const double* __restrict__ src, long rows, long cols, double* __restrict__ dst, long ld, long rows_p, long cols_p) {
  const long total = rows_p * cols_p;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const long r = i / cols_p, c = i - r * cols_p;
    dst[r * ld + c] = (r < rows && c < cols) ? src[r * cols + c] : 0.0;
  }
}
__global__ void scale_kernel(double* x, long n, double f) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) x[i] *= f;
}
// Zaug[m] = [1, z_m, z_m^2] (rows >= M zero), Z padded copy
// Zin / alpha_in are read straight from the pinned host slot of gp_set_globals (mapped memory: M Q + Q doubles over the bus, no copy commands)
__global__ void __launch_bounds__(256) zaug_kernel(const double* __restrict__ Zin, const double* __restrict__ alpha_in, int M, int Mp, int Q, int CZp,
                                                   double* __restrict__ Z, double* __restrict__ Zaug, double* __restrict__ alpha, double* __restrict__ Zt) {
  // one element of Zaug per thread (r05): every read of the mapped host slot is its own bus round trip, and the first form -- one thread per inducing point
  // walking its Q coordinates -- made them one after the other (8.8 us at M = 128, Q = 10; 28 us at M = 1024, Q = 50)
  const long total = (long)Mp * CZp;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const int m = (int)(e / CZp), c = (int)(e - (long)m * CZp);
    if (e < Q) alpha[e] = alpha_in[e];
    double v = 0.0;
    if (c == 0) v = (m < M) ? 1.0 : 0.0;
    else if (c <= 2 * Q) {
      const int q = (c <= Q) ? c - 1 : c - 1 - Q;
      const double z = (m < M) ? Zin[(long)m * Q + q] : 0.0;
      if (c <= Q) { Z[(long)m * Q + q] = z; Zt[(long)q * Mp + m] = z; v = z; }
      else v = z * z;
    }
    Zaug[e] = v;
  }
}

static int blocks_for(long n) { return (int)std::max<long>(1, std::min<long>((n + 255) / 256, 8192)); }

static int download_matrix(gp_ctx* c, const double* src, long ld, long rows, long cols, double* dst, int64_t n) {
  if (n != rows * cols) return fail(c, GP_ERR_BAD_ARG, "gp_download: expected %ld doubles, got %ld", rows * cols, (long)n);
  double* tmp = nullptr;
  GP_HIP(c, hipMalloc((void**)&tmp, std::max<long>(rows * cols, 1) * 8));
  hipLaunchKernelGGL(gather2d_kernel, dim3(blocks_for(rows * cols)), dim3(256), 0, c->stream, src, ld, rows, cols, tmp);
  hipError_t e = hipMemcpyAsync(dst, tmp, rows * cols * 8, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(tmp);
  if (e != hipSuccess) return fail(c, GP_ERR_HIP, "download failed: %s", hipGetErrorString(e));
  return GP_OK;
}
}  // namespace gp

using namespace gp;

extern "C" const char* gp_last_error(const gp_ctx* ctx) { return ctx ? ctx->err.c_str() : gp::g_create_error.c_str(); }
extern "C" const char* gp_version(void) { return "gparml_hip 0.1 (gfx950, fp64 4x4x4-mfma)"; }

extern "C" int gp_create(gp_ctx** out, int device, int64_t N_s, int D, int M, int Q) {
  if (!out) return fail(nullptr, GP_ERR_BAD_ARG, "gp_create: out is NULL");
  *out = nullptr;
  if (N_s <= 0 || D <= 0 || M <= 0 || Q <= 0) return fail(nullptr, GP_ERR_BAD_ARG, "gp_create: N_s, D, M, Q must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, GP_ERR_HIP, "gp_create: no HIP device available");
  if (device < 0 || device >= ndev) return fail(nullptr, GP_ERR_BAD_ARG, "gp_create: device %d out of range (%d devices)", device, ndev);
  gp_ctx* c = new gp_ctx();
  c->device = device;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) { delete c; return fail(nullptr, GP_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e)); }
  c->N = N_s; c->D = D; c->M = M; c->Q = Q;
  c->Np = round_up(N_s, TILE);
  c->Mp = (int)round_up(M, TILE);
  c->Dp = (int)round_up(D, TILE);
  c->LDK = c->Mp + c->Dp;
  c->CX = 2 * Q + 1; c->CXp = (int)round_up(c->CX, 4);
  c->CZ = 2 * Q + 1; c->CZp = (int)round_up(c->CZ, 4);
  const long Mp = c->Mp, Dp = c->Dp, Np = c->Np;
  int rc = GP_OK;
  auto A = [&](auto** p, size_t n) { if (rc == GP_OK) rc = dalloc(c, p, n); };
  A(&c->Kaug, (size_t)Np * c->LDK);
  A(&c->Xmu, (size_t)N_s * Q); A(&c->Xs, (size_t)N_s * Q); A(&c->dir, (size_t)2 * N_s * Q);
  A(&c->mu, (size_t)Np * Q); A(&c->S, (size_t)Np * Q); A(&c->U, (size_t)Np * Q);
  if (rc == GP_OK) rc = dalloc(c, &c->PU, (size_t)Np * (2 * std::max(psi1_qp(Q), 2) + 2), DA_ZERO);   // zero contract: the records' columns Q .. QP - 1 (u = 0: no guards in psi1_kernel's q loop) are never written
  A(&c->lnc1, (size_t)Np); A(&c->Xa, (size_t)Np * c->CXp);
  A(&c->Z, (size_t)Mp * Q); A(&c->alpha, (size_t)Q); A(&c->Zaug, (size_t)Mp * c->CZp + 8); A(&c->Zt, (size_t)Mp * Q);   // + 8: p2_gen8_kernel stages feature columns in groups of eight
  A(&c->stats, (size_t)Mp * Mp + Mp * Dp + SC_COUNT);
  A(&c->grads, (size_t)M * Q + Q);
  // phase-1 tile table: Psi2 upper tiles first, then the C tiles
  std::vector<int> tiles;
  const int mt = c->Mp / TILE, dt = c->Dp / TILE;
  for (int i = 0; i < mt; ++i) for (int j = i; j < mt; ++j) { tiles.push_back(i); tiles.push_back(j); }
  for (int i = 0; i < mt; ++i) for (int j = 0; j < dt; ++j) { tiles.push_back(i); tiles.push_back(mt + j); }
  c->n_tiles = (int)tiles.size() / 2;
  A(&c->tiles, tiles.size());
  if (rc == GP_OK && hipMemcpy(c->tiles, tiles.data(), tiles.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) rc = fail(c, GP_ERR_HIP, "tile table upload failed");
  const int total_chunks = (int)(Np / KC);
  const int S = std::max(1, std::min(512 / std::max(1, std::min(c->n_tiles, mt * dt > 0 ? c->n_tiles : 1)), total_chunks));
  // worst case slices x tiles (regime B uses fewer tiles, hence possibly more slices)
  const int Tb = mt * dt;
  const int Sb = std::max(1, std::min(512 / std::max(1, Tb), total_chunks));
  c->part_doubles = (size_t)std::max((long)(S + 16) * c->n_tiles, (long)(Sb + 16) * Tb) * TILE * TILE;
  if (c->part_doubles < (size_t)1100 * TILE * TILE) c->part_doubles = (size_t)1100 * TILE * TILE;   // p1v2: up to 1024 partial tiles
  A(&c->part, c->part_doubles);
  c->kl_blocks = blocks_for(Np);
  A(&c->klpart, (size_t)c->kl_blocks + 8192);
  A(&c->Kmm, (size_t)2 * Mp * Mp); A(&c->Lmat, (size_t)2 * Mp * Mp); A(&c->Inv, (size_t)2 * Mp * Mp);
  if (rc == GP_OK) rc = dalloc(c, &c->Linv, (size_t)2 * Mp * Mp, DA_ZERO);   // zero contract: the 128-blocks above the block diagonal are never written (potrf_inverse_batched's precondition)
  if (Mp >= 512 && Mp <= 2048) {
    // gsi8.hip: ten digit planes of W = [A | B] for the larger of the two products (K_mm^-1 | Psi2: 2 Mp columns; K_mm + beta Psi2 | E: Mp + Dp), and W's column scales
    c->gss_count = (size_t)Mp + std::max(Mp, Dp);
    c->gsd_bytes = (size_t)10 * Mp * c->gss_count;
    double* tmp = nullptr;
    A(&tmp, c->gsd_bytes / 8); c->gsd = reinterpret_cast<int8_t*>(tmp);
    A(&c->gss, c->gss_count);
  }
  A(&c->KmmKeep, (size_t)Mp * Mp); A(&c->T1, (size_t)Mp * std::max<long>(std::max(Mp, Dp), 256)); A(&c->T2, (size_t)Mp * std::max(Mp, Dp));
  A(&c->dFdK, (size_t)Mp * Mp); A(&c->Bbar, (size_t)Mp * Mp);
  A(&c->E, (size_t)Mp * Dp); A(&c->PsiE, (size_t)Mp * Dp); A(&c->Abar, (size_t)Mp * Dp);
  A(&c->Bm, (size_t)c->LDK * Mp);
  A(&c->gs, (size_t)GS_COUNT + 8 + 8 * 64);   // scalars | failure flags | dots_kernel partials [8 jobs][64 blocks]
  A(&c->gK, (size_t)M * Q + Q);
  // phase 2
  c->p2_slices = std::max(1, std::min<int>(8 * std::max(1, 64 / mt), (int)(Np / TILE)));
  A(&c->Rpart, (size_t)2 * (c->p2_slices + 8) * Mp * c->CXp);
  A(&c->HZp, (size_t)(Mp / TILE) * Np * c->CZp);    // p2_gen8_kernel: per-point partials, one array per 128 inducing columns
  A(&c->gXmu, (size_t)N_s * Q); A(&c->gXs, (size_t)N_s * Q);
  c->ga_blocks = blocks_for(Np);
  A(&c->gapart, (size_t)c->ga_blocks * Q);
  // fast phase 2: per-wave (eight-wave kernel: blocks * 8 rows of <= 12) or per-256-points (four-wave kernel) partials of grad_alpha's mu^2 term
  A(&c->hgpart, std::max((size_t)((N_s + 255) / 256) * Q, (size_t)8 * (c->p2_slices + 8) * (Mp / TILE) * 8 * 12));
  A(&c->g_latest, (size_t)2 * N_s * Q); A(&c->g_new, (size_t)2 * N_s * Q); A(&c->g_old, (size_t)2 * N_s * Q);
  for (int i = 0; i < 14 && rc == GP_OK; ++i) if (hipEventCreate(&c->ev[i]) != hipSuccess) rc = fail(c, GP_ERR_HIP, "hipEventCreate failed");
  if (rc == GP_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(c, GP_ERR_HIP, "device sync failed after allocation");
  if (rc != GP_OK) { gp::g_create_error = c->err; gp_destroy(c); return rc; }
  *out = c;
  return GP_OK;
}

extern "C" int gp_destroy(gp_ctx* c) {
  if (!c) return GP_OK;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  double* bufs[] = {c->Kaug, c->Xmu, c->Xs, c->dir, c->mu, c->S, c->U, c->PU, c->lnc1, c->Xa, c->Z, c->alpha, c->Zaug, c->Zt, reinterpret_cast<double*>(c->gsd), c->gss,
                    c->stats_external ? nullptr : c->stats, c->grads_external ? nullptr : c->grads, c->part, c->klpart, c->Kmm, c->Lmat,
                    c->Linv, c->Inv, c->KmmKeep, c->T1, c->T2, c->dFdK, c->Bbar, c->E, c->PsiE, c->Abar, c->Bm, c->gs, c->gK, c->Rpart,
                    c->HZp, c->gXmu, c->gXs, c->gapart, c->hgpart, c->g_latest, c->g_new, c->g_old, c->LE, c->LET, c->Bbar4, c->Vn, c->Wn, c->V2P, c->ZP, c->Z1P, c->WP, c->MUP, c->alphaP, c->lnc2h, c->DZ2,
                    c->Gpart, c->Gtmp, c->gapart2, c->pp, c->Z1S, c->ppt, c->Gt, c->spack, c->gen_T, c->gen_rt};
  for (double* b : bufs) if (b) (void)hipFree(b);
  if (c->tiles) (void)hipFree(c->tiles);
  if (c->ptiles) (void)hipFree(c->ptiles);
  if (c->tiles64) (void)hipFree(c->tiles64);
  if (c->sym_sched) (void)hipFree(c->sym_sched);
  if (c->bmap) (void)hipFree(c->bmap);
  if (c->staging) (void)hipFree(c->staging);
  if (c->p2prog) (void)hipFree(c->p2prog);
  gp::p1v2_free(c);
  gp::p1i8_free(c);
  gp::comm_free(c);
  for (int i = 0; i < 14; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->h_out) (void)hipHostFree(c->h_out);
  for (int i = 0; i < 2; ++i) { if (c->h_glob[i]) (void)hipHostFree(c->h_glob[i]); if (c->glob_ev[i]) (void)hipEventDestroy(c->glob_ev[i]); }
  delete c;
  return GP_OK;
}

extern "C" int gp_memory_info(gp_ctx* c, int64_t* free_bytes, int64_t* total_bytes) {
  if (!c) return GP_ERR_BAD_ARG;
  GP_HIP(c, hipSetDevice(c->device));
  size_t f = 0, t = 0;
  GP_HIP(c, hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  return GP_OK;
}

extern "C" int gp_i8_status(gp_ctx* c, int* state, double* rel_psi2, double* rel_c, double* cond_lower_bound, int64_t* checks) {
  if (!c) return GP_ERR_BAD_ARG;
  if (state) *state = !p1i8_applicable_static(c) ? -1 : c->i8_guard;
  if (rel_psi2) *rel_psi2 = c->i8_rel_psi2;
  if (rel_c) *rel_c = c->i8_rel_c;
  if (cond_lower_bound) *cond_lower_bound = c->i8_cond_lb;
  if (checks) *checks = c->i8_checks;
  return GP_OK;
}

extern "C" int gp_set_timing(gp_ctx* c, int level) {
  if (!c) return GP_ERR_BAD_ARG;
  if (level < 0 || level > 2) return fail(c, GP_ERR_BAD_ARG, "gp_set_timing: level must be 0 (no events), 1 (total only) or 2 (every phase and kernel)");
  c->timing = level;
  return GP_OK;
}

extern "C" int gp_set_stream(gp_ctx* c, void* s) {
  if (!c) return GP_ERR_BAD_ARG;
  c->stream = (hipStream_t)s;
  return GP_OK;
}

static int upload_embeddings(gp_ctx* c, const double* X_mu, const double* X_S, int xs_is_raw) {
  const size_t nq = (size_t)c->N * c->Q;
  bool all_zero = true, any_neg = false, finite = true;
  for (size_t i = 0; i < nq; ++i) {
    const double s = X_S[i];
    if (s != 0.0) all_zero = false;
    if (s < 0.0) any_neg = true;
    if (!std::isfinite(s) || !std::isfinite(X_mu[i])) finite = false;
  }
  if (!finite) return fail(c, GP_ERR_NON_FINITE, "embeddings contain non-finite values");
  if (!xs_is_raw && any_neg) return fail(c, GP_ERR_BAD_ARG, "X_S must be >= 0 (kernel_exp.py:30 assertion)");
  if (!xs_is_raw && !all_zero) {
    for (size_t i = 0; i < nq; ++i) if (X_S[i] == 0.0) return fail(c, GP_ERR_NON_FINITE, "X_S mixes zero and non-zero variances: log(0) in the KL term (partial_terms.py:85)");
  }
  c->xs_raw = xs_is_raw != 0;
  c->regime_A = (!xs_is_raw) && all_zero;
  c->prep_fixa_valid = false;
  GP_HIP(c, hipMemcpyAsync(c->Xmu, X_mu, nq * 8, hipMemcpyHostToDevice, c->stream));
  GP_HIP(c, hipMemcpyAsync(c->Xs, X_S, nq * 8, hipMemcpyHostToDevice, c->stream));
  GP_HIP(c, hipStreamSynchronize(c->stream));
  c->state = 0;
  return GP_OK;
}

extern "C" int gp_upload_shard(gp_ctx* c, const double* Y, const double* X_mu, const double* X_S, int xs_is_raw) {
  if (!c) return GP_ERR_BAD_ARG;
  if (!Y || !X_mu || !X_S) return fail(c, GP_ERR_BAD_ARG, "gp_upload_shard: NULL array");
  GP_HIP(c, hipSetDevice(c->device));
  const size_t nd = (size_t)c->N * c->D;
  double* dY = nullptr;
  GP_HIP(c, hipMalloc((void**)&dY, nd * 8));
  hipError_t e = hipMemcpyAsync(dY, Y, nd * 8, hipMemcpyHostToDevice, c->stream);
  int rc = GP_OK;
  if (e != hipSuccess) rc = fail(c, GP_ERR_HIP, "Y upload failed: %s", hipGetErrorString(e));
  if (rc == GP_OK) rc = run_upload_y(c, dY);
  if (rc == GP_OK) {
    // sum_YYT (partial_terms.py:40) once per upload
    const int nb = 1024;
    double* part = c->klpart + c->kl_blocks;  // spare tail of the KL partial buffer (8192 doubles)
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, c->stream, dY, (long)nd, part);
    std::vector<double> h(nb);
    e = hipMemcpyAsync(h.data(), part, nb * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(c, GP_ERR_HIP, "sum_YYT failed: %s", hipGetErrorString(e));
    double s = 0.0;
    for (double v : h) s += v;
    c->sumYY = s;
    if (!std::isfinite(s)) rc = fail(c, GP_ERR_NON_FINITE, "Y contains non-finite values");
  }
  (void)hipFree(dY);
  if (rc != GP_OK) return rc;
  GP_TRY(upload_embeddings(c, X_mu, X_S, xs_is_raw));
  c->have_data = true;
  c->have_dir = false;
  c->prep_fixa_valid = false;
  c->i8_y_valid = false;
  c->i8_guard = 0; c->i8_since_check = 0; c->i8_check_pending = false;      // new data: the int8 path is measured again (p1i8.hip, guard)
  return GP_OK;
}

extern "C" int gp_upload_embeddings(gp_ctx* c, const double* X_mu, const double* X_S, int xs_is_raw) {
  if (!c) return GP_ERR_BAD_ARG;
  if (!c->have_data) return fail(c, GP_ERR_STATE, "gp_upload_embeddings before gp_upload_shard");
  if (!X_mu || !X_S) return fail(c, GP_ERR_BAD_ARG, "gp_upload_embeddings: NULL array");
  GP_HIP(c, hipSetDevice(c->device));
  return upload_embeddings(c, X_mu, X_S, xs_is_raw);
}

extern "C" int gp_set_direction(gp_ctx* c, const double* d) {
  if (!c) return GP_ERR_BAD_ARG;
  GP_HIP(c, hipSetDevice(c->device));
  if (!d) { c->have_dir = false; return GP_OK; }
  GP_HIP(c, hipMemcpyAsync(c->dir, d, (size_t)2 * c->N * c->Q * 8, hipMemcpyHostToDevice, c->stream));
  GP_HIP(c, hipStreamSynchronize(c->stream));
  c->have_dir = true;
  c->state = 0;
  return GP_OK;
}

// Test mode (gp_common.h, g_opt_poison): everything an evaluation is supposed to (re)write before it reads is refilled with NaN bytes when a new
// evaluation starts -- scratch and partial sums, the statistics, the global step's matrices, the gradients, Psi1 (not Y), the regime-B tables.
// Not refilled: the shard's data and what the prep kernels derive from it alone (they are skipped from the second evaluation on for fixed
// embeddings), the CG vectors, Linv (its upper blocks are a zero contract), tables and plans.
static int poison_scratch(gp_ctx* c) {
  const size_t Mp = c->Mp, Dp = c->Dp, Np = c->Np, M = c->M, Q = c->Q, N = c->N;
  auto P = [&](void* p, size_t bytes) -> hipError_t { return (p && bytes) ? hipMemsetAsync(p, 0xFF, bytes, c->stream) : hipSuccess; };
  auto PD = [&](double* p, size_t n) -> hipError_t { return P(p, n * sizeof(double)); };
  GP_HIP(c, PD(c->part, c->part_doubles));
  GP_HIP(c, PD(c->Rpart, (size_t)2 * (c->p2_slices + 8) * Mp * c->CXp));
  GP_HIP(c, PD(c->HZp, (size_t)(Mp / TILE) * Np * c->CZp));
  GP_HIP(c, PD(c->gapart, (size_t)c->ga_blocks * Q));
  GP_HIP(c, PD(c->hgpart, std::max((size_t)((N + 255) / 256) * Q, (size_t)8 * (c->p2_slices + 8) * (Mp / TILE) * 8 * 12)));
  GP_HIP(c, PD(c->Kmm, 2 * Mp * Mp)); GP_HIP(c, PD(c->Lmat, 2 * Mp * Mp)); GP_HIP(c, PD(c->Inv, 2 * Mp * Mp)); GP_HIP(c, PD(c->KmmKeep, Mp * Mp));
  GP_HIP(c, PD(c->T1, Mp * std::max<size_t>(std::max(Mp, Dp), 256))); GP_HIP(c, PD(c->T2, Mp * std::max(Mp, Dp)));
  GP_HIP(c, PD(c->dFdK, Mp * Mp)); GP_HIP(c, PD(c->Bbar, Mp * Mp)); GP_HIP(c, PD(c->E, Mp * Dp)); GP_HIP(c, PD(c->PsiE, Mp * Dp)); GP_HIP(c, PD(c->Abar, Mp * Dp));
  GP_HIP(c, PD(c->Bm, (size_t)c->LDK * Mp)); GP_HIP(c, PD(c->gK, M * Q + Q)); GP_HIP(c, PD(c->gs, (size_t)GS_COUNT + 8 + 8 * 64));
  if (!c->stats_external) GP_HIP(c, PD(c->stats, Mp * Mp + Mp * Dp + SC_COUNT));
  if (!c->grads_external) GP_HIP(c, PD(c->grads, M * Q + Q));
  GP_HIP(c, PD(c->gXmu, N * Q)); GP_HIP(c, PD(c->gXs, N * Q));
  GP_HIP(c, hipMemset2DAsync(c->Kaug, (size_t)c->LDK * 8, 0xFF, Mp * 8, Np, c->stream));      // the Psi1 columns of [Psi1 | Y]
  if (c->b_alloc) {
    GP_HIP(c, PD(c->LE, Np * Mp)); GP_HIP(c, PD(c->LET, Np * Mp));
    GP_HIP(c, PD(c->Gpart, (size_t)c->pb_blocks * M * Q)); GP_HIP(c, PD(c->gapart2, (size_t)c->pb_blocks * Q)); GP_HIP(c, PD(c->Gtmp, (size_t)64 * M * Q));
    GP_HIP(c, PD(c->pp, c->pp_doubles));
    if (c->ppt) GP_HIP(c, PD(c->ppt, (size_t)c->n_tiles64 * (3 * Q + 1) * c->b_ch));
    if (c->Gt) GP_HIP(c, PD(c->Gt, (size_t)c->b_S * c->n_tiles64 * 2 * 64 * Q));
  }
  return GP_OK;
}

extern "C" int gp_set_globals(gp_ctx* c, const double* Z, double sf2, const double* alpha, double beta, int64_t N_global, double step) {
  if (!c) return GP_ERR_BAD_ARG;
  if (!Z || !alpha) return fail(c, GP_ERR_BAD_ARG, "gp_set_globals: NULL array");
  if (!(sf2 > 0.0)) return fail(c, GP_ERR_BAD_ARG, "sf2 must be > 0 (kernels.py:62 assert sf > 0)");
  if (!(beta > 0.0) || !std::isfinite(beta)) return fail(c, GP_ERR_BAD_ARG, "beta must be finite and > 0");
  for (int q = 0; q < c->Q; ++q) {
    if (!(alpha[q] >= 0.0)) return fail(c, GP_ERR_BAD_ARG, "alpha must be >= 0 (kernel_exp.py:31 assertion)");
    if (!std::isfinite(alpha[q])) return fail(c, GP_ERR_NON_FINITE, "alpha is not finite");
  }
  for (long i = 0; i < (long)c->M * c->Q; ++i) if (!std::isfinite(Z[i])) return fail(c, GP_ERR_NON_FINITE, "Z is not finite");
  if (N_global < c->N) return fail(c, GP_ERR_BAD_ARG, "N_global (%ld) smaller than the local shard (%ld)", (long)N_global, (long)c->N);
  GP_HIP(c, hipSetDevice(c->device));
  if (g_opt_poison.load()) GP_TRY(poison_scratch(c));
  // stage through pinned memory: hipMemcpyAsync from pageable memory blocks the host until the copy has been staged AND used to be followed by a
  // stream synchronisation here (r03: every evaluation of an optimiser paid it).  r05: no copy command at all -- zaug_kernel reads the pinned
  // (mapped) slot itself; two copy commands cost ~25 us of stream time at configs[1]'s size (blit dispatches with idle gaps around them,
  // profiles/r05_config1_timeline.txt) for 10 KB.  The evaluation's only host synchronisation is the read-back in gp_finish
  const size_t nz = (size_t)c->M * c->Q, nq = (size_t)c->Q;
  const int slot = c->glob_slot;
  if (!c->h_glob[slot]) {
    GP_HIP(c, hipHostMalloc((void**)&c->h_glob[slot], (nz + nq) * sizeof(double), hipHostMallocMapped));
  } else if (c->glob_epoch[slot] >= c->sync_epoch) {
    // the kernel that read this slot two calls ago may still be queued: no stream synchronisation has been seen since (never the case in an
    // optimiser's sequence -- every evaluation ends in gp_finish's synchronisation -- so no event is recorded per call: that was one more signal
    // packet on the stream)
    GP_HIP(c, hipStreamSynchronize(c->stream));
    ++c->sync_epoch;
  }
  std::memcpy(c->h_glob[slot], Z, nz * sizeof(double));
  std::memcpy(c->h_glob[slot] + nz, alpha, nq * sizeof(double));
  double* dslot = nullptr;
  GP_HIP(c, hipHostGetDevicePointer((void**)&dslot, c->h_glob[slot], 0));
  hipLaunchKernelGGL(zaug_kernel, dim3((int)std::min<long>(((long)c->Mp * c->CZp + 255) / 256, 1024)), dim3(256), 0, c->stream, dslot, dslot + nz, c->M, c->Mp, c->Q, c->CZp, c->Z,
                     c->Zaug, c->alpha, c->Zt);
  GP_HIP(c, hipGetLastError());
  c->glob_epoch[slot] = c->sync_epoch;                      // the slot may be rewritten once a later stream synchronisation has passed
  c->glob_slot = slot ^ 1;
  c->sf2 = sf2; c->beta = beta; c->N_global = N_global; c->step = step;
  c->have_globals = true;
  c->state = 0;
  return GP_OK;
}

extern "C" int gp_phase1(gp_ctx* c) {
  if (!c) return GP_ERR_BAD_ARG;
  if (!c->have_data || !c->have_globals) return fail(c, GP_ERR_STATE, "gp_phase1 needs gp_upload_shard and gp_set_globals first");
  GP_HIP(c, hipSetDevice(c->device));
  GP_EV(c, 0);
  GP_TRY(run_prep_and_generate(c));
  if (!c->regime_A) GP_TRY(run_generate_b(c));
  GP_EV(c, 1);
  GP_TRY(run_phase1(c));
  if (!c->regime_A) GP_TRY(run_phase1_b(c));
  GP_EV(c, 2);
  c->state = 1;
  c->spack_filled = false;
  return GP_OK;
}

extern "C" int gp_stats_buffer(gp_ctx* c, void** dev_ptr, int64_t* n) {
  if (!c) return GP_ERR_BAD_ARG;
  if (dev_ptr) *dev_ptr = c->stats;
  if (n) *n = (int64_t)c->Mp * c->Mp + (int64_t)c->Mp * c->Dp + SC_COUNT;
  return GP_OK;
}

// ---- packed statistics for the all-reduce across processes: Psi2 upper triangle (row-major) | C [M][D] | scalars
// element e of the packed payload <-> its place in the padded statistics buffer (second index: the mirrored Psi2 element, or -1)
__device__ __forceinline__ bool spack_map(long e, int M, int Mp, int D, int Dp, long* k, long* s0, long* s1) {
  const long tri = (long)M * (M + 1) / 2, md = (long)M * D;
  *s1 = -1;
  if (e < (long)M * M) {
    const int i = (int)(e / M), j = (int)(e - (long)i * M);
    if (j < i) return false;
    *k = (long)i * M - (long)i * (i - 1) / 2 + (j - i);
    *s0 = (long)i * Mp + j; *s1 = (long)j * Mp + i;
  } else if (e < (long)M * M + md) {
    const long r = e - (long)M * M;
    const int i = (int)(r / D), d = (int)(r - (long)i * D);
    *k = tri + r; *s0 = (long)Mp * Mp + (long)i * Dp + d;
  } else {
    const long r = e - (long)M * M - md;
    *k = tri + md + r; *s0 = (long)Mp * Mp + (long)Mp * Dp + r;
  }
  return true;
}
__global__ void __launch_bounds__(256) stats_pack_kernel(const double* __restrict__ stats, double* __restrict__ pk, int M, int Mp, int D, int Dp) {
  for (long e = blockIdx.x * 256L + threadIdx.x; e < (long)M * M + (long)M * D + SC_COUNT; e += (long)gridDim.x * 256L) {
    long k, s0, s1;
    if (spack_map(e, M, Mp, D, Dp, &k, &s0, &s1)) pk[k] = stats[s0];
  }
}
__global__ void __launch_bounds__(256) stats_unpack_kernel(const double* __restrict__ pk, double* __restrict__ stats, int M, int Mp, int D, int Dp) {
  for (long e = blockIdx.x * 256L + threadIdx.x; e < (long)M * M + (long)M * D + SC_COUNT; e += (long)gridDim.x * 256L) {
    long k, s0, s1;
    if (spack_map(e, M, Mp, D, Dp, &k, &s0, &s1)) { const double v = pk[k]; stats[s0] = v; if (s1 >= 0) stats[s1] = v; }
  }
}
static int64_t spack_doubles(const gp_ctx* c) { return (int64_t)c->M * (c->M + 1) / 2 + (int64_t)c->M * c->D + SC_COUNT; }
static int ensure_spack(gp_ctx* c) {
  if (!c->spack) {
    GP_TRY(dalloc_bytes(c, (void**)&c->spack, (size_t)spack_doubles(c) * sizeof(double), DA_RAW));
    GP_HIP(c, hipMemsetAsync(c->spack, 0, (size_t)spack_doubles(c) * sizeof(double), c->stream));
  }
  return GP_OK;
}
extern "C" int gp_stats_packed_buffer(gp_ctx* c, void** dev_ptr, int64_t* n) {
  if (!c) return GP_ERR_BAD_ARG;
  GP_HIP(c, hipSetDevice(c->device));
  GP_TRY(ensure_spack(c));
  if (dev_ptr) *dev_ptr = c->spack;
  if (n) *n = spack_doubles(c);
  return GP_OK;
}
static int stats_pack(gp_ctx* c, int unpack_) {
  if (!c) return GP_ERR_BAD_ARG;
  if (c->state < 1) return fail(c, GP_ERR_STATE, "gp_stats_pack / gp_stats_unpack before gp_phase1");
  GP_HIP(c, hipSetDevice(c->device));
  GP_TRY(ensure_spack(c));
  const long n = (long)c->M * c->M + (long)c->M * c->D + SC_COUNT;
  if (unpack_) {
    // the packed buffer only holds statistics after a pack (it is zero from its allocation): unpacking first would silently wipe phase 1's sums
    if (!c->spack_filled) return fail(c, GP_ERR_STATE, "gp_stats_unpack before gp_stats_pack");
    hipLaunchKernelGGL(stats_unpack_kernel, dim3(blocks_for(n)), dim3(256), 0, c->stream, (const double*)c->spack, c->stats, c->M, c->Mp, c->D, c->Dp);
  } else {
    hipLaunchKernelGGL(stats_pack_kernel, dim3(blocks_for(n)), dim3(256), 0, c->stream, (const double*)c->stats, c->spack, c->M, c->Mp, c->D, c->Dp);
    c->spack_filled = true;
  }
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}
extern "C" int gp_stats_pack(gp_ctx* c) { return stats_pack(c, 0); }
extern "C" int gp_stats_unpack(gp_ctx* c) { return stats_pack(c, 1); }

extern "C" int gp_grads_buffer(gp_ctx* c, void** dev_ptr, int64_t* n) {
  if (!c) return GP_ERR_BAD_ARG;
  if (dev_ptr) *dev_ptr = c->grads;
  if (n) *n = (int64_t)c->M * c->Q + c->Q;
  return GP_OK;
}

__global__ void combine_kernel(double* dst, const double* src, long n, int op) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) dst[i] = (op == 0 ? dst[i] : 0.0) + src[i];
}
__global__ void grad_latest_kernel(const double* __restrict__ gmu, const double* __restrict__ gS, const double* __restrict__ Xs,
                                   const double* __restrict__ dir, long N, int Q, double step, int raw, int have_dir, double* __restrict__ out) {
  const long nq = N * Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nq; i += (long)gridDim.x * 256L) {
    out[i] = -gmu[i];
    double g = gS[i];
    if (raw) {
      double x = Xs[i];
      if (have_dir && step != 0.0) x += step * dir[nq + i];
      g *= 1.0 / (exp(-x) + 1.0);      // transformVar_grad, supporting_functions.py:165-168
    }
    out[nq + i] = -g;
  }
}

static bool g_force_staging = false;   // test hook: take the cross-device path of gp_buffer_combine on one device
extern "C" int gp_debug_force_staging(int on) { g_force_staging = on != 0; return GP_OK; }

extern "C" int gp_buffer_combine(gp_ctx* dst, const gp_ctx* src, int which, int op) {
  if (!dst || !src) return GP_ERR_BAD_ARG;
  if (dst->M != src->M || dst->Q != src->Q || dst->D != src->D) return fail(dst, GP_ERR_BAD_ARG, "gp_buffer_combine: shape mismatch");
  const long n = which == 0 ? (long)dst->Mp * dst->Mp + (long)dst->Mp * dst->Dp + SC_COUNT : (long)dst->M * dst->Q + dst->Q;
  const double* from = which == 0 ? src->stats : src->grads;
  // the source context's work must have finished before its buffer is read from another stream / device
  if (src->stream != dst->stream || src->device != dst->device) {
    GP_HIP(dst, hipSetDevice(src->device));
    GP_HIP(dst, hipStreamSynchronize(src->stream));
  }
  GP_HIP(dst, hipSetDevice(dst->device));
  if (src->device != dst->device || g_force_staging) {
    // shards on different GPUs of one process (options['devices']): peer copy into a staging buffer on the destination
    // device, then the same combine kernel -- the device-side form of statistics_reducer (local_MapReduce.py:250-277)
    if (dst->staging_doubles < (size_t)n) {
      if (dst->staging) (void)hipFree(dst->staging);
      dst->staging = nullptr; dst->staging_doubles = 0;
      GP_TRY(dalloc_bytes(dst, (void**)&dst->staging, (size_t)n * 8, DA_RAW));
      dst->staging_doubles = (size_t)n;
    }
    GP_HIP(dst, hipMemcpyPeerAsync(dst->staging, dst->device, from, src->device, (size_t)n * 8, dst->stream));
    from = dst->staging;
  }
  hipLaunchKernelGGL(combine_kernel, dim3(blocks_for(n)), dim3(256), 0, dst->stream, which == 0 ? dst->stats : dst->grads, from, n, op);
  GP_HIP(dst, hipGetLastError());
  if (which == 0 && dst->state < 1) dst->state = 1;
  return GP_OK;
}

extern "C" int gp_scale_buffer(gp_ctx* c, int which, double f) {
  if (!c) return GP_ERR_BAD_ARG;
  if (which != 0 && which != 1) return fail(c, GP_ERR_BAD_ARG, "gp_scale_buffer: which must be 0 (statistics) or 1 (gradient sums)");
  if (which == 0 && c->state < 1) return fail(c, GP_ERR_STATE, "gp_scale_buffer(statistics) before gp_phase1");
  if (which == 1 && c->state < 3) return fail(c, GP_ERR_STATE, "gp_scale_buffer(gradient sums) before gp_phase2");
  GP_HIP(c, hipSetDevice(c->device));
  const long n = which == 0 ? (long)c->Mp * c->Mp + (long)c->Mp * c->Dp + SC_COUNT : (long)c->M * c->Q + c->Q;
  if (which == 0) c->spack_filled = false;   // the padded buffer is the source of truth: a later unpack needs a new pack
  if (f == 0.0) {
    // a dropped shard: the reference never loads its files (local_MapReduce.py:119-129) -- a memset, so that non-finite values in
    // the dropped shard's sums (0 * inf = nan) cannot reach the reduction
    GP_HIP(c, hipMemsetAsync(which == 0 ? c->stats : c->grads, 0, (size_t)n * sizeof(double), c->stream));
    return GP_OK;
  }
  hipLaunchKernelGGL(scale_kernel, dim3(blocks_for(n)), dim3(256), 0, c->stream, which == 0 ? c->stats : c->grads, n, f);
  GP_HIP(c, hipGetLastError());
  return GP_OK;
}

extern "C" int gp_scale_stats(gp_ctx* c, double f) { return gp_scale_buffer(c, 0, f); }

extern "C" int gp_global_step_jitter(gp_ctx* c, int jitter_mask) {
  if (!c) return GP_ERR_BAD_ARG;
  if (c->state < 1) return fail(c, GP_ERR_STATE, "gp_global_step before gp_phase1 / gp_set_local_statistics");
  if (jitter_mask < 0 || jitter_mask > 3) return fail(c, GP_ERR_BAD_ARG, "gp_global_step_jitter: mask must be 0..3");
  GP_HIP(c, hipSetDevice(c->device));
  c->jitter_mask = jitter_mask;
  GP_EV(c, 3);
  const int rc_gs = run_global_step(c);
  GP_EV(c, 4);
  if (rc_gs != GP_OK) return rc_gs;
  c->state = 2;
  return GP_OK;
}

extern "C" int gp_global_step(gp_ctx* c) { return gp_global_step_jitter(c, 0); }

// The int8 guard's decision (p1i8.hip) for an evaluation that ran both phase-1 paths.  It needs P = (K_mm + beta Psi2)^-1 of a global step that SUCCEEDED:
// a step that failed or asks for the jitter retry leaves P non-finite and would reject the int8 path for a reason that has nothing to do with its
// accuracy -- the check then stays pending (guard 0: the next evaluation runs both paths again).  Called wherever an evaluation's global step is
// known to be over: gp_finish, gp_global_status, gp_download of a global-step array.
static int resolve_i8_check(gp_ctx* c) {
  if (!c->i8_check_pending || c->state < 2) return GP_OK;
  if (check_global(c) != GP_OK) return GP_OK;     // the caller reports that status itself
  return p1i8_check_finish(c);
}

extern "C" int gp_global_status(gp_ctx* c, int* retry_mask) {
  if (!c) return GP_ERR_BAD_ARG;
  if (c->state < 2) return fail(c, GP_ERR_STATE, "gp_global_status before gp_global_step");
  GP_HIP(c, hipSetDevice(c->device));
  GP_TRY(resolve_i8_check(c));
  const int rc = check_global(c);
  if (retry_mask) *retry_mask = (rc == GP_RETRY_JITTER) ? c->retry_mask : 0;
  return rc;
}

extern "C" int gp_phase2(gp_ctx* c, int want_embedding_grads) {
  if (!c) return GP_ERR_BAD_ARG;
  if (c->state < 2) return fail(c, GP_ERR_STATE, "gp_phase2 before gp_global_step");
  if (!c->have_data) return fail(c, GP_ERR_STATE, "gp_phase2 without shard data");
  GP_HIP(c, hipSetDevice(c->device));
  GP_EV(c, 5);
  if ((want_embedding_grads != 0) != c->want_emb) {
    // the per-point feature matrix depends on the mode: rebuild the trial point
    c->want_emb = want_embedding_grads != 0;
    GP_TRY(run_prep_and_generate(c));
  }
  GP_TRY(run_phase2(c));
  if (!c->regime_A) GP_TRY(run_phase2_b(c));
  if (c->want_emb && c->xs_raw) {
    // the .grad_latest vector of this evaluation, resident for the optimiser's dot products
    const long nq = (long)c->N * c->Q;
    hipLaunchKernelGGL(grad_latest_kernel, dim3(blocks_for(nq)), dim3(256), 0, c->stream, c->gXmu, c->gXs, c->Xs, c->dir, (long)c->N, c->Q, c->step,
                       1, c->have_dir ? 1 : 0, c->g_latest);
    GP_HIP(c, hipGetLastError());
    c->have_glatest = true;
  }
  GP_EV(c, 6);
  c->state = 3;
  return GP_OK;
}

extern "C" int gp_last_timings(gp_ctx* c, double* out8) {
  double* out5 = out8;
  if (!c || !out8) return GP_ERR_BAD_ARG;
  GP_HIP(c, hipSetDevice(c->device));
  GP_HIP(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < 8; ++i) out8[i] = 0.0;
  float ms;
  ++c->sync_epoch;
  if (c->timing == 1) {       // only the evaluation's first and last event were recorded
    if (c->state >= 3 && hipEventElapsedTime(&ms, c->ev[0], c->ev[6]) == hipSuccess) out5[4] = ms;
    return GP_OK;
  }
  if (c->timing == 0) return GP_OK;
  if (c->state >= 1 && hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) out5[0] = ms;
  if (c->state >= 1 && hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) out5[1] = ms;
  if (c->state >= 2 && hipEventElapsedTime(&ms, c->ev[3], c->ev[4]) == hipSuccess) out5[2] = ms;
  if (c->state >= 3 && hipEventElapsedTime(&ms, c->ev[5], c->ev[6]) == hipSuccess) out5[3] = ms;
  out5[4] = out5[0] + out5[1] + out5[2] + out5[3];
  if (c->state >= 1 && hipEventElapsedTime(&ms, c->ev[8], c->ev[9]) == hipSuccess) out8[5] = ms;
  if (c->state >= 1 && hipEventElapsedTime(&ms, c->ev[10], c->ev[11]) == hipSuccess) out8[6] = ms;
  if (c->state >= 3 && hipEventElapsedTime(&ms, c->ev[12], c->ev[13]) == hipSuccess) out8[7] = ms;
  return GP_OK;
}

extern "C" int gp_download(gp_ctx* c, int which, double* dst, int64_t n) {
  if (!c || !dst) return GP_ERR_BAD_ARG;
  GP_HIP(c, hipSetDevice(c->device));
  const long M = c->M, Mp = c->Mp, D = c->D, Dp = c->Dp, N = c->N, Q = c->Q;
  if (c->state >= 2 && (which == GP_ARR_KMM_INV || which == GP_ARR_KMM_PLUS_OP_INV || which == GP_ARR_DF_DKMM || which == GP_ARR_DF_DPSI1TY ||
                        which == GP_ARR_DF_DPSI2 || which == GP_ARR_SCALARS)) {
    GP_TRY(resolve_i8_check(c));
    GP_TRY(check_global(c));
  }
  switch (which) {
    case GP_ARR_PSI1: return download_matrix(c, c->Kaug, c->LDK, N, M, dst, n);
    case GP_ARR_PSI2_SUM: return download_matrix(c, c->stats, Mp, M, M, dst, n);
    case GP_ARR_PSI1TY: return download_matrix(c, c->stats + Mp * Mp, Dp, M, D, dst, n);
    case GP_ARR_KMM: return download_matrix(c, c->KmmKeep, Mp, M, M, dst, n);
    case GP_ARR_KMM_INV: return download_matrix(c, c->Inv, Mp, M, M, dst, n);
    case GP_ARR_KMM_PLUS_OP_INV: return download_matrix(c, c->Inv + Mp * Mp, Mp, M, M, dst, n);
    case GP_ARR_DF_DKMM: return download_matrix(c, c->dFdK, Mp, M, M, dst, n);
    case GP_ARR_DF_DPSI1TY: return download_matrix(c, c->Abar, Dp, M, D, dst, n);
    case GP_ARR_DF_DPSI2: return download_matrix(c, c->Bbar, Mp, M, M, dst, n);
    case GP_ARR_GRAD_X_MU: return download_matrix(c, c->gXmu, Q, N, Q, dst, n);
    case GP_ARR_GRAD_X_S: return download_matrix(c, c->gXs, Q, N, Q, dst, n);
    case GP_ARR_X_MU_TRIAL: return download_matrix(c, c->mu, Q, N, Q, dst, n);
    case GP_ARR_X_S_TRIAL: return download_matrix(c, c->S, Q, N, Q, dst, n);
    case GP_ARR_GRAD_LATEST: {
      if (c->state < 3 || !c->want_emb) return fail(c, GP_ERR_STATE, "GP_ARR_GRAD_LATEST needs gp_phase2(ctx, 1) first");
      if (n != 2 * N * Q) return fail(c, GP_ERR_BAD_ARG, "GP_ARR_GRAD_LATEST wants %ld doubles", 2 * N * Q);
      if (c->regime_A) return fail(c, GP_ERR_NON_FINITE, "grad_X_S with X_S == 0 (1/S, partial_terms.py:417)");
      double* tmp = nullptr;
      GP_HIP(c, hipMalloc((void**)&tmp, 2 * N * Q * 8));
      hipLaunchKernelGGL(grad_latest_kernel, dim3(blocks_for(N * Q)), dim3(256), 0, c->stream, c->gXmu, c->gXs, c->Xs, c->dir, N, (int)Q, c->step,
                         c->xs_raw ? 1 : 0, c->have_dir ? 1 : 0, tmp);
      hipError_t e = hipMemcpyAsync(dst, tmp, 2 * N * Q * 8, hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      (void)hipFree(tmp);
      if (e != hipSuccess) return fail(c, GP_ERR_HIP, "download failed: %s", hipGetErrorString(e));
      return GP_OK;
    }
    case GP_ARR_SCALARS: {
      if (n != 8) return fail(c, GP_ERR_BAD_ARG, "GP_ARR_SCALARS wants 8 doubles");
      double sc[SC_COUNT];
      GP_HIP(c, hipMemcpyAsync(sc, c->stats + Mp * Mp + Mp * Dp, sizeof(sc), hipMemcpyDeviceToHost, c->stream));
      GP_HIP(c, hipStreamSynchronize(c->stream));
      dst[0] = sc[SC_SUM_YYT]; dst[1] = sc[SC_PSI0]; dst[2] = sc[SC_KL];
      dst[3] = c->h_gs[GS_LOGDET_K]; dst[4] = c->h_gs[GS_LOGDET_A]; dst[5] = c->h_gs[GS_F]; dst[6] = c->h_gs[GS_GRAD_BETA]; dst[7] = c->h_gs[GS_GRAD_SF2];
      return GP_OK;
    }
    case GP_ARR_PSI2_POINTS: case GP_ARR_DKMM_DZ: case GP_ARR_DPSI1TY_DZ: case GP_ARR_DPSI2_DZ: case GP_ARR_DKMM_DALPHA:
    case GP_ARR_DPSI1TY_DALPHA: case GP_ARR_DPSI2_DALPHA: {
      double* buf = nullptr; long cnt = 0;
      GP_TRY(compat_build(c, which, &buf, &cnt));
      int rc = GP_OK;
      if (cnt != n) rc = fail(c, GP_ERR_BAD_ARG, "gp_download: expected %ld doubles, got %ld", cnt, (long)n);
      else {
        hipError_t e = hipMemcpyAsync(dst, buf, cnt * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, GP_ERR_HIP, "download failed: %s", hipGetErrorString(e));
      }
      (void)hipFree(buf);
      return rc;
    }
    default: return fail(c, GP_ERR_UNSUPPORTED, "gp_download: array %d not available", which);
  }
}

extern "C" int gp_set_local_statistics(gp_ctx* c, double sum_YYT, const double* Psi2, const double* C, double sum_exp_K_ii, double KL) {
  if (!c || !Psi2 || !C) return GP_ERR_BAD_ARG;
  if (!c->have_globals) return fail(c, GP_ERR_STATE, "gp_set_local_statistics before gp_set_globals");
  GP_HIP(c, hipSetDevice(c->device));
  const long M = c->M, Mp = c->Mp, D = c->D, Dp = c->Dp;
  double* tmp = c->T2;
  GP_HIP(c, hipMemcpyAsync(tmp, Psi2, M * M * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(scatter2d_kernel, dim3(blocks_for(Mp * Mp)), dim3(256), 0, c->stream, tmp, M, M, c->stats, Mp, Mp, Mp);
  GP_HIP(c, hipStreamSynchronize(c->stream));
  GP_HIP(c, hipMemcpyAsync(tmp, C, M * D * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(scatter2d_kernel, dim3(blocks_for(Mp * Dp)), dim3(256), 0, c->stream, tmp, M, D, c->stats + Mp * Mp, Dp, Mp, Dp);
  double sc[SC_COUNT] = {0};
  sc[SC_SUM_YYT] = sum_YYT; sc[SC_PSI0] = sum_exp_K_ii; sc[SC_KL] = KL; sc[SC_NLOCAL] = sum_exp_K_ii / c->sf2;
  GP_HIP(c, hipMemcpyAsync(c->stats + Mp * Mp + Mp * Dp, sc, sizeof(sc), hipMemcpyHostToDevice, c->stream));
  GP_HIP(c, hipStreamSynchronize(c->stream));
  c->spack_filled = false;     // the packed payload of an earlier evaluation no longer describes these statistics
  if (c->state < 1) c->state = 1;
  return GP_OK;
}

// ---- final gradients ---------------------------------------------------------------------------------------------
// final = Kmm parts (global step) + all-reduced data parts (phase 2)
__global__ void add_kernel(const double* a, const double* b, double* out, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) out[i] = a[i] + b[i];
}
// gp_finish's only launch: out = [global-step scalars and failure flags (ngs) | Kmm parts + data parts of grad_Z, grad_alpha (n)], written straight
// into pinned host memory -- one kernel and one stream synchronisation instead of copy, synchronise, kernel, copy, synchronise (r04: the second
// round trip and the two blit dispatches were ~0.1 ms of wall time per evaluation at configs[1]'s size, profiles/r05_config1_timeline.txt)
__global__ void finish_kernel(const double* __restrict__ gs, int ngs, const double* __restrict__ a, const double* __restrict__ b, long n,
                              double* __restrict__ out) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < ngs + n; i += (long)gridDim.x * 256L) out[i] = i < ngs ? gs[i] : a[i - ngs] + b[i - ngs];
}

extern "C" int gp_finish(gp_ctx* c, double* F, double* grad_Z, double* grad_sf2, double* grad_alpha, double* grad_beta) {
  if (!c) return GP_ERR_BAD_ARG;
  if (c->state < 2) return fail(c, GP_ERR_STATE, "gp_finish before gp_global_step");
  GP_HIP(c, hipSetDevice(c->device));
  const bool want_grads = grad_Z || grad_alpha;
  GP_TRY(resolve_i8_check(c));      // this evaluation ran both phase-1 paths: decide whether the context stays on int8 (only behind a successful global step)
  if (want_grads && c->state < 3) {
    GP_TRY(check_global(c));   // a failed global step is the more useful message
    return fail(c, GP_ERR_STATE, "gp_finish: gradients requested before gp_phase2");
  }
  if (!want_grads) {
    GP_TRY(check_global(c));
  } else {
    // the one host synchronisation of an evaluation: scalars + failure flags of the global step and the final gradients in one mapped buffer
    const long n = (long)c->M * c->Q + c->Q;
    constexpr int ngs = GS_COUNT + 8;
    if (!c->h_out) GP_HIP(c, hipHostMalloc((void**)&c->h_out, (size_t)(ngs + n) * sizeof(double), hipHostMallocMapped));
    double* dout = nullptr;
    GP_HIP(c, hipHostGetDevicePointer((void**)&dout, c->h_out, 0));
    hipLaunchKernelGGL(finish_kernel, dim3(blocks_for(ngs + n)), dim3(256), 0, c->stream, c->gs, ngs, c->gK, c->grads, n, dout);
    GP_HIP(c, hipGetLastError());
    GP_HIP(c, hipStreamSynchronize(c->stream));
    ++c->sync_epoch;
    GP_TRY(check_global_from(c, c->gs_pending ? c->h_out : nullptr));
    if (grad_Z) memcpy(grad_Z, c->h_out + ngs, (size_t)c->M * c->Q * 8);
    if (grad_alpha) memcpy(grad_alpha, c->h_out + ngs + (size_t)c->M * c->Q, (size_t)c->Q * 8);
  }
  if (F) *F = c->h_gs[GS_F];
  if (grad_sf2) *grad_sf2 = c->h_gs[GS_GRAD_SF2];
  if (grad_beta) *grad_beta = c->h_gs[GS_GRAD_BETA];
  return GP_OK;
}
// ---- resident CG vectors (scg_adapted_local_MapReduce.py:29-243) --------------------------------------------------------
// which: 0 d = -g_new (reset_d :160-173) | 1 d = a*d - g_new (update_d :175-189) | 2 X += a*d (update_X :191-214)
//        3 g_old = g_new (:216-229) | 4 g_new = g_latest (:231-243) | 5 set_grads: g_new = g_old = g_latest, d = -g_latest (:29-55)
__global__ void cg_update_kernel(int which, double a, long nq, double* __restrict__ d, double* __restrict__ gnew, double* __restrict__ gold,
                                 const double* __restrict__ glatest, double* __restrict__ Xmu, double* __restrict__ Xs) {
  const long n2 = 2 * nq;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n2; i += (long)gridDim.x * 256L) {
    switch (which) {
      case 0: d[i] = -gnew[i]; break;
      case 1: d[i] = a * d[i] - gnew[i]; break;
      case 2: if (i < nq) Xmu[i] += a * d[i]; else Xs[i - nq] += a * d[i]; break;
      case 3: gold[i] = gnew[i]; break;
      case 4: gnew[i] = glatest[i]; break;
      default: { const double g = glatest[i]; gnew[i] = g; gold[i] = g; d[i] = -g; } break;
    }
  }
}
// out[0..4] += (g_new.d, d.d, d.(g_latest - g_new), g_new.g_new, g_new.g_old); out[5] = max(out[5], max |d|)
__global__ void __launch_bounds__(256) cg_dots_kernel(long n2, const double* __restrict__ d, const double* __restrict__ gnew,
                                                      const double* __restrict__ gold, const double* __restrict__ glatest, double* part) {
  __shared__ double red[6][256];
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n2; i += (long)gridDim.x * 256L) {
    const double di = d[i], gn = gnew[i];
    s[0] += gn * di; s[1] += di * di; s[2] += di * (glatest[i] - gn); s[3] += gn * gn; s[4] += gn * gold[i];
    s[5] = fmax(s[5], fabs(di));
  }
  for (int k = 0; k < 6; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      for (int k = 0; k < 5; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + w];
      red[5][threadIdx.x] = fmax(red[5][threadIdx.x], red[5][threadIdx.x + w]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0];
}

static int cg_ready(gp_ctx* c, const char* what) {
  if (!c->have_data) return fail(c, GP_ERR_STATE, "%s: no shard data", what);
  if (!c->xs_raw) return fail(c, GP_ERR_STATE, "%s: the resident CG vectors exist only for free embeddings (raw variances)", what);
  return GP_OK;
}

extern "C" int gp_cg_update(gp_ctx* c, int which, double a) {
  if (!c) return GP_ERR_BAD_ARG;
  if (which < 0 || which > 5) return fail(c, GP_ERR_BAD_ARG, "gp_cg_update: which must be 0..5");
  GP_TRY(cg_ready(c, "gp_cg_update"));
  if ((which == 4 || which == 5) && !c->have_glatest) return fail(c, GP_ERR_STATE, "gp_cg_update: no grad_latest yet (gp_phase2(ctx, 1) first)");
  GP_HIP(c, hipSetDevice(c->device));
  const long nq = (long)c->N * c->Q;
  hipLaunchKernelGGL(cg_update_kernel, dim3(blocks_for(2 * nq)), dim3(256), 0, c->stream, which, a, nq, c->dir, c->g_new, c->g_old, c->g_latest,
                     c->Xmu, c->Xs);
  GP_HIP(c, hipGetLastError());
  if (which == 0 || which == 1 || which == 5) c->have_dir = true;
  if (which == 2) { c->state = 0; c->prep_fixa_valid = false; }   // the embeddings moved: statistics are stale
  return GP_OK;
}

extern "C" int gp_cg_set_grads(gp_ctx* c) { return gp_cg_update(c, 5, 0.0); }

static int cg_reduce(gp_ctx* c, double* out6) {
  const long n2 = 2L * c->N * c->Q;
  const int nb = std::min(blocks_for(n2), 1024);
  double* part = c->klpart + c->kl_blocks;   // spare tail (8192 doubles)
  hipLaunchKernelGGL(cg_dots_kernel, dim3(nb), dim3(256), 0, c->stream, n2, c->dir, c->g_new, c->g_old, c->g_latest, part);
  std::vector<double> h((size_t)nb * 6);
  GP_HIP(c, hipMemcpyAsync(h.data(), part, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
  GP_HIP(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < 6; ++k) out6[k] = 0.0;
  for (int b = 0; b < nb; ++b) {
    for (int k = 0; k < 5; ++k) out6[k] += h[(size_t)b * 6 + k];
    out6[5] = std::max(out6[5], h[(size_t)b * 6 + 5]);
  }
  return GP_OK;
}

extern "C" int gp_cg_dots(gp_ctx* c, double* out6) {
  if (!c || !out6) return GP_ERR_BAD_ARG;
  GP_TRY(cg_ready(c, "gp_cg_dots"));
  if (!c->have_dir) return fail(c, GP_ERR_STATE, "gp_cg_dots: no search direction (gp_cg_set_grads first)");
  GP_HIP(c, hipSetDevice(c->device));
  return cg_reduce(c, out6);
}

// out[0] += sum |g_new| ; out[1] = max |g_new|   (gd_local_MapReduce.py:38-61)
__global__ void __launch_bounds__(256) cg_abs_kernel(long n2, const double* __restrict__ gnew, double* part) {
  __shared__ double red[2][256];
  double s = 0.0, m = 0.0;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n2; i += (long)gridDim.x * 256L) { const double a = fabs(gnew[i]); s += a; m = fmax(m, a); }
  red[0][threadIdx.x] = s; red[1][threadIdx.x] = m;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) { red[0][threadIdx.x] += red[0][threadIdx.x + w]; red[1][threadIdx.x] = fmax(red[1][threadIdx.x], red[1][threadIdx.x + w]); }
    __syncthreads();
  }
  if (threadIdx.x < 2) part[blockIdx.x * 2 + threadIdx.x] = red[threadIdx.x][0];
}

extern "C" int gp_cg_abs(gp_ctx* c, double* out2) {
  if (!c || !out2) return GP_ERR_BAD_ARG;
  GP_TRY(cg_ready(c, "gp_cg_abs"));
  if (!c->have_dir) return fail(c, GP_ERR_STATE, "gp_cg_abs: no gradient vectors (gp_cg_set_grads first)");
  GP_HIP(c, hipSetDevice(c->device));
  const long n2 = 2L * c->N * c->Q;
  const int nb = std::min(blocks_for(n2), 1024);
  double* part = c->klpart + c->kl_blocks;   // spare tail (8192 doubles)
  hipLaunchKernelGGL(cg_abs_kernel, dim3(nb), dim3(256), 0, c->stream, n2, c->g_new, part);
  std::vector<double> h((size_t)nb * 2);
  GP_HIP(c, hipMemcpyAsync(h.data(), part, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
  GP_HIP(c, hipStreamSynchronize(c->stream));
  out2[0] = 0.0; out2[1] = 0.0;
  for (int b = 0; b < nb; ++b) { out2[0] += h[(size_t)b * 2]; out2[1] = std::max(out2[1], h[(size_t)b * 2 + 1]); }
  return GP_OK;
}

extern "C" int gp_cg_max_d(gp_ctx* c, double alpha, double* out) {
  if (!c || !out) return GP_ERR_BAD_ARG;
  GP_TRY(cg_ready(c, "gp_cg_max_d"));
  GP_HIP(c, hipSetDevice(c->device));
  double o[6];
  GP_TRY(cg_reduce(c, o));
  *out = std::fabs(alpha) * o[5];
  return GP_OK;
}
