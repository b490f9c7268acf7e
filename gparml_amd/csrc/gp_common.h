// Shared host-side declarations of the gparml HIP library (context, error handling, launch helpers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstdarg>
#include <atomic>
#include <string>
#include <vector>
#include "../../include/gparml_hip.h"
#include "mma_f64.h"

namespace gp {

#ifndef GP_I8_DIGITS
#define GP_I8_DIGITS 6     // signed 7-bit digits per operand of the int8 phase-1 prototype (p1i8.hip): 42 bits below the operand's scale
#endif

inline long round_up(long x, long m) { return (x + m - 1) / m * m; }
// latent width of the packed per-point records of psi1_kernel (Q <= 16: Q rounded up to 2) / psi1_wide_kernel (24, 32, 52, 64); 0: none
inline int psi1_qp(int Q) { return Q <= 16 ? (Q + 1) / 2 * 2 : Q <= 24 ? 24 : Q <= 32 ? 32 : Q <= 52 ? 52 : Q <= 64 ? 64 : 0; }

// dense batched GEMM on tile-aligned buffers: C = alpha * op(A) op(B) + beta * C
struct GemmP {
  const double* A;
  const double* B;
  double* C;
  long lda, ldb, ldc;
  long sA, sB, sC;  // batch strides (doubles) of the inner batch index
  int K;            // multiple of KC
  double alpha, beta;
  int tri;          // 0 all tiles, 1 only tiles with row-tile >= col-tile (lower), 2 only upper
  int inner = 1 << 30;  // blockIdx.z = i + inner * o: inner index i uses sA/sB/sC, outer index o uses oA/oB/oC
  long oA = 0, oB = 0, oC = 0;
  int splits = 1;       // split-k: partial tiles go to ws, gemm_splitk_reduce applies alpha/beta (small grids only)
  double* ws = nullptr;
  // X^T X with X lower-triangular (both operands FREE_CONTIG, stored [k][free], zero for k < free): with tri = 1 only the tiles on or below the diagonal
  // are computed, klow = 1 starts their k-range at the tile's first possibly non-zero k = max(row0, col0) (the skipped products are exact zeros:
  // same bits), mirror = 1 stores every off-diagonal tile a second time transposed (the two halves were equal bit for bit before: same k order, and a
  // product does not depend on which factor is the A operand)
  int klow = 0, mirror = 0;
  int big = 0;          // 1: the 128 x 128-tile kernel (with splits) whatever the tile count (the M x M x M products at M >= 1024)
};
// m, n multiples of TILE; la/lb: Layout of A (free index = rows of C) and B (free index = cols of C)
void launch_gemm(hipStream_t st, Layout la, Layout lb, int m, int n, int batch, const GemmP& p);

// index of the scalars at the tail of the packed statistics buffer
enum { SC_SUM_YYT = 0, SC_PSI0 = 1, SC_KL = 2, SC_NLOCAL = 3, SC_COUNT = 8 };
// device scalars produced by the global step (GP_ARR_SCALARS order after the first three)
enum { GS_LOGDET_K = 0, GS_LOGDET_A = 1, GS_F = 2, GS_GRAD_BETA = 3, GS_GRAD_SF2 = 4, GS_FAIL = 5, GS_TR_KIPSI2 = 6, GS_TR_PPSI2 = 7,
       GS_TR_CE = 8, GS_TR_EPSI2E = 9, GS_SUM_V = 10, GS_SUM_AC = 11, GS_SUM_BPSI2 = 12, GS_COUNT = 16 };

}  // namespace gp

struct gp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // sizes
  int64_t N = 0;   // local shard rows
  int64_t Np = 0;  // padded to TILE
  int D = 0, M = 0, Q = 0;
  int Mp = 0, Dp = 0, LDK = 0;  // padded M, D (multiples of TILE) and the row stride of Kaug = Mp + Dp
  int CX = 0, CXp = 0;          // per-point feature columns [f1(Q), f2(Q), 1], padded to 4
  int CZ = 0, CZp = 0;          // inducing feature columns [1, Z(Q), Z^2(Q)], padded to 4
  int64_t N_global = 0;
  double sf2 = 1, beta = 1, step = 0;
  bool regime_A = true;   // every variance exactly zero (fixed embeddings)
  bool xs_raw = false;    // X_S stored in softplus-inverse space
  bool have_data = false, have_globals = false, have_dir = false, have_glatest = false;
  int state = 0;          // 0 nothing, 1 phase1 done, 2 stats final (global step done), 3 phase2 done
  bool want_emb = false;
  bool prep_fixa_valid = false;   // regime A, fixed embeddings: the prep kernels' outputs (mu, features, records) are current

  // ---- device buffers ----
  double* Kaug = nullptr;     // [Np][LDK]  Psi1 | Y
  double* Xmu = nullptr;      // [N][Q] base means
  double* Xs = nullptr;       // [N][Q] base variances (raw or actual)
  double* dir = nullptr;      // [2][N][Q] search direction
  double* mu = nullptr;       // [Np][Q] trial means
  double* S = nullptr;        // [Np][Q] trial variances (actual)
  double* U = nullptr;        // [Np][Q] u = alpha / (alpha S + 1)
  double* PU = nullptr;       // [Np][2*QP+2] packed [mu | u | ln c1] rows for psi1_kernel (QP = Q rounded up to 2, <= 16)
  double* lnc1 = nullptr;     // [Np] ln(sf2) - 1/2 sum ln(a S + 1)
  double* Xa = nullptr;       // [Np][CXp] per-point features for the n-contraction
  double* Z = nullptr;        // [Mp][Q] (rows >= M zero)
  double* alpha = nullptr;    // [Q]
  double* Zaug = nullptr;     // [Mp][CZp]
  double* Zt = nullptr;       // [Q][Mp] the inducing points transposed (kmm_grads_lds_kernel: lanes = inducing points)
  int8_t* gsd = nullptr;      // gsi8.hip (M >= 1024): digit planes of the global step's two double-double-grade products on the int8 matrix core
  size_t gsd_bytes = 0;
  double* gss = nullptr;      // their column scales
  size_t gss_count = 0;
  double* stats = nullptr;    // packed: Psi2 [Mp*Mp] | C [Mp*Dp] | scalars [SC_COUNT]
  bool stats_external = false;
  double* spack = nullptr;    // Psi2 upper triangle | C [M][D] | scalars: the all-reduce payload across processes (allocated on first use)
  double* grads = nullptr;    // packed: gZ_data [M*Q] | galpha_data [Q]
  bool spack_filled = false;  // gp_stats_pack has run since the last gp_phase1 (gp_stats_unpack refuses to run before it)
  bool grads_external = false;
  double* staging = nullptr;  // landing buffer for a peer copy from a shard on another device (gp_buffer_combine)
  size_t staging_doubles = 0;
  double* part = nullptr;     // phase-1 split-k partials
  size_t part_doubles = 0;
  int* tiles = nullptr;       // phase-1 tile table (int2)
  int n_tiles = 0, p1_slices = 0, p1_cps = 0;
  void* p1plan = nullptr;     // regime-A phase-1 plan (job and output tables of p1v2.hip), built on first use
  void* i8plan = nullptr;     // int8 phase 1 (p1i8.hip): digit buffers, job tables; built on first use
  bool i8_active = false;     // this evaluation's phase 1 runs on the int8 matrix core (psi1_kernel wrote the digits)
  bool i8_y_valid = false;    // Y's digits are current (reset by gp_upload_shard)
  bool i8_unsupported = false;  // the int8 plan could not be built for this context (falls back to the float64 kernels)
  // the int8 path's run-time guard (p1i8.hip, "guard"): 0 = not checked since the last upload, 1 = accepted, 2 = rejected (float64 from then on)
  int i8_guard = 0;
  bool i8_check_pending = false;   // this evaluation ran both phase-1 paths: gp_finish reads the comparison and decides
  long i8_since_check = 0, i8_checks = 0;
  double i8_rel_psi2 = 0, i8_rel_c = 0, i8_cond_lb = 0;
  double* i8_cmp = nullptr;        // device: [4] squared Frobenius norms (dPsi2, Psi2, dC, C) | [2] max diag(Psi2), max diag(P) | partials
  int* bmap = nullptr;        // phase-1 block -> (slice, tile type) placement table
  int bmap_T = -1, bmap_S = -1, bmap_blocks = 0;
  double* klpart = nullptr;   // [blocks] partial KL sums
  int kl_blocks = 0;
  double sumYY = 0;           // host copy, computed at upload
  // global step
  double* Kmm = nullptr;      // batch of 2: [Kmm ; A] -> factorised in place into [Lk ; La]
  double* Lmat = nullptr;     // [2][Mp][Mp] Cholesky factors
  double* Linv = nullptr;     // [2][Mp][Mp] inverse factors
  double* Inv = nullptr;      // [2][Mp][Mp] Ki, P
  double* KmmKeep = nullptr;  // [Mp][Mp] Kmm (kept for downloads / derivative parts)
  double* T1 = nullptr;       // [Mp][Mp] scratch
  double* T2 = nullptr;       // [Mp][Mp] scratch
  double* dFdK = nullptr;     // [Mp][Mp]
  double* Bbar = nullptr;     // [Mp][Mp]
  double* Bbar4 = nullptr;    // free embeddings, Q <= 16: Bbar with four ROWS interleaved, element (m, m') at ((m / 4) Mp + m') 4 + m % 4 (csrc/psi2.hip)
  double* E = nullptr;        // [Mp][Dp]
  double* PsiE = nullptr;     // [Mp][Dp]
  double* Abar = nullptr;     // [Mp][Dp]
  double* Bm = nullptr;       // [LDK][Mp] = [2 Bbar ; Abar^T]
  double* gs = nullptr;       // [GS_COUNT] device scalars
  double* gK = nullptr;       // [M*Q + Q] Kmm-parts of grad_Z / grad_alpha (+ regime-B alpha term)
  double h_gs[gp::GS_COUNT] = {0};
  bool gs_pending = false;    // a global step was enqueued and its scalars / failure flags have not been read back yet
  int gs_status = 0;          // outcome of the last global step once read back (GP_OK, GP_ERR_NOT_PD, GP_ERR_NON_FINITE, GP_RETRY_JITTER)
  std::string gs_msg;
  int jitter_mask = 0;        // bit 0: Kmm, bit 1: Kmm + beta*Psi2 get 1e-7 * I in this global step (partial_terms.py:452-456)
  int retry_mask = 0;         // what a GP_RETRY_JITTER asks the caller to pass to gp_global_step_jitter
  // phase 2
  double* Rpart = nullptr;    // [p2_slices][Mp][CXp]
  int p2_slices = 0;
  double* HZp = nullptr;      // [Mp/128][Np][CZp] per-point partials (one array per 128 inducing columns)
  double* gXmu = nullptr;     // [N][Q]
  double* gXs = nullptr;      // [N][Q]
  double* gapart = nullptr;   // [blocks][Q] per-block alpha partial sums from the per-point kernel
  int ga_blocks = 0;
  unsigned long long* p2prog = nullptr;   // p2_fast8_kernel: [slices][MT] tile progress of the workgroups of a slice (kept in step for the L2), bases grow per launch
  unsigned long long p2_epoch = 0;
  double* hgpart = nullptr;   // partial sums of the fast path's mu^2 term of grad_alpha (per wave, or per 256 points from p2_ga_kernel)
  // regime B (variances > 0): pairwise psi2 kernels; allocated on first use
  bool b_alloc = false;
  double* LE = nullptr;       // [Np][Mp]  1/2 ln c2_n - 1/2 sum_q w_nq (mu_nq - z_mq)^2   (n-major)
  double* LET = nullptr;      // [Np][Mp]  LEA = LE + sum_q V_nq z_mq^2 (n-major)
  double* Vn = nullptr;       // [Np][Q]   -1/4 (alpha_q - w_nq)
  double* Wn = nullptr;       // [Np][Q]   w_nq = alpha_q / (2 alpha_q S_nq + 1)
  double* V2P = nullptr;      // [Np][QB]  -2 V_nq, zero-padded to the kernels' compile-time width QB
  double* WP = nullptr;       // [Np][QB]  w_nq, zero-padded
  double* MUP = nullptr;      // [Np][QB]  mu_nq, zero-padded
  double* alphaP = nullptr;   // [QB]      alpha, zero-padded
  double* Z1P = nullptr;      // [Mp][QB]  Z with a column of ones at index Q (only meaningful when QB > Q)
  bool b_mfma = false;        // regime-B phase 1 on the matrix core (psi2_pairs_mfma_kernel: latent tables 32 / 52 / 64 wide with a spare column)
  bool b_sym = false;         // regime-B phase 2 on tile pairs (psi2_sym_kernel: Q <= 10, 64 < M <= 1024)
  double* Z1S = nullptr;      // [Mp][RT]  [Z | 1 at index QB | 0], RT = QB + 1 rounded up to 4: B operand of the row-side MFMAs
  int* sym_sched = nullptr;   // [rounds][waves] tile of every wave in every round (I | J << 16, -1 idle)
  int sym_nw = 0, sym_rounds = 0;
  double* ZP = nullptr;       // [Mp][QB]  Z zero-padded (rows >= M and columns >= Q are zero)
  int QB = 0;                 // 4, 10, 16, 32 or 64: smallest instantiated width >= Q
  double* lnc2h = nullptr;    // [Np]      1/2 ln c2_n
  double* DZ2 = nullptr;      // [M][M][Q] (z_mq - z_m'q)^2
  double* Gpart = nullptr;    // [pb_blocks][M][Q] per-block grad_Z partials of the psi2 part
  double* Gtmp = nullptr;     // [64][M][Q] second-level grad_Z partials
  double* gapart2 = nullptr;  // [pb_blocks][Q]
  double* pp = nullptr;       // [Np][3Q+1] per-point running sums sr, zr, z2r, zt of the psi2 rows kernel
  size_t pp_doubles = 0;
  int pb_blocks = 0;
  int nslab = 0, ppb = 0;     // regime-B phase-2 pair kernel: 64-column slabs of M, points per workgroup
  int* ptiles = nullptr;      // upper-triangular 16x16 tile table for the psi2 pair kernel
  int n_ptiles = 0;
  int* tiles64 = nullptr;     // upper-triangular 64x64 tile table for the MFMA pair kernel (wide latent spaces) and the tile-pair phase 2
  int n_tiles64 = 0;
  // regime-B phase 2 on tile pairs (psi2_tile.hip, Q <= 51)
  bool b_tile = false;        // regime-B phase 2 runs on psi2_tile_kernel
  double* ppt = nullptr;      // [tiles][3Q+1][b_ch] per-point sums of every tile for the points of one launch
  double* Gt = nullptr;       // [b_S][tiles][2][64][Q] grad_Z partials per workgroup
  long b_ch = 0;              // points per launch
  int b_S = 0;                // point slices per launch
  // regime B beyond the compiled latent widths (psi2_generic.hip, Q >= 64): psi2_n of a chunk of points and its row contractions
  double* gen_T = nullptr;    // [gen_P][M][M]
  double* gen_rt = nullptr;   // [gen_P][M][Q + 1]
  long gen_P = 0;             // points per chunk
  // CG vectors (resident): grad_latest/new/old (2,N,Q) each
  double* g_latest = nullptr;
  double* g_new = nullptr;
  double* g_old = nullptr;
  // RCCL communicator of this context's rank (comm.hip; NULL until gp_comm_init)
  void* comm = nullptr;
  int comm_ranks = 0, comm_rank = -1;
  // gp_set_globals: pinned host staging (two slots, [M*Q + Q] doubles each) so that the upload of Z and alpha is a true asynchronous copy --
  // an evaluation then has ONE host synchronisation, the read-back in gp_finish; the slot's event guards its reuse two calls later
  double* h_glob[2] = {nullptr, nullptr};
  hipEvent_t glob_ev[2] = {nullptr, nullptr};
  int glob_slot = 0;
  double* h_out = nullptr;    // gp_finish: pinned, mapped [GS_COUNT + 8 | M*Q + Q] -- finish_kernel writes the evaluation's results straight into it
  // timing: 2 = HIP events around every phase and the dominant kernels (gp_last_timings reports all eight numbers; the default), 1 = only the
  // evaluation's first and last event (total_ms), 0 = none.  Every recorded event is a signal packet the stream waits on: ~4-7 us of idle
  // stream each, thirteen per evaluation -- 0.3 % of an evaluation at configs[2]'s size, 15 % at configs[1]'s (gp_set_timing)
  int timing = [] { const char* e = getenv("GPARML_TIMING"); return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2; }();
  long sync_epoch = 0;        // stream synchronisations seen so far (gp_set_globals' pinned slots are reused without an event once one has passed)
  long glob_epoch[2] = {-1, -1};
  hipEvent_t ev[14] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double t_ms[5] = {0, 0, 0, 0, 0};
};

namespace gp {
extern thread_local std::string g_create_error;
int fail(gp_ctx* ctx, int code, const char* fmt, ...);
// Test mode (GPARML_POISON=1 at load time or gp_debug_set_option("poison_alloc", 1)): every device allocation that does not carry a documented
// zero-initialisation contract is filled with 0xFF bytes (a NaN as a double, -1 as an int) instead of zeros, and gp_set_globals refills the
// per-evaluation scratch and output buffers with it: a kernel that reads a region this evaluation did not write, or that relies on zeros nobody
// promised, then fails deterministically (NaN in the outputs) instead of once in a thousand runs.  DA_ZERO marks the buffers whose zeros ARE part of the
// design (padding nobody writes; each such call site says which region that is), DA_INIT the ones that were zeroed for tidiness only, DA_RAW the
// ones that are not initialised at all outside the test mode (every element is written before it is read).
extern std::atomic<int> g_opt_poison;
enum { DA_ZERO = 0, DA_INIT = 1, DA_RAW = 2 };
int dalloc_bytes(gp_ctx* c, void** p, size_t bytes, int mode);

// psi.hip
int run_upload_y(gp_ctx* c, const double* dY);
int run_prep_and_generate(gp_ctx* c);
int run_phase1(gp_ctx* c);
int run_phase2(gp_ctx* c);
bool p2_fast_mode(const gp_ctx* c);
// p1i8.hip (regime A phase 1 on the int8 matrix core)
bool p1i8_applicable(const gp_ctx* c);
bool p1i8_applicable_static(const gp_ctx* c);     // the shape / regime conditions alone (not the opt-in switch, not the guard)
int p1i8_prepare(gp_ctx* c, int8_t** Sl, long* strideJ, double** Dpart, int row_blocks);
int run_phase1_i8(gp_ctx* c);
int p1i8_check_begin(gp_ctx* c);      // after run_phase1_i8: keep the int8 statistics aside (the caller then runs the float64 phase 1)
int p1i8_check_compare(gp_ctx* c);    // after the float64 phase 1: norms of the difference (device)
int p1i8_check_finish(gp_ctx* c);     // gp_finish, after the stream synchronisation of a checked evaluation: decide
void p1i8_free(gp_ctx* c);
// p1v2.hip (regime A phase 1 without wasted tile slots)
bool p1v2_applicable(const gp_ctx* c);
int run_phase1_v2(gp_ctx* c);
void p1v2_free(gp_ctx* c);
// psi2.hip (regime B)
int ensure_regime_b_buffers(gp_ctx* c);
int run_generate_b(gp_ctx* c);
int run_phase1_b(gp_ctx* c);
int run_phase2_b(gp_ctx* c);
int run_dz2(gp_ctx* c);
// psi2_generic.hip (regime B for Q >= 64: plain kernels, any Q)
bool b_generic(const gp_ctx* c);
int run_le_generic(gp_ctx* c);
int run_phase1_b_generic(gp_ctx* c);
int run_phase2_b_generic(gp_ctx* c);
// psi2_tile.hip (regime B phase 2 on tile pairs)
bool pt2_applicable(const gp_ctx* c, bool sym_available);
int run_phase2_b_tiles(gp_ctx* c);
// compat.hip
int compat_build(gp_ctx* c, int which, double** out, long* count);
// comm.hip
void comm_free(gp_ctx* c);
// linalg.hip
int run_global_step(gp_ctx* c);
// gsi8.hip: the global step's two double-double-grade products on the int8 matrix core (M >= 1024)
bool gs_i8_wanted(const gp_ctx* c);
int run_gs_i8_product(gp_ctx* c, hipStream_t st, const double* A, long lda, int nA, const double* B, long ldb, int nB, int K, double* out, long ldo,
                      const double* Csub);
int check_global(gp_ctx* c);
// the same with the scalars + failure flags already on the host (h = [GS_COUNT + 8] doubles, or NULL when nothing is pending)
int check_global_from(gp_ctx* c, const double* h);
// PRECONDITION: the 128-blocks of Linv strictly above the block diagonal must be ZERO on entry -- they are never written here and
// Inv = Linv^T Linv reads the whole matrix.  gp_create allocates Linv zeroed and nothing else writes those blocks; the test hook
// gp_debug_potrf_inverse memsets its own buffer.  A caller that hands in a reused scratch buffer must clear it first.
int potrf_inverse_batched(gp_ctx* c, hipStream_t st, int Mp, int batch, double* A /*in: SPD, out: L*/, double* Linv, double* Inv,
                          double* Twork /*batch * Mp * Mp / 2 doubles*/, double* logdet2 /*device, [batch]*/, double* fail_flag /*device, [batch]*/,
                          double* splitk_ws /*may be NULL*/, size_t splitk_cap = 0);
// layout of the free-embedding LE table (csrc/psi2.hip, b_le_kernel): four points interleaved up to the 16-wide latent tables, point-major beyond
__host__ __device__ constexpr bool le_interleaved(int QT) { return QT <= 16; }
__host__ __device__ inline long le_index(bool il, long n, long m, long Mp) { return il ? ((((n >> 2) * Mp + m) << 2) + (n & 3)) : n * Mp + m; }

}  // namespace gp

// event i of the context's timing set, if the timing level asks for it (levels: gp_ctx::timing)
#define GP_EV(c, i) do { if ((c)->timing >= 2 || ((c)->timing == 1 && ((i) == 0 || (i) == 6))) (void)hipEventRecord((c)->ev[i], (c)->stream); } while (0)
#define GP_TRY_RC(x) do { int rc__ = (x); if (rc__ != GP_OK) return rc__; } while (0)
#define GP_HIP(ctx, call)                                                                         \
  do {                                                                                            \
    hipError_t e__ = (call);                                                                      \
    if (e__ != hipSuccess) return gp::fail(ctx, GP_ERR_HIP, "%s failed: %s (%s:%d)", #call,      \
                                            hipGetErrorString(e__), __FILE__, __LINE__);           \
  } while (0)
