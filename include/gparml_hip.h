/*
 * gparml_hip.h -- C ABI of the MI355X (gfx950) implementation of GParML's per-shard
 * `partial_terms` hot path.  Plain C: opaque context, plain pointers and sizes, int status codes.
 *
 * The reference (markvdw/GParML) has no FFI; the hot path sits behind a Python class and a Python
 * MapReduce backend module.  Each entry point below names the reference interface it replaces
 * (paths relative to the reference root).  The Python host side (gparml_amd/) binds this library
 * with ctypes and mirrors the reference's class/module surface; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - all host arrays are C-contiguous float64, owned by the caller, copied during the call;
 *   - one context per GPU/shard; calls on one context are serialised by the caller;
 *   - every function returns GP_OK (0) or an error code; gp_last_error(ctx) gives the message;
 *   - alpha is the inverse squared lengthscale (ard**-2), sf2 the signal variance, beta the noise
 *     precision -- the argument meaning of partial_terms.__init__ (partial_terms.py:16-36).
 *
 * One evaluation (parallel_GPLVM.py:222-279) is
 *   gp_set_globals -> gp_phase1 -> [gp_stats_pack, all-reduce gp_stats_packed_buffer, gp_stats_unpack] -> gp_global_step
 *                  -> gp_phase2 -> [all-reduce gp_grads_buffer] -> gp_finish
 */
#ifndef GPARML_HIP_H
#define GPARML_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gp_ctx gp_ctx;

/* status codes; the Python wrapper maps them to the exception classes the reference raises
 * (scg_adapted.py:55 catches LinAlgError / ZeroDivisionError / ValueError / AssertionError) */
enum {
  GP_OK = 0,
  GP_ERR_BAD_ARG = 1,      /* -> AssertionError / ValueError (kernel_exp.py:30-34 input assertions)        */
  GP_ERR_NOT_PD = 2,       /* -> numpy.linalg.LinAlgError (linalg.inv / slogdet sign, partial_terms.py:459) */
  GP_ERR_NON_FINITE = 3,   /* -> FloatingPointError (nputil.py:9 np.seterr(all='raise'))                    */
  GP_ERR_HIP = 4,          /* -> RuntimeError                                                               */
  GP_ERR_STATE = 5,        /* call sequence violated (e.g. phase2 before global_step) -> RuntimeError       */
  GP_ERR_UNSUPPORTED = 6,
  GP_RETRY_JITTER = 7,     /* a Cholesky factorisation failed for the first time: repeat the global step with gp_global_step_jitter */
  GP_ERR_RCCL = 8          /* RCCL missing (dlopen) or a collective failed -> RuntimeError                                        */
};

/* what gp_download can fetch (every array the reference exposes on the path) */
enum {
  GP_ARR_KMM = 0,            /* (M,M)   partial_terms.Kmm                 partial_terms.py:94  */
  GP_ARR_KMM_INV = 1,        /* (M,M)   partial_terms.Kmm_inv             partial_terms.py:95  */
  GP_ARR_PSI1 = 2,           /* (N_s,M) partial_terms.exp_K_mi            partial_terms.py:49  */
  GP_ARR_PSI2_SUM = 3,       /* (M,M)   sum_exp_K_mi_K_im                 partial_terms.py:79  */
  GP_ARR_PSI1TY = 4,         /* (M,D)   exp_K_miY                         partial_terms.py:80  */
  GP_ARR_KMM_PLUS_OP_INV = 5,/* (M,M)   Kmm_plus_op_inv                   partial_terms.py:82  */
  GP_ARR_DF_DKMM = 6,        /* (M,M)   dF_dKmm()                         partial_terms.py:102 */
  GP_ARR_DF_DPSI1TY = 7,     /* (M,D)   dF_dexp_K_miY()                   partial_terms.py:115 */
  GP_ARR_DF_DPSI2 = 8,       /* (M,M)   dF_dexp_K_mi_K_im()               partial_terms.py:123 */
  GP_ARR_GRAD_X_MU = 9,      /* (N_s,Q) grad_X_mu()                       partial_terms.py:367 */
  GP_ARR_GRAD_X_S = 10,      /* (N_s,Q) grad_X_S()                        partial_terms.py:400 */
  GP_ARR_SCALARS = 11,       /* 8 doubles: sum_YYT, sum_exp_K_ii, KL, logdet Kmm, logdet A, F, grad_beta, grad_sf2 */
  GP_ARR_PSI2_POINTS = 12,   /* (N_s,M,M) exp_K_mi_K_im, compat only      partial_terms.py:45  */
  GP_ARR_DKMM_DZ = 13,       /* (M,Q,M) dKmm_dZ()                         partial_terms.py:146 */
  GP_ARR_DPSI1TY_DZ = 14,    /* (M,Q,D) dexp_K_miY_dZ()                   partial_terms.py:162 */
  GP_ARR_DPSI2_DZ = 15,      /* (M,Q,M) dexp_K_mi_K_im_dZ()               partial_terms.py:190 */
  GP_ARR_DKMM_DALPHA = 16,   /* (Q,M,M) dKmm_dalpha()                     partial_terms.py:247 */
  GP_ARR_DPSI1TY_DALPHA = 17,/* (Q,M,D) dexp_K_miY_dalpha()               partial_terms.py:256 */
  GP_ARR_DPSI2_DALPHA = 18,  /* (Q,M,M) dexp_K_mi_K_im_dalpha()           partial_terms.py:273 */
  GP_ARR_X_MU_TRIAL = 19,    /* (N_s,Q) X_mu + step*d_mu                  local_MapReduce.py:205-211 */
  GP_ARR_X_S_TRIAL = 20,     /* (N_s,Q) softplus(X_S_raw + step*d_S)      local_MapReduce.py:214 */
  GP_ARR_GRAD_LATEST = 21    /* (2,N_s,Q) -[grad_X_mu, grad_X_S * softplus'(raw trial)], the .grad_latest.npy of local_MapReduce.py:357-360 */
};

/* ---- lifetime -------------------------------------------------------------------------------- */
/* partial_terms.__init__ sizes (partial_terms.py:16-30): local shard rows N_s, outputs D, inducing M, latent Q */
int gp_create(gp_ctx** out, int device, int64_t N_s, int D, int M, int Q);
int gp_destroy(gp_ctx* ctx);
const char* gp_last_error(const gp_ctx* ctx);   /* ctx may be NULL: returns the last creation error */
const char* gp_version(void);
/* hipStream_t to launch on (NULL = default stream); lets the host order RCCL collectives with kernels */
int gp_set_stream(gp_ctx* ctx, void* hip_stream);

/* ---- data ------------------------------------------------------------------------------------ */
/* partial_terms.set_data inputs (partial_terms.py:38-43) == what statistics_mapper loads
 * (local_MapReduce.py:197-201).  xs_is_raw != 0: X_S is in softplus-inverse space and is transformed
 * on the device (supporting_functions.py:153-156, local_MapReduce.py:214). */
int gp_upload_shard(gp_ctx* ctx, const double* Y, const double* X_mu, const double* X_S, int xs_is_raw);
/* replace only the embeddings (Y stays resident) */
int gp_upload_embeddings(gp_ctx* ctx, const double* X_mu, const double* X_S, int xs_is_raw);
/* search direction (2,N_s,Q) = [d_mu, d_S] of the optimiser, the .grad_d.npy of
 * local_MapReduce.py:204-211; NULL clears it */
int gp_set_direction(gp_ctx* ctx, const double* d);

/* ---- one evaluation ------------------------------------------------------------------------- */
/* global_statistics Z,sf2,alpha,beta (parallel_GPLVM.py:236-238), global N (options['N']) and the
 * trial step size (options['step_size'], parallel_GPLVM.py:228) */
int gp_set_globals(gp_ctx* ctx, const double* Z, double sf2, const double* alpha, double beta,
                   int64_t N_global, double step_size);
/* statistics_mapper body (local_MapReduce.py:224-240 -> partial_terms.set_data / update_local_statistics,
 * partial_terms.py:38-52,74-87): local Psi-statistics into the packed stats buffer */
int gp_phase1(gp_ctx* ctx);
/* packed device buffer the host all-reduces (sum) across shards: the statistics_reducer
 * (local_MapReduce.py:250-277).  Layout: Psi2 (Mp*Mp) | C (Mp*Dp) | sum_YYT, Psi0, KL, n_local, pad(4) */
int gp_stats_buffer(gp_ctx* ctx, void** dev_ptr, int64_t* n_doubles);
/* The same statistics without padding and without the lower triangle, for the all-reduce across processes (what the reducer's twelve
 * accumulated_statistics files carry, local_MapReduce.py:250-277): Psi2 upper triangle, row-major (M(M+1)/2) | C (M*D) | the 8 scalars.
 * gp_stats_pack(ctx) fills it from the statistics buffer (after gp_phase1 and any drop-out scaling), gp_stats_unpack(ctx) writes the
 * reduced values back (both triangles) before gp_global_step.  1.46 MB instead of 2.6 MB at M=512, D=100. */
int gp_stats_packed_buffer(gp_ctx* ctx, void** dev_ptr, int64_t* n_doubles);
int gp_stats_pack(gp_ctx* ctx);
int gp_stats_unpack(gp_ctx* ctx);
/* device-side reduce for several shards in one process (statistics_reducer, local_MapReduce.py:250-277), on one GPU or
 * across GPUs (peer copy into a staging buffer of dst): which=0 statistics buffer, which=1 phase-2 gradient-sum buffer;
 * op=0 dst += src, op=1 dst = src */
int gp_buffer_combine(gp_ctx* dst, const gp_ctx* src, int which, int op);
/* node drop-out (local_MapReduce.py:119-129, 263-264): the reference sums every statistic over the kept nodes only and divides
 * by kept/(kept+dropped).  A dropped shard's contribution is excluded by the caller (factor 0 before the reduction), the reduced
 * buffers are scaled by (kept+dropped)/kept after it: which=0 statistics (before gp_global_step), which=1 gradient sums (after
 * gp_phase2; they are the contracted form of the reference's sum_d_*_d_Z / d_alpha statistics). */
int gp_scale_buffer(gp_ctx* ctx, int which, double factor);
int gp_scale_stats(gp_ctx* ctx, double factor);   /* = gp_scale_buffer(ctx, 0, factor) */
/* ---- the reduce across GPUs inside the library (`allreduce(ctx, phase)`) ------------------------
 * statistics_reducer (local_MapReduce.py:250-277) as an RCCL all-reduce(sum, float64) on the context's stream, one rank per GPU.
 * RCCL is resolved with dlopen at the first of these calls (GP_ERR_RCCL if it cannot be found; GPARML_RCCL_LIB names a specific
 * library), so a single-GPU user never needs it.  Rank 0 obtains the 128-byte ncclUniqueId with gp_comm_unique_id and distributes it;
 * every rank then calls gp_comm_init(ctx, id, nranks, rank) (collective: all ranks must call it).  gp_allreduce(ctx, 0) packs the
 * statistics (gp_stats_pack), all-reduces the packed buffer and unpacks it -- call it between gp_phase1 (and any drop-out scaling)
 * and gp_global_step; gp_allreduce(ctx, 1) all-reduces the gradient sums between gp_phase2 and gp_finish.  Nothing synchronises
 * the host.  A host that prefers its own communicator all-reduces gp_stats_packed_buffer / gp_grads_buffer itself. */
#define GP_COMM_ID_BYTES 128
int gp_comm_unique_id(void* id_out_128_bytes);
int gp_comm_init(gp_ctx* ctx, const void* unique_id_128_bytes, int nranks, int rank);
int gp_allreduce(gp_ctx* ctx, int which);
int gp_comm_destroy(gp_ctx* ctx);
/* Diagnostics for the first multi-GPU run.  gp_comm_available() = GP_OK when RCCL can be resolved in this process (nothing collective
 * happens: every rank can ask before any rank calls gp_comm_init, and a host agrees on the answer over its own channel -- one rank
 * without RCCL would otherwise leave the others waiting inside ncclCommInitRank).  gp_comm_info reports what the communicator of this
 * context saw: nranks / rank as given to ncclCommInitRank (0 / -1 without a communicator), the payload of the two all-reduces in
 * bytes, and -- when probe_sum is not NULL, COLLECTIVE, synchronises the context's stream -- the all-reduced sum of one double 1.0 per
 * rank: it equals nranks exactly when RCCL really connected that many ranks. */
int gp_comm_available(void);
int gp_comm_info(gp_ctx* ctx, int* nranks, int* rank, int64_t* stats_payload_bytes, int64_t* grads_payload_bytes, double* probe_sum);
/* calculate_global_statistics + Kmm parts of calculate_global_derivatives (parallel_GPLVM.py:302-369):
 * Kmm, Cholesky of Kmm and Kmm+beta*Psi2, F, dF_d*, grad_beta.  Asynchronous: the launches are enqueued and nothing is
 * read back; the outcome is reported by the first of gp_global_status / gp_finish / gp_download that follows (one host
 * synchronisation per evaluation). */
int gp_global_step(gp_ctx* ctx);
/* The reference adds 1e-7*I to Kmm (bit 0) and / or Kmm+beta*Psi2 (bit 1) when slogdet reports a negative sign and carries on
 * (partial_terms.logmarglik, partial_terms.py:452-456); if that does not help it asserts (:459-461).  Here a failed Cholesky is
 * reported once as GP_RETRY_JITTER (with the mask to use); the caller repeats the global step -- and phase 2 and its reduction --
 * through this entry.  A second failure of the same matrix is GP_ERR_NOT_PD.  The jittered matrix is used for the log-determinant
 * AND the inverse (the reference keeps the un-jittered LU inverse of the indefinite matrix for the traces and gradients). */
int gp_global_step_jitter(gp_ctx* ctx, int jitter_mask);
/* synchronise and report the outcome of the last global step: GP_OK, GP_RETRY_JITTER (*retry_mask = mask to pass on),
 * GP_ERR_NOT_PD, GP_ERR_NON_FINITE.  retry_mask may be NULL. */
int gp_global_status(gp_ctx* ctx, int* retry_mask);
/* embeddings_mapper body + data-dependent sums of the Z/alpha gradients
 * (local_MapReduce.py:348-358; partial_terms.py:162-205, 256-284, 367-431) */
int gp_phase2(gp_ctx* ctx, int want_embedding_grads);
/* packed device buffer of the phase-2 sums the host all-reduces: grad_Z data part (M*Q) | grad_alpha data part (Q) */
int gp_grads_buffer(gp_ctx* ctx, void** dev_ptr, int64_t* n_doubles);
/* contraction into the final gradients (partial_terms.grad_Z/grad_alpha/grad_sf2/grad_beta,
 * partial_terms.py:207-240, 286-299, 322-333, 340-360).  Any output pointer may be NULL. */
int gp_finish(gp_ctx* ctx, double* F, double* grad_Z, double* grad_sf2, double* grad_alpha, double* grad_beta);

/* ---- results ---------------------------------------------------------------------------------- */
int gp_download(gp_ctx* ctx, int which, double* dst, int64_t n_doubles);
/* set the reduced statistics from the host (partial_terms.set_local_statistics, partial_terms.py:54-61) */
int gp_set_local_statistics(gp_ctx* ctx, double sum_YYT, const double* Psi2, const double* C,
                            double sum_exp_K_ii, double KL);
/* device time of the last evaluation in milliseconds (HIP events on ctx's stream, recorded around each stage):
 * out[0]=prep+Psi1 generation, [1]=phase-1 contraction+reduce, [2]=global step, [3]=phase 2, [4]=sum of [0..3],
 * out[5]=psi1_kernel alone, [6]=p1_kernel alone, [7]=p2_kernel alone */
int gp_last_timings(gp_ctx* ctx, double* out8);
/* How many timing events an evaluation records: 2 (default) = around every stage and the dominant kernels (all of gp_last_timings), 1 = only the
 * first and the last one (gp_last_timings fills out[4], the whole evaluation, alone), 0 = none.  Every event is a signal packet the stream waits
 * on (~4-7 us of idle stream each, thirteen per evaluation): nothing at configs[2]'s size, 15 % of an evaluation at configs[1]'s.  An optimiser
 * that does not read the timings switches them off. */
int gp_set_timing(gp_ctx* ctx, int level);
/* The opt-in int8 phase 1 (GPARML_P1_I8=1; csrc/p1i8.hip: Psi2 / Psi1^T Y of partial_terms.py:45-52, 79-80 from exact integer digit products) guards
 * itself: the first int8 evaluation after an upload and every 64th one run both phase-1 paths, compare the statistics on the device and use the
 * float64 ones; the context is taken off the int8 path (until the next upload) when cond_lower_bound * max(rel_psi2, rel_c) exceeds the library's
 * threshold.  state: -1 the path does not apply to this context's shape / regime, 0 not checked yet, 1 accepted, 2 rejected (float64 kernels run);
 * rel_*: relative Frobenius distance of the int8 statistics from the float64 ones at the last check; checks: how many were made. */
int gp_i8_status(gp_ctx* ctx, int* state, double* rel_psi2, double* rel_c, double* cond_lower_bound, int64_t* checks);
/* hipMemGetInfo of ctx's device: bytes free / total right now (the footprint of a shard at BASELINE configs[4]'s per-GPU size is
 * OBSERVED with this; the reference has no counterpart -- its shard lives in the mapper process' numpy arrays,
 * local_MapReduce.py:197-201) */
int gp_memory_info(gp_ctx* ctx, int64_t* free_bytes, int64_t* total_bytes);

/* partial_terms.grad_Z (which=0, partial_terms.py:207-240) / grad_alpha (which=1, :286-299) from explicitly given parts --
 * the signature an unmodified parallel_GPLVM.calculate_global_derivatives (:340-351) calls.  Shapes as in the reference:
 * which=0: dKmm_dX (M,Q,M), dC_dX (M,Q,D), dPsi2_dX (M,Q,M) -> out (M,Q); which=1: (Q,M,M), (Q,M,D), (Q,M,M) -> out (Q) */
int gp_grad_from_parts(gp_ctx* ctx, int which, const double* dF_dKmm, const double* dKmm_dX, const double* dF_dC, const double* dC_dX,
                       const double* dF_dPsi2, const double* dPsi2_dX, double* out);

/* ---- resident CG vectors (scg_adapted_local_MapReduce.py:29-243), SURVEY.md section 8(f)-1 ----------------------------
 * grad_latest / grad_new / grad_old / d are (2,N_s,Q) device arrays; gp_phase2(ctx,1) refreshes grad_latest. */
int gp_cg_set_grads(gp_ctx* ctx);                              /* embeddings_set_grads :29-55: new = old = latest, d = -latest */
/* out6 = [mu = new.d (:59-74), kappa = d.d (:76-90), theta = d.(latest-new) (:92-110), |new|^2 (:112-126),
 *         new.old (:128-140; the caller forms Gamma), max|d|] -- local sums, the host all-reduces them across shards */
int gp_cg_dots(gp_ctx* ctx, double* out6);
int gp_cg_max_d(gp_ctx* ctx, double alpha, double* out);       /* max |alpha*d| :142-155 */
/* the gradient-descent optimiser's two reductions on grad_now (= grad_new here): out2 = [sum |grad_now|, max |grad_now|]
 * (gd_local_MapReduce.py:38-61); its updates are gp_cg_update: set_grads -> 5, update_d(gamma) -> 1 with a = -gamma
 * (d = -(grad_now + gamma d), :63-74), update_X -> 2 (:76-94), update_grad_now -> 4 (:96-105) */
int gp_cg_abs(gp_ctx* ctx, double* out2);
/* which: 0 reset_d :160-173 | 1 update_d(a=Gamma) :175-189 | 2 update_X(a=alpha) :191-214 | 3 update_grad_old :216-229 |
 *        4 update_grad_new :231-243 | 5 set_grads */
int gp_cg_update(gp_ctx* ctx, int which, double a);

/* ---- shard ingest ------------------------------------------------------
 * numpy.genfromtxt(file, delimiter=',') as statistics_mapper / embeddings_mapper call it on every evaluation
 * (local_MapReduce.py:197-199, 325-327): comma-separated floating-point text -> row-major (rows, cols) doubles.
 * Blank lines and '#' comment lines are skipped, empty or unparsable fields become NaN; a ragged file is GP_ERR_BAD_ARG
 * (genfromtxt raises ValueError).  Host only, parsed in parallel (threads <= 0: all cores); the caller parses a shard once
 * and keeps Y resident (gp_upload_shard). */
int gp_csv_shape(const char* path, int64_t* rows, int64_t* cols);
int gp_csv_read(const char* path, double* out, int64_t rows, int64_t cols, int threads);

/* ---- test hooks (used by tests/ only) --------------------------------------------------------- */
/* C = alpha*op(A)op(B) + beta*C through the library's FP64 MFMA GEMM core; A (m,k) or (k,m) if ta,
 * B (k,n) or (n,k) if tb; any sizes (padded internally) */
int gp_debug_gemm(int device, int ta, int tb, int m, int n, int k, double alpha, const double* A, const double* B,
                  double beta, double* C);
/* make gp_buffer_combine take its cross-device path (peer copy through the staging buffer) even on one device */
int gp_debug_force_staging(int on);
/* process-wide switches of the global step's extended-precision pieces (both on by default): "dd_kipsi2" -- K_mm^-1 Psi2 accumulated in
 * double-double (the product whose float64 rounding is grad_Z's whole error at cond ~1e10), "refine_E" -- one refinement
 * step of E with a double-double residual.  bench.py times the global step with and without to print their cost.
 * "poison_alloc" (also GPARML_POISON=1 at load time): every allocation without a documented zero contract is filled with NaN bytes and
 * gp_set_globals refills the per-evaluation buffers with them -- a kernel that reads what this evaluation did not write fails
 * deterministically (tests/test_gpu_poison.py). */
int gp_debug_set_option(const char* name, int value);
/* in-place lower Cholesky + inverse of an SPD (n,n) matrix; logdet out; returns GP_ERR_NOT_PD on failure */
int gp_debug_potrf_inverse(int device, int n, const double* A, double* L, double* Ainv, double* logdet);
/* device-resident timing of the FP64 MFMA GEMM core: `iters` products of the given shape, milliseconds per product (tools/dev_gemm_bench.py) */
int gp_debug_gemm_bench(int device, int ta, int tb, int m, int n, int k, int iters, double* ms_out);
/* raw copy of one of the global step's internal buffers ("Linv", "Inv", "E", "PsiE", "T1", "T2", "dFdK", "Bbar", "Abar", "Bm", "gK", "gs") after a
 * stream synchronisation; n = capacity of out in doubles (tests/devtools/dev_tail_diff.py) */
int gp_debug_peek(gp_ctx* ctx, const char* name, double* out, long n);
/* the rule the internal matrix products are launched under (csrc/gemm.hip, launch_gemm aborts when it fails: the blocked Cholesky's in-place panel solve raced until
 * round 6): does an operand stored as rows x cols doubles with leading dimension ld, starting at element offset x_off of a buffer, share an element with the result
 * m x n, leading dimension ldc, at offset c_off of the SAME buffer?  Host arithmetic only (no device needed); returns 0 or 1. */
int gp_debug_operands_overlap(long x_off, long rows, long cols, long ld, long c_off, long m, long n, long ldc);

#ifdef __cplusplus
}
#endif
#endif /* GPARML_HIP_H */
