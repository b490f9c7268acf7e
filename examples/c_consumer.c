/* A consumer of the C ABI without Python or torch: one bound + gradient evaluation of a sparse GP (fixed embeddings) on one shard that it reads from a
 * binary file, printed as text.  This is the whole of what a host in another language binds (include/gparml_hip.h; INTEGRATION.md section 3 is the same
 * sequence through ctypes).  Reference counterpart: partial_terms.set_data + logmarglik + grad_Z / grad_alpha / grad_sf2 / grad_beta
 * (/root/reference/partial_terms.py:38-52, 207-360, 436-473).
 *
 *   gcc -O2 -Iinclude examples/c_consumer.c -o /tmp/c_consumer -Lgparml_amd -lgparml_hip -Wl,-rpath,$PWD/gparml_amd
 *   /tmp/c_consumer shard.bin
 * shard.bin: int64 N, D, M, Q, then float64 Y[N*D], X_mu[N*Q], Z[M*Q], alpha[Q], sf2, beta (tests/test_gpu_c_consumer.py writes one and compares the output
 * with the Python engine's). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gparml_hip.h"

#define CHECK(call)                                                                 \
  do {                                                                              \
    int rc_ = (call);                                                               \
    if (rc_ != GP_OK) { fprintf(stderr, "%s: %s (code %d)\n", #call, gp_last_error(ctx), rc_); return 1; } \
  } while (0)

static double* read_doubles(FILE* f, size_t n) {
  double* p = (double*)malloc((n ? n : 1) * sizeof(double));
  if (!p || fread(p, sizeof(double), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
  return p;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s shard.bin\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  int64_t dims[4];
  if (fread(dims, sizeof(int64_t), 4, f) != 4) { fprintf(stderr, "short header\n"); return 2; }
  const int64_t N = dims[0];
  const int D = (int)dims[1], M = (int)dims[2], Q = (int)dims[3];
  double* Y = read_doubles(f, (size_t)N * D);
  double* X = read_doubles(f, (size_t)N * Q);
  double* Z = read_doubles(f, (size_t)M * Q);
  double* alpha = read_doubles(f, (size_t)Q);
  double* sb = read_doubles(f, 2);
  fclose(f);
  double* S = (double*)calloc((size_t)N * Q, sizeof(double));          /* fixed embeddings: zero variances */
  double* gZ = (double*)malloc((size_t)M * Q * sizeof(double));
  double* ga = (double*)malloc((size_t)Q * sizeof(double));

  gp_ctx* ctx = NULL;
  if (gp_create(&ctx, 0, N, D, M, Q) != GP_OK) { fprintf(stderr, "gp_create: %s\n", gp_last_error(NULL)); return 1; }
  CHECK(gp_upload_shard(ctx, Y, X, S, 0));
  CHECK(gp_set_globals(ctx, Z, sb[0], alpha, sb[1], N, 0.0));
  CHECK(gp_phase1(ctx));          /* a multi-GPU host: gp_allreduce(ctx, 0) here */
  CHECK(gp_global_step(ctx));
  CHECK(gp_phase2(ctx, 0));       /* ... and gp_allreduce(ctx, 1) here */
  double F, gsf2, gbeta;
  CHECK(gp_finish(ctx, &F, gZ, &gsf2, ga, &gbeta));
  printf("%s\nF %.17g\ngrad_sf2 %.17g\ngrad_beta %.17g\n", gp_version(), F, gsf2, gbeta);
  for (int q = 0; q < Q; ++q) printf("grad_alpha %d %.17g\n", q, ga[q]);
  for (long i = 0; i < (long)M * Q; ++i) printf("grad_Z %ld %.17g\n", i, gZ[i]);
  CHECK(gp_destroy(ctx));
  free(Y); free(X); free(Z); free(alpha); free(sb); free(S); free(gZ); free(ga);
  return 0;
}
