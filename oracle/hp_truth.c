/* Extended-precision truth of one regime-A bound+gradient evaluation at the benchmark's full size.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): nothing under gparml_amd/ builds, links or runs this.
 *
 * The same evaluation as tests/golden/make_hp_golden.py's evaluate_ld (numpy long double, N = 4000), restated in C so that
 * it can be streamed over N = 1e5 .. 1e6 rows on a few cores: every operation -- Psi1 = K_nm, Psi2 = K^T K, C = K^T Y, both
 * Cholesky factorisations, the inverses, every contraction -- in x87 80-bit long double (eps 1.08e-19).  Formulation:
 * SURVEY.md section 7 / oracle/factorised.py; reference lines /root/reference/partial_terms.py:74-87 (statistics),
 * 102-138 (partials), 207-240, 286-299, 322-360 (gradients), 436-473 (bound), kernels.py:72-113 (K_mm),
 * kernel_exp.py:51-82 (Psi1 with S = 0).
 *
 * Two results are produced from ONE pass over the statistics: "plain" (inducing points in their given order) and "reversed"
 * (the whole global step on the reversed order of the inducing points, so that every rounding of both factorisations, the
 * inverses and the products changes); their difference is the truth's own uncertainty.  (A Newton step X <- X (2I - A X) in the
 * SAME precision is no better witness: it keeps the right residual but its error X dT is cond times larger than dT.)
 *
 *   hp_truth <dir> N D M Q
 *   reads  <dir>/Y.bin (N*D f64) X.bin (N*Q f64) Z.bin (M*Q f64) alpha.bin (Q f64) params.bin (sf2, beta: 2 f64)
 *   writes <dir>/truth_plain.bin, <dir>/truth_reversed.bin: [F | grad_Z (M*Q) | grad_alpha (Q) | grad_sf2 | grad_beta] as
 *          float64 "hi" values followed by the same count of "lo" values (x - (double)x), and
 *          <dir>/{Abar,Bbar,dFdK}_{plain,reversed}.bin (float64, for error norms of the float64 paths)
 * Build: gcc -O2 -fopenmp -o hp_truth hp_truth.c -lm      (tests/golden/make_hp_truth_large.py does it)
 */
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef long double ld;
#define RB 32 /* rows per block */

static void die(const char* m) { fprintf(stderr, "hp_truth: %s\n", m); exit(2); }

static double* read_f64(const char* dir, const char* name, size_t n) {
  char p[4096];
  snprintf(p, sizeof p, "%s/%s", dir, name);
  FILE* f = fopen(p, "rb");
  if (!f) die(p);
  double* x = (double*)malloc(n * sizeof(double));
  if (!x || fread(x, sizeof(double), n, f) != n) die("short read");
  fclose(f);
  return x;
}
static void write_f64(const char* dir, const char* name, const double* x, size_t n) {
  char p[4096];
  snprintf(p, sizeof p, "%s/%s", dir, name);
  FILE* f = fopen(p, "wb");
  if (!f || fwrite(x, sizeof(double), n, f) != n) die(p);
  fclose(f);
}
static ld* ldalloc(size_t n) {
  ld* p = (ld*)calloc(n, sizeof(ld));
  if (!p) die("out of memory");
  return p;
}

/* four dot products against one left vector (the x87 stack holds 4 accumulators + operands) */
static inline void dot4(const ld* a, const ld* b0, const ld* b1, const ld* b2, const ld* b3, int n, ld* out) {
  ld s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int k = 0; k < n; ++k) { const ld x = a[k]; s0 += x * b0[k]; s1 += x * b1[k]; s2 += x * b2[k]; s3 += x * b3[k]; }
  out[0] = s0; out[1] = s1; out[2] = s2; out[3] = s3;
}
static inline ld dot1(const ld* a, const ld* b, int n) {
  ld s = 0;
  for (int k = 0; k < n; ++k) s += a[k] * b[k];
  return s;
}
static inline ld dot1d(const ld* a, const double* b, int n) {
  ld s = 0;
  for (int k = 0; k < n; ++k) s += a[k] * (ld)b[k];
  return s;
}

/* C[i][j] = sum_k A[i][k] Bt[j][k]   (m x n, inner k) */
static void mm_nt(ld* C, const ld* A, const ld* Bt, int m, int n, int k) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) C[(size_t)i * n + j] = dot1(A + (size_t)i * k, Bt + (size_t)j * k, k);
}

/* SPD inverse and log-determinant by Cholesky; A is M x M (destroyed: holds L), Inv receives A^-1 */
static ld spd_inverse(ld* A, ld* Inv, int M) {
  /* left-looking Cholesky by columns; L stored in the lower triangle of A, row-major */
  for (int j = 0; j < M; ++j) {
    ld d = A[(size_t)j * M + j] - dot1(A + (size_t)j * M, A + (size_t)j * M, j);
    if (!(d > 0)) die("matrix is not positive definite in long double");
    d = sqrtl(d);
    A[(size_t)j * M + j] = d;
#pragma omp parallel for schedule(static)
    for (int i = j + 1; i < M; ++i)
      A[(size_t)i * M + j] = (A[(size_t)i * M + j] - dot1(A + (size_t)i * M, A + (size_t)j * M, j)) / d;
  }
  ld logdet = 0;
  for (int j = 0; j < M; ++j) logdet += 2 * logl(A[(size_t)j * M + j]);
  /* Xt[c][i] = (L^-1)[i][c]: forward substitution per column c */
  ld* Xt = ldalloc((size_t)M * M);
#pragma omp parallel for schedule(dynamic, 4)
  for (int c = 0; c < M; ++c) {
    ld* x = Xt + (size_t)c * M;
    for (int i = c; i < M; ++i) {
      ld s = (i == c) ? 1.0L : 0.0L;
      s -= dot1(A + (size_t)i * M + c, x + c, i - c);
      x[i] = s / A[(size_t)i * M + i];
    }
  }
  /* Inv[a][b] = sum_{i >= max(a,b)} X[i][a] X[i][b] */
#pragma omp parallel for schedule(dynamic, 4)
  for (int a = 0; a < M; ++a)
    for (int b = a; b < M; ++b) {
      const ld s = dot1(Xt + (size_t)a * M + b, Xt + (size_t)b * M + b, M - b);
      Inv[(size_t)a * M + b] = s;
      Inv[(size_t)b * M + a] = s;
    }
  free(Xt);
  return logdet;
}

typedef struct {
  ld F, grad_sf2, grad_beta;
  ld *gZ_K, *ga_K; /* Kmm parts */
  ld *Abar, *B2;   /* M x D, M x M (= 2 Bbar, symmetric) */
  ld *Bbar, *dFdK;
} Global;

static void global_step(Global* g, const ld* Psi2, const ld* C, const ld* Z, const ld* a, ld s2, ld b, ld sumYY, long N, int D, int M,
                        int Q) {
  const size_t mm = (size_t)M * M, md = (size_t)M * D;
  const ld half = 0.5L;
  ld* Kmm = ldalloc(mm);
  ld* A = ldalloc(mm);
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < M; ++j) {
      ld e = 0;
      for (int q = 0; q < Q; ++q) { const ld d = Z[(size_t)i * Q + q] - Z[(size_t)j * Q + q]; e += a[q] * d * d; }
      Kmm[(size_t)i * M + j] = s2 * expl(-half * e);
      A[(size_t)i * M + j] = Kmm[(size_t)i * M + j] + b * Psi2[(size_t)i * M + j];
    }
  ld* Lk = ldalloc(mm);
  ld* La = ldalloc(mm);
  memcpy(Lk, Kmm, mm * sizeof(ld));
  memcpy(La, A, mm * sizeof(ld));
  ld* Ki = ldalloc(mm);
  ld* P = ldalloc(mm);
  const ld ldK = spd_inverse(Lk, Ki, M);
  const ld ldA = spd_inverse(La, P, M);
  free(Lk); free(La);
  /* E = P C */
  ld* Ct = ldalloc(md);
  for (int i = 0; i < M; ++i) for (int d = 0; d < D; ++d) Ct[(size_t)d * M + i] = C[(size_t)i * D + d];
  ld* E = ldalloc(md);
  mm_nt(E, P, Ct, M, D, M);
  ld trKi = 0, trP = 0, trCE = 0;
  for (size_t i = 0; i < mm; ++i) { trKi += Ki[i] * Psi2[i]; trP += P[i] * Psi2[i]; }
  for (size_t i = 0; i < md; ++i) trCE += C[i] * E[i];
  const ld Psi0 = s2 * (ld)N;
  const ld two_pi = 2 * acosl(-1.0L);
  const ld Dd = (ld)D, Nn = (ld)N;
  g->F = -half * Nn * Dd * logl(two_pi) + half * Dd * Nn * logl(b) + half * Dd * ldK - half * Dd * ldA - half * b * sumYY -
         half * b * Dd * Psi0 + half * b * Dd * trKi + half * b * b * trCE;
  ld* EEt = ldalloc(mm);
  mm_nt(EEt, E, E, M, M, D);
  ld* T = ldalloc(mm);
  ld* KPK = ldalloc(mm);
  mm_nt(T, Ki, Psi2, M, M, M);   /* Ki Psi2 (Psi2 symmetric) */
  mm_nt(KPK, T, Ki, M, M, M);    /* (Ki Psi2) Ki (Ki symmetric) */
  g->Abar = ldalloc(md); g->B2 = ldalloc(mm); g->Bbar = ldalloc(mm); g->dFdK = ldalloc(mm);
  for (size_t i = 0; i < md; ++i) g->Abar[i] = b * b * E[i];
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < M; ++j) {
      const size_t ij = (size_t)i * M + j, ji = (size_t)j * M + i;
      const ld kp = half * ((Ki[ij] - P[ij]) + (Ki[ji] - P[ji]));
      const ld ee = half * (EEt[ij] + EEt[ji]);
      g->Bbar[ij] = half * b * Dd * kp - half * b * b * b * ee;
      g->B2[ij] = 2 * g->Bbar[ij];
      g->dFdK[ij] = half * Dd * (Ki[ij] - P[ij]) - half * b * Dd * KPK[ij] - half * b * b * EEt[ij];
    }
  /* grad_beta */
  ld* Et = ldalloc(md);
  for (int i = 0; i < M; ++i) for (int d = 0; d < D; ++d) Et[(size_t)d * M + i] = E[(size_t)i * D + d];
  ld* PsiE = ldalloc(md);
  mm_nt(PsiE, Psi2, Et, M, D, M);
  ld trEPE = 0;
  for (size_t i = 0; i < md; ++i) trEPE += E[i] * PsiE[i];
  g->grad_beta = half * Nn * Dd / b - half * Dd * trP - half * sumYY - half * Dd * Psi0 + half * Dd * trKi + b * trCE - half * b * b * trEPE;
  /* Kmm parts */
  g->gZ_K = ldalloc((size_t)M * Q); g->ga_K = ldalloc(Q);
  ld sumV = 0;
  for (int i = 0; i < M; ++i) {
    ld rs = 0;
    ld* sz = ldalloc(Q);
    for (int j = 0; j < M; ++j) {
      const size_t ij = (size_t)i * M + j, ji = (size_t)j * M + i;
      const ld S = (g->dFdK[ij] + g->dFdK[ji]) * Kmm[ij];
      const ld V = g->dFdK[ij] * Kmm[ij];
      sumV += V;
      rs += S;
      for (int q = 0; q < Q; ++q) {
        sz[q] += S * Z[(size_t)j * Q + q];
        const ld dz = Z[(size_t)i * Q + q] - Z[(size_t)j * Q + q];
        g->ga_K[q] += -half * V * dz * dz;
      }
    }
    for (int q = 0; q < Q; ++q) g->gZ_K[(size_t)i * Q + q] = -a[q] * (Z[(size_t)i * Q + q] * rs - sz[q]);
    free(sz);
  }
  ld sAC = 0, sBP = 0;
  for (size_t i = 0; i < md; ++i) sAC += g->Abar[i] * C[i];
  for (size_t i = 0; i < mm; ++i) sBP += g->Bbar[i] * Psi2[i];
  g->grad_sf2 = (sumV + sAC + 2 * sBP - half * b * Dd * Psi0) / s2;
  free(Kmm); free(A); free(Ki); free(P); free(Ct); free(E); free(EEt); free(T); free(KPK); free(Et); free(PsiE);
}

int main(int argc, char** argv) {
  if (argc < 6) die("usage: hp_truth <dir> N D M Q");
  const char* dir = argv[1];
  const long N = atol(argv[2]);
  const int D = atoi(argv[3]), M = atoi(argv[4]), Q = atoi(argv[5]);
  if (M % 4) die("M must be a multiple of 4");
  double* Y = read_f64(dir, "Y.bin", (size_t)N * D);
  double* X = read_f64(dir, "X.bin", (size_t)N * Q);
  double* Zd = read_f64(dir, "Z.bin", (size_t)M * Q);
  double* ad = read_f64(dir, "alpha.bin", Q);
  double* par = read_f64(dir, "params.bin", 2);
  const ld s2 = par[0], b = par[1], half = 0.5L;
  ld* Z = ldalloc((size_t)M * Q);
  ld* a = ldalloc(Q);
  for (int i = 0; i < M * Q; ++i) Z[i] = Zd[i];
  for (int q = 0; q < Q; ++q) a[q] = ad[q];
  const size_t mm = (size_t)M * M, md = (size_t)M * D;
  const long nblk = (N + RB - 1) / RB;
  const int nth = omp_get_max_threads();
  const double t0 = omp_get_wtime();

  /* ---- pass 1: Psi2 = K^T K (upper triangle, mirrored at the end), C = K^T Y, sum Y^2 */
  ld* Psi2 = ldalloc(mm);
  ld* C = ldalloc(md);
  ld sumYY = 0;
  {
    ld** tP = (ld**)malloc(nth * sizeof(ld*));
    ld** tC = (ld**)malloc(nth * sizeof(ld*));
    ld* tS = ldalloc(nth);
#pragma omp parallel
    {
      const int t = omp_get_thread_num();
      ld* p2 = tP[t] = ldalloc(mm);
      ld* cc = tC[t] = ldalloc(md);
      ld* Kt = ldalloc((size_t)M * RB);       /* [M][RB] */
      ld* Yt = ldalloc((size_t)D * RB);       /* [D][RB] */
      ld syy = 0;
#pragma omp for schedule(static)
      for (long blk = 0; blk < nblk; ++blk) {
        const long r0 = blk * RB;
        const int nr = (int)((N - r0 < RB) ? (N - r0) : RB);
        for (int r = 0; r < RB; ++r) {
          if (r < nr) {
            ld x[64];
            for (int q = 0; q < Q; ++q) x[q] = X[(size_t)(r0 + r) * Q + q];
            for (int m = 0; m < M; ++m) {
              ld e = 0;
              for (int q = 0; q < Q; ++q) { const ld d = x[q] - Z[(size_t)m * Q + q]; e += a[q] * d * d; }
              Kt[(size_t)m * RB + r] = s2 * expl(-half * e);
            }
            for (int d = 0; d < D; ++d) { const ld y = Y[(size_t)(r0 + r) * D + d]; Yt[(size_t)d * RB + r] = y; syy += y * y; }
          } else {
            for (int m = 0; m < M; ++m) Kt[(size_t)m * RB + r] = 0;
            for (int d = 0; d < D; ++d) Yt[(size_t)d * RB + r] = 0;
          }
        }
        for (int i = 0; i < M; ++i) {
          const ld* ki = Kt + (size_t)i * RB;
          int j = i;
          for (; j < M && (j & 3); ++j) p2[(size_t)i * M + j] += dot1(ki, Kt + (size_t)j * RB, RB);
          for (; j + 4 <= M; j += 4) {
            ld o[4];
            dot4(ki, Kt + (size_t)j * RB, Kt + (size_t)(j + 1) * RB, Kt + (size_t)(j + 2) * RB, Kt + (size_t)(j + 3) * RB, RB, o);
            ld* dst = p2 + (size_t)i * M + j;
            dst[0] += o[0]; dst[1] += o[1]; dst[2] += o[2]; dst[3] += o[3];
          }
          int d = 0;
          for (; d + 4 <= D; d += 4) {
            ld o[4];
            dot4(ki, Yt + (size_t)d * RB, Yt + (size_t)(d + 1) * RB, Yt + (size_t)(d + 2) * RB, Yt + (size_t)(d + 3) * RB, RB, o);
            ld* dst = cc + (size_t)i * D + d;
            dst[0] += o[0]; dst[1] += o[1]; dst[2] += o[2]; dst[3] += o[3];
          }
          for (; d < D; ++d) cc[(size_t)i * D + d] += dot1(ki, Yt + (size_t)d * RB, RB);
        }
      }
      tS[t] = syy;
      free(Kt); free(Yt);
    }
    for (int t = 0; t < nth; ++t) {
      for (size_t i = 0; i < mm; ++i) Psi2[i] += tP[t][i];
      for (size_t i = 0; i < md; ++i) C[i] += tC[t][i];
      sumYY += tS[t];
      free(tP[t]); free(tC[t]);
    }
    free(tP); free(tC); free(tS);
    for (int i = 0; i < M; ++i) for (int j = i + 1; j < M; ++j) Psi2[(size_t)j * M + i] = Psi2[(size_t)i * M + j];
  }
  fprintf(stderr, "[hp_truth] pass 1: %.0f s (%d threads)\n", omp_get_wtime() - t0, nth);

  /* ---- global step: in the given order, and on the reversed order of the inducing points (un-permuted afterwards) */
  Global G[2];
  global_step(&G[0], Psi2, C, Z, a, s2, b, sumYY, N, D, M, Q);
  {
    ld* Zr = ldalloc((size_t)M * Q);
    ld* Pr = ldalloc(mm);
    ld* Cr = ldalloc(md);
    for (int i = 0; i < M; ++i) {
      const int pi = M - 1 - i;
      for (int q = 0; q < Q; ++q) Zr[(size_t)i * Q + q] = Z[(size_t)pi * Q + q];
      for (int d = 0; d < D; ++d) Cr[(size_t)i * D + d] = C[(size_t)pi * D + d];
      for (int j = 0; j < M; ++j) Pr[(size_t)i * M + j] = Psi2[(size_t)pi * M + (M - 1 - j)];
    }
    Global R;
    global_step(&R, Pr, Cr, Zr, a, s2, b, sumYY, N, D, M, Q);
    G[1] = R;
    G[1].Abar = ldalloc(md); G[1].B2 = ldalloc(mm); G[1].Bbar = ldalloc(mm); G[1].dFdK = ldalloc(mm); G[1].gZ_K = ldalloc((size_t)M * Q);
    for (int i = 0; i < M; ++i) {
      const int pi = M - 1 - i;
      for (int q = 0; q < Q; ++q) G[1].gZ_K[(size_t)pi * Q + q] = R.gZ_K[(size_t)i * Q + q];
      for (int d = 0; d < D; ++d) G[1].Abar[(size_t)pi * D + d] = R.Abar[(size_t)i * D + d];
      for (int j = 0; j < M; ++j) {
        const size_t dst = (size_t)pi * M + (M - 1 - j), src = (size_t)i * M + j;
        G[1].B2[dst] = R.B2[src]; G[1].Bbar[dst] = R.Bbar[src]; G[1].dFdK[dst] = R.dFdK[src];
      }
    }
    free(R.Abar); free(R.B2); free(R.Bbar); free(R.dFdK); free(R.gZ_K); free(Zr); free(Pr); free(Cr);
  }
  fprintf(stderr, "[hp_truth] global steps done: %.0f s\n", omp_get_wtime() - t0);

  /* ---- pass 2 for both operand sets: W = (K (2 Bbar) + Y Abar^T) o K;  W1 = W^T 1, WX = W^T X, WX2 = W^T X^2 */
  const size_t acc_n = (size_t)M * (2 * Q + 1);
  ld* R[2];
  for (int v = 0; v < 2; ++v) R[v] = ldalloc(acc_n);
  {
    ld** tR = (ld**)malloc(2 * nth * sizeof(ld*));
#pragma omp parallel
    {
      const int t = omp_get_thread_num();
      ld* r0v = tR[2 * t] = ldalloc(acc_n);
      ld* r1v = tR[2 * t + 1] = ldalloc(acc_n);
      ld* Kb = ldalloc((size_t)RB * M);   /* [RB][M] */
      ld* Yb = ldalloc((size_t)RB * D);   /* [RB][D] */
      ld* xb = ldalloc((size_t)RB * Q);
#pragma omp for schedule(static)
      for (long blk = 0; blk < nblk; ++blk) {
        const long rr0 = blk * RB;
        const int nr = (int)((N - rr0 < RB) ? (N - rr0) : RB);
        for (int r = 0; r < RB; ++r) {
          if (r < nr) {
            for (int q = 0; q < Q; ++q) xb[(size_t)r * Q + q] = X[(size_t)(rr0 + r) * Q + q];
            for (int m = 0; m < M; ++m) {
              ld e = 0;
              for (int q = 0; q < Q; ++q) { const ld d = xb[(size_t)r * Q + q] - Z[(size_t)m * Q + q]; e += a[q] * d * d; }
              Kb[(size_t)r * M + m] = s2 * expl(-half * e);
            }
            for (int d = 0; d < D; ++d) Yb[(size_t)r * D + d] = Y[(size_t)(rr0 + r) * D + d];
          } else {
            for (int q = 0; q < Q; ++q) xb[(size_t)r * Q + q] = 0;
            for (int m = 0; m < M; ++m) Kb[(size_t)r * M + m] = 0;
            for (int d = 0; d < D; ++d) Yb[(size_t)r * D + d] = 0;
          }
        }
        for (int v = 0; v < 2; ++v) {
          ld* acc = v ? r1v : r0v;
          for (int m = 0; m < M; ++m) {
            const ld* brow = G[v].B2 + (size_t)m * M;     /* symmetric: row m = column m */
            const ld* arow = G[v].Abar + (size_t)m * D;
            ld* am = acc + (size_t)m * (2 * Q + 1);
            for (int r = 0; r < RB; r += 4) {
              ld o[4], p[4];
              dot4(brow, Kb + (size_t)r * M, Kb + (size_t)(r + 1) * M, Kb + (size_t)(r + 2) * M, Kb + (size_t)(r + 3) * M, M, o);
              dot4(arow, Yb + (size_t)r * D, Yb + (size_t)(r + 1) * D, Yb + (size_t)(r + 2) * D, Yb + (size_t)(r + 3) * D, D, p);
              for (int u = 0; u < 4; ++u) {
                const ld w = (o[u] + p[u]) * Kb[(size_t)(r + u) * M + m];
                am[0] += w;
                const ld* xr = xb + (size_t)(r + u) * Q;
                for (int q = 0; q < Q; ++q) { am[1 + q] += w * xr[q]; am[1 + Q + q] += w * xr[q] * xr[q]; }
              }
            }
          }
        }
      }
      free(Kb); free(Yb); free(xb);
    }
    for (int t = 0; t < nth; ++t)
      for (int v = 0; v < 2; ++v) {
        for (size_t i = 0; i < acc_n; ++i) R[v][i] += tR[2 * t + v][i];
        free(tR[2 * t + v]);
      }
    free(tR);
  }
  fprintf(stderr, "[hp_truth] pass 2: %.0f s\n", omp_get_wtime() - t0);

  /* ---- finish and write */
  const size_t nout = 1 + (size_t)M * Q + Q + 2;
  for (int v = 0; v < 2; ++v) {
    ld* out = ldalloc(nout);
    out[0] = G[v].F;
    ld* gZ = out + 1;
    ld* ga = out + 1 + (size_t)M * Q;
    for (int q = 0; q < Q; ++q) ga[q] = 0;
    for (int m = 0; m < M; ++m) {
      const ld* am = R[v] + (size_t)m * (2 * Q + 1);
      for (int q = 0; q < Q; ++q) {
        const ld z = Z[(size_t)m * Q + q];
        gZ[(size_t)m * Q + q] = a[q] * (am[1 + q] - z * am[0]) + G[v].gZ_K[(size_t)m * Q + q];
        ga[q] += -half * (am[1 + Q + q] - 2 * z * am[1 + q] + z * z * am[0]);
      }
    }
    for (int q = 0; q < Q; ++q) ga[q] += G[v].ga_K[q];
    out[1 + (size_t)M * Q + Q] = G[v].grad_sf2;
    out[1 + (size_t)M * Q + Q + 1] = G[v].grad_beta;
    double* o64 = (double*)malloc(2 * nout * sizeof(double));
    for (size_t i = 0; i < nout; ++i) { o64[i] = (double)out[i]; o64[nout + i] = (double)(out[i] - (ld)o64[i]); }
    write_f64(dir, v ? "truth_reversed.bin" : "truth_plain.bin", o64, 2 * nout);
    free(o64); free(out);
    const char* tag = v ? "reversed" : "plain";
    char nm[64];
    double* tmp = (double*)malloc(mm * sizeof(double));
    for (size_t i = 0; i < md; ++i) tmp[i] = (double)G[v].Abar[i];
    snprintf(nm, sizeof nm, "Abar_%s.bin", tag); write_f64(dir, nm, tmp, md);
    for (size_t i = 0; i < mm; ++i) tmp[i] = (double)G[v].Bbar[i];
    snprintf(nm, sizeof nm, "Bbar_%s.bin", tag); write_f64(dir, nm, tmp, mm);
    for (size_t i = 0; i < mm; ++i) tmp[i] = (double)G[v].dFdK[i];
    snprintf(nm, sizeof nm, "dFdK_%s.bin", tag); write_f64(dir, nm, tmp, mm);
    free(tmp);
  }
  /* the float64 image of the long-double statistics: lets the caller measure cond(Kmm + beta Psi2) */
  {
    double* tmp = (double*)malloc(mm * sizeof(double));
    for (size_t i = 0; i < mm; ++i) tmp[i] = (double)Psi2[i];
    write_f64(dir, "Psi2.bin", tmp, mm);
    for (size_t i = 0; i < md; ++i) tmp[i] = (double)C[i];
    write_f64(dir, "C.bin", tmp, md);
    free(tmp);
  }
  fprintf(stderr, "[hp_truth] done: %.0f s\n", omp_get_wtime() - t0);
  return 0;
}
