/*
 * oracle/tvlqr_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C fp64 restatement of the reference's TVLQR hot path.  Follows, step by
 * step, the operation order of
 *     /root/reference/src/tvlqr/tvlqr.cpp:18-63    (tvlqr_TotalMemSize)
 *     /root/reference/src/tvlqr/tvlqr.cpp:65-195   (tvlqr_BackwardPass)
 *     /root/reference/src/tvlqr/tvlqr.cpp:197-248  (tvlqr_ForwardPass)
 * The reference's arithmetic lives in Eigen 3.4.0 (commit 3147391d, pinned in
 * /root/reference/deps/CMakeLists.txt:15-19), which is not vendored and not on
 * this image, so the reference translation unit itself is UNBUILDABLE here.
 * Eigen's published algorithms are restated: dense products as index-ordered
 * dot products, LLT as the unblocked lower in-place Cholesky Eigen uses below
 * size 32 (fail when the pivot x <= 0), solveInPlace as forward substitution
 * with L then back substitution with L^T.
 *
 * Parity status: PINNED against the reference's own known-answer constants
 * (tvlqr_test.cpp:188-190 K0/d0, :206-207 xN/yN) -- see tests/test_oracle_kat.py.
 * Everything denser (all K_k, P_k for every k, 4x4 Quu, failure cases) is not
 * pinned by any reference test; there the oracle is the authority by virtue of
 * passing those pins and an independent numpy cross-check.
 *
 * All matrices are column-major, like the reference (tvlqr.cpp:13-16).
 * Compile with -ffp-contract=off so that results do not depend on FMA fusion.
 */
#include <math.h>
#include <stdbool.h>
#include <stddef.h>
#include <string.h>

#define ORACLE_TVLQR_SUCCESS (-1)

/* ---- tiny column-major helpers ------------------------------------------------ */

/* C(mr x nc) = alpha * op(A) * op(B) + beta * C, op given by transpose flags.
 * A is (ar x ac) col-major, B is (br x bc) col-major. Index-ordered dot products. */
static void gemm(int ta, int tb, int mr, int nc, int kd, double alpha, const double* A, int lda,
                 const double* B, int ldb, double beta, double* C, int ldc) {
  for (int j = 0; j < nc; ++j) {
    for (int i = 0; i < mr; ++i) {
      double s = 0.0;
      for (int k = 0; k < kd; ++k) {
        double a = ta ? A[k + (size_t)i * lda] : A[i + (size_t)k * lda];
        double b = tb ? B[j + (size_t)k * ldb] : B[k + (size_t)j * ldb];
        s += a * b;
      }
      double c0 = (beta == 0.0) ? 0.0 : beta * C[i + (size_t)j * ldc];
      C[i + (size_t)j * ldc] = c0 + alpha * s;
    }
  }
}

/* Unblocked lower Cholesky in place (Eigen llt_inplace<Lower>::unblocked).
 * Returns -1 on success, else the failing pivot index. */
static int chol_lower_inplace(double* M, int n) {
  for (int k = 0; k < n; ++k) {
    double x = M[k + (size_t)k * n];
    for (int j = 0; j < k; ++j) x -= M[k + (size_t)j * n] * M[k + (size_t)j * n];
    if (x <= 0.0) return k;
    x = sqrt(x);
    M[k + (size_t)k * n] = x;
    for (int i = k + 1; i < n; ++i) {
      double s = M[i + (size_t)k * n];
      for (int j = 0; j < k; ++j) s -= M[i + (size_t)j * n] * M[k + (size_t)j * n];
      M[i + (size_t)k * n] = s / x;
    }
  }
  return -1;
}

/* Solve (L L^T) X = B in place; L lower (n x n), B is (n x nrhs) col-major. */
static void chol_solve_inplace(const double* L, int n, double* B, int nrhs) {
  for (int c = 0; c < nrhs; ++c) {
    double* b = B + (size_t)c * n;
    for (int i = 0; i < n; ++i) { /* forward: L y = b */
      double s = b[i];
      for (int j = 0; j < i; ++j) s -= L[i + (size_t)j * n] * b[j];
      b[i] = s / L[i + (size_t)i * n];
    }
    for (int i = n - 1; i >= 0; --i) { /* backward: L^T x = y */
      double s = b[i];
      for (int j = i + 1; j < n; ++j) s -= L[j + (size_t)i * n] * b[j];
      b[i] = s / L[i + (size_t)i * n];
    }
  }
}

/* ---- tvlqr_TotalMemSize  (tvlqr.cpp:18-63) ------------------------------------ */
int oracle_tvlqr_TotalMemSize(const int* nx, const int* nu, int num_horizon, bool is_diag) {
  if (!nx) return 0;
  if (!nu) return 0;
  int mem_size = 0;
  for (int k = 0; k <= num_horizon; ++k) {
    int n = nx[k];
    mem_size += is_diag ? n : n * n; /* Q */
    mem_size += n;                   /* q */
    mem_size += n * n;               /* P */
    mem_size += n;                   /* p */
    mem_size += n;                   /* x */
    mem_size += n;                   /* y */
    if (k < num_horizon) {
      int m = nu[k];
      mem_size += n * n + n * m + n;                       /* A B f */
      mem_size += (is_diag ? m : m * m) + (is_diag ? 0 : m * n) + m; /* R H r */
      mem_size += m * n + m;                               /* K d */
      mem_size += 2 * (n * n + m * m + m * n + n + m);     /* Q-blocks + tmp */
      mem_size += m;                                       /* u */
    }
  }
  mem_size += 2; /* delta_V */
  return mem_size * (int)sizeof(double);
}

/* ---- tvlqr_BackwardPass  (tvlqr.cpp:65-195) ----------------------------------- */
int oracle_tvlqr_BackwardPass(const int* nx, const int* nu, int num_horizon,
                              const double* const* A, const double* const* B,
                              const double* const* f, const double* const* Q,
                              const double* const* R, const double* const* H,
                              const double* const* q, const double* const* r, double reg,
                              double** K, double** d, double** P, double** p, double* delta_V,
                              double** Qxx, double** Quu, double** Qux, double** Qx, double** Qu,
                              double** Qxx_tmp, double** Quu_tmp, double** Qux_tmp,
                              double** Qx_tmp, double** Qu_tmp, bool linear_only_update,
                              bool is_diag) {
  const int N = num_horizon;
  (void)linear_only_update; /* tvlqr.cpp:78: accepted and ignored */

  /* terminal cost-to-go (tvlqr.cpp:81-90) */
  {
    int n = nx[N];
    delta_V[0] = 0;
    delta_V[1] = 0;
    if (is_diag) {
      memset(P[N], 0, sizeof(double) * n * n);
      for (int i = 0; i < n; ++i) P[N][i + (size_t)i * n] = Q[N][i];
    } else {
      memcpy(P[N], Q[N], sizeof(double) * n * n);
    }
    memcpy(p[N], q[N], sizeof(double) * n);
  }

  for (int k = N - 1; k >= 0; --k) {
    const int n = nx[k];
    const int m = nu[k];
    const int n2 = nx[k + 1];
    const double* Pn = P[k + 1];
    const double* pn = p[k + 1];

    /* action-value expansion init (tvlqr.cpp:125-133) */
    if (is_diag) {
      memset(Qxx[k], 0, sizeof(double) * n * n);
      memset(Quu[k], 0, sizeof(double) * m * m);
      memset(Qux[k], 0, sizeof(double) * m * n);
      for (int i = 0; i < n; ++i) Qxx[k][i + (size_t)i * n] = Q[k][i];
      for (int i = 0; i < m; ++i) Quu[k][i + (size_t)i * m] = R[k][i];
    } else {
      memcpy(Qxx[k], Q[k], sizeof(double) * n * n);
      memcpy(Quu[k], R[k], sizeof(double) * m * m);
      memcpy(Qux[k], H[k], sizeof(double) * m * n);
    }
    /* Qxx_tmp = A^T P' ; Qxx += Qxx_tmp A   (tvlqr.cpp:135-136) */
    gemm(1, 0, n, n2, n2, 1.0, A[k], n2, Pn, n2, 0.0, Qxx_tmp[k], n);
    gemm(0, 0, n, n, n2, 1.0, Qxx_tmp[k], n, A[k], n2, 1.0, Qxx[k], n);
    /* Qux_tmp = B^T P' ; Quu += Qux_tmp B ; Qux += Qux_tmp A  (tvlqr.cpp:139-143) */
    gemm(1, 0, m, n2, n2, 1.0, B[k], n2, Pn, n2, 0.0, Qux_tmp[k], m);
    gemm(0, 0, m, m, n2, 1.0, Qux_tmp[k], m, B[k], n2, 1.0, Quu[k], m);
    gemm(0, 0, m, n, n2, 1.0, Qux_tmp[k], m, A[k], n2, 1.0, Qux[k], m);
    /* Qx_tmp = p' + P' f ; Qx = q + A^T Qx_tmp ; Qu = r + B^T Qx_tmp (tvlqr.cpp:147-152) */
    memcpy(Qx_tmp[k], pn, sizeof(double) * n2);
    gemm(0, 0, n2, 1, n2, 1.0, Pn, n2, f[k], n2, 1.0, Qx_tmp[k], n2);
    memcpy(Qx[k], q[k], sizeof(double) * n);
    gemm(1, 0, n, 1, n2, 1.0, A[k], n2, Qx_tmp[k], n2, 1.0, Qx[k], n);
    memcpy(Qu[k], r[k], sizeof(double) * m);
    gemm(1, 0, m, 1, n2, 1.0, B[k], n2, Qx_tmp[k], n2, 1.0, Qu[k], m);

    /* gains (tvlqr.cpp:155-166) */
    memcpy(K[k], Qux[k], sizeof(double) * m * n);
    for (int i = 0; i < m; ++i) d[k][i] = -Qu[k][i];
    memcpy(Quu_tmp[k], Quu[k], sizeof(double) * m * m);
    for (int i = 0; i < m; ++i) Quu_tmp[k][i + (size_t)i * m] += reg;
    if (chol_lower_inplace(Quu_tmp[k], m) != -1) {
      return k; /* tvlqr.cpp:162-164 */
    }
    chol_solve_inplace(Quu_tmp[k], m, K[k], n);
    chol_solve_inplace(Quu_tmp[k], m, d[k], 1);

    /* cost-to-go (tvlqr.cpp:173-186); uses the UNregularised Quu */
    memcpy(P[k], Qxx[k], sizeof(double) * n * n);
    gemm(0, 0, m, n, m, 1.0, Quu[k], m, K[k], m, 0.0, Qux_tmp[k], m);   /* Qux_tmp = Quu K   */
    gemm(1, 0, n, n, m, 1.0, K[k], m, Qux[k], m, 0.0, Qxx_tmp[k], n);   /* Qxx_tmp = K^T Qux */
    gemm(1, 0, n, 1, m, 1.0, K[k], m, Qu[k], m, 0.0, Qx_tmp[k], n);     /* Qx_tmp  = K^T Qu  */
    gemm(1, 0, n, n, m, 1.0, Qux_tmp[k], m, K[k], m, 1.0, P[k], n);     /* P += (Quu K)^T K  */
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) P[k][i + (size_t)j * n] -= Qxx_tmp[k][i + (size_t)j * n];
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) P[k][i + (size_t)j * n] -= Qxx_tmp[k][j + (size_t)i * n];

    memcpy(p[k], Qx[k], sizeof(double) * n);
    gemm(1, 0, n, 1, m, -1.0, Qux_tmp[k], m, d[k], m, 1.0, p[k], n);    /* -= (Quu K)^T d */
    gemm(1, 0, n, 1, m, -1.0, K[k], m, Qu[k], m, 1.0, p[k], n);         /* -= K^T Qu      */
    gemm(1, 0, n, 1, m, 1.0, Qux[k], m, d[k], m, 1.0, p[k], n);         /* += Qux^T d     */

    /* expected decrease (tvlqr.cpp:189-191) */
    gemm(0, 0, m, 1, m, 1.0, Quu[k], m, d[k], m, 0.0, Qu_tmp[k], m);
    double s0 = 0, s1 = 0;
    for (int i = 0; i < m; ++i) s0 += d[k][i] * Qu[k][i];
    for (int i = 0; i < m; ++i) s1 += d[k][i] * Qu_tmp[k][i];
    delta_V[0] += s0;
    delta_V[1] += 0.5 * s1;
  }
  return ORACLE_TVLQR_SUCCESS;
}

/* ---- tvlqr_ForwardPass  (tvlqr.cpp:197-248) ----------------------------------- */
int oracle_tvlqr_ForwardPass(const int* nx, const int* nu, int num_horizon,
                             const double* const* A, const double* const* B,
                             const double* const* f, const double* const* K,
                             const double* const* d, const double* const* P,
                             const double* const* p, const double* x0, double** x, double** u,
                             double** y) {
  const int N = num_horizon;
  memcpy(x[0], x0, sizeof(double) * nx[0]);
  for (int k = 0; k < N; ++k) {
    const int n = nx[k], m = nu[k], n2 = nx[k + 1];
    /* u = d - K x  (tvlqr.cpp:223-224) */
    memcpy(u[k], d[k], sizeof(double) * m);
    gemm(0, 0, m, 1, n, -1.0, K[k], m, x[k], n, 1.0, u[k], m);
    /* x+ = f + A x + B u  (tvlqr.cpp:226-228) */
    memcpy(x[k + 1], f[k], sizeof(double) * n2);
    gemm(0, 0, n2, 1, n, 1.0, A[k], n2, x[k], n, 1.0, x[k + 1], n2);
    gemm(0, 0, n2, 1, m, 1.0, B[k], n2, u[k], m, 1.0, x[k + 1], n2);
    if (y != NULL) { /* y = P x + p  (tvlqr.cpp:230-235) */
      gemm(0, 0, n, 1, n, 1.0, P[k], n, x[k], n, 0.0, y[k], n);
      for (int i = 0; i < n; ++i) y[k][i] += p[k][i];
    }
  }
  if (y != NULL) { /* terminal (tvlqr.cpp:238-246) */
    const int n = nx[N];
    gemm(0, 0, n, 1, n, 1.0, P[N], n, x[N], n, 0.0, y[N], n);
    for (int i = 0; i < n; ++i) y[N][i] += p[N][i];
  }
  return ORACLE_TVLQR_SUCCESS;
}

/* ---- flat / batched convenience wrappers (uniform n, m; used by tests + bench) -- *
 * Layout of every flat array: [batch][k][block], block column-major, i.e. exactly the
 * reference's per-knot-point blocks laid end to end.  The wrapper only builds the
 * pointer arrays the reference-style entry points take; it adds no arithmetic.      */
#include <stdlib.h>

typedef struct {
  int N, n, m;
  const double **A, **B, **f, **Q, **R, **H, **q, **r;
  double **K, **d, **P, **p, **Qxx, **Quu, **Qux, **Qx, **Qu;
  double **Qxx_t, **Quu_t, **Qux_t, **Qx_t, **Qu_t;
  double **x, **u, **y;
  double* scratch;
  int *nx, *nu;
} oracle_ws;

static void* xmalloc(size_t s) { return malloc(s ? s : 1); }

void* oracle_ws_create(int N, int n, int m) {
  oracle_ws* w = (oracle_ws*)calloc(1, sizeof(oracle_ws));
  w->N = N; w->n = n; w->m = m;
  size_t np = (size_t)(N + 1);
#define PA(name) w->name = xmalloc(sizeof(void*) * np)
  PA(A); PA(B); PA(f); PA(Q); PA(R); PA(H); PA(q); PA(r);
  PA(K); PA(d); PA(P); PA(p); PA(Qxx); PA(Quu); PA(Qux); PA(Qx); PA(Qu);
  PA(Qxx_t); PA(Quu_t); PA(Qux_t); PA(Qx_t); PA(Qu_t); PA(x); PA(u); PA(y);
#undef PA
  size_t per = (size_t)2 * (n * n + m * m + m * n + n + m);
  w->scratch = (double*)xmalloc(sizeof(double) * per * N);
  w->nx = (int*)xmalloc(sizeof(int) * np);
  w->nu = (int*)xmalloc(sizeof(int) * np);
  for (int k = 0; k <= N; ++k) { w->nx[k] = n; w->nu[k] = m; }
  double* s = w->scratch;
  for (int k = 0; k < N; ++k) {
    w->Qxx[k] = s; s += n * n;  w->Quu[k] = s; s += m * m;  w->Qux[k] = s; s += m * n;
    w->Qx[k] = s; s += n;       w->Qu[k] = s; s += m;
    w->Qxx_t[k] = s; s += n * n; w->Quu_t[k] = s; s += m * m; w->Qux_t[k] = s; s += m * n;
    w->Qx_t[k] = s; s += n;      w->Qu_t[k] = s; s += m;
  }
  return w;
}

void oracle_ws_destroy(void* h) {
  oracle_ws* w = (oracle_ws*)h;
  if (!w) return;
  free(w->A); free(w->B); free(w->f); free(w->Q); free(w->R); free(w->H); free(w->q); free(w->r);
  free(w->K); free(w->d); free(w->P); free(w->p); free(w->Qxx); free(w->Quu); free(w->Qux);
  free(w->Qx); free(w->Qu); free(w->Qxx_t); free(w->Quu_t); free(w->Qux_t); free(w->Qx_t);
  free(w->Qu_t); free(w->x); free(w->u); free(w->y); free(w->scratch); free(w->nx); free(w->nu);
  free(w);
}

/* One problem, flat [k][block] arrays.  Q has N+1 blocks (n*n dense or n diag), q N+1.
 * If qblocks (optional, may be NULL) is given it receives, per k, Qxx|Quu|Qux|Qx|Qu.   */
int oracle_backward_flat(void* h, const double* A, const double* B, const double* f,
                         const double* Q, const double* R, const double* H, const double* q,
                         const double* r, double reg, int is_diag, double* K, double* d,
                         double* P, double* p, double* delta_V, double* qblocks) {
  oracle_ws* w = (oracle_ws*)h;
  const int N = w->N, n = w->n, m = w->m;
  const size_t qs = is_diag ? n : (size_t)n * n, rs = is_diag ? m : (size_t)m * m;
  for (int k = 0; k <= N; ++k) {
    w->Q[k] = Q + qs * k; w->q[k] = q + (size_t)n * k;
    w->P[k] = P + (size_t)n * n * k; w->p[k] = p + (size_t)n * k;
    if (k < N) {
      w->A[k] = A + (size_t)n * n * k; w->B[k] = B + (size_t)n * m * k; w->f[k] = f + (size_t)n * k;
      w->R[k] = R + rs * k; w->H[k] = H ? H + (size_t)m * n * k : NULL; w->r[k] = r + (size_t)m * k;
      w->K[k] = K + (size_t)m * n * k; w->d[k] = d + (size_t)m * k;
    }
  }
  int res = oracle_tvlqr_BackwardPass(w->nx, w->nu, N, w->A, w->B, w->f, w->Q, w->R, w->H, w->q,
                                      w->r, reg, w->K, w->d, w->P, w->p, delta_V, w->Qxx, w->Quu,
                                      w->Qux, w->Qx, w->Qu, w->Qxx_t, w->Quu_t, w->Qux_t, w->Qx_t,
                                      w->Qu_t, false, is_diag != 0);
  if (qblocks) {
    const size_t per = (size_t)n * n + m * m + m * n + n + m;
    for (int k = 0; k < N; ++k) {
      double* o = qblocks + per * k;
      memcpy(o, w->Qxx[k], sizeof(double) * n * n); o += n * n;
      memcpy(o, w->Quu[k], sizeof(double) * m * m); o += m * m;
      memcpy(o, w->Qux[k], sizeof(double) * m * n); o += m * n;
      memcpy(o, w->Qx[k], sizeof(double) * n); o += n;
      memcpy(o, w->Qu[k], sizeof(double) * m);
    }
  }
  return res;
}

int oracle_forward_flat(void* h, const double* A, const double* B, const double* f,
                        const double* K, const double* d, const double* P, const double* p,
                        const double* x0, double* x, double* u, double* y) {
  oracle_ws* w = (oracle_ws*)h;
  const int N = w->N, n = w->n, m = w->m;
  for (int k = 0; k <= N; ++k) {
    w->P[k] = (double*)P + (size_t)n * n * k; w->p[k] = (double*)p + (size_t)n * k;
    w->x[k] = x + (size_t)n * k; w->y[k] = y ? y + (size_t)n * k : NULL;
    if (k < N) {
      w->A[k] = A + (size_t)n * n * k; w->B[k] = B + (size_t)n * m * k; w->f[k] = f + (size_t)n * k;
      w->K[k] = (double*)K + (size_t)m * n * k; w->d[k] = (double*)d + (size_t)m * k;
      w->u[k] = u + (size_t)m * k;
    }
  }
  return oracle_tvlqr_ForwardPass(w->nx, w->nu, N, w->A, w->B, w->f, (const double* const*)w->K,
                                  (const double* const*)w->d, (const double* const*)w->P,
                                  (const double* const*)w->p, x0, w->x, w->u, y ? w->y : NULL);
}

/* Batched loops over the flat single-problem entry points: arrays are [batch][k][block].
 * status[b] receives the per-problem return code.  Single-threaded by construction
 * (the reference has no threads: SURVEY.md section 2).                                  */
void oracle_backward_batch(int N, int n, int m, int batch, const double* A, const double* B,
                           const double* f, const double* Q, const double* R, const double* H,
                           const double* q, const double* r, double reg, int is_diag, double* K,
                           double* d, double* P, double* p, double* delta_V, int* status) {
  void* w = oracle_ws_create(N, n, m);
  const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
  const size_t qs = is_diag ? n : nn, rs = is_diag ? m : mm;
  for (int b = 0; b < batch; ++b) {
    status[b] = oracle_backward_flat(
        w, A + nn * N * b, B + nm * N * b, f + (size_t)n * N * b, Q + qs * (N + 1) * b,
        R + rs * N * b, H ? H + nm * N * b : NULL, q + (size_t)n * (N + 1) * b,
        r + (size_t)m * N * b, reg, is_diag, K + nm * N * b, d + (size_t)m * N * b,
        P + nn * (N + 1) * b, p + (size_t)n * (N + 1) * b, delta_V + 2 * (size_t)b, NULL);
  }
  oracle_ws_destroy(w);
}

void oracle_forward_batch(int N, int n, int m, int batch, const double* A, const double* B,
                          const double* f, const double* K, const double* d, const double* P,
                          const double* p, const double* x0, double* x, double* u, double* y) {
  void* w = oracle_ws_create(N, n, m);
  const size_t nn = (size_t)n * n, nm = (size_t)n * m;
  for (int b = 0; b < batch; ++b) {
    oracle_forward_flat(w, A + nn * N * b, B + nm * N * b, f + (size_t)n * N * b, K + nm * N * b,
                        d + (size_t)m * N * b, P + nn * (N + 1) * b, p + (size_t)n * (N + 1) * b,
                        x0 + (size_t)n * b, x + (size_t)n * (N + 1) * b, u + (size_t)m * N * b,
                        y ? y + (size_t)n * (N + 1) * b : NULL);
  }
  oracle_ws_destroy(w);
}
