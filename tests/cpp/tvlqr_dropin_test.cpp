// Re-authored counterpart of the reference's src/tvlqr/test/tvlqr_test.cpp: the same double-integrator
// known-answer problem pushed through the three tvlqr_* entry points with the reference's calling
// convention (arrays of per-knot-point pointers into ONE flat buffer sized by tvlqr_TotalMemSize).
// Links against libaltro_hip.so; needs an MI355X.  Prints "OK" and returns 0 on success.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tvlqr/tvlqr.h"

#define CHECK(cond)                                                     \
  do {                                                                  \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

int main() {
  constexpr int N = 10, dim = 2, n = 2 * dim, m = dim;
  const double h = 0.01;   // the golden constants are exact for h = 0.01 as a double
  int nx[N + 1], nu[N];
  for (int k = 0; k < N; ++k) { nx[k] = n; nu[k] = m; }
  nx[N] = n;
  const bool is_diag = true;
  const int mem_size = tvlqr_TotalMemSize(nx, nu, N, is_diag);
  std::vector<double> mem0(mem_size / sizeof(double), 0.0);
  double* mem = mem0.data();
  double *A[N], *B[N], *f[N], *Q[N + 1], *R[N], *H[N], *q[N + 1], *r[N], *K[N], *d[N], *P[N + 1], *p[N + 1];
  double *Qxx[N], *Quu[N], *Qux[N], *Qx[N], *Qu[N], *Qxx_t[N], *Quu_t[N], *Qux_t[N], *Qx_t[N], *Qu_t[N];
  double *x[N + 1], *u[N], *y[N + 1], *delta_V;
  auto take = [&](int cnt) { double* ptr = mem; mem += cnt; return ptr; };
  const double b = h * h / 2;
  for (int k = 0; k < N; ++k) {
    x[k] = take(n); u[k] = take(m); y[k] = take(n);
    A[k] = take(n * n); B[k] = take(n * m); f[k] = take(n);
    Q[k] = take(n); q[k] = take(n); R[k] = take(m); r[k] = take(m); H[k] = nullptr;
    K[k] = take(m * n); d[k] = take(m); P[k] = take(n * n); p[k] = take(n);
    Qxx[k] = take(n * n); Quu[k] = take(m * m); Qux[k] = take(m * n); Qx[k] = take(n); Qu[k] = take(m);
    Qxx_t[k] = take(n * n); Quu_t[k] = take(m * m); Qux_t[k] = take(m * n); Qx_t[k] = take(n); Qu_t[k] = take(m);
    for (int i = 0; i < dim; ++i) {
      A[k][i + i * n] = 1.0; A[k][(i + dim) + (i + dim) * n] = 1.0; A[k][i + (i + dim) * n] = h;
      B[k][i + i * n] = b; B[k][(i + dim) + i * n] = h;
    }
    const double xeq[4] = {1, 2, 0, 0};
    for (int i = 0; i < n; ++i) {   // f = A xeq + B * 0
      double s = 0;
      for (int j = 0; j < n; ++j) s += A[k][i + j * n] * xeq[j];
      f[k][i] = s;
    }
    for (int i = 0; i < n; ++i) { Q[k][i] = 1.1; q[k][i] = 0.01; }
    for (int i = 0; i < m; ++i) { R[k][i] = 0.1; r[k][i] = 0.001; }
  }
  x[N] = take(n); y[N] = take(n); Q[N] = take(n); q[N] = take(n); P[N] = take(n * n); p[N] = take(n);
  delta_V = take(2);
  for (int i = 0; i < n; ++i) { Q[N][i] = 110.0; q[N][i] = 0.01; }
  CHECK((size_t)(mem - mem0.data()) == mem_size / sizeof(double));   // tvlqr_test.cpp:167

  int res = tvlqr_BackwardPass(nx, nu, N, A, B, f, Q, R, H, q, r, 0.0, K, d, P, p, delta_V, Qxx, Quu, Qux,
                               Qx, Qu, Qxx_t, Quu_t, Qux_t, Qx_t, Qu_t, false, is_diag);
  CHECK(res == TVLQR_SUCCESS);
  const double K0[2][4] = {{0.7753129718046554, 0.0, 5.840445640045901, 0.0},
                           {0.0, 0.7753129718046554, 0.0, 5.840445640045901}};
  const double d0[2] = {-7.634078625343007, -15.256221385516275};
  double Kerr = 0, derr = 0;
  for (int i = 0; i < m; ++i) {
    for (int j = 0; j < n; ++j) Kerr += std::pow(K[0][i + j * m] - K0[i][j], 2);
    derr += std::pow(d[0][i] - d0[i], 2);
  }
  std::printf("K_err = %.3e  d_err = %.3e\n", std::sqrt(Kerr), std::sqrt(derr));
  CHECK(std::sqrt(Kerr) < 1e-12 && std::sqrt(derr) < 1e-12);   // reference tolerance: 1e-6

  const double x0[4] = {10.5, -20.5, -4, 5};
  res = tvlqr_ForwardPass(nx, nu, N, A, B, f, K, d, P, p, x0, x, u, y);
  CHECK(res == TVLQR_SUCCESS);
  const double xN[4] = {20.165445369740308, -0.13732391651279308, -2.3724421496097037, 2.3113121303468707};
  const double yN[4] = {2218.2089906714345, -15.09563081640724, -260.9586364570674, 254.2543343381558};
  double xerr = 0, yerr = 0;
  for (int i = 0; i < n; ++i) {
    xerr = std::fmax(xerr, std::fabs(x[N][i] - xN[i]));
    yerr = std::fmax(yerr, std::fabs(y[N][i] - yN[i]));
  }
  std::printf("x_err = %.3e  y_err = %.3e\n", xerr, yerr);
  CHECK(xerr < 1e-11 && yerr < 1e-9);   // reference tolerance: 1e-6 / 1e-5

  // failure convention (tvlqr.cpp:162-164): a non-PD Quu returns that knot point
  R[3][0] = -1e6;
  res = tvlqr_BackwardPass(nx, nu, N, A, B, f, Q, R, H, q, r, 0.0, K, d, P, p, delta_V, Qxx, Quu, Qux,
                           Qx, Qu, Qxx_t, Quu_t, Qux_t, Qx_t, Qu_t, false, is_diag);
  CHECK(res == 3);
  res = tvlqr_BackwardPass(nx, nu, N, A, B, f, Q, R, H, q, r, 2e6, K, d, P, p, delta_V, Qxx, Quu, Qux,
                           Qx, Qu, Qxx_t, Quu_t, Qux_t, Qx_t, Qu_t, false, is_diag);
  CHECK(res == TVLQR_SUCCESS);
  {   // latency of ONE call through the kernel boundary (informative: the boundary is for compatibility, the
      // throughput comes from the batched C ABI)
    const auto t0 = std::chrono::steady_clock::now();
    const int reps = 200;
    for (int i = 0; i < reps; ++i)
      tvlqr_BackwardPass(nx, nu, N, A, B, f, Q, R, H, q, r, 2e6, K, d, P, p, delta_V, Qxx, Quu, Qux, Qx, Qu, Qxx_t, Quu_t,
                         Qux_t, Qx_t, Qu_t, false, is_diag);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    std::printf("tvlqr_BackwardPass (N = %d, n = %d, m = %d) through the drop-in: %.1f us per call\n", N, n, m, us);
  }
  std::printf("OK\n");
  return 0;
}
