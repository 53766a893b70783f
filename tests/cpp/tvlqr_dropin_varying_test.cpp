// The tvlqr_* drop-in with PER-KNOT-POINT dimensions: tvlqr.cpp:65-248 takes nx[k], nu[k] per knot point (A_k is nx[k+1] x nx[k],
// B_k nx[k+1] x nu[k], K_k nu[k] x nx[k], ...); the reference's own tests only ever pass uniform ones.  A seeded random problem
// whose state dimension shrinks along the horizon (the reference sizes its scratch blocks by nx[k], so nx[k+1] <= nx[k] is what
// its own buffers allow) and whose input dimension changes every step goes through the device drop-in and through the CPU
// oracle (oracle/tvlqr_oracle.c, linked from oracle/_build/liboracle.so) on identical pointer tables: gains, cost-to-go, expected
// decrease, the Q blocks, the closed-loop trajectory and the duals must agree bit for bit (plan GENERIC's promise), dense and
// diagonal cost.  Needs an MI355X.  Prints "OK" and returns 0 on success.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "tvlqr/tvlqr.h"

extern "C" {
int oracle_tvlqr_TotalMemSize(const int* nx, const int* nu, int num_horizon, bool is_diag);
int oracle_tvlqr_BackwardPass(const int* nx, const int* nu, int num_horizon, const double* const* A, const double* const* B,
                              const double* const* f, const double* const* Q, const double* const* R, const double* const* H,
                              const double* const* q, const double* const* r, double reg, double** K, double** d, double** P, double** p,
                              double* delta_V, double** Qxx, double** Quu, double** Qux, double** Qx, double** Qu, double** Qxx_tmp,
                              double** Quu_tmp, double** Qux_tmp, double** Qx_tmp, double** Qu_tmp, bool linear_only_update, bool is_diag);
int oracle_tvlqr_ForwardPass(const int* nx, const int* nu, int num_horizon, const double* const* A, const double* const* B,
                             const double* const* f, const double* const* K, const double* const* d, const double* const* P,
                             const double* const* p, const double* x0, double** x, double** u, double** y);
}

#define CHECK(cond)                                                     \
  do {                                                                  \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

namespace {
#ifdef BIG_DIMS   // dimensions past 32 (round 5): the knot point's blocks leave LDS for a work block in global memory
constexpr int N = 5;
const int kNx[N + 1] = {48, 48, 40, 40, 36, 33};
const int kNu[N] = {10, 12, 9, 33, 7};
#define DIMS_TEXT "nx = 48..33, nu = 7..33"
#elif defined(MID_DIMS)   // just past 32: the blocks stay in LDS in the form without a block for Qxx ("late Q")
constexpr int N = 6;
const int kNx[N + 1] = {35, 36, 34, 33, 36, 30, 35};
const int kNu[N] = {3, 5, 2, 6, 4, 5};
#define DIMS_TEXT "nx = 30..36, nu = 2..6"
#else
constexpr int N = 9;
const int kNx[N + 1] = {6, 6, 5, 5, 5, 4, 3, 3, 2, 2};
const int kNu[N] = {2, 3, 1, 2, 2, 3, 1, 2, 1};
#define DIMS_TEXT "nx = 6..2, nu = 1..3"
#endif

uint64_t g_state = 0x9E3779B97F4A7C15ull;
double rnd() {   // splitmix64 -> U(-1, 1)
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) / 9007199254740992.0 * 2.0 - 1.0;
}

struct Problem {   // one flat buffer + the reference's pointer tables into it
  std::vector<double> mem;
  double *A[N], *B[N], *f[N], *Q[N + 1], *R[N], *H[N], *q[N + 1], *r[N], *K[N], *d[N], *P[N + 1], *p[N + 1];
  double *Qxx[N], *Quu[N], *Qux[N], *Qx[N], *Qu[N], *Qxx_t[N], *Quu_t[N], *Qux_t[N], *Qx_t[N], *Qu_t[N];
  double *x[N + 1], *u[N], *y[N + 1], *dV;
  void layout(bool is_diag) {
    size_t total = 0;
    auto take = [&](int cnt) { const size_t at = total; total += cnt; return at; };
    std::vector<size_t> off;
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) mem.assign(total, 0.0);
      total = 0;
      auto ptr = [&](int cnt) { const size_t at = take(cnt); return pass ? mem.data() + at : nullptr; };
      for (int k = 0; k <= N; ++k) {
        const int n = kNx[k];
        x[k] = ptr(n); y[k] = ptr(n); Q[k] = ptr(is_diag ? n : n * n); q[k] = ptr(n); P[k] = ptr(n * n); p[k] = ptr(n);
        if (k == N) break;
        const int m = kNu[k], n2 = kNx[k + 1];
        u[k] = ptr(m); A[k] = ptr(n2 * n); B[k] = ptr(n2 * m); f[k] = ptr(n2);
        R[k] = ptr(is_diag ? m : m * m); H[k] = is_diag ? nullptr : ptr(m * n); r[k] = ptr(m);
        K[k] = ptr(m * n); d[k] = ptr(m);
        Qxx[k] = ptr(n * n); Quu[k] = ptr(m * m); Qux[k] = ptr(m * n); Qx[k] = ptr(n); Qu[k] = ptr(m);
        Qxx_t[k] = ptr(n * n); Quu_t[k] = ptr(m * m); Qux_t[k] = ptr(m * n); Qx_t[k] = ptr(n); Qu_t[k] = ptr(m);
      }
      dV = ptr(2);
    }
  }
  void fill(bool is_diag) {
    g_state = 0x1234567ull;
    for (int k = 0; k <= N; ++k) {
      const int n = kNx[k];
      if (is_diag) for (int i = 0; i < n; ++i) Q[k][i] = 1.0 + 0.5 * (rnd() + 1.0);
      else {
        std::vector<double> L(n * n);
        for (auto& v : L) v = 0.3 * rnd();
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j) {
            double s = i == j ? 1.0 : 0.0;
            for (int c = 0; c < n; ++c) s += L[i + c * n] * L[j + c * n];
            Q[k][i + j * n] = s;
          }
      }
      for (int i = 0; i < n; ++i) q[k][i] = 0.2 * rnd();
      if (k == N) break;
      const int m = kNu[k], n2 = kNx[k + 1];
      for (int i = 0; i < n2 * n; ++i) A[k][i] = (n > 8 ? 0.12 : 0.4) * rnd();
      for (int i = 0; i < n2 && i < n; ++i) A[k][i + i * n2] += 1.0;
      for (int i = 0; i < n2 * m; ++i) B[k][i] = 0.5 * rnd();
      for (int i = 0; i < n2; ++i) f[k][i] = 0.1 * rnd();
      if (is_diag) for (int i = 0; i < m; ++i) R[k][i] = 0.2 + 0.1 * (rnd() + 1.0);
      else {
        std::vector<double> M(m * m);
        for (auto& v : M) v = 0.2 * rnd();
        for (int i = 0; i < m; ++i)
          for (int j = 0; j < m; ++j) {
            double s = i == j ? 0.3 : 0.0;
            for (int c = 0; c < m; ++c) s += M[i + c * m] * M[j + c * m];
            R[k][i + j * m] = s;
          }
        for (int i = 0; i < m * n; ++i) H[k][i] = 0.03 * rnd();
      }
      for (int i = 0; i < m; ++i) r[k][i] = 0.1 * rnd();
    }
  }
};

int run(bool is_diag) {
  static Problem dev, ref;
  dev.layout(is_diag); ref.layout(is_diag);
  dev.fill(is_diag); ref.fill(is_diag);
  CHECK(std::memcmp(dev.mem.data(), ref.mem.data(), dev.mem.size() * sizeof(double)) == 0);
  CHECK(tvlqr_TotalMemSize(kNx, kNu, N, is_diag) == oracle_tvlqr_TotalMemSize(kNx, kNu, N, is_diag));
  const int rd = tvlqr_BackwardPass(kNx, kNu, N, dev.A, dev.B, dev.f, dev.Q, dev.R, dev.H, dev.q, dev.r, 0.0, dev.K, dev.d, dev.P, dev.p, dev.dV,
                                    dev.Qxx, dev.Quu, dev.Qux, dev.Qx, dev.Qu, dev.Qxx_t, dev.Quu_t, dev.Qux_t, dev.Qx_t, dev.Qu_t, false, is_diag);
  const int rr = oracle_tvlqr_BackwardPass(kNx, kNu, N, ref.A, ref.B, ref.f, ref.Q, ref.R, ref.H, ref.q, ref.r, 0.0, ref.K, ref.d, ref.P, ref.p,
                                           ref.dV, ref.Qxx, ref.Quu, ref.Qux, ref.Qx, ref.Qu, ref.Qxx_t, ref.Quu_t, ref.Qux_t, ref.Qx_t, ref.Qu_t,
                                           false, is_diag);
  CHECK(rd == TVLQR_SUCCESS && rr == -1);
  double worst = 0.0;
  auto cmp = [&](const double* a, const double* b, int cnt) {
    for (int i = 0; i < cnt; ++i) worst = std::fmax(worst, std::fabs(a[i] - b[i]) / std::fmax(1.0, std::fabs(b[i])));
    return std::memcmp(a, b, cnt * sizeof(double)) == 0;
  };
  bool same = true;
  for (int k = 0; k <= N; ++k) {
    const int n = kNx[k];
    same &= cmp(dev.P[k], ref.P[k], n * n); same &= cmp(dev.p[k], ref.p[k], n);
    if (k == N) break;
    const int m = kNu[k];
    same &= cmp(dev.K[k], ref.K[k], m * n); same &= cmp(dev.d[k], ref.d[k], m);
    same &= cmp(dev.Qxx[k], ref.Qxx[k], n * n); same &= cmp(dev.Quu[k], ref.Quu[k], m * m); same &= cmp(dev.Qux[k], ref.Qux[k], m * n);
    same &= cmp(dev.Qx[k], ref.Qx[k], n); same &= cmp(dev.Qu[k], ref.Qu[k], m);
  }
  same &= cmp(dev.dV, ref.dV, 2);
  std::printf("%s cost, " DIMS_TEXT ": backward pass %s (worst relative difference %.2e)\n", is_diag ? "diagonal" : "dense",
              same ? "bit-identical" : "DIFFERS", worst);
  CHECK(same);
  double x0[64];
  for (int i = 0; i < kNx[0]; ++i) x0[i] = 2.0 * rnd();
  CHECK(tvlqr_ForwardPass(kNx, kNu, N, dev.A, dev.B, dev.f, dev.K, dev.d, dev.P, dev.p, x0, dev.x, dev.u, dev.y) == TVLQR_SUCCESS);
  CHECK(oracle_tvlqr_ForwardPass(kNx, kNu, N, ref.A, ref.B, ref.f, ref.K, ref.d, ref.P, ref.p, x0, ref.x, ref.u, ref.y) == -1);
  same = true; worst = 0.0;
  for (int k = 0; k <= N; ++k) {
    same &= cmp(dev.x[k], ref.x[k], kNx[k]); same &= cmp(dev.y[k], ref.y[k], kNx[k]);
    if (k < N) same &= cmp(dev.u[k], ref.u[k], kNu[k]);
  }
  std::printf("%s cost: forward pass %s (worst relative difference %.2e)\n", is_diag ? "diagonal" : "dense", same ? "bit-identical" : "DIFFERS", worst);
  CHECK(same);
  // failure convention with varying dimensions: an indefinite R at knot point 4 returns 4 from both
  if (is_diag) dev.R[4][0] = ref.R[4][0] = -1e6; else dev.R[4][0] = ref.R[4][0] = -1e6;
  const int fd = tvlqr_BackwardPass(kNx, kNu, N, dev.A, dev.B, dev.f, dev.Q, dev.R, dev.H, dev.q, dev.r, 0.0, dev.K, dev.d, dev.P, dev.p, dev.dV,
                                    dev.Qxx, dev.Quu, dev.Qux, dev.Qx, dev.Qu, dev.Qxx_t, dev.Quu_t, dev.Qux_t, dev.Qx_t, dev.Qu_t, false, is_diag);
  const int fr = oracle_tvlqr_BackwardPass(kNx, kNu, N, ref.A, ref.B, ref.f, ref.Q, ref.R, ref.H, ref.q, ref.r, 0.0, ref.K, ref.d, ref.P, ref.p,
                                           ref.dV, ref.Qxx, ref.Quu, ref.Qux, ref.Qx, ref.Qu, ref.Qxx_t, ref.Quu_t, ref.Qux_t, ref.Qx_t, ref.Qu_t,
                                           false, is_diag);
  CHECK(fd == 4 && fr == 4);
  (void)0;
  return 0;
}
}  // namespace

int main() {
  if (run(false)) return 1;
  if (run(true)) return 1;
  std::printf("OK\n");
  return 0;
}
