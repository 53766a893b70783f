// The tvlqr_BackwardPass seam for ONE problem with uniform dimensions 7 <= n <= 31, m <= 8, n + m <= 32 runs plan MFMA32's backward
// kernel for a batch of one + a wave-per-knot-point kernel for the Q-blocks and their scratch twins (altro_amd/csrc/tvlqr_dropin.hip).
// Every array the reference's signature carries is held against the GENERIC path (ALTRO_TVLQR_DROPIN_GENERIC=1), the one pinned to
// the oracle bit for bit: K, d, P, p, delta_V, Qxx, Quu, Qux, Qx, Qu and the final contents of Qxx_tmp (K^T Qux), Quu_tmp (the
// in-place Cholesky factor of Quu + reg I), Qux_tmp (Quu K), Qx_tmp (K^T Qu), Qu_tmp (Quu d) -- tvlqr.cpp:125-191.  The sweep's sums
// run in the matrix cores' order, so the statement is a tolerance: 1e-9 relative to the largest entry of each array (observed 1e-13;
// the reference's own test holds 1e-6).  Dense and diagonal costs, reg = 0 and reg > 0, one-tile and two-tile shapes, m <= 4 and
// m > 4, and a failing factorisation (which the path hands back to the GENERIC kernel: same status, same partial outputs, bit for
// bit).  Needs an MI355X.  Prints "OK".
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tvlqr/tvlqr.h"

#define CHECK(cond)                                                     \
  do {                                                                  \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

static unsigned long long g_state = 20260929ull;
static double rnd() {   // splitmix64 -> (-1, 1)
  unsigned long long z = (g_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return 2.0 * ((double)(z >> 11) / 9007199254740992.0) - 1.0;
}

struct Problem {
  int N, n, m;
  bool is_diag;
  std::vector<int> nx, nu;
  std::vector<std::vector<double>> A, B, f, Q, R, H, q, r;
  struct Out {
    std::vector<std::vector<double>> K, d, P, p, Qxx, Quu, Qux, Qx, Qu, Qxx_t, Quu_t, Qux_t, Qx_t, Qu_t;
    double dV[2];
    int status;
  };
};

static Problem make(int N, int n, int m, bool is_diag, bool indefinite_at_3) {
  Problem pr{N, n, m, is_diag, std::vector<int>(N + 1, n), std::vector<int>(N, m), {}, {}, {}, {}, {}, {}, {}, {}};
  for (int k = 0; k <= N; ++k) {
    std::vector<double> Qk(is_diag ? n : n * n), qk(n);
    if (is_diag) for (int i = 0; i < n; ++i) Qk[i] = 1.0 + 0.5 * std::fabs(rnd());
    else {
      std::vector<double> Lm(n * n);
      for (auto& v : Lm) v = 0.3 * rnd();
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) { double s = (i == j) ? 1.0 : 0.0; for (int c = 0; c < n; ++c) s += Lm[i + c * n] * Lm[j + c * n]; Qk[i + j * n] = s; }
    }
    for (auto& v : qk) v = 0.1 * rnd();
    pr.Q.push_back(Qk); pr.q.push_back(qk);
    if (k == N) break;
    std::vector<double> Ak(n * n), Bk(n * m), fk(n), Rk(is_diag ? m : m * m), Hk(m * n), rk(m);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Ak[i + j * n] = (i == j ? 1.0 : 0.0) + 0.05 * rnd();
    for (auto& v : Bk) v = 0.1 * rnd();
    for (auto& v : fk) v = 0.01 * rnd();
    if (is_diag) for (int i = 0; i < m; ++i) Rk[i] = 0.1 + 0.1 * std::fabs(rnd());
    else {
      std::vector<double> Mm(m * m);
      for (auto& v : Mm) v = 0.1 * rnd();
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) { double s = (i == j) ? 0.1 : 0.0; for (int c = 0; c < m; ++c) s += Mm[i + c * m] * Mm[j + c * m]; Rk[i + j * m] = s; }
    }
    if (indefinite_at_3 && k == 3) Rk[0] = -1e6;
    for (auto& v : Hk) v = 0.01 * rnd();
    for (auto& v : rk) v = 0.1 * rnd();
    pr.A.push_back(Ak); pr.B.push_back(Bk); pr.f.push_back(fk); pr.R.push_back(Rk); pr.H.push_back(Hk); pr.r.push_back(rk);
  }
  return pr;
}

static Problem::Out run(Problem& pr, double reg, bool generic) {
  if (generic) setenv("ALTRO_TVLQR_DROPIN_GENERIC", "1", 1); else unsetenv("ALTRO_TVLQR_DROPIN_GENERIC");
  const int N = pr.N, n = pr.n, m = pr.m;
  Problem::Out o;
  auto alloc = [&](std::vector<std::vector<double>>& v, int cnt, int len) { v.assign(cnt, std::vector<double>(len, -7.0)); };
  alloc(o.K, N, m * n); alloc(o.d, N, m); alloc(o.P, N + 1, n * n); alloc(o.p, N + 1, n);
  alloc(o.Qxx, N, n * n); alloc(o.Quu, N, m * m); alloc(o.Qux, N, m * n); alloc(o.Qx, N, n); alloc(o.Qu, N, m);
  alloc(o.Qxx_t, N, n * n); alloc(o.Quu_t, N, m * m); alloc(o.Qux_t, N, m * n); alloc(o.Qx_t, N, n); alloc(o.Qu_t, N, m);
  auto cptrs = [](std::vector<std::vector<double>>& v) { std::vector<const double*> p; for (auto& e : v) p.push_back(e.data()); return p; };
  auto ptrs = [](std::vector<std::vector<double>>& v) { std::vector<double*> p; for (auto& e : v) p.push_back(e.data()); return p; };
  auto A = cptrs(pr.A), B = cptrs(pr.B), f = cptrs(pr.f), Q = cptrs(pr.Q), R = cptrs(pr.R), H = cptrs(pr.H), q = cptrs(pr.q), r = cptrs(pr.r);
  auto K = ptrs(o.K), d = ptrs(o.d), P = ptrs(o.P), p = ptrs(o.p), Qxx = ptrs(o.Qxx), Quu = ptrs(o.Quu), Qux = ptrs(o.Qux), Qx = ptrs(o.Qx),
       Qu = ptrs(o.Qu), Qxx_t = ptrs(o.Qxx_t), Quu_t = ptrs(o.Quu_t), Qux_t = ptrs(o.Qux_t), Qx_t = ptrs(o.Qx_t), Qu_t = ptrs(o.Qu_t);
  o.status = tvlqr_BackwardPass(pr.nx.data(), pr.nu.data(), N, A.data(), B.data(), f.data(), Q.data(), R.data(), pr.is_diag ? nullptr : H.data(),
                                q.data(), r.data(), reg, K.data(), d.data(), P.data(), p.data(), o.dV, Qxx.data(), Quu.data(), Qux.data(),
                                Qx.data(), Qu.data(), Qxx_t.data(), Quu_t.data(), Qux_t.data(), Qx_t.data(), Qu_t.data(), false, pr.is_diag);
  unsetenv("ALTRO_TVLQR_DROPIN_GENERIC");
  return o;
}

static bool same(const std::vector<std::vector<double>>& a, const std::vector<std::vector<double>>& b, const char* what) {
  for (size_t k = 0; k < a.size(); ++k)
    if (std::memcmp(a[k].data(), b[k].data(), a[k].size() * sizeof(double)) != 0) {
      std::printf("  %s differs at knot point %zu\n", what, k);
      return false;
    }
  return true;
}
static double g_worst = 0.0;
static bool close_to(const std::vector<std::vector<double>>& a, const std::vector<std::vector<double>>& b, const char* what) {
  double scale = 0.0, err = 0.0;
  for (size_t k = 0; k < a.size(); ++k)
    for (size_t e = 0; e < a[k].size(); ++e) {
      scale = std::fmax(scale, std::fabs(b[k][e]));
      err = std::fmax(err, std::fabs(a[k][e] - b[k][e]));
      if (!(a[k][e] == a[k][e])) err = 1e300;   // NaN
    }
  const double rel = err / std::fmax(scale, 1e-300);
  if (scale > 0.0) g_worst = std::fmax(g_worst, rel);
  if (rel > 1e-9) { std::printf("  %s: |tile - generic| = %.3g against entries up to %.3g\n", what, err, scale); return false; }
  return true;
}

int main() {
  const int shapes[][2] = {{12, 4}, {7, 1}, {8, 4}, {13, 4}, {16, 4}, {14, 7}, {9, 8}, {20, 8}, {24, 8}, {28, 4}, {31, 1}, {5, 5}};
  int cases = 0;
  for (auto& sh : shapes)
    for (int diag = 0; diag < 2; ++diag)
      for (int wr = 0; wr < 2; ++wr) {
        const int N = 3 + (int)(9 * std::fabs(rnd()));
        Problem pr = make(N, sh[0], sh[1], diag != 0, false);
        const double reg = wr ? 0.3 : 0.0;
        Problem::Out tile = run(pr, reg, false), gen = run(pr, reg, true);
        CHECK(tile.status == TVLQR_SUCCESS && gen.status == TVLQR_SUCCESS);
        CHECK(close_to(tile.K, gen.K, "K") && close_to(tile.d, gen.d, "d") && close_to(tile.P, gen.P, "P") && close_to(tile.p, gen.p, "p"));
        CHECK(close_to(tile.Qxx, gen.Qxx, "Qxx") && close_to(tile.Quu, gen.Quu, "Quu") && close_to(tile.Qux, gen.Qux, "Qux"));
        CHECK(close_to(tile.Qx, gen.Qx, "Qx") && close_to(tile.Qu, gen.Qu, "Qu"));
        CHECK(close_to(tile.Qxx_t, gen.Qxx_t, "Qxx_tmp") && close_to(tile.Quu_t, gen.Quu_t, "Quu_tmp") && close_to(tile.Qux_t, gen.Qux_t, "Qux_tmp"));
        CHECK(close_to(tile.Qx_t, gen.Qx_t, "Qx_tmp") && close_to(tile.Qu_t, gen.Qu_t, "Qu_tmp"));
        CHECK(std::fabs(tile.dV[0] - gen.dV[0]) <= 1e-9 * std::fmax(1.0, std::fabs(gen.dV[0])));
        CHECK(std::fabs(tile.dV[1] - gen.dV[1]) <= 1e-9 * std::fmax(1.0, std::fabs(gen.dV[1])));
        ++cases;
      }
  {   // a failing factorisation: same status and same partial outputs (the path repeats the call on the GENERIC kernel)
    Problem pr = make(8, 12, 4, true, true);
    Problem::Out tile = run(pr, 0.0, false), gen = run(pr, 0.0, true);
    CHECK(tile.status == 3 && gen.status == 3);
    CHECK(same(tile.K, gen.K, "K") && same(tile.P, gen.P, "P") && same(tile.Qxx_t, gen.Qxx_t, "Qxx_tmp") && same(tile.Quu_t, gen.Quu_t, "Quu_tmp"));
    Problem::Out f2 = run(pr, 2e6, false), g2 = run(pr, 2e6, true);
    CHECK(f2.status == TVLQR_SUCCESS && g2.status == TVLQR_SUCCESS && close_to(f2.K, g2.K, "K") && close_to(f2.P, g2.P, "P"));
  }
  std::printf("%d cases: the matrix-core path agrees with the GENERIC path, worst relative difference %.3g\nOK\n", cases, g_worst);
  return 0;
}
