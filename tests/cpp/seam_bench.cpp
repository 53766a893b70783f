// What ONE call through the kernel seam costs: tvlqr_BackwardPass / tvlqr_ForwardPass of include/tvlqr/tvlqr.h
// (the reference's own signatures, executed on the GPU: one small problem, one wavefront) against the CPU port of the same two
// functions (oracle/tvlqr_oracle.c) on the same pointer tables.  bench.py runs this for its `single_problem_seam` entry: the
// single-problem API stays a drop-in, but a lone small problem is latency-bound on a GPU -- the number says by how much.
//   seam_bench.bin            -> one JSON object on stdout
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "tvlqr/tvlqr.h"

extern "C" {
int oracle_tvlqr_BackwardPass(const int* nx, const int* nu, int num_horizon, const double* const* A, const double* const* B,
                              const double* const* f, const double* const* Q, const double* const* R, const double* const* H,
                              const double* const* q, const double* const* r, double reg, double** K, double** d, double** P, double** p,
                              double* delta_V, double** Qxx, double** Quu, double** Qux, double** Qx, double** Qu, double** Qxx_tmp,
                              double** Quu_tmp, double** Qux_tmp, double** Qx_tmp, double** Qu_tmp, bool linear_only_update, bool is_diag);
}

namespace {
uint64_t g_state = 0x9E3779B97F4A7C15ull;
double rnd() {
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) / 9007199254740992.0 * 2.0 - 1.0;
}

struct Problem {
  int N, n, m;
  std::vector<int> nx, nu;
  std::vector<double> mem;
  std::vector<double*> A, B, f, Q, R, H, q, r, K, d, P, p, Qxx, Quu, Qux, Qx, Qu, Qxx_t, Quu_t, Qux_t, Qx_t, Qu_t;
  double dV[2];
  Problem(int N_, int n_, int m_) : N(N_), n(n_), m(m_), nx(N_ + 1, n_), nu(N_, m_) {
    const size_t per = (size_t)8 * n * n + 6 * m * n + 4 * m * m + 8 * n + 6 * m;
    mem.assign(per * (N + 1), 0.0);
    size_t at = 0;
    auto take = [&](int cnt) { double* ptr = mem.data() + at; at += cnt; return ptr; };
    for (auto* v : {&A, &B, &f, &R, &H, &r, &K, &d, &Qxx, &Quu, &Qux, &Qx, &Qu, &Qxx_t, &Quu_t, &Qux_t, &Qx_t, &Qu_t}) v->resize(N);
    for (auto* v : {&Q, &q, &P, &p}) v->resize(N + 1);
    for (int k = 0; k <= N; ++k) {
      Q[k] = take(n * n); q[k] = take(n); P[k] = take(n * n); p[k] = take(n);
      for (int i = 0; i < n; ++i) { Q[k][i + i * n] = 1.0 + 0.3 * (rnd() + 1.0); q[k][i] = 0.1 * rnd(); }
      if (k == N) break;
      A[k] = take(n * n); B[k] = take(n * m); f[k] = take(n); R[k] = take(m * m); H[k] = take(m * n); r[k] = take(m);
      K[k] = take(m * n); d[k] = take(m);
      Qxx[k] = take(n * n); Quu[k] = take(m * m); Qux[k] = take(m * n); Qx[k] = take(n); Qu[k] = take(m);
      Qxx_t[k] = take(n * n); Quu_t[k] = take(m * m); Qux_t[k] = take(m * n); Qx_t[k] = take(n); Qu_t[k] = take(m);
      for (int i = 0; i < n * n; ++i) A[k][i] = 0.1 * rnd();
      for (int i = 0; i < n; ++i) A[k][i + i * n] += 1.0;
      for (int i = 0; i < n * m; ++i) B[k][i] = 0.3 * rnd();
      for (int i = 0; i < m; ++i) { R[k][i + i * m] = 0.1 + 0.05 * (rnd() + 1.0); r[k][i] = 0.05 * rnd(); }
    }
  }
  int device() {
    return tvlqr_BackwardPass(nx.data(), nu.data(), N, A.data(), B.data(), f.data(), Q.data(), R.data(), H.data(), q.data(), r.data(), 0.0,
                              K.data(), d.data(), P.data(), p.data(), dV, Qxx.data(), Quu.data(), Qux.data(), Qx.data(), Qu.data(),
                              Qxx_t.data(), Quu_t.data(), Qux_t.data(), Qx_t.data(), Qu_t.data(), false, false);
  }
  int cpu() {
    return oracle_tvlqr_BackwardPass(nx.data(), nu.data(), N, A.data(), B.data(), f.data(), Q.data(), R.data(), H.data(), q.data(), r.data(), 0.0,
                                     K.data(), d.data(), P.data(), p.data(), dV, Qxx.data(), Quu.data(), Qux.data(), Qx.data(), Qu.data(),
                                     Qxx_t.data(), Quu_t.data(), Qux_t.data(), Qx_t.data(), Qu_t.data(), false, false);
  }
};

template <typename F>
double micros(F fn, int reps) {
  std::vector<double> t(reps);
  for (int i = 0; i < reps; ++i) {
    auto a = std::chrono::steady_clock::now();
    fn();
    t[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
  }
  std::sort(t.begin(), t.end());
  return t[reps / 2];
}
}  // namespace

int main() {
  const int shapes[][3] = {{10, 4, 2}, {50, 4, 2}, {10, 12, 4}, {10, 20, 8}, {50, 28, 4}};
  std::printf("{\"what\": \"median microseconds of ONE tvlqr_BackwardPass call (reference signature, include/tvlqr/tvlqr.h): the GPU seam -- one small "
              "problem, one wavefront, pinned staging, one wait -- against the CPU port of the same function (oracle/tvlqr_oracle.c, this host, "
              "one thread)\", \"shapes\": [");
  bool first = true;
  for (auto& sh : shapes) {
    Problem pr(sh[0], sh[1], sh[2]);
    for (int i = 0; i < 5; ++i) if (pr.device() != TVLQR_SUCCESS) { std::printf("]} \n"); return 1; }
    const double dev = micros([&] { pr.device(); }, 200);
    const double cpu = micros([&] { pr.cpu(); }, 200);
    std::printf("%s{\"N\": %d, \"n\": %d, \"m\": %d, \"seam_us\": %.1f, \"cpu_port_us\": %.1f}", first ? "" : ", ", sh[0], sh[1], sh[2], dev, cpu);
    first = false;
  }
  std::printf("]}\n");
  return 0;
}
