// ALTROSolver with PER-KNOT-POINT dimensions (ALTROSolver::SetDimension(n, m, k_start, k_stop), altro_solver.cpp:26-47): a linear
// time-varying problem between state spaces of changing size, a dense quadratic cost, written against include/altro/altro.hpp the
// way a user of the reference would -- one SetDimension / SetExplicitDynamics / SetQuadraticCost per knot point.  Every backward
// sweep runs on the GPU through tvlqr_BackwardPass with the nx[k], nu[k] the reference hands it (tvlqr.cpp:65-248).
//
//   altro_varying_dims_test.bin <problem.txt> [constrained]
// constrained: |u_i| <= 0.25 where the input has three entries, u_0 <= 0.1 where it has one (INEQUALITY blocks, SetConstraint per knot
// point with that knot point's own n + m columns), x_N[0] = 0.3 and x_N[2] = -0.2 (an EQUALITY block at the terminal knot point).
// problem.txt: N, then nx[0..N], nu[0..N-1], then per k < N: A_k (nx[k+1] x nx[k] column-major), B_k, f_k, R_k, H_k (nu x nx), r_k;
// per k <= N: Q_k, q_k, c_k; then x0 and u0_k.  Prints: status, iterations, then x_0 .. x_N and u_0 .. u_(N-1), one value per token.
// tests/test_gpu_ragged_ilqr.py compares them with the batched ABI on the same problem and with the oracle on the padded one.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "altro/altro.hpp"

using namespace altro;

namespace {
bool read_vec(std::FILE* f, std::vector<double>& v, size_t count) {
  v.resize(count);
  for (size_t i = 0; i < count; ++i)
    if (std::fscanf(f, "%lf", &v[i]) != 1) return false;
  return true;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s problem.txt\n", argv[0]); return 2; }
  std::FILE* f = std::fopen(argv[1], "r");
  if (!f) { std::perror(argv[1]); return 2; }
  int N = 0;
  if (std::fscanf(f, "%d", &N) != 1 || N < 1) return 2;
  std::vector<int> nx(N + 1), nu(N);
  for (int k = 0; k <= N; ++k) if (std::fscanf(f, "%d", &nx[k]) != 1) return 2;
  for (int k = 0; k < N; ++k) if (std::fscanf(f, "%d", &nu[k]) != 1) return 2;
  std::vector<std::vector<double>> A(N), B(N), fv(N), R(N), H(N), r(N), Q(N + 1), q(N + 1), u0(N);
  std::vector<double> c(N + 1), x0;
  for (int k = 0; k < N; ++k) {
    const size_t n = nx[k], m = nu[k], n2 = nx[k + 1];
    if (!read_vec(f, A[k], n2 * n) || !read_vec(f, B[k], n2 * m) || !read_vec(f, fv[k], n2) || !read_vec(f, R[k], m * m) ||
        !read_vec(f, H[k], m * n) || !read_vec(f, r[k], m))
      return 2;
  }
  for (int k = 0; k <= N; ++k) {
    const size_t n = nx[k];
    if (!read_vec(f, Q[k], n * n) || !read_vec(f, q[k], n) || std::fscanf(f, "%lf", &c[k]) != 1) return 2;
  }
  if (!read_vec(f, x0, nx[0])) return 2;
  for (int k = 0; k < N; ++k) if (!read_vec(f, u0[k], nu[k])) return 2;
  std::fclose(f);

  int bad = 0;
  auto ok = [&bad](ErrorCodes e) { if (e != ErrorCodes::NoError) ++bad; };
  ALTROSolver solver(N);
  for (int k = 0; k < N; ++k) ok(solver.SetDimension(nx[k], nu[k], k));
  ok(solver.SetDimension(nx[N], nu[N - 1], N));    // (the reference wants an input dimension at the terminal knot point too)
  ok(solver.SetTimeStep(0.01f));
  for (int k = 0; k < N; ++k) {
    const int n = nx[k], m = nu[k], n2 = nx[k + 1];
    const double *Ak = A[k].data(), *Bk = B[k].data(), *fk = fv[k].data();
    auto dyn = [=](double* xn, const double* x, const double* u, float) {
      for (int i = 0; i < n2; ++i) {
        double s = 0.0, s2 = 0.0;
        for (int j = 0; j < n; ++j) s += Ak[i + j * n2] * x[j];
        for (int j = 0; j < m; ++j) s2 += Bk[i + j * n2] * u[j];
        xn[i] = (s + s2) + fk[i];
      }
    };
    auto jac = [=](double* J, const double*, const double*, float) {   // n2 x (n + m) column-major = [A B]
      for (int e = 0; e < n2 * n; ++e) J[e] = Ak[e];
      for (int e = 0; e < n2 * m; ++e) J[n2 * n + e] = Bk[e];
    };
    ok(solver.SetExplicitDynamics(dyn, jac, k));
    ok(solver.SetQuadraticCost(n, m, Q[k].data(), R[k].data(), H[k].data(), q[k].data(), r[k].data(), c[k], k));
  }
  {   // terminal cost: Q_N, q_N, c_N (R, H, r of the right size and of no effect)
    const int n = nx[N], m = nu[N - 1];
    std::vector<double> Rn((size_t)m * m, 0.0), Hn((size_t)m * n, 0.0), rn(m, 0.0);
    for (int i = 0; i < m; ++i) Rn[i + i * m] = 1.0;
    ok(solver.SetQuadraticCost(n, m, Q[N].data(), Rn.data(), Hn.data(), q[N].data(), rn.data(), c[N], N));
  }
  const bool constrained = argc > 2;
  if (constrained) {
    for (int k = 0; k < N; ++k) {
      const int n = nx[k], m = nu[k];
      if (m == 3) {
        auto con = [n](a_float* c, const a_float*, const a_float* u) { for (int i = 0; i < 3; ++i) { c[i] = u[i] - 0.25; c[3 + i] = -u[i] - 0.25; } };
        auto jac = [n](a_float* J, const a_float*, const a_float*) {     // 6 x (n + 3) column-major
          for (int e = 0; e < 6 * (n + 3); ++e) J[e] = 0.0;
          for (int i = 0; i < 3; ++i) { J[i + (n + i) * 6] = 1.0; J[3 + i + (n + i) * 6] = -1.0; }
        };
        ok(solver.SetConstraint(con, jac, 6, ConstraintType::INEQUALITY, "input box", k));
      } else if (m == 1) {
        auto con = [](a_float* c, const a_float*, const a_float* u) { c[0] = u[0] - 0.1; };
        auto jac = [n](a_float* J, const a_float*, const a_float*) { for (int e = 0; e < n + 1; ++e) J[e] = 0.0; J[n] = 1.0; };
        ok(solver.SetConstraint(con, jac, 1, ConstraintType::INEQUALITY, "input bound", k));
      }
    }
    const int n = nx[N], m = nu[N - 1];
    auto con = [](a_float* c, const a_float* x, const a_float*) { c[0] = x[0] - 0.3; c[1] = x[2] + 0.2; };
    auto jac = [n, m](a_float* J, const a_float*, const a_float*) {       // 2 x (n + m) column-major
      for (int e = 0; e < 2 * (n + m); ++e) J[e] = 0.0;
      J[0 + 0 * 2] = 1.0; J[1 + 2 * 2] = 1.0;
    };
    ok(solver.SetConstraint(con, jac, 2, ConstraintType::EQUALITY, "goal", N));
  }
  ok(solver.SetInitialState(x0.data(), nx[0]));
  ok(solver.Initialize());
  for (int k = 0; k < N; ++k) ok(solver.SetInput(u0[k].data(), nu[k], k));
  if (bad) { std::printf("setup failed (%d API errors)\n", bad); return 1; }

  AltroOptions opts;
  opts.verbose = Verbosity::Silent;
  opts.iterations_max = 60;
  opts.tol_stationarity = 1e-4;
  solver.SetOptions(opts);
  const SolveStatus status = solver.Solve();
  std::printf("status %d iterations %d\n", (int)status, solver.GetIterations());
  std::vector<double> buf(64);
  std::printf("x");
  for (int k = 0; k <= N; ++k) {
    ok(solver.GetState(buf.data(), k));
    for (int i = 0; i < nx[k]; ++i) std::printf(" %.17g", buf[i]);
  }
  std::printf("\nu");
  for (int k = 0; k < N; ++k) {
    ok(solver.GetInput(buf.data(), k));
    for (int i = 0; i < nu[k]; ++i) std::printf(" %.17g", buf[i]);
  }
  std::printf("\n%s\n", bad ? "FAILED" : "OK");
  return bad ? 1 : 0;
}
