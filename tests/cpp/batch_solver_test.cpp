// tests/cpp/batch_solver_test.cpp -- altro::hip::BatchSolver (include/altro_hip/altro_hip.hpp) on the GPU: the
// constrained double-integrator problems of test/double_integrator_test.cpp:170-376 for a whole batch, through the
// C++ wrapper of the C ABI.  Prints PASS / FAIL lines; the exit code is the number of failures.
#include <cmath>
#include <cstdio>
#include <vector>

#include "altro_hip/altro_hip.hpp"

static int failures = 0;
#define EXPECT(cond)                                                       \
  do {                                                                     \
    if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++failures; } \
  } while (0)

int main() {
  using altro::hip::BatchSolver;
  using altro::hip::Cone;
  const int N = 10, n = 4, m = 2, batch = 100;
  const float h = 5.0f / 10.0f;
  BatchSolver solver(N, n, m, batch);
  EXPECT(solver.GetPlan() == ALTRO_HIP_PLAN_LANE);
  solver.SetModel(ALTRO_HIP_MODEL_DOUBLE_INTEGRATOR, h);
  const double Qd[2 * 4] = {1, 1, 1, 1, 1, 1, 1, 1}, Rd[2] = {1e-2, 1e-2};
  const double xref[2 * 4] = {0, 0, 0, 0, 0, 0, 0, 0}, uref[2] = {0, 0};
  solver.SetLQRCost(Qd, Rd, xref, uref, true, true);
  std::vector<double> x0(batch * n);
  for (int b = 0; b < batch; ++b) { x0[b * n + 0] = 2.0; x0[b * n + 1] = 2.0; x0[b * n + 2] = 0; x0[b * n + 3] = 0; }
  solver.SetInitialState(x0.data());
  const double u0[2] = {0, 0};
  solver.SetInput(u0, true, true);
  // goal constraint at k = N (EQUALITY), control bounds |u| <= 1 at k < N (INEQUALITY): double_integrator_test.cpp:276-334
  double Gg[4 * 6] = {0};
  for (int i = 0; i < 4; ++i) Gg[i + i * 4] = 1.0;
  const double gg[4] = {0, 0, 0, 0};
  solver.SetConstraint(N, N, Cone::Equality, 4, Gg, gg);
  double Gb[4 * 6] = {0};
  for (int i = 0; i < 2; ++i) { Gb[i + (4 + i) * 4] = 1.0; Gb[(i + 2) + (4 + i) * 4] = -1.0; }
  const double gb[4] = {1, 1, 1, 1};
  solver.SetConstraint(0, N - 1, Cone::Inequality, 4, Gb, gb);
  solver.opts.penalty_initial = 100;
  solver.opts.penalty_scaling = 100;
  auto res = solver.Solve();
  EXPECT(res.NumConverged() == batch);
  for (int b : {0, 57, 99}) EXPECT(res.problems[b].iterations == 5);          // double_integrator_test.cpp:374
  std::vector<double> x(batch * (N + 1) * n), u(batch * N * m);
  solver.GetTrajectory(x.data(), u.data());
  for (int b : {0, 99}) {
    double dist = 0;
    for (int i = 0; i < n; ++i) dist += x[(b * (N + 1) + N) * n + i] * x[(b * (N + 1) + N) * n + i];
    EXPECT(std::sqrt(dist) < 1e-4);
    EXPECT(std::fabs(u[b * N * m + 0] + 1.0) < 1e-4 && std::fabs(u[b * N * m + 1] + 1.0) < 1e-4);   // saturated
  }
  {   // the same problems with the cost handed over as dense blocks (SetQuadraticCost; H = 0, diagonal Q and R): the same solves, bit for bit
    BatchSolver dense(N, n, m, batch);
    dense.SetModel(ALTRO_HIP_MODEL_DOUBLE_INTEGRATOR, h);
    double Q[2 * 16] = {0}, R[4] = {0}, H[8] = {0}, q[2 * 4] = {0}, r[2] = {0}, c[2] = {0};
    for (int i = 0; i < 4; ++i) { Q[i + 4 * i] = 1.0; Q[16 + i + 4 * i] = 1.0; }
    R[0] = R[3] = 1e-2;
    dense.SetQuadraticCost(Q, R, H, q, r, c, true, true);
    dense.SetInitialState(x0.data());
    dense.SetInput(u0, true, true);
    dense.SetConstraint(N, N, Cone::Equality, 4, Gg, gg);
    dense.SetConstraint(0, N - 1, Cone::Inequality, 4, Gb, gb);
    dense.opts.penalty_initial = 100;
    dense.opts.penalty_scaling = 100;
    auto rd = dense.Solve();
    EXPECT(rd.NumConverged() == batch);
    std::vector<double> xd(batch * (N + 1) * n), ud(batch * N * m);
    dense.GetTrajectory(xd.data(), ud.data());
    bool same = true;
    for (size_t i = 0; i < xd.size(); ++i) same = same && xd[i] == x[i];
    for (size_t i = 0; i < ud.size(); ++i) same = same && ud[i] == u[i];
    for (int b = 0; b < batch; ++b) same = same && rd.problems[b].iterations == res.problems[b].iterations;
    EXPECT(same);
  }
  bool threw = false;
  try { solver.SetConstraint(0, 99, Cone::Equality, 4, Gg, gg); } catch (const std::runtime_error&) { threw = true; }
  EXPECT(threw);
  std::printf(failures ? "batch_solver_test: %d FAILURES\n" : "batch_solver_test: PASS\n", failures);
  return failures;
}
