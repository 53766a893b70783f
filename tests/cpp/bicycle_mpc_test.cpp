// The reference's receding-horizon scenario (test/bicycle_test.cpp:140-229 fixture `BicycleMPC`, :266-359
// `TrackingMPC_2Solves`) written against include/altro/altro.hpp: the kinematic bicycle in the centre-of-gravity frame
// (test/test_utils.cpp:134-238) under the explicit midpoint rule (:84-132) tracks the "scotty" path for Nsim = 200 steps;
// every backward sweep of every solve runs on the GPU through tvlqr_BackwardPass.
//
//   bicycle_mpc_test.bin <path.txt> [Nsim]      path.txt: "count" then count lines "px py theta delta v deltadot"
//
// Prints, for tests/test_gpu_scotty.py to compare with tests/golden/scotty_mpc_expected.json:
//   step <i> iters <it> status <s> u <u0 u1> x <x0..x3> err <tracking error>
//   Average rate = <Hz>            (bicycle_test.cpp:342)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "altro/altro.hpp"

#include "altro_hip/altro_hip.h"
using namespace altro;

namespace {
constexpr int n = 4, m = 2, N = 30;
constexpr double kLength = 2.7, kRear = 1.5;   // test_utils.hpp:142-143

void bike_f(double* xd, const double* x, const double* u) {
  const double v = u[0], theta = x[2], delta = x[3];
  const double beta = std::atan2(kRear * delta, kLength);
  xd[0] = v * std::cos(theta + beta);
  xd[1] = v * std::sin(theta + beta);
  xd[2] = v * std::cos(beta) * std::tan(delta) / kLength;
  xd[3] = u[1];
}

// J is 4 x 6 column-major = [df/dx df/du]
void bike_J(double* J, const double* x, const double* u) {
  const double v = u[0], theta = x[2], delta = x[3];
  const double by = kRear * delta, bx = kLength;
  const double beta = std::atan2(by, bx);
  const double dbeta = bx / (bx * bx + by * by) * kRear;
  const double s = std::sin(theta + beta), c = std::cos(theta + beta);
  std::memset(J, 0, sizeof(double) * n * (n + m));
  J[0 + 2 * n] = v * -s;
  J[0 + 3 * n] = v * (-s * dbeta);
  J[0 + 4 * n] = c;
  J[1 + 2 * n] = v * c;
  J[1 + 3 * n] = v * (c * dbeta);
  J[1 + 4 * n] = s;
  J[2 + 3 * n] = v / kLength * (-std::sin(beta) * std::tan(delta) * dbeta + std::cos(beta) / (std::cos(delta) * std::cos(delta)));
  J[2 + 4 * n] = std::cos(beta) * std::tan(delta) / kLength;
  J[3 + 5 * n] = 1.0;
}

void step(double* xn, const double* x, const double* u, float h) {
  double xm[n];
  bike_f(xm, x, u);
  for (int i = 0; i < n; ++i) xm[i] *= h / 2;
  for (int i = 0; i < n; ++i) xm[i] += x[i];
  bike_f(xn, xm, u);
  for (int i = 0; i < n; ++i) xn[i] = x[i] + h * xn[i];
}

void step_jac(double* J, const double* x, const double* u, float h) {
  double xm[n], J0[n * (n + m)], Jm[n * (n + m)], T[n * n];
  bike_f(xm, x, u);
  for (int i = 0; i < n; ++i) xm[i] = x[i] + h / 2 * xm[i];
  bike_J(J0, x, u);
  bike_J(Jm, xm, u);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) T[i + j * n] = (i == j ? 1.0 : 0.0) + h / 2 * J0[i + j * n];
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      double s = 0;
      for (int k = 0; k < n; ++k) s += (h * Jm[i + k * n]) * T[k + j * n];
      J[i + j * n] = (i == j ? 1.0 : 0.0) + s;
    }
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < n; ++i) {
      double s = 0;
      for (int k = 0; k < n; ++k) s += (Jm[i + k * n] * (double)(h / 2)) * J0[k + (n + j) * n];
      J[i + (n + j) * n] = h * (s + Jm[i + (n + j) * n]);
    }
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s path.txt [Nsim]\n", argv[0]); return 2; }
  std::FILE* f = std::fopen(argv[1], "r");
  if (!f) { std::perror(argv[1]); return 2; }
  int count = 0;
  if (std::fscanf(f, "%d", &count) != 1 || count < N + 2) return 2;
  std::vector<double> xr(count * n), ur(count * m);
  for (int k = 0; k < count; ++k) {
    for (int i = 0; i < n; ++i) if (std::fscanf(f, "%lf", &xr[k * n + i]) != 1) return 2;
    for (int i = 0; i < m; ++i) if (std::fscanf(f, "%lf", &ur[k * m + i]) != 1) return 2;
  }
  std::fclose(f);
  const int Nsim = argc > 2 ? std::atoi(argv[2]) : 200;
  if (Nsim + N + 1 > count) return 2;
  const float h = 0.1f;   // the saved run's tf / N (tests/golden/make_scotty_fixtures.py)
  const double Qd[n] = {1e-2, 1e-2, 1e-2, 1e-2}, Rd[m] = {1e-3, 1e-3};

  int bad = 0;
  auto ok = [&bad](ErrorCodes e) { if (e != ErrorCodes::NoError) ++bad; };
  ALTROSolver solver(N);
  ok(solver.SetDimension(n, m));
#ifdef DEVICE_MODEL   // the whole Solve on the device (ALTROSolver::SetDeviceModel: altro_hip_ilqr_solve on a resident batch of one)
  ok(solver.SetDeviceModel(ALTRO_HIP_MODEL_BICYCLE, 0, kLength, kRear));
#else
  ok(solver.SetExplicitDynamics(step, step_jac));
#endif
  ok(solver.SetTimeStep(h));
  for (int k = 0; k <= N; ++k) ok(solver.SetLQRCost(n, m, Qd, Rd, &xr[k * n], &ur[k * m], k));
  const double delta_max = 60 * M_PI / 180.0;
  auto con = [delta_max](a_float* c, const a_float* x, const a_float*) { c[0] = x[3] - delta_max; c[1] = -delta_max - x[3]; };
  auto con_jac = [](a_float* J, const a_float*, const a_float*) {
    std::memset(J, 0, sizeof(a_float) * 2 * (n + m));
    J[0 + 3 * 2] = 1.0;
    J[1 + 3 * 2] = -1.0;
  };
#ifdef DEVICE_MODEL
  (void)con; (void)con_jac;
  {
    double G[2 * (n + m)] = {0}, Gt[2 * n] = {0};
    const double g[2] = {delta_max, delta_max};
    G[0 + 3 * 2] = 1.0; G[1 + 3 * 2] = -1.0; Gt[0 + 3 * 2] = 1.0; Gt[1 + 3 * 2] = -1.0;
    ok(solver.SetLinearConstraint(G, g, 2, ConstraintType::INEQUALITY, "steering angle bound", 0, N));
    ok(solver.SetLinearConstraint(Gt, g, 2, ConstraintType::INEQUALITY, "steering angle bound", N, N + 1));
  }
#else
  ok(solver.SetConstraint(con, con_jac, 2, ConstraintType::INEQUALITY, "steering angle bound", 0, N + 1));
#endif
  ok(solver.SetInitialState(&xr[0], n));
  ok(solver.Initialize());
  const double u0[m] = {ur[0], 0.0};
  ok(solver.SetInput(u0, m));
  for (int k = 0; k <= N; ++k) ok(solver.SetState(&xr[k * n], n, k));
  if (bad) { std::printf("setup failed (%d)\n", bad); return 1; }

  AltroOptions opts;
  opts.verbose = Verbosity::Silent;
  opts.iterations_max = 80;
  opts.use_backtracking_linesearch = true;
  solver.SetOptions(opts);

  const double c_u = 0.5 * (u0[0] * (Rd[0] * u0[0]) + u0[1] * (Rd[1] * u0[1]));
  std::vector<double> x(n), xnext(n), u(m), q(n);
  for (int i = 0; i < n; ++i) x[i] = xr[i];
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int it = 0; it < Nsim; ++it) {
    SolveStatus status = solver.Solve();
    const int iters = solver.GetIterations();
    ok(solver.GetInput(u.data(), 0));
    step(xnext.data(), x.data(), u.data(), h);
    double e2 = 0;
    for (int i = 0; i < n; ++i) e2 += (xnext[i] - xr[(it + 1) * n + i]) * (xnext[i] - xr[(it + 1) * n + i]);
    std::printf("step %d iters %d status %d u %.17g %.17g x %.17g %.17g %.17g %.17g err %.17g\n", it, iters, (int)status, u[0], u[1],
                xnext[0], xnext[1], xnext[2], xnext[3], std::sqrt(e2));
    for (int k = 0; k <= N; ++k) {
      const double* xk = &xr[(k + it + 1) * n];
      double dot = 0;
      for (int i = 0; i < n; ++i) { q[i] = -(Qd[i] * xk[i]); dot += q[i] * xk[i]; }
      double c = -(0.5 * dot);
      if (k < N) c += c_u;
      ok(solver.UpdateLinearCosts(q.data(), nullptr, c, k));
    }
    x = xnext;
    ok(solver.SetInitialState(x.data(), n));
    ok(solver.ShiftTrajectory());
  }
  auto t1 = std::chrono::high_resolution_clock::now();
  const double secs = std::chrono::duration<double>(t1 - t0).count();
  std::printf("Total time = %g s\n", secs);
  std::printf("Average rate = %g Hz\n", Nsim / secs);
  std::printf(bad ? "FAILED (%d API errors)\n" : "OK\n", bad);
  return bad ? 1 : 0;
}
