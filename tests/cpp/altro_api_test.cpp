// Re-authored counterparts of the reference's integration tests, against OUR altro::ALTROSolver:
//   test/altro_api.cpp:10-16                        constructor smoke test
//   test/double_integrator_test.cpp:69-168          unconstrained: Success within iterations_max = 3
//   test/double_integrator_test.cpp:170-256         terminal EQUALITY goal: dist < 1e-4, EXACTLY 3 iterations
//   test/double_integrator_test.cpp:258-375         goal + INEQUALITY control bounds: u0 = -1 (1e-4), EXACTLY 5
//   test/double_integrator_test.cpp:377-492         goal + SECOND_ORDER_CONE bound: |u0| = 1 (1e-2), EXACTLY 9
//   test/pendulum_test.cpp:45-115                   pendulum swing-up end state (1e-5), <= 10 iterations
//   test/pendulum_test.cpp:117-203                  pendulum with a goal constraint: dist < 1e-4, <= 10
// The user callbacks run on the host; every backward sweep goes through tvlqr_BackwardPass, i.e. the
// HIP kernel (needs an MI355X).  Prints "OK" and returns 0 when everything holds.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "altro/altro.hpp"

using namespace altro;

static int failures = 0;
#define EXPECT(cond)                                                                         \
  do {                                                                                       \
    if (!(cond)) { std::printf("  EXPECT FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
  } while (0)

constexpr int dim = 2;
static void di_dyn(double* xn, const double* x, const double* u, float h) {
  double b = h * h / 2;
  for (int i = 0; i < dim; ++i) {
    xn[i] = x[i] + x[i + dim] * h + u[i] * b;
    xn[i + dim] = x[i + dim] + u[i] * h;
  }
}
static void di_jac(double* J, const double* x, const double* u, float h) {
  (void)x; (void)u;
  const int n = 2 * dim;
  std::memset(J, 0, sizeof(double) * n * 3 * dim);
  double b = h * h / 2;
  for (int i = 0; i < dim; ++i) {
    J[i + i * n] = 1.0;
    J[(i + dim) + (i + dim) * n] = 1.0;
    J[i + (i + dim) * n] = h;
    J[i + (2 * dim + i) * n] = b;
    J[(i + dim) + (2 * dim + i) * n] = h;
  }
}

static double dist(const std::vector<double>& a, const std::vector<double>& b) {
  double s = 0;
  for (size_t i = 0; i < a.size(); ++i) s += (a[i] - b[i]) * (a[i] - b[i]);
  return std::sqrt(s);
}

struct DIProblem {
  int N = 10, n = 4, m = 2;
  float h = (float)(5.0f / 10.0);
  std::vector<double> Q = std::vector<double>(4, 1.0), R = std::vector<double>(2, 1e-2), x0, xf = std::vector<double>(4, 0.0),
                      uf = std::vector<double>(2, 0.0);
};

static void di_setup(ALTROSolver& s, const DIProblem& p) {
  EXPECT(s.SetDimension(p.n, p.m, 0, LastIndex) == ErrorCodes::NoError);
  EXPECT(s.SetTimeStep(p.h, 0, LastIndex) == ErrorCodes::NoError);
  EXPECT(s.SetExplicitDynamics(di_dyn, di_jac, 0, LastIndex) == ErrorCodes::NoError);
  EXPECT(s.SetLQRCost(p.n, p.m, p.Q.data(), p.R.data(), p.xf.data(), p.uf.data(), 0, LastIndex) == ErrorCodes::NoError);
  EXPECT(s.SetInitialState(p.x0.data(), p.n) == ErrorCodes::NoError);
}
static void di_goal(ALTROSolver& s, const DIProblem& p) {
  auto xf = p.xf;
  auto con = [xf](double* c, const double* x, const double* u) { (void)u; for (int i = 0; i < 4; ++i) c[i] = x[i] - xf[i]; };
  auto jac = [](double* J, const double* x, const double* u) { (void)x; (void)u; for (int i = 0; i < 4; ++i) J[i + i * 4] = 1.0; };
  EXPECT(s.SetConstraint(con, jac, p.n, ConstraintType::EQUALITY, "Goal constraint", p.N, 0, nullptr) == ErrorCodes::NoError);
}
static void di_guess(ALTROSolver& s, const DIProblem& p) {
  std::vector<double> u0(p.m, 0.0);
  EXPECT(s.SetState(p.x0.data(), p.n, 0, LastIndex) == ErrorCodes::NoError);
  EXPECT(s.SetInput(u0.data(), p.m, 0, LastIndex) == ErrorCodes::NoError);
}

static void test_constructor_and_errors() {
  std::printf("[api] constructor, index conventions, error ladder\n");
  ALTROSolver solver(10);
  EXPECT(solver.GetHorizonLength() == 10);
  EXPECT(!solver.IsInitialized());
  EXPECT(solver.Initialize() == ErrorCodes::StateDimUnknown);          // knotpoint_data_test.cpp:71-93 ladder
  EXPECT(solver.SetDimension(4, 2) == ErrorCodes::NoError);             // AllIndices default = every knot point
  EXPECT(solver.GetStateDim(10) == 4 && solver.GetInputDim(0) == 2);
  EXPECT(solver.Initialize() == ErrorCodes::TimestepNotPositive);
  EXPECT(solver.SetTimeStep(-1.0f) == ErrorCodes::TimestepNotPositive);
  EXPECT(solver.SetTimeStep(0.1f) == ErrorCodes::NoError);
  EXPECT(solver.Initialize() == ErrorCodes::DynamicsFunNotSet);
  EXPECT(solver.SetExplicitDynamics(di_dyn, di_jac) == ErrorCodes::NoError);
  EXPECT(solver.Initialize() == ErrorCodes::CostFunNotSet);
  EXPECT(solver.SetDimension(4, 2, 11) == ErrorCodes::BadIndex);
  std::vector<double> Q(4, 1.0), R(2, 0.1), z4(4, 0.0), z2(2, 0.0);
  EXPECT(solver.SetLQRCost(4, 2, Q.data(), R.data(), z4.data(), z2.data(), 0, 10) == ErrorCodes::NoError);   // [0, 10)
  EXPECT(solver.Initialize() == ErrorCodes::CostFunNotSet);             // terminal cost still missing
  EXPECT(solver.SetLQRCost(4, 2, Q.data(), R.data(), z4.data(), z2.data(), 10) == ErrorCodes::NoError);      // single index
  EXPECT(solver.SetLQRCost(3, 2, Q.data(), R.data(), z4.data(), z2.data(), 0) == ErrorCodes::DimensionMismatch);
  EXPECT(solver.SetInitialState(z4.data(), 4) == ErrorCodes::NoError);
  EXPECT(solver.Initialize() == ErrorCodes::NoError);
  EXPECT(solver.IsInitialized());
  EXPECT(solver.SetDimension(4, 2) == ErrorCodes::SolverAlreadyInitialized);
  EXPECT(std::fabs(solver.GetFinalTime() - 1.0f) < 1e-6);
}

static void test_di_unconstrained() {
  std::printf("[double integrator] unconstrained\n");
  DIProblem p; p.x0 = {1.0, 2.0, 0.0, 0.0};
  ALTROSolver s(p.N);
  di_setup(s, p);
  EXPECT(s.Initialize() == ErrorCodes::NoError);
  di_guess(s, p);
  AltroOptions o; o.iterations_max = 3;
  s.SetOptions(o);
  EXPECT(s.Solve() == SolveStatus::Success);
  std::vector<double> xN(4);
  s.GetState(xN.data(), p.N);
  EXPECT(dist(xN, p.xf) < dist(p.x0, p.xf));
  std::vector<double> K(8), d(2);
  EXPECT(s.GetFeedbackGain(K.data(), 0) == ErrorCodes::NoError && s.GetFeedforwardGain(d.data(), 0) == ErrorCodes::NoError);
  EXPECT(K[0] > 0 && std::fabs(K[1]) < 1e-12);   // decoupled axes
  std::printf("   iterations = %d, |xN - xf| = %.3e, solve time %.2f ms\n", s.GetIterations(), dist(xN, p.xf), s.GetSolveTimeMs());
}

// BASELINE.json configs[0] names the double integrator at N = 50 (the reference's own test uses N = 10,
// double_integrator_test.cpp:69-85; SURVEY.md 4.2: "support both").  Same problem, 50 steps of h = tf / 50: one LQ
// iteration, and the end state / first input the CPU oracle gives for it (tests/golden/dense_fixtures.npz, solve_di_n50).
static void test_di_unconstrained_n50() {
  std::printf("[double integrator] unconstrained, N = 50 (BASELINE.json configs[0])\n");
  DIProblem p; p.N = 50; p.h = (float)(5.0f / 50.0f); p.x0 = {1.0, 2.0, 0.0, 0.0};
  ALTROSolver s(p.N);
  di_setup(s, p);
  EXPECT(s.Initialize() == ErrorCodes::NoError);
  di_guess(s, p);
  AltroOptions o; o.iterations_max = 3;
  s.SetOptions(o);
  EXPECT(s.Solve() == SolveStatus::Success);
  EXPECT(s.GetIterations() == 1);
  std::vector<double> xN(4), u0(2);
  s.GetState(xN.data(), p.N);
  s.GetInput(u0.data(), 0);
  const double xN_ref[4] = {0.01395607778542234, 0.02791215557084468, -0.0015793690701396, -0.0031587381402792};
  const double u0_ref[2] = {-5.893250749690505, -11.78650149938101};
  double e = 0;
  for (int i = 0; i < 4; ++i) e = std::fmax(e, std::fabs(xN[i] - xN_ref[i]));
  for (int i = 0; i < 2; ++i) e = std::fmax(e, std::fabs(u0[i] - u0_ref[i]));
  std::printf("   iterations = %d, |xN - xf| = %.3e, max deviation from the oracle's fixture %.2e\n", s.GetIterations(),
              dist(xN, p.xf), e);
  EXPECT(e < 1e-10);
  EXPECT(dist(xN, p.xf) < dist(p.x0, p.xf) && dist(xN, p.xf) > 1e-3);   // double_integrator_test.cpp:165-166
}

static void test_di_goal() {
  std::printf("[double integrator] terminal goal constraint (EQUALITY)\n");
  DIProblem p; p.x0 = {1.0, 2.0, 0.0, 0.0};
  ALTROSolver s(p.N);
  di_setup(s, p); di_goal(s, p);
  EXPECT(s.Initialize() == ErrorCodes::NoError);
  di_guess(s, p);
  AltroOptions o; o.penalty_scaling = 100;
  s.SetOptions(o);
  EXPECT(s.Solve() == SolveStatus::Success);
  std::vector<double> xN(4);
  s.GetState(xN.data(), p.N);
  std::printf("   iterations = %d, dist = %.3e\n", s.GetIterations(), dist(xN, p.xf));
  EXPECT(dist(xN, p.xf) < 1e-4);
  EXPECT(s.GetIterations() == 3);
}

static void test_di_bounds() {
  std::printf("[double integrator] goal + control bounds (INEQUALITY)\n");
  DIProblem p; p.x0 = {2.0, 2.0, 0.0, 0.0};
  ALTROSolver s(p.N);
  di_setup(s, p); di_goal(s, p);
  const double ub = 1.0;
  auto con = [ub](double* c, const double* x, const double* u) { (void)x; for (int i = 0; i < 2; ++i) { c[i] = u[i] - ub; c[i + 2] = -ub - u[i]; } };
  auto jac = [](double* J, const double* x, const double* u) {
    (void)x; (void)u;
    std::memset(J, 0, sizeof(double) * 4 * 6);
    for (int i = 0; i < 2; ++i) { J[i + (4 + i) * 4] = 1.0; J[(i + 2) + (4 + i) * 4] = -1.0; }
  };
  EXPECT(s.SetConstraint(con, jac, 4, ConstraintType::INEQUALITY, "Control bounds", 0, p.N, nullptr) == ErrorCodes::NoError);
  EXPECT(s.Initialize() == ErrorCodes::NoError);
  di_guess(s, p);
  AltroOptions o; o.penalty_initial = 100; o.penalty_scaling = 100;
  s.SetOptions(o);
  EXPECT(s.Solve() == SolveStatus::Success);
  std::vector<double> xN(4), u0(2);
  s.GetState(xN.data(), p.N);
  s.GetInput(u0.data(), 0);
  std::printf("   iterations = %d, dist = %.3e, u0 = (%.6f, %.6f)\n", s.GetIterations(), dist(xN, p.xf), u0[0], u0[1]);
  EXPECT(dist(xN, p.xf) < 1e-4);
  EXPECT(std::fabs(u0[0] + ub) < 1e-4 && std::fabs(u0[1] + ub) < 1e-4);
  EXPECT(s.GetIterations() == 5);
}

static void test_di_soc() {
  std::printf("[double integrator] goal + norm bound (SECOND_ORDER_CONE)\n");
  DIProblem p; p.x0 = {2.0, 2.0, 0.0, 0.0};
  ALTROSolver s(p.N);
  di_setup(s, p); di_goal(s, p);
  const double ub = 1.0;
  auto con = [ub](double* c, const double* x, const double* u) { (void)x; c[0] = u[0]; c[1] = u[1]; c[2] = ub; };
  auto jac = [](double* J, const double* x, const double* u) {
    (void)x; (void)u;
    std::memset(J, 0, sizeof(double) * 3 * 6);
    for (int i = 0; i < 2; ++i) J[i + (4 + i) * 3] = 1.0;
  };
  EXPECT(s.SetConstraint(con, jac, 3, ConstraintType::SECOND_ORDER_CONE, "Control bounds", 0, p.N, nullptr) == ErrorCodes::NoError);
  EXPECT(s.Initialize() == ErrorCodes::NoError);
  di_guess(s, p);
  AltroOptions o; o.penalty_initial = 1.0; o.penalty_scaling = 100;
  s.SetOptions(o);
  EXPECT(s.Solve() == SolveStatus::Success);
  std::vector<double> xN(4), u0(2);
  s.GetState(xN.data(), p.N);
  s.GetInput(u0.data(), 0);
  const double un = std::sqrt(u0[0] * u0[0] + u0[1] * u0[1]);
  std::printf("   iterations = %d, dist = %.3e, |u0| = %.6f\n", s.GetIterations(), dist(xN, p.xf), un);
  EXPECT(dist(xN, p.xf) < 1e-4);
  EXPECT(std::fabs(un - ub) < 1e-2);
  EXPECT(s.GetIterations() == 9);
}

// pendulum with explicit-midpoint discretisation (test_utils.cpp:43-132 equations)
static void pend_f(double* xd, const double* x, const double* u) {
  const double l = 0.5, g = 9.81, b = 0.1, mm = 1.0 * l * l;
  xd[0] = x[1];
  xd[1] = u[0] / mm - g * std::sin(x[0]) / l - b * x[1] / mm;
}
static void pend_J(double* J, const double* x, const double* u) {
  (void)u;
  const double l = 0.5, g = 9.81, b = 0.1, mm = 1.0 * l * l;
  J[0] = 0; J[1] = -g * std::cos(x[0]) / l; J[2] = 1; J[3] = -b / mm; J[4] = 0; J[5] = 1 / mm;
}
static void pend_dyn(double* xn, const double* x, const double* u, float h) {
  double xm[2];
  pend_f(xm, x, u);
  for (int i = 0; i < 2; ++i) xm[i] = x[i] + (double)(h / 2) * xm[i];
  pend_f(xn, xm, u);
  for (int i = 0; i < 2; ++i) xn[i] = x[i] + h * xn[i];
}
static void pend_jac(double* J, const double* x, const double* u, float h) {
  double xm[2], J0[6], Jm[6];
  pend_f(xm, x, u);
  for (int i = 0; i < 2; ++i) xm[i] = x[i] + (double)(h / 2) * xm[i];
  pend_J(J0, x, u);
  pend_J(Jm, xm, u);
  double T[4];
  for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) T[i + 2 * j] = (i == j) + (double)(h / 2) * J0[i + 2 * j];
  for (int j = 0; j < 2; ++j)
    for (int i = 0; i < 2; ++i) {
      double s = 0;
      for (int k = 0; k < 2; ++k) s += (h * Jm[i + 2 * k]) * T[k + 2 * j];
      J[i + 2 * j] = (i == j) + s;
    }
  for (int i = 0; i < 2; ++i) {
    double s = 0;
    for (int k = 0; k < 2; ++k) s += (Jm[i + 2 * k] * (double)(h / 2)) * J0[4 + k];
    J[4 + i] = h * (s + Jm[4 + i]);
  }
}

static void test_pendulum(bool constrained) {
  std::printf("[pendulum] %s\n", constrained ? "goal constrained" : "unconstrained");
  const int n = 2, m = 1, N = constrained ? 20 : 50;
  const float tf = constrained ? 2.0f : 3.0f;
  const float h = (float)(tf / static_cast<double>(N));
  std::vector<double> Qd(n, 1e-2), Rd(m, 1e-3), Qdf(n, 1.0), x0(n, 0.0), xf = {M_PI, 0.0}, uf(m, 0.0);
  ALTROSolver s(N);
  EXPECT(s.SetDimension(n, m, 0, LastIndex) == ErrorCodes::NoError);
  EXPECT(s.SetTimeStep(h, 0, LastIndex) == ErrorCodes::NoError);
  EXPECT(s.SetExplicitDynamics(pend_dyn, pend_jac, 0, LastIndex) == ErrorCodes::NoError);
  EXPECT(s.SetLQRCost(n, m, Qd.data(), Rd.data(), xf.data(), uf.data(), 0, N) == ErrorCodes::NoError);
  EXPECT(s.SetLQRCost(n, m, Qdf.data(), Rd.data(), xf.data(), uf.data(), N) == ErrorCodes::NoError);
  EXPECT(s.SetInitialState(x0.data(), n) == ErrorCodes::NoError);
  if (constrained) {
    auto con = [xf](double* c, const double* x, const double* u) { (void)u; c[0] = x[0] - xf[0]; c[1] = x[1] - xf[1]; };
    auto jac = [](double* J, const double* x, const double* u) { (void)x; (void)u; std::memset(J, 0, sizeof(double) * 6); J[0] = 1; J[3] = 1; };
    EXPECT(s.SetConstraint(con, jac, n, ConstraintType::EQUALITY, "Goal constraint", N, N + 1) == ErrorCodes::NoError);
  }
  EXPECT(s.Initialize() == ErrorCodes::NoError);
  std::vector<double> u0(m, 0.1);
  s.SetInput(u0.data(), m, 0, LastIndex);
  AltroOptions o; o.iterations_max = constrained ? 100 : 20;
  s.SetOptions(o);
  SolveStatus st = s.Solve();
  std::vector<double> xN(n);
  s.GetState(xN.data(), N);
  if (!constrained) {
    const std::vector<double> xN_expected = {3.12099917161669, 0.0011966258762942175};
    std::printf("   status %d, iterations = %d, |xN - xN_expected| = %.3e\n", (int)st, s.GetIterations(), dist(xN, xN_expected));
    EXPECT(st == SolveStatus::Success);
    EXPECT(dist(xN, xN_expected) < 1e-5);
    EXPECT(s.GetIterations() <= 10);
  } else {
    std::printf("   status %d, iterations = %d, dist = %.3e\n", (int)st, s.GetIterations(), dist(xN, xf));
    EXPECT(dist(xN, xf) < 1e-4);
    EXPECT(s.GetIterations() <= 10);   // pendulum_test.cpp:201-202
  }
}

// ALTROSolver::SetDeviceModel (an extension: VERDICT r5 item 7): the pendulum problems above with the compiled-in device model in the
// place of the callback pair -- Solve() runs SolverImpl::Solve on the device (altro_hip_ilqr_solve, batch of one) -- against the
// callback solver, whose loop runs on the host with the sweeps on the device: same status, the same iteration count, the same
// trajectory (both evaluate test_utils.cpp:43-132's equations; the sums differ in order only), the goal constraint given as data
// (SetLinearConstraint) in both.  A second Solve from the solution is a warm start in both.
#include "altro_hip/altro_hip.h"
static void test_pendulum_device_model(bool constrained) {
  std::printf("[pendulum] device model, %s\n", constrained ? "goal constrained" : "unconstrained");
  const int n = 2, m = 1, N = constrained ? 20 : 50;
  const float tf = constrained ? 2.0f : 3.0f;
  const float h = (float)(tf / static_cast<double>(N));
  std::vector<double> Qd(n, 1e-2), Rd(m, 1e-3), Qdf(n, 1.0), x0(n, 0.0), xf = {M_PI, 0.0}, uf(m, 0.0);
  if (constrained) x0 = {0.3, -0.2};       // (a start away from the origin: SetInitialState must reach the device)
  std::vector<double> traj[2];
  int iters[2] = {0, 0}, iters2[2] = {0, 0};
  SolveStatus st[2];
  for (int dev = 0; dev < 2; ++dev) {
    ALTROSolver s(N);
    EXPECT(s.SetDimension(n, m, 0, LastIndex) == ErrorCodes::NoError);
    EXPECT(s.SetTimeStep(h, 0, LastIndex) == ErrorCodes::NoError);
    if (dev) EXPECT(s.SetDeviceModel(ALTRO_HIP_MODEL_PENDULUM) == ErrorCodes::NoError);
    else EXPECT(s.SetExplicitDynamics(pend_dyn, pend_jac, 0, LastIndex) == ErrorCodes::NoError);
    EXPECT(s.SetLQRCost(n, m, Qd.data(), Rd.data(), xf.data(), uf.data(), 0, N) == ErrorCodes::NoError);
    EXPECT(s.SetLQRCost(n, m, Qdf.data(), Rd.data(), xf.data(), uf.data(), N) == ErrorCodes::NoError);
    EXPECT(s.SetInitialState(x0.data(), n) == ErrorCodes::NoError);
    if (constrained) {
      const double G[4] = {1, 0, 0, 1};   // x_N - xf = 0
      EXPECT(s.SetLinearConstraint(G, xf.data(), n, ConstraintType::EQUALITY, "Goal constraint", N, N + 1) == ErrorCodes::NoError);
    }
    EXPECT(s.Initialize() == ErrorCodes::NoError);
    std::vector<double> u0(m, 0.1);
    s.SetInput(u0.data(), m, 0, LastIndex);
    AltroOptions o; o.iterations_max = constrained ? 100 : 20;
    s.SetOptions(o);
    if (dev) EXPECT(s.OpenLoopRollout() == ErrorCodes::DynamicsFunNotSet);
    st[dev] = s.Solve();
    iters[dev] = s.GetIterations();
    traj[dev].resize((size_t)(N + 1) * n + (size_t)N * m);
    for (int k = 0; k <= N; ++k) s.GetState(traj[dev].data() + (size_t)k * n, k);
    for (int k = 0; k < N; ++k) s.GetInput(traj[dev].data() + (size_t)(N + 1) * n + (size_t)k * m, k);
    std::printf("   %s: status %d, iterations %d, objective %.10g, solve %.3f ms\n", dev ? "device model  " : "host callbacks", (int)st[dev], iters[dev],
                (double)s.GetFinalObjective(), (double)s.GetSolveTimeMs());
    s.Solve();                               // from the solution: converged at once
    iters2[dev] = s.GetIterations();
    std::printf("      warm re-solve: %d iteration(s), %.3f ms\n", iters2[dev], (double)s.GetSolveTimeMs());
  }
  EXPECT(st[0] == SolveStatus::Success && st[1] == SolveStatus::Success);
  EXPECT(iters[0] == iters[1]);
  EXPECT(iters2[0] == iters2[1] && iters2[1] <= 2);
  const double dd = dist(traj[0], traj[1]);
  std::printf("   |trajectory(device model) - trajectory(host callbacks)| = %.3e, warm re-solve: %d iteration(s)\n", dd, iters2[1]);
  EXPECT(dd < 1e-7);
  if (!constrained) {
    const std::vector<double> xN_expected = {3.12099917161669, 0.0011966258762942175};   // pendulum_test.cpp:45-115
    std::vector<double> xN(traj[1].begin() + (size_t)N * n, traj[1].begin() + (size_t)(N + 1) * n);
    EXPECT(dist(xN, xN_expected) < 1e-5);
  }
}

// double_integrator_test.cpp's goal + control-bound problem (test_di_bounds above) with the device model and the constraints as data:
// the bound is registered for 0 <= k < N in one call -- ONE block definition on the device, not N -- and the solve takes the callback
// solver's five iterations to the same inputs.
static void test_di_bounds_device_model() {
  std::printf("[double integrator] device model, goal + control bounds as data (SetLinearConstraint)\n");
  DIProblem p; p.x0 = {2.0, 2.0, 0.0, 0.0};
  const double ub = 1.0;
  double Gb[4 * 6] = {0}, gb[4] = {ub, ub, ub, ub}, Gg[4 * 4] = {0};
  for (int i = 0; i < 2; ++i) { Gb[i + (4 + i) * 4] = 1.0; Gb[(i + 2) + (4 + i) * 4] = -1.0; }   // u - ub <= 0, -u - ub <= 0
  for (int i = 0; i < 4; ++i) Gg[i + i * 4] = 1.0;
  std::vector<double> u0s[2];
  int iters[2] = {0, 0};
  for (int dev = 0; dev < 2; ++dev) {
    ALTROSolver s(p.N);
    EXPECT(s.SetDimension(p.n, p.m, 0, LastIndex) == ErrorCodes::NoError);
    EXPECT(s.SetTimeStep(p.h, 0, LastIndex) == ErrorCodes::NoError);
    if (dev) EXPECT(s.SetDeviceModel(ALTRO_HIP_MODEL_DOUBLE_INTEGRATOR) == ErrorCodes::NoError);
    else EXPECT(s.SetExplicitDynamics(di_dyn, di_jac, 0, LastIndex) == ErrorCodes::NoError);
    EXPECT(s.SetLQRCost(p.n, p.m, p.Q.data(), p.R.data(), p.xf.data(), p.uf.data(), 0, LastIndex) == ErrorCodes::NoError);
    EXPECT(s.SetInitialState(p.x0.data(), p.n) == ErrorCodes::NoError);
    EXPECT(s.SetLinearConstraint(Gg, p.xf.data(), p.n, ConstraintType::EQUALITY, "Goal constraint", p.N, 0) == ErrorCodes::NoError);
    EXPECT(s.SetLinearConstraint(Gb, gb, 4, ConstraintType::INEQUALITY, "Control bounds", 0, p.N) == ErrorCodes::NoError);
    EXPECT(s.Initialize() == ErrorCodes::NoError);
    di_guess(s, p);
    AltroOptions o; o.penalty_initial = 100; o.penalty_scaling = 100;
    s.SetOptions(o);
    EXPECT(s.Solve() == SolveStatus::Success);
    iters[dev] = s.GetIterations();
    std::vector<double> xN(4);
    s.GetState(xN.data(), p.N);
    u0s[dev].resize((size_t)p.N * p.m);
    for (int k = 0; k < p.N; ++k) s.GetInput(u0s[dev].data() + (size_t)k * p.m, k);
    std::printf("   %s: iterations = %d, dist = %.3e, u0 = (%.6f, %.6f)\n", dev ? "device model  " : "host callbacks", iters[dev], dist(xN, p.xf), u0s[dev][0], u0s[dev][1]);
    EXPECT(dist(xN, p.xf) < 1e-4);
    EXPECT(std::fabs(u0s[dev][0] + ub) < 1e-4 && std::fabs(u0s[dev][1] + ub) < 1e-4);
  }
  EXPECT(iters[0] == 5 && iters[1] == 5);                 // double_integrator_test.cpp's count
  EXPECT(dist(u0s[0], u0s[1]) < 1e-9);
}

// ALTROSolver::SetQuadraticCost (altro_solver.cpp:118-136) -- a DENSE cost with a cross term,
// 1/2 x'Qx + 1/2 u'Ru + u'Hx + q'x + r'u + c, on the double integrator (with and without the goal constraint).  The blocks are
// small closed forms so that tests/test_gpu_cpp_api.py can hand the very same numbers to the oracle
// (oracle.ILQR(cost_kind = COST_QUADRATIC)) and compare the line printed here: iterations, end state, first input.
static void quad_blocks(std::vector<double>& Q, std::vector<double>& R, std::vector<double>& H, std::vector<double>& q, std::vector<double>& r) {
  const double v[4] = {1.0, -1.0, 0.5, 0.25};
  Q.assign(16, 0.0); R.assign(4, 0.0); H.assign(8, 0.0);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) Q[i + 4 * j] = (i == j ? 1.0 : 0.0) + 0.1 * v[i] * v[j];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) R[i + 2 * j] = (i == j ? 1e-2 : 0.0) + 0.005;
  const double Hr[2][4] = {{0.02, 0.0, -0.02, 0.01}, {0.0, 0.02, 0.01, -0.02}};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j) H[i + 2 * j] = Hr[i][j];
  q = {-0.2, 0.1, 0.05, -0.03};
  r = {0.01, -0.02};
}
static void test_di_quadratic_cost(bool goal) {
  std::printf("[double integrator] dense quadratic cost (SetQuadraticCost)%s\n", goal ? " + goal constraint" : "");
  DIProblem p; p.x0 = {1.0, 2.0, 0.0, 0.0};
  ALTROSolver s(p.N);
  EXPECT(s.SetDimension(p.n, p.m, 0, LastIndex) == ErrorCodes::NoError);
  EXPECT(s.SetTimeStep(p.h, 0, LastIndex) == ErrorCodes::NoError);
  EXPECT(s.SetExplicitDynamics(di_dyn, di_jac, 0, LastIndex) == ErrorCodes::NoError);
  std::vector<double> Q, R, H, q, r;
  quad_blocks(Q, R, H, q, r);
  EXPECT(s.SetQuadraticCost(3, 2, Q.data(), R.data(), H.data(), q.data(), r.data(), 0.3, 0, p.N) == ErrorCodes::DimensionMismatch);
  EXPECT(s.SetQuadraticCost(p.n, p.m, Q.data(), R.data(), H.data(), q.data(), r.data(), 0.3, 0, p.N) == ErrorCodes::NoError);   // [0, N)
  std::vector<double> QN(Q);
  for (double& e : QN) e *= 10.0;
  EXPECT(s.SetQuadraticCost(p.n, p.m, QN.data(), R.data(), H.data(), q.data(), r.data(), 0.1, p.N) == ErrorCodes::NoError);      // terminal
  EXPECT(s.SetInitialState(p.x0.data(), p.n) == ErrorCodes::NoError);
  if (goal) di_goal(s, p);
  EXPECT(s.Initialize() == ErrorCodes::NoError);
  di_guess(s, p);
  AltroOptions o; o.iterations_max = 20;
  s.SetOptions(o);
  const SolveStatus st = s.Solve();
  EXPECT(st == SolveStatus::Success);
  std::vector<double> xN(4), u0(2);
  s.GetState(xN.data(), p.N);
  s.GetInput(u0.data(), 0);
  std::printf("   quadratic%s: status %d iterations %d xN %.17g %.17g %.17g %.17g u0 %.17g %.17g objective %.17g\n", goal ? "_goal" : "", (int)st,
              s.GetIterations(), xN[0], xN[1], xN[2], xN[3], u0[0], u0[1], (double)s.GetFinalObjective());
  if (goal) EXPECT(dist(xN, p.xf) < 1e-4);
}

int main() {
  test_constructor_and_errors();
  test_di_quadratic_cost(false);
  test_di_quadratic_cost(true);
  test_di_unconstrained();
  test_di_unconstrained_n50();
  test_di_goal();
  test_di_bounds();
  test_di_soc();
  test_pendulum(false);
  test_pendulum(true);
  test_pendulum_device_model(false);
  test_pendulum_device_model(true);
  test_di_bounds_device_model();
  if (failures) { std::printf("%d EXPECTATION(S) FAILED\n", failures); return 1; }
  std::printf("OK\n");
  return 0;
}
