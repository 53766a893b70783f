// altro_solver.cpp -- host side of altro::ALTROSolver (include/altro/altro_solver.hpp).
//
// Orchestration only: knot-point storage, user callbacks, augmented-Lagrangian terms, the merit
// function, line search and the AL-iLQR iteration of the reference's SolverImpl::Solve
// (src/altro/solver/solver.cpp:414-511).  The Riccati backward sweep and the linear rollout are NOT
// computed here: they go through tvlqr_BackwardPass / tvlqr_ForwardPass (include/tvlqr/tvlqr.h),
// which this library implements on the MI355X.  No Eigen: blocks are std::vector<double>,
// column-major like the reference's (internal_types.hpp:13-14).
#define ALTRO_TU_FP_CONTRACT 1   // a host C++ unit: clang's default (on); the shared headers restore THIS mode (fp_contract.h)
#include "altro/altro_solver.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>

#include "../linesearch_sm.h"
#include "cones.hpp"
#include "altro_hip/altro_hip.h"
#include "tvlqr/tvlqr.h"

namespace altro {

using Vec = std::vector<double>;

// ---- error strings (exceptions.cpp:12-98 in the reference) ----------------------------------------
const char* ErrorCodeToString(ErrorCodes err) {
  static const char* const kMessages[] = {
#define ALTRO_ERROR_MESSAGE(code, message) message,
      ALTRO_ERROR_TABLE(ALTRO_ERROR_MESSAGE)
#undef ALTRO_ERROR_MESSAGE
  };
  const int i = static_cast<int>(err);
  return i >= 0 && i < static_cast<int>(sizeof(kMessages) / sizeof(kMessages[0])) ? kMessages[i] : "unknown error";
}
void PrintErrorCode(ErrorCodes err) {
  std::fprintf(stderr, "Got error code %d: %s\n", static_cast<int>(err), ErrorCodeToString(err));
}

// ---- per-knot-point record (role of the reference's KnotPointData, knotpoint_data.hpp:15-239) -------
enum class CostKind { Generic, Quadratic, Diagonal };
constexpr int kMaxConstraints = std::numeric_limits<int>::max();

struct Constraint {
  ConstraintFunction fun;
  ConstraintJacobian jac_fun;
  int dim = 0;
  ConstraintType type = ConstraintType::EQUALITY;
  std::string label;
  Vec val, jac, hess, v, z, z_est, z_proj, proj_jvp, proj_jac, proj_hess, jac_tmp;
  double rho = 1.0;
  bool linear = false;   // SetLinearConstraint: G (dim x (n + m), column-major) and g kept for the device loop
  Vec G, g;
};

struct Knot {
  int index = 0;
  bool terminal = false;
  int n = 0, m = 0, n2 = 0;
  float h = 0.0f;
  bool initialized = false;
  // dynamics
  bool dyn_set = false, dyn_linear = false, dyn_device = false;   // (dyn_device: SetDeviceModel -- no host callbacks)
  ExplicitDynamicsFunction dyn;
  ExplicitDynamicsJacobian dyn_jac;
  Vec jac, affine;
  // cost
  bool cost_set = false;
  CostKind cost_kind = CostKind::Generic;
  CostFunction cost_fun;
  CostGradient cost_grad;
  CostHessian cost_hess;
  Vec Q, R, H, q, r;
  double c = 0.0;
  std::vector<Constraint> cons;
  // trajectories: nominal (x,u,y) and candidate (x_,u_,y_)
  Vec x, u, y, x_, u_, y_;
  // expansions, gains, cost-to-go, scratch
  Vec A, B, f, lxx, luu, lux, lx, lu, K, d, P, p;
  Vec Qxx, Quu, Qux, Qx, Qu, Qxx_t, Quu_t, Qux_t, Qx_t, Qu_t, dx_da, du_da;

  ErrorCodes Initialize();
  void Dynamics(double* xnext);
  void DynamicsExpansion();
  void Constraints();
  void ConstraintJacobians();
  void ProjectedDuals();
  void ConicJacobians();
  void ConicHessians();
  double OriginalCost() const;
  double Cost();
  void CostGradient_();
  void CostHessian_();
  double Violations();
};

static void matvec(int rows, int cols, const double* M, const double* v, double* out, bool add) {
  for (int i = 0; i < rows; ++i) {
    double s = 0.0;
    for (int j = 0; j < cols; ++j) s += M[i + (size_t)j * rows] * v[j];
    out[i] = add ? out[i] + s : s;
  }
}
static void matTvec(int rows, int cols, const double* M, const double* v, double* out, double scale) {
  for (int j = 0; j < cols; ++j) {   // out += scale * M^T v
    double s = 0.0;
    for (int i = 0; i < rows; ++i) s += M[i + (size_t)j * rows] * v[i];
    out[j] += scale * s;
  }
}

ErrorCodes Knot::Initialize() {   // knotpoint_data.cpp:229-400: same ladder of checks
  char buf[160];
  if (n <= 0) {
    std::snprintf(buf, sizeof(buf), "Failed to Initialize knot point %d: State dimension unknown", index);
    return ALTRO_THROW(buf, ErrorCodes::StateDimUnknown);
  }
  if (!terminal) {
    if (m <= 0) {
      std::snprintf(buf, sizeof(buf), "Failed to Initialize knot point %d: Input dimension unknown", index);
      return ALTRO_THROW(buf, ErrorCodes::InputDimUnknown);
    }
    if (n2 <= 0) {
      std::snprintf(buf, sizeof(buf), "Failed to Initialize knot point %d: Next state dimension unknown", index);
      return ALTRO_THROW(buf, ErrorCodes::NextStateDimUnknown);
    }
    if (!(h > 0.0f)) {
      std::snprintf(buf, sizeof(buf), "Failed to Initialize knot point %d: Time step not set", index);
      return ALTRO_THROW(buf, ErrorCodes::TimestepNotPositive);
    }
    if (!dyn_set) {
      std::snprintf(buf, sizeof(buf), "Failed to Initialize knot point %d: Dynamics function not set", index);
      return ALTRO_THROW(buf, ErrorCodes::DynamicsFunNotSet);
    }
  }
  if (!cost_set) {
    std::snprintf(buf, sizeof(buf), "Failed to Initialize knot point %d: Cost function not set", index);
    return ALTRO_THROW(buf, ErrorCodes::CostFunNotSet);
  }
  const int mm = std::max(m, 1), nn2 = std::max(n2, 1);
  x.assign(n, 0); u.assign(mm, 0); y.assign(n, 0); x_.assign(n, 0); u_.assign(mm, 0); y_.assign(n, 0);
  jac.assign((size_t)nn2 * (n + mm), 0);
  if (!dyn_linear) { A.assign((size_t)nn2 * n, 0); B.assign((size_t)nn2 * mm, 0); }
  f.assign(nn2, 0);
  lxx.assign((size_t)n * n, 0); luu.assign((size_t)mm * mm, 0); lux.assign((size_t)mm * n, 0);
  lx.assign(n, 0); lu.assign(mm, 0);
  K.assign((size_t)mm * n, 0); d.assign(mm, 0); P.assign((size_t)n * n, 0); p.assign(n, 0);
  Qxx.assign((size_t)n * n, 0); Quu.assign((size_t)mm * mm, 0); Qux.assign((size_t)mm * n, 0); Qx.assign(n, 0); Qu.assign(mm, 0);
  Qxx_t = Qxx; Quu_t = Quu; Qux_t = Qux; Qx_t = Qx; Qu_t = Qu;
  dx_da.assign(n, 0); du_da.assign(mm, 0);
  for (auto& cn : cons) {
    const int pd = cn.dim, w = n + mm;
    cn.val.assign(pd, 0); cn.jac.assign((size_t)pd * w, 0); cn.hess.assign((size_t)w * w, 0); cn.v.assign(pd, 0);
    cn.z.assign(pd, 0); cn.z_est.assign(pd, 0); cn.z_proj.assign(pd, 0); cn.proj_jvp.assign(pd, 0);
    cn.proj_jac.assign((size_t)pd * pd, 0); cn.proj_hess.assign((size_t)pd * pd, 0); cn.jac_tmp.assign((size_t)pd * w, 0);
    cn.rho = 1.0;
  }
  if (cost_kind != CostKind::Generic) {   // constant Hessian; TVLQR-style gradient initialisation
    CostHessian_();
    if (terminal) lx = q;
    if (!terminal && dyn_linear) { lx = q; lu = r; f = affine; }
  }
  initialized = true;
  return ErrorCodes::NoError;
}

void Knot::Dynamics(double* xnext) {   // knotpoint_data.cpp:710-719, at the candidate point
  if (dyn_linear) {
    matvec(n2, n, A.data(), x_.data(), xnext, false);
    matvec(n2, m, B.data(), u_.data(), xnext, true);
    for (int i = 0; i < n2; ++i) xnext[i] += affine[i];
  } else {
    dyn(xnext, x_.data(), u_.data(), h);
  }
}
void Knot::DynamicsExpansion() {   // knotpoint_data.cpp:406-419
  if (terminal) return;
  if (!dyn_linear) {
    dyn_jac(jac.data(), x_.data(), u_.data(), h);
    std::copy(jac.begin(), jac.begin() + (size_t)n2 * n, A.begin());
    std::copy(jac.begin() + (size_t)n2 * n, jac.begin() + (size_t)n2 * (n + m), B.begin());
  } else {
    std::fill(f.begin(), f.end(), 0.0);
  }
}
void Knot::Constraints() {
  for (auto& cn : cons) cn.fun(cn.val.data(), x_.data(), u_.data());
}
void Knot::ConstraintJacobians() {
  for (auto& cn : cons) cn.jac_fun(cn.jac.data(), x_.data(), u_.data());
}
void Knot::ProjectedDuals() {   // knotpoint_data.cpp:523-535
  for (auto& cn : cons) {
    for (int i = 0; i < cn.dim; ++i) cn.z_est[i] = cn.z[i] - cn.rho * cn.val[i];
    cones::Projection(cones::DualCone(cn.type), cn.dim, cn.z_est.data(), cn.z_proj.data());
  }
}
void Knot::ConicJacobians() {   // knotpoint_data.cpp:537-547
  for (auto& cn : cons) {
    cones::ProjectionJacobian(cones::DualCone(cn.type), cn.dim, cn.z_est.data(), cn.proj_jac.data());
    std::fill(cn.proj_jvp.begin(), cn.proj_jvp.end(), 0.0);
    matTvec(cn.dim, cn.dim, cn.proj_jac.data(), cn.z_proj.data(), cn.proj_jvp.data(), 1.0);
  }
}
void Knot::ConicHessians() {   // knotpoint_data.cpp:549-570: Gauss-Newton + (SOC) projection curvature
  const int w = n + std::max(m, 1);
  for (auto& cn : cons) {
    const int pd = cn.dim;
    auto mul = [&](const Vec& Mat) {   // jac_tmp = Mat (pd x pd) * J (pd x w)
      for (int j = 0; j < w; ++j)
        for (int i = 0; i < pd; ++i) {
          double s = 0.0;
          for (int k = 0; k < pd; ++k) s += Mat[i + (size_t)k * pd] * cn.jac[k + (size_t)j * pd];
          cn.jac_tmp[i + (size_t)j * pd] = s;
        }
    };
    mul(cn.proj_jac);
    for (int j = 0; j < w; ++j)
      for (int i = 0; i < w; ++i) {
        double s = 0.0;
        for (int k = 0; k < pd; ++k) s += cn.jac_tmp[k + (size_t)i * pd] * cn.jac_tmp[k + (size_t)j * pd];
        cn.hess[i + (size_t)j * w] = cn.rho * s;
      }
    if (!cones::ProjectionIsLinear(cones::DualCone(cn.type))) {
      cones::ProjectionHessian(cones::DualCone(cn.type), pd, cn.z_est.data(), cn.z_proj.data(), cn.proj_hess.data());
      mul(cn.proj_hess);
      for (int j = 0; j < w; ++j)
        for (int i = 0; i < w; ++i) {
          double s = 0.0;
          for (int k = 0; k < pd; ++k) s += cn.jac[k + (size_t)i * pd] * cn.jac_tmp[k + (size_t)j * pd];
          cn.hess[i + (size_t)j * w] += cn.rho * s;
        }
    }
  }
}
double Knot::OriginalCost() const {   // knotpoint_data.cpp:616-648
  double J = 0.0;
  Vec tmp(std::max(n, m) + 1);
  switch (cost_kind) {
    case CostKind::Generic: J = cost_fun(x_.data(), u_.data()); break;
    case CostKind::Quadratic: {
      matvec(n, n, Q.data(), x_.data(), tmp.data(), false);
      double a = 0; for (int i = 0; i < n; ++i) a += x_[i] * tmp[i];
      J = 0.5 * a;
      for (int i = 0; i < n; ++i) J += q[i] * x_[i];
      if (!terminal) {
        matvec(m, m, R.data(), u_.data(), tmp.data(), false);
        a = 0; for (int i = 0; i < m; ++i) a += u_[i] * tmp[i];
        J += 0.5 * a;
        for (int i = 0; i < m; ++i) J += r[i] * u_[i];
        matvec(m, n, H.data(), x_.data(), tmp.data(), false);
        for (int i = 0; i < m; ++i) J += u_[i] * tmp[i];
      }
      J += c;
    } break;
    case CostKind::Diagonal: {
      double a = 0; for (int i = 0; i < n; ++i) a += x_[i] * Q[i] * x_[i];
      J = 0.5 * a;
      for (int i = 0; i < n; ++i) J += q[i] * x_[i];
      if (!terminal) {
        a = 0; for (int i = 0; i < m; ++i) a += u_[i] * R[i] * u_[i];
        J += 0.5 * a;
        for (int i = 0; i < m; ++i) J += r[i] * u_[i];
      }
      J += c;
    } break;
  }
  return J;
}
double Knot::Cost() {   // CalcCost (:421-428): original + ||Pi(z - rho c)||^2 / (2 rho)
  double J = OriginalCost();
  ProjectedDuals();
  for (auto& cn : cons) {
    double s = 0; for (double zp : cn.z_proj) s += zp * zp;
    J += s / (2 * cn.rho);
  }
  return J;
}
void Knot::CostGradient_() {   // CalcCostGradient (:430-437, :583-595, :650-681)
  switch (cost_kind) {
    case CostKind::Generic: cost_grad(lx.data(), lu.data(), x_.data(), u_.data()); break;
    case CostKind::Quadratic:
      matvec(n, n, Q.data(), x_.data(), lx.data(), false);
      for (int i = 0; i < n; ++i) lx[i] += q[i];
      if (!terminal) {
        matvec(m, m, R.data(), u_.data(), lu.data(), false);
        for (int i = 0; i < m; ++i) lu[i] += r[i];
        matvec(m, n, H.data(), x_.data(), lu.data(), true);
        matTvec(m, n, H.data(), u_.data(), lx.data(), 1.0);
      }
      break;
    case CostKind::Diagonal:
      for (int i = 0; i < n; ++i) lx[i] = Q[i] * x_[i] + q[i];
      if (!terminal) for (int i = 0; i < m; ++i) lu[i] = R[i] * u_[i] + r[i];
      break;
  }
  ConicJacobians();
  for (auto& cn : cons) {
    matTvec(cn.dim, n, cn.jac.data(), cn.proj_jvp.data(), lx.data(), -1.0);
    if (!terminal) matTvec(cn.dim, m, cn.jac.data() + (size_t)cn.dim * n, cn.proj_jvp.data(), lu.data(), -1.0);
  }
}
void Knot::CostHessian_() {   // CalcCostHessian (:439-448, :597-613, :683-708)
  switch (cost_kind) {
    case CostKind::Generic: cost_hess(lxx.data(), luu.data(), lux.data(), x_.data(), u_.data()); break;
    case CostKind::Quadratic:
      lxx = Q;
      if (!terminal) { luu = R; lux = H; }
      break;
    case CostKind::Diagonal:
      std::fill(lxx.begin(), lxx.end(), 0.0);
      for (int i = 0; i < n; ++i) lxx[i + (size_t)i * n] = Q[i];
      if (!terminal) {
        std::fill(luu.begin(), luu.end(), 0.0);
        for (int i = 0; i < m; ++i) luu[i + (size_t)i * m] = R[i];
        std::fill(lux.begin(), lux.end(), 0.0);
      }
      break;
  }
  if (cons.empty() || !initialized) return;
  ConicHessians();
  const int w = n + std::max(m, 1);
  for (auto& cn : cons) {
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) lxx[i + (size_t)j * n] += cn.hess[i + (size_t)j * w];
    if (!terminal) {
      for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) luu[i + (size_t)j * m] += cn.hess[(n + i) + (size_t)(n + j) * w];
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) lux[i + (size_t)j * m] += cn.hess[(n + i) + (size_t)j * w];
    }
  }
}
double Knot::Violations() {   // CalcViolations (:489-501)
  double viol = 0.0;
  for (auto& cn : cons) {
    cones::Projection(cn.type, cn.dim, cn.val.data(), cn.v.data());
    for (int i = 0; i < cn.dim; ++i) {
      cn.v[i] -= cn.val[i];
      viol = std::max(viol, std::fabs(cn.v[i]));
    }
  }
  return viol;
}

// ---- the solver (role of the reference's SolverImpl, solver.hpp:21-125) ----------------------------
class SolverImpl {
 public:
  explicit SolverImpl(int N) : N(N), data(N + 1), nx(N + 1, 0), nu(N + 1, 0), hs(N, 0.0f) {
    for (int k = 0; k <= N; ++k) { data[k].index = k; data[k].terminal = (k == N); }
  }
  int N;
  std::vector<Knot> data;
  std::vector<int> nx, nu;
  std::vector<float> hs;
  Vec initial_state;
  AltroOptions opts;
  AltroStats stats;
  bool initialized = false;
  double phi0 = 0, dphi0 = 0, phi = 0, dphi = 0, rho = 1.0, delta_V[2] = {0, 0};
  int ls_iters = 0;
  int backward_status = TVLQR_SUCCESS;
  // SetDeviceModel: the compiled-in model and the resident batch of one that Solve() runs on
  int device_model = -1, device_frame = 0;
  double device_length = 2.7, device_lr = 1.5;
  altro_hip_batch* dev = nullptr;
  ~SolverImpl() { if (dev) altro_hip_batch_destroy(dev); }
  ErrorCodes DeviceSolve();
  // pointer tables handed to the tvlqr_* kernel boundary (solver.cpp:63-106)
  std::vector<double*> pA, pB, pf, plxx, pluu, plux, plx, plu, pK, pd, pP, pp, pQxx, pQuu, pQux, pQx, pQu, pQxxt,
      pQuut, pQuxt, pQxt, pQut, px, pu, py;

  ErrorCodes Initialize();
  void OpenLoopRollout();
  ErrorCodes LinearRollout();
  void CopyTrajectory();
  double CalcCost();
  double Stationarity();
  double Feasibility();
  ErrorCodes BackwardPass();
  void MeritFunction(double alpha, double* phi_out, double* dphi_out);
  ErrorCodes ForwardPass(double* alpha);
  ErrorCodes Solve();
};

ErrorCodes SolverImpl::Initialize() {
  for (auto& kp : data) {
    ErrorCodes err = kp.Initialize();
    if (err != ErrorCodes::NoError) return ALTRO_THROW("Failed to initialize the solver", err);
  }
  auto tab = [&](std::vector<double*>& t, Vec Knot::*member) {
    t.assign(N + 1, nullptr);
    for (int k = 0; k <= N; ++k) t[k] = (data[k].*member).data();
  };
  tab(pA, &Knot::A); tab(pB, &Knot::B); tab(pf, &Knot::f); tab(plxx, &Knot::lxx); tab(pluu, &Knot::luu);
  tab(plux, &Knot::lux); tab(plx, &Knot::lx); tab(plu, &Knot::lu); tab(pK, &Knot::K); tab(pd, &Knot::d);
  tab(pP, &Knot::P); tab(pp, &Knot::p); tab(pQxx, &Knot::Qxx); tab(pQuu, &Knot::Quu); tab(pQux, &Knot::Qux);
  tab(pQx, &Knot::Qx); tab(pQu, &Knot::Qu); tab(pQxxt, &Knot::Qxx_t); tab(pQuut, &Knot::Quu_t);
  tab(pQuxt, &Knot::Qux_t); tab(pQxt, &Knot::Qx_t); tab(pQut, &Knot::Qu_t);
  tab(px, &Knot::x_); tab(pu, &Knot::u_); tab(py, &Knot::y_);
  for (int k = 0; k <= N; ++k) { nx[k] = data[k].n; nu[k] = data[k].m; }
  (void)tvlqr_hip_warmup(nx.data(), nu.data(), N);   // device, kernels and workspace now, not in the first Solve (a missing device shows in Solve)
  initialized = true;
  return ErrorCodes::NoError;
}

void SolverImpl::OpenLoopRollout() {   // solver.cpp:116-131
  data[0].x_ = initial_state;
  for (int k = 0; k < N; ++k) data[k].Dynamics(data[k + 1].x_.data());
}
ErrorCodes SolverImpl::LinearRollout() {   // solver.cpp:133-146 -> tvlqr_ForwardPass (on the device)
  int res = tvlqr_ForwardPass(nx.data(), nu.data(), N, pA.data(), pB.data(), pf.data(), pK.data(), pd.data(),
                              pP.data(), pp.data(), initial_state.data(), px.data(), pu.data(), py.data());
  return res == TVLQR_SUCCESS ? ErrorCodes::NoError : ErrorCodes::BackwardPassFailed;
}
void SolverImpl::CopyTrajectory() {
  for (int k = 0; k <= N; ++k) {
    data[k].x = data[k].x_;
    data[k].y = data[k].y_;
    if (k < N) data[k].u = data[k].u_;
  }
}
double SolverImpl::CalcCost() {   // solver.cpp:163-174
  double cost = 0.0;
  for (int k = 0; k <= N; ++k) { data[k].Constraints(); cost += data[k].Cost(); }
  return cost;
}
double SolverImpl::Stationarity() {   // solver.cpp:207-222
  double res_x = 0, res_u = 0;
  for (int k = 0; k < N; ++k) {
    Knot& z = data[k];
    Knot& zn = data[k + 1];
    Vec tx(z.lx), tu(z.lu);
    matTvec(z.n2, z.n, z.A.data(), zn.y_.data(), tx.data(), 1.0);
    matTvec(z.n2, z.m, z.B.data(), zn.y_.data(), tu.data(), 1.0);
    for (int i = 0; i < z.n; ++i) res_x = std::max(res_x, std::fabs(tx[i] - z.y_[i]));
    for (int i = 0; i < z.m; ++i) res_u = std::max(res_u, std::fabs(tu[i]));
  }
  Knot& z = data[N];
  for (int i = 0; i < z.n; ++i) res_x = std::max(res_x, std::fabs(z.lx[i] - z.y_[i]));
  return std::max(res_x, res_u);
}
double SolverImpl::Feasibility() {
  double viol = 0;
  for (auto& kp : data) viol = std::max(viol, kp.Violations());
  return viol;
}
ErrorCodes SolverImpl::BackwardPass() {   // solver.cpp:360-378 -> tvlqr_BackwardPass (on the device)
  backward_status = tvlqr_BackwardPass(nx.data(), nu.data(), N, pA.data(), pB.data(), pf.data(), plxx.data(),
                                       pluu.data(), plux.data(), plx.data(), plu.data(), /*reg=*/0.0, pK.data(),
                                       pd.data(), pP.data(), pp.data(), delta_V, pQxx.data(), pQuu.data(),
                                       pQux.data(), pQx.data(), pQu.data(), pQxxt.data(), pQuut.data(),
                                       pQuxt.data(), pQxt.data(), pQut.data(), false, false);
  return backward_status == TVLQR_SUCCESS ? ErrorCodes::NoError : ErrorCodes::BackwardPassFailed;
}

void SolverImpl::MeritFunction(double alpha, double* phi_out, double* dphi_out) {   // solver.cpp:273-355
  const bool deriv = dphi_out != nullptr;
  double ph = 0, dph = 0;
  data[0].x_ = initial_state;
  std::fill(data[0].dx_da.begin(), data[0].dx_da.end(), 0.0);
  Vec dx, tmp;
  for (int k = 0; k < N; ++k) {
    Knot& kp = data[k];
    Knot& nk = data[k + 1];
    const int n = kp.n, m = kp.m;
    dx.assign(n, 0); tmp.assign(std::max(n, m), 0);
    for (int i = 0; i < n; ++i) dx[i] = kp.x_[i] - kp.x[i];
    matvec(m, n, kp.K.data(), dx.data(), tmp.data(), false);
    for (int i = 0; i < m; ++i) kp.u_[i] = kp.u[i] + (-tmp[i] + alpha * kp.d[i]);
    matvec(n, n, kp.P.data(), dx.data(), kp.y_.data(), false);
    for (int i = 0; i < n; ++i) kp.y_[i] += kp.p[i];
    kp.Dynamics(nk.x_.data());
    kp.Constraints();
    ph += kp.Cost();
    if (deriv) {
      kp.DynamicsExpansion();
      matvec(m, n, kp.K.data(), kp.dx_da.data(), tmp.data(), false);
      for (int i = 0; i < m; ++i) kp.du_da[i] = -tmp[i] + kp.d[i];
      matvec(kp.n2, n, kp.A.data(), kp.dx_da.data(), nk.dx_da.data(), false);
      matvec(kp.n2, m, kp.B.data(), kp.du_da.data(), nk.dx_da.data(), true);
      kp.ConstraintJacobians();
      kp.CostGradient_();
      double s = 0; for (int i = 0; i < n; ++i) s += kp.lx[i] * kp.dx_da[i];
      dph += s;
      s = 0; for (int i = 0; i < m; ++i) s += kp.lu[i] * kp.du_da[i];
      dph += s;
    }
  }
  Knot& kp = data[N];
  kp.Constraints();
  ph += kp.Cost();
  dx.assign(kp.n, 0);
  for (int i = 0; i < kp.n; ++i) dx[i] = kp.x_[i] - kp.x[i];
  matvec(kp.n, kp.n, kp.P.data(), dx.data(), kp.y_.data(), false);
  for (int i = 0; i < kp.n; ++i) kp.y_[i] += kp.p[i];
  *phi_out = ph;
  if (deriv) {
    kp.ConstraintJacobians();
    kp.CostGradient_();
    double s = 0; for (int i = 0; i < kp.n; ++i) s += kp.lx[i] * kp.dx_da[i];
    dph += s;
    *dphi_out = dph;
  }
  phi = ph;
  dphi = dph;
}

ErrorCodes SolverImpl::ForwardPass(double* alpha) {   // solver.cpp:237-271
  MeritFunction(0.0, &phi0, &dphi0);
  if (std::fabs(dphi0) < opts.tol_meritfun_gradient) {
    *alpha = 0.0;
    return ErrorCodes::MeritFunctionGradientTooSmall;
  }
  // the same resumable state machine the batched device solver runs per problem
  altro_hip::LsOptions lo = altro_hip::ls_default_options();
  lo.try_cubic_first = 1;
  lo.use_backtracking = opts.use_backtracking_linesearch != 0.0;
  altro_hip::LsState st;
  bool need = altro_hip::ls_begin(st, lo, 1.0, phi0, dphi0);
  while (need) {
    double ph = 0, dph = 0;
    MeritFunction(st.alpha, &ph, st.want_derivative ? &dph : nullptr);
    if (opts.verbose == Verbosity::LineSearch)
      std::printf("    ls: alpha = %.6g  phi = %.10g  dphi = %.6g\n", st.alpha, ph, dph);
    need = altro_hip::ls_feed(st, lo, ph, dph);
  }
  *alpha = st.alpha;
  phi = st.phi;
  dphi = st.dphi;
  ls_iters = st.n_iters;
  if (lo.use_backtracking && std::fabs(*alpha - 1.0) > 0) {   // solver.cpp:256-262
    for (auto& kp : data) { kp.DynamicsExpansion(); kp.ConstraintJacobians(); kp.CostGradient_(); }
  }
  if (std::isnan(*alpha) || !(st.status == altro_hip::LS_MINIMUM_FOUND || st.status == altro_hip::LS_HIT_MAX_STEPSIZE)) {
    char buf[64];
    std::snprintf(buf, sizeof(buf), "Line search failed with code %d", st.status);
    return ALTRO_THROW(buf, ErrorCodes::LineSearchFailed);
  }
  return ErrorCodes::NoError;
}

ErrorCodes SolverImpl::Solve() {   // solver.cpp:414-511
  const auto t_start = std::chrono::steady_clock::now();
  rho = opts.penalty_initial;
  OpenLoopRollout();
  CopyTrajectory();
  const double cost_initial = CalcCost();
  for (auto& kp : data) {
    kp.DynamicsExpansion();
    kp.ConstraintJacobians();
    kp.CostGradient_();
    for (auto& cn : kp.cons) cn.rho = opts.penalty_initial;
  }
  if (opts.verbose > Verbosity::Silent) std::printf("STARTING ALTRO iLQR SOLVE....\n  Initial Cost: %g\n", cost_initial);
  bool is_converged = false, stop_iterating = false;
  stats.status = SolveStatus::Unsolved;
  double alpha = 0.0, stationarity = 0.0, feasibility = 0.0;
  int iter;
  for (iter = 0; iter < opts.iterations_max; ++iter) {
    for (auto& kp : data) kp.CostHessian_();   // CalcExpansions
    BackwardPass();                            // a Cholesky failure (k >= 0) is ignored, as solver.cpp:449 does
    if (backward_status < TVLQR_SUCCESS) {     // ... but "the device pass did not run" is not a reference case: K, d, P, p
      // were never written, and iterating on them would spin to iterations_max with alpha = 0 and no diagnostic
      std::fprintf(stderr, "altro: tvlqr_BackwardPass did not run (%s); the HIP path has no CPU fallback\n",
                   backward_status == TVLQR_NO_DEVICE ? "no usable HIP device or a device allocation / copy failed"
                                                      : "a state or input dimension exceeds 32");
      stats.status = SolveStatus::Unsolved;
      stats.iterations = iter;
      stats.solve_time = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start);
      return ErrorCodes::BackwardPassFailed;
    }
    ErrorCodes err = ForwardPass(&alpha);
    if (!(err == ErrorCodes::NoError || err == ErrorCodes::MeritFunctionGradientTooSmall)) {
      PrintErrorCode(err);
      stop_iterating = true;
    }
    stationarity = Stationarity();
    feasibility = Feasibility();
    CopyTrajectory();
    if (std::fabs(stationarity) < opts.tol_stationarity && feasibility < opts.tol_primal_feasibility) {
      is_converged = true;
      stop_iterating = true;
      stats.status = SolveStatus::Success;
    }
    bool dual_update = false;
    const double penalty = rho;
    if (stationarity < std::sqrt(opts.tol_stationarity)) {
      for (auto& kp : data) for (auto& cn : kp.cons) cn.z = cn.z_proj;   // DualUpdate
      if (feasibility > opts.tol_primal_feasibility) {                   // PenaltyUpdate
        for (auto& kp : data) for (auto& cn : kp.cons) cn.rho = std::min(cn.rho * opts.penalty_scaling, opts.penalty_max);
        rho = std::min(rho * opts.penalty_scaling, opts.penalty_max);
      }
      for (auto& kp : data) { kp.ProjectedDuals(); kp.CostGradient_(); }
      dual_update = true;
    }
    if (opts.verbose > Verbosity::Silent)
      std::printf("  iter = %3d, phi = %8.4g -> %8.4g (%10.3g), dphi = %10.3g -> %10.3g, alpha = %8.3g, ls_iter = %2d, "
                  "stat = %8.3e, feas = %8.3e, rho = %7.2g, dual update? %d\n",
                  iter, phi0, phi, phi0 - phi, dphi0, dphi, alpha, ls_iters, stationarity, feasibility, penalty,
                  (int)dual_update);
    if (stop_iterating) break;
  }
  if (!is_converged && iter == opts.iterations_max) stats.status = SolveStatus::MaxIterations;
  stats.iterations = iter + 1;
  stats.stationarity = stationarity;
  stats.primal_feasibility = feasibility;
  stats.objective_value = phi;
  stats.solve_time = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start);
  if (opts.verbose > Verbosity::Silent) std::printf("ALTRO SOLVE FINISHED!\n");
  return ErrorCodes::NoError;
}

// SolverImpl::Solve (solver.cpp:414-511) on the device for a solver with SetDeviceModel: the problem as data into a resident
// altro_hip batch of one (created at the first Solve, kept), altro_hip_ilqr_solve, the iterate back.  Everything the device loop
// computes is the batched path's (tests/test_gpu_ilqr*.py hold it to the oracle); this function only moves the problem across.
ErrorCodes SolverImpl::DeviceSolve() {
  const auto t_start = std::chrono::steady_clock::now();
  stats.status = SolveStatus::Unsolved;
  const int n = data[0].n, m = data[0].m;
  for (int k = 0; k <= N; ++k) {
    const Knot& kp = data[k];
    if (kp.n != n || (k < N && (kp.m != m || kp.h != data[0].h)))
      return ALTRO_THROW("SetDeviceModel: the device loop takes one state / input dimension and one time step", ErrorCodes::DimensionMismatch);
    if (kp.cost_kind == CostKind::Generic)
      return ALTRO_THROW("SetDeviceModel: costs must be set as data (SetDiagonalCost / SetQuadraticCost / SetLQRCost)", ErrorCodes::CostNotQuadratic);
    for (const auto& cn : kp.cons)
      if (!cn.linear) return ALTRO_THROW("SetDeviceModel: constraints must be given as data (SetLinearConstraint)", ErrorCodes::InvalidPointer);
  }
  auto hip_fail = [&](const char* what) {
    std::fprintf(stderr, "altro: %s failed on the device: %s (the device path has no CPU fallback)\n", what, altro_hip_last_error());
    stats.solve_time = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start);
    return ErrorCodes::BackwardPassFailed;
  };
  const bool fresh = dev == nullptr;
  if (fresh) {
    if (altro_hip_batch_create(&dev, N, n, m, 1, ALTRO_HIP_F64, ALTRO_HIP_PLAN_AUTO, 0, 0, nullptr)) { dev = nullptr; return hip_fail("altro_hip_batch_create"); }
    if (altro_hip_set_model(dev, device_model, data[0].h, device_frame, device_length, device_lr)) return hip_fail("altro_hip_set_model");
  }
  // the cost, dense (a diagonal cost is its diagonal): [N+1][n*n], [N][m*m], [N][m*n], [N+1][n], [N][m], [N+1]
  std::vector<double> Q((size_t)(N + 1) * n * n, 0.0), R((size_t)N * m * m, 0.0), H((size_t)N * m * n, 0.0), q((size_t)(N + 1) * n, 0.0),
      r((size_t)N * m, 0.0), c(N + 1, 0.0);
  for (int k = 0; k <= N; ++k) {
    const Knot& kp = data[k];
    const bool diag = kp.cost_kind == CostKind::Diagonal;
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) Q[(size_t)k * n * n + i + (size_t)j * n] = diag ? (i == j ? kp.Q[i] : 0.0) : kp.Q[i + (size_t)j * n];
    std::copy(kp.q.begin(), kp.q.begin() + n, q.begin() + (size_t)k * n);
    c[k] = kp.c;
    if (k == N) break;
    for (int j = 0; j < m; ++j)
      for (int i = 0; i < m; ++i) R[(size_t)k * m * m + i + (size_t)j * m] = diag ? (i == j ? kp.R[i] : 0.0) : kp.R[i + (size_t)j * m];
    if (!diag) std::copy(kp.H.begin(), kp.H.begin() + (size_t)m * n, H.begin() + (size_t)k * m * n);
    std::copy(kp.r.begin(), kp.r.begin() + m, r.begin() + (size_t)k * m);
  }
  // A diagonal cost with no zero weight is SetLQRCost's tracking form 1/2 (x - xref)' Q (x - xref) + ... (altro_solver.cpp:138-172) up to a
  // constant: handed over as such, the small plans run their whole solve in ONE launch (plan LANE's fused kernel) instead of the
  // launch sequence the dense cost takes.  The constant (c_k - 1/2 xref' Q xref - 1/2 uref' R uref, summed) goes back into the objective.
  bool tracking = true;
  for (int k = 0; k <= N && tracking; ++k) {
    const Knot& kp = data[k];
    if (kp.cost_kind != CostKind::Diagonal) tracking = false;
    for (int i = 0; i < n && tracking; ++i) if (kp.Q[i] == 0.0) tracking = false;
    for (int i = 0; i < m && tracking && k < N; ++i) if (kp.R[i] == 0.0) tracking = false;
  }
  double objective_offset = 0.0;
  if (tracking) {
    std::vector<double> Qd((size_t)(N + 1) * n), Rd((size_t)N * m), xr((size_t)(N + 1) * n), ur((size_t)N * m);
    for (int k = 0; k <= N; ++k) {
      const Knot& kp = data[k];
      double ct = 0.0;
      for (int i = 0; i < n; ++i) { Qd[(size_t)k * n + i] = kp.Q[i]; xr[(size_t)k * n + i] = -kp.q[i] / kp.Q[i]; ct += 0.5 * xr[(size_t)k * n + i] * kp.Q[i] * xr[(size_t)k * n + i]; }
      if (k < N)
        for (int i = 0; i < m; ++i) { Rd[(size_t)k * m + i] = kp.R[i]; ur[(size_t)k * m + i] = -kp.r[i] / kp.R[i]; ct += 0.5 * ur[(size_t)k * m + i] * kp.R[i] * ur[(size_t)k * m + i]; }
      objective_offset += kp.c - ct;
    }
    if (altro_hip_set_tracking_cost(dev, Qd.data(), Rd.data(), xr.data(), ur.data(), 0, 0)) return hip_fail("altro_hip_set_tracking_cost");
  } else if (altro_hip_set_quadratic_cost(dev, Q.data(), R.data(), H.data(), q.data(), r.data(), c.data(), 0, 0)) {
    return hip_fail("altro_hip_set_quadratic_cost");
  }
  if (fresh) {   // constraint blocks are fixed at Initialize(): once (their duals then persist from Solve to Solve, like the host loop's)
    // runs of knot points with the same block become ONE block over the range (the device holds a handful of block definitions per
    // handle, include/altro_hip/altro_hip.h: an input bound set for 0 <= k < N is one definition, not N)
    struct Run { int k0, k1; const Constraint* cn; };
    std::vector<Run> runs;
    for (int k = 0; k <= N; ++k)
      for (const auto& cn : data[k].cons) {
        bool joined = false;
        for (auto& r : runs)
          if (r.k1 == k - 1 && r.cn->type == cn.type && r.cn->dim == cn.dim && r.cn->G == cn.G && r.cn->g == cn.g) { r.k1 = k; joined = true; break; }
        if (!joined) runs.push_back(Run{k, k, &cn});
      }
    for (const auto& r : runs)
      if (altro_hip_add_linear_constraint(dev, r.k0, r.k1, (int)r.cn->type, r.cn->dim, r.cn->G.data(), r.cn->g.data(), 0) < 0)
        return hip_fail("altro_hip_add_linear_constraint");
  }   // (a later Solve keeps the duals and restarts the penalty, as the host loop does: solver.cpp:429)
  std::vector<double> u0((size_t)N * m);
  for (int k = 0; k < N; ++k) std::copy(data[k].u_.begin(), data[k].u_.begin() + m, u0.begin() + (size_t)k * m);
  if ((int)initial_state.size() != n) return ALTRO_THROW("SetDeviceModel: the initial state is not set", ErrorCodes::DimensionMismatch);
  if (altro_hip_set_initial_state(dev, initial_state.data(), 0)) return hip_fail("altro_hip_set_initial_state");   // (OpenLoopRollout's x_0: solver.cpp:119)
  if (altro_hip_set_input_guess(dev, u0.data(), 0, 0)) return hip_fail("altro_hip_set_input_guess");
  altro_hip_solve_options o;
  altro_hip_default_solve_options(&o);
  o.iterations_max = opts.iterations_max;
  o.tol_stationarity = opts.tol_stationarity;
  o.tol_primal_feasibility = opts.tol_primal_feasibility;
  o.tol_meritfun_gradient = opts.tol_meritfun_gradient;
  o.use_backtracking_linesearch = opts.use_backtracking_linesearch != 0.0 ? 1 : 0;
  o.penalty_initial = opts.penalty_initial; o.penalty_scaling = opts.penalty_scaling; o.penalty_max = opts.penalty_max;
  altro_hip_solve_result res;
  if (altro_hip_ilqr_solve(dev, &o, &res)) return hip_fail("altro_hip_ilqr_solve");
  std::vector<double> xs((size_t)(N + 1) * n), us((size_t)N * m);
  if (altro_hip_get_nominal(dev, xs.data(), us.data())) return hip_fail("altro_hip_get_nominal");
  for (int k = 0; k <= N; ++k) {
    Knot& kp = data[k];
    std::copy(xs.begin() + (size_t)k * n, xs.begin() + (size_t)(k + 1) * n, kp.x.begin());
    kp.x_ = kp.x;
    if (k == N) break;
    std::copy(us.begin() + (size_t)k * m, us.begin() + (size_t)(k + 1) * m, kp.u.begin());
    std::copy(kp.u.begin(), kp.u.begin() + m, kp.u_.begin());
  }
  stats.status = res.status == 0 ? SolveStatus::Success : (res.status == 2 ? SolveStatus::MaxIterations : SolveStatus::Unsolved);
  stats.iterations = res.iterations;
  stats.stationarity = res.stationarity;
  stats.primal_feasibility = res.primal_feasibility;
  stats.objective_value = res.final_phi + objective_offset;
  phi = stats.objective_value;
  rho = res.penalty;
  stats.solve_time = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start);
  return ErrorCodes::NoError;
}

// ---- ALTROSolver: validation + forwarding (altro_solver.cpp) ------------------------------------------
ALTROSolver::ALTROSolver(int horizon_length) : solver_(std::make_unique<SolverImpl>(horizon_length)) {}
ALTROSolver::ALTROSolver(ALTROSolver&& other) = default;
ALTROSolver& ALTROSolver::operator=(ALTROSolver&& other) = default;
ALTROSolver::~ALTROSolver() = default;

ErrorCodes ALTROSolver::SetDimension(int num_states, int num_inputs, int k_start, int k_stop) {
  if (IsInitialized())
    return ALTRO_THROW("Cannot change the dimension once the solver has been initialized.", ErrorCodes::SolverAlreadyInitialized);
  ErrorCodes err = CheckKnotPointIndices(k_start, k_stop, LastIndexMode::Inclusive);
  if (err != ErrorCodes::NoError) return err;
  if (num_states <= 0) return ErrorCodes::StateDimUnknown;
  const int N = GetHorizonLength();
  for (int k = k_start; k < k_stop; ++k) {
    Knot& kp = solver_->data[k];
    if (num_inputs <= 0)
      return ALTRO_THROW(kp.terminal ? "Input dimension must also be specified at the terminal knot point."
                                     : "Input dimension must be specified", ErrorCodes::InputDimUnknown);
    kp.n = num_states;
    kp.m = num_inputs;
    solver_->nx[k] = num_states;
    solver_->nu[k] = num_inputs;
    if (k > 0) solver_->data[k - 1].n2 = num_states;
    (void)N;
  }
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::SetTimeStep(float h, int k_start, int k_stop) {
  ErrorCodes err = CheckKnotPointIndices(k_start, k_stop, LastIndexMode::Exclusive);
  if (err != ErrorCodes::NoError) return err;
  if (h <= 0.0f) return ErrorCodes::TimestepNotPositive;
  for (int k = k_start; k < k_stop; ++k) { solver_->data[k].h = h; solver_->hs[k] = h; }
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::SetExplicitDynamics(ExplicitDynamicsFunction f, ExplicitDynamicsJacobian df, int k_start, int k_stop) {
  ErrorCodes err = CheckKnotPointIndices(k_start, k_stop, LastIndexMode::Exclusive);
  err = AssertDimensionsAreSet(k_start, k_stop, "Cannot set the dynamics");
  if (err != ErrorCodes::NoError) return err;
  for (int k = k_start; k < k_stop; ++k) {
    Knot& kp = solver_->data[k];
    kp.dyn = f; kp.dyn_jac = df; kp.dyn_set = true; kp.dyn_linear = false;
  }
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::SetCostFunction(CostFunction cf, CostGradient cg, CostHessian ch, int k_start, int k_stop) {
  ErrorCodes err = CheckKnotPointIndices(k_start, k_stop, LastIndexMode::Inclusive);
  if (err != ErrorCodes::NoError) return err;
  for (int k = k_start; k < k_stop; ++k) {
    Knot& kp = solver_->data[k];
    kp.cost_fun = cf; kp.cost_grad = cg; kp.cost_hess = ch; kp.cost_set = true; kp.cost_kind = CostKind::Generic;
  }
  return ErrorCodes::NoError;
}
static void set_diag(Knot& kp, int n, int m, const double* Qd, const double* Rd, const double* q, const double* r, double c) {
  kp.Q.assign((size_t)n * n, 0.0);   // diagonal kept in the head, like knotpoint_data.cpp:92-95
  std::copy(Qd, Qd + n, kp.Q.begin());
  kp.H.assign((size_t)std::max(m, 1) * n, 0.0);
  kp.q.assign(q, q + n);
  kp.c = c;
  if (!kp.terminal) {
    kp.R.assign((size_t)m * m, 0.0);
    std::copy(Rd, Rd + m, kp.R.begin());
    kp.r.assign(r, r + m);
  }
  kp.cost_set = true;
  kp.cost_kind = CostKind::Diagonal;
}
ErrorCodes ALTROSolver::SetDiagonalCost(int num_states, int num_inputs, const a_float* Qd, const a_float* Rd,
                                        const a_float* q, const a_float* r, a_float c, int k_start, int k_stop) {
  ErrorCodes err = CheckKnotPointIndices(k_start, k_stop, LastIndexMode::Inclusive);
  err = AssertDimensionsAreSet(k_start, k_stop, "Cannot set the cost function");
  if (err != ErrorCodes::NoError) return err;
  for (int k = k_start; k < k_stop; ++k) {
    const int n = GetStateDim(k), m = GetInputDim(k);
    if (n != num_states) return ErrorCodes::DimensionMismatch;
    if (k != GetHorizonLength() && m != num_inputs) return ErrorCodes::DimensionMismatch;
    set_diag(solver_->data[k], n, m, Qd, Rd, q, r, c);
  }
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::SetQuadraticCost(int num_states, int num_inputs, const a_float* Q, const a_float* R,
                                         const a_float* H, const a_float* q, const a_float* r, a_float c, int k_start,
                                         int k_stop) {
  ErrorCodes err = CheckKnotPointIndices(k_start, k_stop, LastIndexMode::Inclusive);
  err = AssertDimensionsAreSet(k_start, k_stop, "Cannot set the cost function");
  if (err != ErrorCodes::NoError) return err;
  for (int k = k_start; k < k_stop; ++k) {
    const int n = GetStateDim(k), m = GetInputDim(k);
    if (n != num_states) return ErrorCodes::DimensionMismatch;
    if (k != GetHorizonLength() && m != num_inputs) return ErrorCodes::DimensionMismatch;
    Knot& kp = solver_->data[k];
    kp.Q.assign(Q, Q + (size_t)n * n);
    kp.R.assign(R, R + (size_t)m * m);
    kp.H.assign(H, H + (size_t)m * n);
    kp.q.assign(q, q + n);
    kp.r.assign(r, r + m);
    kp.c = c;
    kp.cost_set = true;
    kp.cost_kind = CostKind::Quadratic;
  }
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::SetLQRCost(int num_states, int num_inputs, const a_float* Qd, const a_float* Rd,
                                   const a_float* x_ref, const a_float* u_ref, int k_start, int k_stop) {
  ErrorCodes err = CheckKnotPointIndices(k_start, k_stop, LastIndexMode::Inclusive);
  err = AssertDimensionsAreSet(k_start, k_stop, "Cannot set the cost function");
  if (err != ErrorCodes::NoError) return err;
  const int N = GetHorizonLength();
  for (int k = k_start; k < k_stop; ++k) {
    const int n = GetStateDim(k), m = GetInputDim(k);
    if (n != num_states) return ALTRO_THROW("State dimension mismatch", ErrorCodes::DimensionMismatch);
    if (k != N && m != num_inputs) return ALTRO_THROW("Input dimension mismatch", ErrorCodes::DimensionMismatch);
    Vec q(n), r(std::max(m, 1));   // q = -Q xref, r = -R uref, c = 1/2 xref'Q xref (+ 1/2 uref'R uref)
    double c = 0.0;
    for (int i = 0; i < n; ++i) { q[i] = -(Qd[i] * x_ref[i]); c += x_ref[i] * Qd[i] * x_ref[i]; }
    c *= 0.5;
    if (k != N) {
      double cu = 0.0;
      for (int i = 0; i < m; ++i) { r[i] = -(Rd[i] * u_ref[i]); cu += u_ref[i] * Rd[i] * u_ref[i]; }
      c += 0.5 * cu;
    }
    set_diag(solver_->data[k], n, m, Qd, Rd, q.data(), r.data(), c);
  }
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::SetInitialState(const double* x0, int n) {
  const int n0 = GetStateDim(0);
  if (n0 <= 0) solver_->nx[0] = n;
  else if (n != n0) return ALTRO_THROW("Dimension mismatch: initial state", ErrorCodes::DimensionMismatch);
  solver_->initial_state.assign(x0, x0 + n);
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::SetConstraint(ConstraintFunction cfun, ConstraintJacobian cjac, int dim, ConstraintType type,
                                      std::string label, int k_start, int k_stop, std::vector<ConstraintIndex>* con_inds) {
  ErrorCodes err = CheckKnotPointIndices(k_start, k_stop, LastIndexMode::Inclusive);
  if (IsInitialized()) return ALTRO_THROW("Cannot Set Constraints: Solver Already Initialized.", ErrorCodes::SolverAlreadyInitialized);
  err = AssertDimensionsAreSet(k_start, k_stop, "Cannot set constraint");
  if (err != ErrorCodes::NoError) return err;
  if (dim <= 0) return ALTRO_THROW("Got a non-positive constraint dimension", ErrorCodes::InvalidConstraintDim);
  const int num = k_stop - k_start;
  if (con_inds) con_inds->reserve(num);
  for (int k = k_start; k < k_stop; ++k) {
    Knot& kp = solver_->data[k];
    Constraint cn;
    cn.fun = cfun; cn.jac_fun = cjac; cn.dim = dim; cn.type = type;
    cn.label = (num != 1) ? label + "_" + std::to_string(k) : label;
    const int idx = (int)kp.cons.size();
    kp.cons.push_back(std::move(cn));
    if (con_inds) con_inds->emplace_back(ConstraintIndex(k, idx));
  }
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::SetDeviceModel(int altro_hip_model, int bicycle_frame, double bicycle_length, double bicycle_lr) {
  if (IsInitialized()) return ALTRO_THROW("Cannot set the dynamics: Solver Already Initialized.", ErrorCodes::SolverAlreadyInitialized);
  const int N = GetHorizonLength();
  ErrorCodes err = AssertDimensionsAreSet(0, N, "Cannot set the device model");
  if (err != ErrorCodes::NoError) return err;
  if (altro_hip_model < 0) return ALTRO_THROW("SetDeviceModel: unknown model", ErrorCodes::InvalidPointer);
  for (int k = 0; k < N; ++k) {
    Knot& kp = solver_->data[k];
    kp.dyn = nullptr; kp.dyn_jac = nullptr;
    kp.dyn_set = true; kp.dyn_linear = false; kp.dyn_device = true;
  }
  solver_->device_model = altro_hip_model; solver_->device_frame = bicycle_frame;
  solver_->device_length = bicycle_length; solver_->device_lr = bicycle_lr;
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::SetLinearConstraint(const a_float* G, const a_float* g, int dim, ConstraintType type, std::string label, int k_start,
                                            int k_stop, std::vector<ConstraintIndex>* con_inds) {
  if (!G || !g) return ALTRO_THROW("SetLinearConstraint: G and g are required", ErrorCodes::InvalidPointer);
  ErrorCodes err = CheckKnotPointIndices(k_start, k_stop, LastIndexMode::Inclusive);
  if (err != ErrorCodes::NoError) return err;
  err = AssertDimensionsAreSet(k_start, k_stop, "Cannot set constraint");
  if (err != ErrorCodes::NoError) return err;
  if (dim <= 0) return ALTRO_THROW("Got a non-positive constraint dimension", ErrorCodes::InvalidConstraintDim);
  const int N = GetHorizonLength();
  for (int k = k_start; k < k_stop; ++k) {
    const int n = GetStateDim(k), m = k == N ? 0 : GetInputDim(k), w = n + m;
    if (n != GetStateDim(k_start) || m != (k_start == N ? 0 : GetInputDim(k_start)))
      return ALTRO_THROW("SetLinearConstraint: the knot points of one call must share their dimensions", ErrorCodes::DimensionMismatch);
    Vec Gk(G, G + (size_t)dim * w), gk(g, g + dim);
    const int wh = n + std::max(m, 1);      // the host record's Jacobian width (a terminal knot point keeps one input column)
    auto fun = [Gk, gk, dim, n, m](a_float* c, const a_float* x, const a_float* u) {
      for (int i = 0; i < dim; ++i) {
        double s = 0.0;
        for (int e = 0; e < n; ++e) s += Gk[i + (size_t)e * dim] * x[e];
        for (int e = 0; e < m; ++e) s += Gk[i + (size_t)(n + e) * dim] * u[e];
        c[i] = s - gk[i];
      }
    };
    auto jac = [Gk, dim, w, wh](a_float* J, const a_float* x, const a_float* u) {
      (void)x; (void)u;
      std::fill(J, J + (size_t)dim * wh, 0.0);
      std::copy(Gk.begin(), Gk.begin() + (size_t)dim * w, J);
    };
    std::vector<ConstraintIndex> one;
    err = SetConstraint(fun, jac, dim, type, (k_stop - k_start != 1) ? label + "_" + std::to_string(k) : label, k, k + 1, &one);
    if (err != ErrorCodes::NoError) return err;
    Constraint& cn = solver_->data[k].cons.back();
    cn.linear = true; cn.G = Gk; cn.g = gk;
    if (m == 0) {   // the device takes G as dim x (n + m) of the HANDLE's dimensions: zero input columns at the terminal knot point
      cn.G.resize((size_t)dim * (n + GetInputDim(0)), 0.0);
    }
    if (con_inds) con_inds->insert(con_inds->end(), one.begin(), one.end());
  }
  return ErrorCodes::NoError;
}
bool ALTROSolver::IsInitialized() const { return solver_->initialized; }
ErrorCodes ALTROSolver::Initialize() {
  AssertDimensionsAreSet(0, GetHorizonLength(), "Cannot initialize solver");
  return solver_->Initialize();
}
ErrorCodes ALTROSolver::SetState(const a_float* x, int n, int k_start, int k_stop) {
  ErrorCodes err = AssertInitialized();
  err = CheckKnotPointIndices(k_start, k_stop, LastIndexMode::Inclusive);
  if (err != ErrorCodes::NoError) return err;
  for (int k = k_start; k < k_stop; ++k) { AssertStateDim(k, n); solver_->data[k].x_.assign(x, x + n); }
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::SetInput(const a_float* u, int m, int k_start, int k_stop) {
  ErrorCodes err = AssertInitialized();
  err = CheckKnotPointIndices(k_start, k_stop, LastIndexMode::Exclusive);
  if (err != ErrorCodes::NoError) return err;
  for (int k = k_start; k < k_stop; ++k) { AssertInputDim(k, m); solver_->data[k].u_.assign(u, u + m); }
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::OpenLoopRollout() {
  if (!IsInitialized()) return ErrorCodes::SolverNotInitialized;
  if (solver_->device_model >= 0) return ALTRO_THROW("OpenLoopRollout: a device model has no host callbacks (the rollout happens inside Solve)", ErrorCodes::DynamicsFunNotSet);
  solver_->OpenLoopRollout();
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::UpdateLinearCosts(const a_float* q, const a_float* r, a_float c, int k_start, int k_stop) {
  ErrorCodes err = AssertInitialized();
  err = CheckKnotPointIndices(k_start, k_stop, LastIndexMode::Inclusive);
  if (err != ErrorCodes::NoError) return ALTRO_THROW("Error in UpdateLinearCosts", err);
  for (int k = k_start; k < k_stop; ++k) {
    Knot& kp = solver_->data[k];
    if (kp.cost_kind == CostKind::Generic) return ALTRO_THROW("Cannot update linear costs: cost not quadratic", ErrorCodes::CostNotQuadratic);
    if (r != nullptr && kp.terminal) return ALTRO_THROW("Cannot update linear input costs at the terminal index", ErrorCodes::InvalidOptAtTerminalKnotPoint);
    if (q) std::copy(q, q + kp.n, kp.q.begin());
    if (r && !kp.terminal) std::copy(r, r + kp.m, kp.r.begin());
    kp.c = c;
  }
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::ShiftTrajectory() {   // altro_solver.cpp:283-293
  const int N = GetHorizonLength();
  for (int k = 0; k < N; ++k) {
    solver_->data[k].x_ = solver_->data[k + 1].x_;
    if (k < N - 1) solver_->data[k].u_ = solver_->data[k + 1].u_;
  }
  return ErrorCodes::NoError;
}
void ALTROSolver::SetOptions(const AltroOptions& opts) { solver_->opts = opts; }
AltroOptions& ALTROSolver::GetOptions() { return solver_->opts; }
const AltroOptions& ALTROSolver::GetOptions() const { return solver_->opts; }
SolveStatus ALTROSolver::Solve() {
  if (solver_->device_model >= 0) solver_->DeviceSolve();
  else solver_->Solve();
  return solver_->stats.status;
}
SolveStatus ALTROSolver::GetStatus() const { return solver_->stats.status; }
int ALTROSolver::GetIterations() const { return solver_->stats.iterations; }
a_float ALTROSolver::GetSolveTimeMs() const { return solver_->stats.solve_time.count(); }
a_float ALTROSolver::GetPrimalFeasibility() const { return solver_->stats.primal_feasibility; }
a_float ALTROSolver::GetFinalObjective() const { return solver_->stats.objective_value; }
a_float ALTROSolver::CalcCost() { return solver_->CalcCost(); }
int ALTROSolver::GetHorizonLength() const { return solver_->N; }
int ALTROSolver::GetStateDim(int k) const { return solver_->data[k].n; }
int ALTROSolver::GetInputDim(int k) const { return solver_->data[k].m; }
float ALTROSolver::GetFinalTime() const {
  float t = 0.0f;
  for (float h : solver_->hs) t += h;
  return t;
}
float ALTROSolver::GetTimeStep(int k) const { return solver_->hs[k]; }
ErrorCodes ALTROSolver::GetState(a_float* x, int k) const {
  int k_stop = k + 1;
  ErrorCodes err = CheckKnotPointIndices(k, k_stop, LastIndexMode::Inclusive);
  if (err != ErrorCodes::NoError) return ALTRO_THROW("Error in GetState", err);
  std::copy(solver_->data[k].x_.begin(), solver_->data[k].x_.begin() + GetStateDim(k), x);
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::GetInput(a_float* u, int k) const {
  int k_stop = k + 1;
  ErrorCodes err = CheckKnotPointIndices(k, k_stop, LastIndexMode::Exclusive);
  if (err != ErrorCodes::NoError) return ALTRO_THROW("Error in GetInput", err);
  std::copy(solver_->data[k].u_.begin(), solver_->data[k].u_.begin() + GetInputDim(k), u);
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::GetDualDynamics(a_float* y, int k) const {
  int k_stop = k + 1;
  ErrorCodes err = CheckKnotPointIndices(k, k_stop, LastIndexMode::Inclusive);
  if (err != ErrorCodes::NoError) return ALTRO_THROW("Error in GetDualDynamics", err);
  std::copy(solver_->data[k].y_.begin(), solver_->data[k].y_.begin() + GetStateDim(k), y);
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::GetFeedbackGain(a_float* K, int k) const {
  int k_stop = k + 1;
  ErrorCodes err = CheckKnotPointIndices(k, k_stop, LastIndexMode::Exclusive);
  if (err != ErrorCodes::NoError) return ALTRO_THROW("Error in GetFeedbackGain", err);
  std::copy(solver_->data[k].K.begin(), solver_->data[k].K.begin() + (size_t)GetInputDim(k) * GetStateDim(k), K);
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::GetFeedforwardGain(a_float* d, int k) const {
  int k_stop = k + 1;
  ErrorCodes err = CheckKnotPointIndices(k, k_stop, LastIndexMode::Exclusive);
  if (err != ErrorCodes::NoError) return ALTRO_THROW("Error in GetFeedforwardGain", err);
  std::copy(solver_->data[k].d.begin(), solver_->data[k].d.begin() + GetInputDim(k), d);
  return ErrorCodes::NoError;
}
void ALTROSolver::PrintStateTrajectory() const {
  for (int k = 0; k <= GetHorizonLength(); ++k) {
    std::printf("%03d: [", k);
    for (int i = 0; i < GetStateDim(k); ++i) std::printf("%s%g", i ? ", " : "", solver_->data[k].x_[i]);
    std::printf("]\n");
  }
}
void ALTROSolver::PrintInputTrajectory() const {
  for (int k = 0; k < GetHorizonLength(); ++k) {
    std::printf("%03d: [", k);
    for (int i = 0; i < GetInputDim(k); ++i) std::printf("%s%g", i ? ", " : "", solver_->data[k].u_[i]);
    std::printf("]\n");
  }
}

ErrorCodes ALTROSolver::CheckKnotPointIndices(int& k_start, int& k_stop, LastIndexMode last_index) const {
  // altro_solver.cpp:385-433: (AllIndices, 0) and (0, LastIndex) mean "all"; k_stop <= 0 means one index
  const int terminal_index = (last_index == LastIndexMode::Inclusive) ? GetHorizonLength() : GetHorizonLength() - 1;
  if (k_start == AllIndices && k_stop == 0) { k_start = 0; k_stop = LastIndex; }
  if (k_start == 0 && k_stop == LastIndex) { k_start = 0; k_stop = terminal_index + 1; }
  if (k_stop <= 0) k_stop = k_start + 1;
  if (k_start < 0 || k_start > terminal_index) return ALTRO_THROW("Knot point index out of range.", ErrorCodes::BadIndex);
  if (k_stop < 0 || k_start > terminal_index + 1) return ALTRO_THROW("Terminal knot point index out of range.", ErrorCodes::BadIndex);
  if (k_stop > 0 && k_stop <= k_start)
    std::printf("WARNING [ALTRO]: Stopping index %d not greater than starting index %d. Index range is empty.\n", k_stop, k_start);
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::AssertInitialized() const {
  if (!IsInitialized()) return ALTRO_THROW("Solver must be initialized.", ErrorCodes::SolverNotInitialized);
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::AssertDimensionsAreSet(int k_start, int k_stop, std::string msg) const {
  const int N = GetHorizonLength();
  for (int k = k_start; k < k_stop; ++k) {
    const int n = GetStateDim(k), m = GetInputDim(k);
    if (n <= 0 || (k < N && m <= 0)) ALTRO_THROW(msg + ". Dimensions haven't been set at a knot point.", ErrorCodes::DimensionUnknown);
  }
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::AssertStateDim(int k, int n) const {
  if (GetStateDim(k) != n) return ALTRO_THROW("State dimension mismatch.", ErrorCodes::DimensionMismatch);
  return ErrorCodes::NoError;
}
ErrorCodes ALTROSolver::AssertInputDim(int k, int m) const {
  if (GetInputDim(k) != m) return ALTRO_THROW("Input dimension mismatch.", ErrorCodes::DimensionMismatch);
  return ErrorCodes::NoError;
}

}  // namespace altro
