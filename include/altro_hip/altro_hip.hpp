// altro_hip.hpp -- header-only C++17 convenience over the C ABI of altro_hip.h (no additional symbols: everything
// here inlines into calls of the extern "C" entry points, so it adds nothing to the drop-in boundary).
//
// altro::hip::BatchSolver is the batched sibling of altro::ALTROSolver (src/altro/altro_solver.hpp:21-442): the same
// vocabulary (SetLQRCost, SetInitialState, SetInput, SetConstraint, Solve, GetState / GetInput, UpdateLinearCosts,
// ShiftTrajectory) applied to `batch` independent problems resident on one MI355X.  Errors follow the reference's
// convention for ALTRO_ENABLE_RUNTIME_EXCEPTIONS: they throw (std::runtime_error with altro_hip_last_error()).
#pragma once
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "altro_hip/altro_hip.h"

namespace altro {
namespace hip {

enum class Cone { Equality = 0, Identity = 1, Inequality = 2, SecondOrder = 3 };   // ConstraintType, typedefs.hpp:29-34

struct SolveResult {
  std::vector<altro_hip_solve_result> problems;   // AltroStats per problem
  int sweeps = 0, merit_launches = 0;
  int NumConverged() const {
    int c = 0;
    for (const auto& r : problems) c += r.status == 0;
    return c;
  }
};

class BatchSolver {
 public:
  BatchSolver(int horizon_length, int num_states, int num_inputs, int batch, altro_hip_dtype dtype = ALTRO_HIP_F64,
              int device = 0, unsigned flags = 0, void* stream = nullptr)
      : N_(horizon_length), n_(num_states), m_(num_inputs), batch_(batch) {
    Check(altro_hip_batch_create(&h_, N_, n_, m_, batch_, dtype, ALTRO_HIP_PLAN_AUTO, flags, device, stream));
    altro_hip_default_solve_options(&opts);
  }
  ~BatchSolver() { altro_hip_batch_destroy(h_); }
  BatchSolver(const BatchSolver&) = delete;
  BatchSolver& operator=(const BatchSolver&) = delete;
  BatchSolver(BatchSolver&& o) noexcept { *this = std::move(o); }
  BatchSolver& operator=(BatchSolver&& o) noexcept {
    if (this != &o) {
      if (h_) altro_hip_batch_destroy(h_);
      h_ = o.h_; o.h_ = nullptr;
      N_ = o.N_; n_ = o.n_; m_ = o.m_; batch_ = o.batch_; opts = o.opts;
    }
    return *this;
  }

  int GetHorizonLength() const { return N_; }
  int GetStateDim() const { return n_; }
  int GetInputDim() const { return m_; }
  int GetBatch() const { return batch_; }
  int GetPlan() const { return altro_hip_batch_plan(h_); }
  altro_hip_batch* Handle() { return h_; }

  // ---- problem definition (arrays in the reference's layout, [batch][k][column-major block]) ----------------
  // KnotPointData::SetLinearDynamics for every knot point of every problem (plan MFMA16 / TVLQR sweeps)
  void SetLinearDynamics(const double* A, const double* B, const double* f, bool shared_over_k = false,
                         bool shared_over_batch = false) {
    Check(altro_hip_set_dynamics(h_, A, B, f, shared_over_k, shared_over_batch));
  }
  // SetExplicitDynamics with one of the compiled-in device models (plan LANE)
  void SetModel(altro_hip_model model, float timestep, int bicycle_frame = 0, double bicycle_length = 2.7,
                double bicycle_lr = 1.5) {
    Check(altro_hip_set_model(h_, model, timestep, bicycle_frame, bicycle_length, bicycle_lr));
  }
  // ALTROSolver::SetLQRCost for all knot points: Qd, xref [batch][N+1][n]; Rd, uref [batch][N][m]
  void SetLQRCost(const double* Qd, const double* Rd, const double* xref, const double* uref, bool shared_over_k = false,
                  bool shared_over_batch = false) {
    Check(altro_hip_set_tracking_cost(h_, Qd, Rd, xref, uref, shared_over_k, shared_over_batch));
  }
  // ALTROSolver::SetQuadraticCost (altro_solver.cpp:118-136) for all knot points: the general quadratic cost with the cross term
  // u^T H x.  Q [batch][N+1][n*n], R [batch][N][m*m], H [batch][N][m*n] (column-major blocks), q, r, c likewise (c may be null)
  void SetQuadraticCost(const double* Q, const double* R, const double* H, const double* q, const double* r, const double* c,
                        bool shared_over_k = false, bool shared_over_batch = false) {
    Check(altro_hip_set_quadratic_cost(h_, Q, R, H, q, r, c, shared_over_k, shared_over_batch));
  }
  // SetExplicitDynamics with the caller's own continuous model as HIP source, compiled at run time (altro_hip_set_model_source)
  void SetModelSource(const char* source, float timestep) { Check(altro_hip_set_model_source(h_, source, timestep)); }
  // whether the device model runs the loop's row-layout model kernels (plans GENERIC / MFMA32; altro_hip_model_row_layout)
  bool ModelRowLayout() const { return altro_hip_model_row_layout(h_) != 0; }
  void SetInitialState(const double* x0, bool shared_over_batch = false) {
    Check(altro_hip_set_initial_state(h_, x0, shared_over_batch));
  }
  void SetInput(const double* u, bool shared_over_k = false, bool shared_over_batch = false) {
    Check(altro_hip_set_input_guess(h_, u, shared_over_k, shared_over_batch));
  }
  void SetState(const double* x, bool shared_over_k = false, bool shared_over_batch = false) {
    Check(altro_hip_set_state_guess(h_, x, shared_over_k, shared_over_batch));
  }
  // ALTROSolver::SetConstraint for c = G [x;u] - g in `cone` at knot points k_start .. k_stop (inclusive);
  // G is p x (n+m) column-major; returns the block id
  int SetConstraint(int k_start, int k_stop, Cone cone, int p, const double* G, const double* g, bool g_per_problem = false) {
    const int id = altro_hip_add_linear_constraint(h_, k_start, k_stop, static_cast<int>(cone), p, G, g, g_per_problem);
    if (id < 0) Check(id);
    return id;
  }
  void ClearConstraints() { Check(altro_hip_clear_constraints(h_)); }
  void ResetDuals(double penalty = 1.0) { Check(altro_hip_reset_duals(h_, penalty)); }

  // ---- solve ------------------------------------------------------------------------------------------------------
  altro_hip_solve_options opts;   // AltroOptions (solver_options.hpp:16-39)
  SolveResult Solve() {
    SolveResult r;
    r.problems.resize(batch_);
    Check(altro_hip_ilqr_solve(h_, &opts, r.problems.data()));
    Check(altro_hip_last_solve_counts(h_, &r.sweeps, &r.merit_launches));
    return r;
  }
  // TVLQR sweep on the data as it stands: tvlqr_BackwardPass + tvlqr_ForwardPass for every problem
  void Sweep(double reg = 0.0) { Check(altro_hip_sweep(h_, reg)); }
  void Synchronize() { Check(altro_hip_synchronize(h_)); }

  // ---- results (host arrays, reference layout) ---------------------------------------------------------------
  void GetTrajectory(double* x /*[batch][N+1][n]*/, double* u /*[batch][N][m]*/) { Check(altro_hip_get_nominal(h_, x, u)); }
  void GetKnotPoint(int k, double* x /*[batch][n]*/, double* u /*[batch][m] or nullptr*/) { Check(altro_hip_get_knot(h_, k, x, u)); }
  void GetFeedbackGain(double* K /*[batch][N][m*n]*/) { Check(altro_hip_get_K(h_, K)); }
  void GetFeedforwardGain(double* d /*[batch][N][m]*/) { Check(altro_hip_get_d(h_, d)); }
  void GetCostToGo(double* P, double* p) {
    if (P) Check(altro_hip_get_P(h_, P));
    if (p) Check(altro_hip_get_p(h_, p));
  }
  void GetBackwardStatus(int* status /*[batch]: -1 or the failing knot point*/) { Check(altro_hip_get_status(h_, status)); }

  // ---- MPC methods (altro_solver.cpp:266-293) ---------------------------------------------------------------------
  void UpdateLinearCosts(const double* q, const double* r, const double* c, int k_start, int k_stop, bool shared_over_k = false,
                         bool shared_over_batch = false) {
    Check(altro_hip_update_linear_costs(h_, q, r, c, k_start, k_stop, shared_over_k, shared_over_batch));
  }
  void ShiftTrajectory() { Check(altro_hip_shift_trajectory(h_)); }

 private:
  static void Check(int rc) {
    if (rc != 0) throw std::runtime_error(std::string("altro_hip error ") + std::to_string(rc) + ": " + altro_hip_last_error());
  }
  altro_hip_batch* h_ = nullptr;
  int N_ = 0, n_ = 0, m_ = 0, batch_ = 0;
};

}  // namespace hip
}  // namespace altro
