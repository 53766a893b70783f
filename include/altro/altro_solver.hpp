// altro_solver.hpp -- altro::ALTROSolver, the reference's public solver class
// (src/altro/altro_solver.hpp:21-442), kept method for method so that callers compile unchanged:
// Eigen-free header, raw `const double*` buffers, `float h`, std::function callbacks, index ranges
// [k_start, k_stop) with the LastIndex / AllIndices conventions of altro_solver.cpp:385-433.
//
// What runs where.  The user's dynamics / cost / constraint callbacks are host std::functions and are
// evaluated on the host, exactly like the reference.  The Riccati backward sweep -- and the linear
// rollout -- of every iLQR iteration go through the tvlqr_* kernel boundary (include/tvlqr/tvlqr.h),
// i.e. they execute on the MI355X.  Many-problem workloads should use the batched C ABI
// (include/altro_hip/altro_hip.h), where the whole loop including the models is on the device.
//
// Differences from the reference, all deliberate:
//   * SetCostFunction / SetDiagonalCost are no-ops in the reference because of a loop-bound typo
//     (altro_solver.cpp:91,105); here SetDiagonalCost works and SetCostFunction stores the callbacks.
//   * GetFeedbackGain / GetFeedforwardGain / GetStatus / GetFinalTime are declared but never defined
//     in the reference; they are defined here.
//   * Methods the reference declares and never defines and no test calls (copy construction,
//     SetImplicitDynamics, bound setters, SetDual*, SetCallback, GetDual*Bound) are not declared.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "altro/solver/exceptions.hpp"
#include "altro/solver/solver_options.hpp"
#include "altro/solver/solver_stats.hpp"
#include "altro/solver/typedefs.hpp"

namespace altro {

class SolverImpl;

class ALTROSolver {
 public:
  explicit ALTROSolver(int horizon_length);
  ALTROSolver(ALTROSolver&& other);
  ALTROSolver& operator=(ALTROSolver&& other);
  ~ALTROSolver();

  // ---- problem definition (before Initialize) ------------------------------------------------------
  ErrorCodes SetDimension(int num_states, int num_inputs, int k_start = AllIndices, int k_stop = 0);
  ErrorCodes SetTimeStep(float h, int k_start = AllIndices, int k_stop = 0);
  ErrorCodes SetExplicitDynamics(ExplicitDynamicsFunction dynamics_function,
                                 ExplicitDynamicsJacobian dynamics_jacobian, int k_start = AllIndices,
                                 int k_stop = 0);
  ErrorCodes SetCostFunction(CostFunction cost_function, CostGradient cost_gradient, CostHessian cost_hessian,
                             int k_start = AllIndices, int k_stop = 0);
  // 1/2 x'Qx + q'x + 1/2 u'Ru + r'u + c with diagonal Q, R
  ErrorCodes SetDiagonalCost(int num_states, int num_inputs, const a_float* Q_diag, const a_float* R_diag,
                             const a_float* q, const a_float* r, a_float c, int k_start = AllIndices,
                             int k_stop = 0);
  // dense column-major Q (n,n), R (m,m), H (m,n): ... + u'Hx
  ErrorCodes SetQuadraticCost(int num_states, int num_inputs, const a_float* Q, const a_float* R, const a_float* H,
                              const a_float* q, const a_float* r, a_float c, int k_start = AllIndices,
                              int k_stop = 0);
  // 1/2 (x-xref)'Q(x-xref) + 1/2 (u-uref)'R(u-uref), diagonal Q, R
  ErrorCodes SetLQRCost(int num_states, int num_inputs, const a_float* Q_diag, const a_float* R_diag,
                        const a_float* x_ref, const a_float* u_ref, int k_start, int k_stop = 0);
  // c(x,u) in K; jac is (dim, n+m) column-major
  ErrorCodes SetConstraint(ConstraintFunction constraint_function, ConstraintJacobian constraint_jacobian, int dim,
                           ConstraintType constraint_type, std::string label, int k_start, int k_stop = 0,
                           std::vector<ConstraintIndex>* con_inds = nullptr);
  // ---- extensions (not in the reference): the whole Solve on the device -------------------------------
  // A compiled-in device model (ALTRO_HIP_MODEL_* of include/altro_hip/altro_hip.h: double integrator, pendulum, bicycle, the
  // quadrotors) in the place of SetExplicitDynamics' host callbacks, for every knot point, discretised with the explicit midpoint
  // rule at SetTimeStep's step.  Solve() of such a solver runs SolverImpl::Solve (solver.cpp:414-511) on the device --
  // altro_hip_ilqr_solve on a resident batch of one, one host round trip per Solve instead of two per sweep -- and leaves states,
  // inputs, status, iterations and the final objective where the getters look.  It needs what a device loop can hold: uniform
  // dimensions and time step, costs set as data (SetDiagonalCost / SetQuadraticCost / SetLQRCost), constraints given as data
  // (SetLinearConstraint below; a callback pair cannot run on the device).  OpenLoopRollout of such a solver returns
  // DynamicsFunNotSet: there is no host model to evaluate.  `bicycle_*`: MODEL_BICYCLE's reference frame / length / lr.
  ErrorCodes SetDeviceModel(int altro_hip_model, int bicycle_frame = 0, double bicycle_length = 2.7, double bicycle_lr = 1.5);
  // c(x, u) = G [x; u] - g in `constraint_type`'s cone at k_start <= k < k_stop: SetConstraint with the callback pair made here
  // (G is (dim, n + m) column-major, (dim, n) for the terminal knot point alone; both are copied), and the data kept for the device.
  ErrorCodes SetLinearConstraint(const a_float* G, const a_float* g, int dim, ConstraintType constraint_type, std::string label,
                                 int k_start, int k_stop = 0, std::vector<ConstraintIndex>* con_inds = nullptr);
  bool IsInitialized() const;

  // ---- initialization and initial guess ------------------------------------------------------------
  ErrorCodes Initialize();
  ErrorCodes SetInitialState(const double* x0, int n);
  ErrorCodes SetState(const a_float* x, int n, int k_start = AllIndices, int k_stop = 0);
  ErrorCodes SetInput(const a_float* u, int m, int k_start = AllIndices, int k_stop = 0);
  ErrorCodes OpenLoopRollout();

  // ---- MPC helpers -------------------------------------------------------------------------------------
  ErrorCodes UpdateLinearCosts(const a_float* q, const a_float* r, a_float c, int k_start = AllIndices,
                               int k_stop = 0);
  ErrorCodes ShiftTrajectory();

  // ---- options / solve -----------------------------------------------------------------------------------
  void SetOptions(const AltroOptions& opts);
  AltroOptions& GetOptions();
  const AltroOptions& GetOptions() const;
  SolveStatus Solve();
  SolveStatus GetStatus() const;
  int GetIterations() const;
  a_float GetSolveTimeMs() const;
  a_float GetPrimalFeasibility() const;
  a_float GetFinalObjective() const;
  a_float CalcCost();

  // ---- getters ---------------------------------------------------------------------------------------------
  int GetHorizonLength() const;
  int GetStateDim(int k) const;
  int GetInputDim(int k) const;
  float GetFinalTime() const;
  float GetTimeStep(int k) const;
  ErrorCodes GetState(a_float* x, int k) const;
  ErrorCodes GetInput(a_float* u, int k) const;
  ErrorCodes GetDualDynamics(a_float* y, int k) const;
  ErrorCodes GetFeedbackGain(a_float* K, int k) const;      // (m, n) column-major
  ErrorCodes GetFeedforwardGain(a_float* d, int k) const;   // (m)

  void PrintStateTrajectory() const;
  void PrintInputTrajectory() const;

  std::unique_ptr<SolverImpl> solver_;   // public in the reference too (altro_solver.hpp:430)

 private:
  enum class LastIndexMode { Inclusive, Exclusive };
  ErrorCodes CheckKnotPointIndices(int& k_start, int& k_stop, LastIndexMode last_index) const;
  ErrorCodes AssertInitialized() const;
  ErrorCodes AssertDimensionsAreSet(int k_start, int k_stop, std::string msg = "") const;
  ErrorCodes AssertStateDim(int k, int n) const;
  ErrorCodes AssertInputDim(int k, int m) const;
};

}  // namespace altro
