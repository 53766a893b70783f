// ilqr_loop_logic.h -- the per-problem bookkeeping of one sweep of SolverImpl::Solve (solver.cpp:237-271, :447-502) as
// inline device functions on IlqrProb: ONE copy of the logic, used by the one-thread-per-problem kernels of the
// launch-sequenced solve (ilqr_loop_kernels.hip) and by the fused solve kernel (ilqr_fused.hip), so that the two paths
// take the same decisions bit for bit.
#pragma once
#include <hip/hip_runtime.h>

#include "ilqr_types.h"

#if defined(__clang__)
#pragma clang fp contract(on)   // single-expression a * b + c only: the same rounding in every kernel these functions are inlined into (see models.h)
#endif
namespace altro_hip {

// start of a solve (what a fresh Solve() call resets)
__device__ __forceinline__ void ilqr_prob_init(IlqrProb& p) {
  p.running = 1; p.iterations = 0; p.status = 1; p.ls_failed = 0; p.evaluating = 0;
  p.alpha = 0.0; p.stationarity = 0.0; p.ls_iters = 0;
  p.feasibility = 0.0; p.dual = 0; p.n_dual_updates = 0; p.reg_retries = 0;
  p.rho_est = p.rho;   // the initial gradient is formed with the penalty left by Initialize / the last solve
}

// ForwardPass's head after merit(alpha = 0) (solver.cpp:241-249): returns whether the search wants a trial step
// (then p.ls.alpha is that step)
__device__ __forceinline__ bool ilqr_ls_begin_logic(IlqrProb& p, const LsOptions& ls, double tol_meritfun_gradient,
                                                    double phi0, double dphi0) {
  p.phi0 = phi0;
  p.dphi0 = dphi0;
  p.ls_failed = 0;
  bool need;
  if (fabs(p.dphi0) < tol_meritfun_gradient) {   // MeritFunctionGradientTooSmall: alpha = 0
    p.alpha = 0.0;
    p.ls_iters = 0;
    need = false;
  } else {
    need = ls_begin(p.ls, ls, 1.0, p.phi0, p.dphi0);
    if (!need) {   // not a descent direction
      p.alpha = p.ls.alpha;
      p.ls_iters = p.ls.n_iters;
      p.ls_failed = 1;
    }
  }
  return need;
}

// the search of problem p has ended (on whatever step was fed last): solver.cpp:264-268
__device__ __forceinline__ void ilqr_ls_end_logic(IlqrProb& p) {
  p.alpha = p.ls.alpha;
  p.ls_iters = p.ls.n_iters;
  const int st = p.ls.status;
  p.ls_failed = (isnan(p.alpha) || !(st == LS_MINIMUM_FOUND || st == LS_HIT_MAX_STEPSIZE)) ? 1 : 0;
}

// end of one sweep (solver.cpp:459-502): convergence test, outer-update decision, iteration bookkeeping
__device__ __forceinline__ void ilqr_finish_iter_logic(IlqrProb& p, const IlqrLoopArgs& a, int iter) {
  bool stop = p.ls_failed != 0;
  if (fabs(p.stationarity) < a.tol_stationarity && p.feasibility < a.tol_primal_feasibility) {
    p.status = 0;
    stop = true;
  }
  // outer AL update (solver.cpp:470-489), also on the sweep that stops
  p.dual = 0;
  if (a.al_enabled && p.stationarity < sqrt(a.tol_stationarity)) {
    p.dual = p.feasibility > a.tol_primal_feasibility ? 2 : 1;
    ++p.n_dual_updates;
  }
  p.iterations = iter + 1;
  if (!stop && iter + 1 >= a.iterations_max) {
    p.status = 2;   // MaxIterations; the reference reports iter + 1 after its loop ends (solver.cpp:503-506)
    p.iterations = iter + 2;
    stop = true;
  }
  if (stop) p.running = 0;
}

// Regularisation retry -- an EXTENSION (see ilqr_reg_retry_kernel): returns whether the backward pass is to be repeated
__device__ __forceinline__ bool ilqr_reg_retry_logic(IlqrProb& p, const IlqrLoopArgs& a, double& reg, int bwd_status) {
  if (bwd_status != -1) {
    const double r = fmax(reg * a.reg_scale, a.reg_min);
    if (r <= a.reg_max) { reg = r; ++p.reg_retries; return true; }
    return false;
  }
  reg = fmax(reg / a.reg_scale, a.reg_initial);
  return false;
}

// PenaltyUpdate (solver.cpp:396-409) after the duals were updated with the old penalty; returns whether the problem's
// cost gradients have to be recomputed
__device__ __forceinline__ bool ilqr_penalty_update_logic(IlqrProb& p, const IlqrLoopArgs& a) {
  if (p.dual == 2) p.rho = fmin(p.rho * a.penalty_scaling, a.penalty_max);
  if (p.dual) p.rho_est = p.rho;
  return p.dual != 0;
}

}  // namespace altro_hip
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
