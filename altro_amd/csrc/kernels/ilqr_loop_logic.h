// ilqr_loop_logic.h -- the per-problem bookkeeping of one sweep of SolverImpl::Solve (solver.cpp:237-271, :447-502) as
// inline device functions on IlqrProb: ONE copy of the logic, used by the one-thread-per-problem kernels of the
// launch-sequenced solve (ilqr_loop_kernels.hip) and by the fused solve kernel (ilqr_fused.hip), so that the two paths
// take the same decisions bit for bit.
#pragma once
#include <hip/hip_runtime.h>

#include "ilqr_types.h"

#include "../fp_contract.h"
ALTRO_FP_REGION_ON   // single-expression a * b + c only: the same rounding in every kernel these functions are inlined into (see models.h)
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

// AltroStats of one problem (solver.cpp:503-509) in the layout the C ABI hands out
__device__ __forceinline__ IlqrResult ilqr_result_of(const IlqrProb& p) {
  IlqrResult r;
  r.status = p.status;
  r.iterations = p.iterations;
  r.stationarity = p.stationarity;
  r.final_alpha = p.alpha;
  r.final_phi = p.ls_iters > 0 ? p.ls.phi : p.phi0;
  r.primal_feasibility = p.feasibility;
  r.penalty = p.rho;
  r.dual_updates = p.n_dual_updates;
  r.reg_retries = p.reg_retries;
  return r;
}

// ---- one problem's share of each bookkeeping step, on the per-batch arrays of IlqrLoopArgs: the bodies of the
//      one-thread-per-problem kernels (ilqr_loop_kernels.hip), also called by the fused solve kernel -----------------------

// set `active` := running (used before the per-sweep kernels)
__device__ __forceinline__ void ilqr_mark_running_body(const IlqrLoopArgs& a, int b) {
  a.active[b] = a.prob[b].running;
  a.alpha[b] = 0.0;
}

// after merit(alpha = 0): ForwardPass's head (solver.cpp:241-249); returns whether problem b needs a merit evaluation
__device__ __forceinline__ bool ilqr_ls_begin_body(const IlqrLoopArgs& a, int b) {
  IlqrProb& p = a.prob[b];
  a.spec_sel[b] = 0;
  a.spec_refresh[b] = 0;
  if (a.guard) { a.guard[b] = 0; a.active_exact[b] = 0; }   // (null in the one-launch kernel's arguments)
  if (!p.running) { a.active[b] = 0; return false; }
  bool need = ilqr_ls_begin_logic(p, a.ls, a.tol_meritfun_gradient, a.phi[b], a.dphi[b]);
  // Fused first trial (IlqrLoopArgs::spec_pre): the merit launch that produced phi(0) also evaluated the first step
  // alpha0 = 1 the search asks for (into phi / dphi row 1 and spare candidate 0), so it is consumed right here.
  if (need && a.spec_pre) {
    need = ls_feed(p.ls, a.ls, a.phi[(size_t)a.batch + b], a.dphi[(size_t)a.batch + b]);
    if (!need) {   // the search ended on that step (same bookkeeping as ilqr_ls_feed_body)
      ilqr_ls_end_logic(p);
      a.spec_sel[b] = 1;
      a.spec_refresh[b] = 1;
    }
  }
  if (a.spec_flip) {   // (IlqrLoopArgs::spec_flip) the first step's pass wrote the candidate and the expansion itself
    const bool on_first = a.spec_sel[b] == 1;
    a.spec_sel[b] = 0;
    a.spec_refresh[b] = (!on_first && !need) ? 1 : 0;
    a.stat_done[b] = (on_first && a.stat_inline) ? 1 : 0;
  }
  p.evaluating = need ? 1 : 0;
  a.active[b] = need ? 1 : 0;
  if (need) a.alpha[b] = p.ls.alpha;
  return need;
}

// after merit(alpha[b]): advance the state machine of problem b; returns whether it needs another evaluation
// GUARD: with the decision guard of the affine rounds compiled in (the launch-sequenced loop's kernel; the one-launch solve kernel of
// plan LANE has no affine rounds and keeps the function it had)
template <bool GUARD = false>
__device__ __forceinline__ bool ilqr_ls_feed_body(const IlqrLoopArgs& a, int b) {
  IlqrProb& p = a.prob[b];
  if (!p.running || !p.evaluating) { a.active[b] = 0; a.spec_sel[b] = 0; if (GUARD && a.guard) a.active_exact[b] = 0; return false; }
  // Affine rounds (IlqrLoopArgs::guard; plan MFMA16, dynamics as data).  The values of an affine trial equal the rollout's to rounding
  // only.  The decision guard (decision_margin > 0): a trial whose feed is not robust against that margin (linesearch_sm.h:
  // ls_feed_is_robust) stays pending (guard = 1) and is evaluated again as a rollout by the next round, on the mask active_exact.
  // aff_exact (ALTRO_HIP_FORM_AFFINE_EXACT) goes all the way: the search uses affine values for nothing but robust decisions --
  //   * a trial whose VALUES the search would keep is evaluated again as well: a rejected zoom or extrapolation step enters the bracket
  //     the next interpolation is taken from (a rejected cubic first guess does not: linesearch.cpp goes on from the first step's
  //     values; the backtracking sequence alpha beta^j is fixed in advance);
  //   * a search that ENDS on an affine trial is finalised (guard = 2): one rollout evaluation of the step it ended on writes the
  //     candidate trajectory, its expansion and the final merit values
  // -- and every step length, candidate and decision is the rollout form's bit for bit (tools/fuzz_affine.py: 0 of 5617 problems
  // differ, |dx| = 0), at the price of a rollout per accepted step: slower than plain rollout rounds, a checking form.
  if (GUARD && a.guard) {
    const int gs = a.guard[b];
    a.guard[b] = 0; a.active_exact[b] = 0;
    if (gs == 2) {   // the rollout evaluation of the step the search ended on
      p.ls.phi = a.phi[b]; p.ls.dphi = a.dphi[b];
      p.evaluating = 0; a.active[b] = 0; a.spec_sel[b] = 0;
      ilqr_ls_end_logic(p);
      return false;
    }
    if (gs == 0 && a.aff_fed && a.decision_margin > 0.0) {
      for (int j = 0; j < a.spec_trials; ++j) {
        const int stage0 = p.ls.stage;
        LsState t;
        bool need = false;
        const bool robust = ls_feed_is_robust(p.ls, a.ls, a.phi[(size_t)j * a.batch + b], j == 0 ? a.dphi[b] : 0.0, a.decision_margin, &t, &need);
        const bool keeps_values = a.aff_exact && need && !(stage0 == LS_STAGE_CUBIC || t.stage == LS_STAGE_BACKTRACK);
        if (!robust || keeps_values) {   // this trial again, as a rollout
          a.guard[b] = 1; a.active_exact[b] = 1; a.active[b] = 0; a.spec_sel[b] = 0;
          a.alpha[b] = p.ls.alpha;
          atomicAdd(&a.counters[5], 1);
          return true;
        }
        p.ls = t;
        if (!need && a.aff_exact) {      // ended on an affine trial: finalise on a rollout of that step
          a.guard[b] = 2; a.active_exact[b] = 1; a.active[b] = 0; a.spec_sel[b] = 0;
          a.alpha[b] = p.ls.alpha;
          atomicAdd(&a.counters[5], 1);
          return true;
        }
        if (!need) {                     // (the margin guard alone: the search ends on the affine trial like an unguarded one)
          a.spec_sel[b] = j;
          if (j > 0) a.spec_refresh[b] = 1;
          p.evaluating = 0; a.active[b] = 0;
          ilqr_ls_end_logic(p);
          return false;
        }
        // the next trial of this launch exists only along the backtracking sequence (see below)
        if (!((stage0 == LS_STAGE_BACKTRACK || stage0 == LS_STAGE_CUBIC) && p.ls.stage == LS_STAGE_BACKTRACK)) break;
      }
      a.alpha[b] = p.ls.alpha;
      a.active[b] = 1; a.spec_sel[b] = 0;
      return true;
    }
  }
  // Speculative backtracking: the merit launch also evaluated alpha beta^j, j = 1 .. spec_trials - 1, for the problems
  // that were in the backtracking stage or about to enter it (cubic first guess pending).  Feeding them in order
  // reproduces the sequential search exactly; "need" after trial j - 1 is precisely the condition under which trial j
  // exists (bt_iter below max_iters).
  const int stage0 = p.ls.stage;
  bool need = ls_feed(p.ls, a.ls, a.phi[b], a.dphi[b]);
  int last = 0;
  if (stage0 == LS_STAGE_BACKTRACK || stage0 == LS_STAGE_CUBIC)   // (a rejected cubic guess is followed by alpha0 beta^j, j >= 1)
    for (int j = 1; j < a.spec_trials && need && p.ls.stage == LS_STAGE_BACKTRACK; ++j) {
      need = ls_feed(p.ls, a.ls, a.phi[(size_t)j * a.batch + b], 0.0);
      last = j;
    }
  a.spec_sel[b] = need ? 0 : last;          // the trajectory of the last trial fed is the one the search ends on
  if (!need && last > 0) a.spec_refresh[b] = 1;
  if (need) {
    a.alpha[b] = p.ls.alpha;
    a.active[b] = 1;
  } else {
    p.evaluating = 0;
    a.active[b] = 0;
    ilqr_ls_end_logic(p);
  }
  return need;
}

// end of one sweep (solver.cpp:459-502); `active` := still running; returns that
__device__ __forceinline__ bool ilqr_finish_iter_body(const IlqrLoopArgs& a, int b) {
  IlqrProb& p = a.prob[b];
  if (p.running) ilqr_finish_iter_logic(p, a, a.iter);
  else p.dual = 0;
  a.active[b] = p.running;
  return p.running != 0;
}

// regularisation retry after a backward pass (extension); `active` := repeat the backward pass; returns that
__device__ __forceinline__ bool ilqr_reg_retry_body(const IlqrLoopArgs& a, int b) {
  IlqrProb& p = a.prob[b];
  const bool was_active = a.active[b] != 0;
  bool again = false;
  if (p.running && was_active) again = ilqr_reg_retry_logic(p, a, a.reg[b], a.bwd_status[b]);
  a.active[b] = again ? 1 : 0;
  return again;
}

// PenaltyUpdate; `active` := problems whose cost gradients have to be recomputed
__device__ __forceinline__ void ilqr_penalty_update_body(const IlqrLoopArgs& a, int b) {
  a.active[b] = ilqr_penalty_update_logic(a.prob[b], a) ? 1 : 0;
}

}  // namespace altro_hip
ALTRO_FP_REGION_END   // back to the including translation unit's own mode (fp_contract.h)
