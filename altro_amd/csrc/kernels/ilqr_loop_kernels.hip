// ilqr_loop_kernels.hip -- per-problem bookkeeping kernels of the batched AL-iLQR loop (one thread per problem):
// the resumable line search, convergence tests, outer (dual / penalty) updates, regularisation retry.
// Non-template kernels: included by exactly one translation unit (ilqr_launch_f64.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ilqr_types.h"

namespace altro_hip {

// ---- batched line search + sweep bookkeeping (one thread per problem) ---------------------------------
__global__ void ilqr_loop_init_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  IlqrProb& p = a.prob[b];
  p.running = 1; p.iterations = 0; p.status = 1; p.ls_failed = 0; p.evaluating = 0;
  p.alpha = 0.0; p.stationarity = 0.0; p.ls_iters = 0;
  p.feasibility = 0.0; p.dual = 0; p.n_dual_updates = 0; p.reg_retries = 0;
  a.reg[b] = a.reg_initial;
  p.rho_est = p.rho;   // the initial gradient is formed with the penalty left by Initialize / the last solve
  a.active[b] = 1;
  a.alpha[b] = 0.0;
}

// after merit(alpha = 0): ForwardPass's head (solver.cpp:241-249)
__global__ void ilqr_ls_begin_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  IlqrProb& p = a.prob[b];
  a.spec_sel[b] = 0;
  a.spec_refresh[b] = 0;
  if (!p.running) { a.active[b] = 0; return; }
  p.phi0 = a.phi[b];
  p.dphi0 = a.dphi[b];
  p.ls_failed = 0;
  bool need;
  if (fabs(p.dphi0) < a.tol_meritfun_gradient) {   // MeritFunctionGradientTooSmall: alpha = 0
    p.alpha = 0.0;
    p.ls_iters = 0;
    need = false;
  } else {
    need = ls_begin(p.ls, a.ls, 1.0, p.phi0, p.dphi0);
    if (!need) {   // not a descent direction
      p.alpha = p.ls.alpha;
      p.ls_iters = p.ls.n_iters;
      p.ls_failed = 1;
    }
  }
  // Fused first trial (IlqrLoopArgs::spec_pre): the merit launch that produced phi(0) also evaluated the first step
  // alpha0 = 1 the search asks for (into phi / dphi row 1 and spare candidate 0), so it is consumed right here.
  if (need && a.spec_pre) {
    need = ls_feed(p.ls, a.ls, a.phi[(size_t)a.batch + b], a.dphi[(size_t)a.batch + b]);
    if (!need) {   // the search ended on that step (same bookkeeping as ilqr_ls_feed_kernel)
      p.alpha = p.ls.alpha;
      p.ls_iters = p.ls.n_iters;
      const int st = p.ls.status;
      p.ls_failed = (isnan(p.alpha) || !(st == LS_MINIMUM_FOUND || st == LS_HIT_MAX_STEPSIZE)) ? 1 : 0;
      a.spec_sel[b] = 1;
      a.spec_refresh[b] = 1;
    }
  }
  p.evaluating = need ? 1 : 0;
  a.active[b] = need ? 1 : 0;
  if (need) {
    a.alpha[b] = p.ls.alpha;
    atomicAdd(&a.counters[0], 1);
  }
}

// after merit(alpha[b]): advance every searching problem's state machine
__global__ void ilqr_ls_feed_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  IlqrProb& p = a.prob[b];
  if (!p.running || !p.evaluating) { a.active[b] = 0; a.spec_sel[b] = 0; return; }
  // Speculative backtracking: the merit launch also evaluated alpha beta^j, j = 1 .. spec_trials - 1, for the problems
  // that were in the backtracking stage or about to enter it (cubic first guess pending).  Feeding them in order reproduces the sequential search exactly; "need" after
  // trial j - 1 is precisely the condition under which trial j exists (bt_iter below max_iters).
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
    atomicAdd(&a.counters[0], 1);
  } else {
    p.evaluating = 0;
    a.active[b] = 0;
    p.alpha = p.ls.alpha;
    p.ls_iters = p.ls.n_iters;
    const int st = p.ls.status;
    // solver.cpp:264-268
    p.ls_failed = (isnan(p.alpha) || !(st == LS_MINIMUM_FOUND || st == LS_HIT_MAX_STEPSIZE)) ? 1 : 0;
  }
}

// end of one sweep (solver.cpp:459-502): convergence test, bookkeeping; `active` := still running
__global__ void ilqr_finish_iter_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  IlqrProb& p = a.prob[b];
  if (p.running) {
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
    p.iterations = a.iter + 1;
    if (!stop && a.iter + 1 >= a.iterations_max) {
      p.status = 2;   // MaxIterations; the reference reports iter + 1 after its loop ends (solver.cpp:503-506)
      p.iterations = a.iter + 2;
      stop = true;
    }
    if (stop) p.running = 0;
  }
  else p.dual = 0;
  a.active[b] = p.running;
  if (p.running) atomicAdd(&a.counters[1], 1);
}

// Regularisation retry -- an EXTENSION: the reference passes reg = 0 and ignores a failed factorisation
// (tvlqr.cpp:159-164, solver.cpp:363, :449).  After a backward pass, every running problem whose Cholesky
// failed gets reg <- max(reg * scale, reg_min) and is marked for another backward pass (counters[2] counts
// them); a problem that succeeded relaxes reg <- max(reg / scale, reg_initial) for its next sweep.
__global__ void ilqr_reg_retry_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  IlqrProb& p = a.prob[b];
  const bool was_active = a.active[b] != 0;
  int again = 0;
  if (p.running && was_active) {
    if (a.bwd_status[b] != -1) {
      const double r = fmax(a.reg[b] * a.reg_scale, a.reg_min);
      if (r <= a.reg_max) { a.reg[b] = r; again = 1; ++p.reg_retries; }
    } else {
      a.reg[b] = fmax(a.reg[b] / a.reg_scale, a.reg_initial);
    }
  }
  a.active[b] = again;
  if (again) atomicAdd(&a.counters[2], 1);
}

// SetPenalty (solver.cpp:429) after the initial gradient
__global__ void ilqr_set_penalty_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  a.prob[b].rho = a.penalty_initial;
}

// PenaltyUpdate (solver.cpp:396-409) after the duals of every knot point have been updated with the old
// penalty; then the projected duals are refreshed with the new one (solver.cpp:483-486).  `active` :=
// problems whose cost gradients have to be recomputed.
__global__ void ilqr_penalty_update_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  IlqrProb& p = a.prob[b];
  a.active[b] = p.dual != 0;
  if (p.dual == 2) p.rho = fmin(p.rho * a.penalty_scaling, a.penalty_max);
  if (p.dual) p.rho_est = p.rho;
}

// set `active` := running (used before the per-sweep kernels)
__global__ void ilqr_mark_running_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  a.active[b] = a.prob[b].running;
  a.alpha[b] = 0.0;
}

}  // namespace altro_hip
