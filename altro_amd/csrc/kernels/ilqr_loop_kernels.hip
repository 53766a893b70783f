// ilqr_loop_kernels.hip -- per-problem bookkeeping kernels of the batched AL-iLQR loop (one thread per problem):
// the resumable line search, convergence tests, outer (dual / penalty) updates, regularisation retry.
// Non-template kernels: included by exactly one translation unit (ilqr_launch_f64.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ilqr_loop_logic.h"
#include "ilqr_types.h"

namespace altro_hip {

// ---- batched line search + sweep bookkeeping (one thread per problem) ---------------------------------
__global__ void ilqr_loop_init_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  IlqrProb& p = a.prob[b];
  ilqr_prob_init(p);
  a.reg[b] = a.reg_initial;
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
  bool need = ilqr_ls_begin_logic(p, a.ls, a.tol_meritfun_gradient, a.phi[b], a.dphi[b]);
  // Fused first trial (IlqrLoopArgs::spec_pre): the merit launch that produced phi(0) also evaluated the first step
  // alpha0 = 1 the search asks for (into phi / dphi row 1 and spare candidate 0), so it is consumed right here.
  if (need && a.spec_pre) {
    need = ls_feed(p.ls, a.ls, a.phi[(size_t)a.batch + b], a.dphi[(size_t)a.batch + b]);
    if (!need) {   // the search ended on that step (same bookkeeping as ilqr_ls_feed_kernel)
      ilqr_ls_end_logic(p);
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
    ilqr_ls_end_logic(p);
  }
}

// end of one sweep (solver.cpp:459-502): convergence test, bookkeeping; `active` := still running
__global__ void ilqr_finish_iter_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  IlqrProb& p = a.prob[b];
  if (p.running) ilqr_finish_iter_logic(p, a, a.iter);
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
  if (p.running && was_active) again = ilqr_reg_retry_logic(p, a, a.reg[b], a.bwd_status[b]) ? 1 : 0;
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
  a.active[b] = ilqr_penalty_update_logic(p, a) ? 1 : 0;
}

// set `active` := running (used before the per-sweep kernels)
__global__ void ilqr_mark_running_kernel(IlqrLoopArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  a.active[b] = a.prob[b].running;
  a.alpha[b] = 0.0;
}

}  // namespace altro_hip
