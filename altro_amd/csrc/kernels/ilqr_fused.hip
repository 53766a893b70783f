// ilqr_fused.hip -- plan LANE: whole sweeps of SolverImpl::Solve (solver.cpp:447-502) in ONE kernel launch.
//
// On plan LANE a problem lives in one lane from the first knot point to the last, and no kernel of the loop ever reads
// another problem's data: the launch boundaries of the launch-sequenced solve (capi_ilqr.hip) carry no dependency, only
// cost -- ~13 launches and two or three host read-backs per sweep, a third of the wall time of a small solve
// (profiles/r02a_c2_rocprofv3.txt: 2.95 ms of kernels in a 4.5 ms solve).  Here a wave runs its 64 problems through
//     [AL Hessians] -> backward sweep (+ regularisation retries) -> merit(0) -> line search (a merit pass per trial step)
//     -> stationarity / feasibility -> accept -> convergence test -> [dual / penalty update, gradient refresh]
// sweep after sweep with no launch and no host in between, and retires as soon as its own problems have stopped (a
// straggler holds its own wave, not the batch).  The arithmetic is the SAME device functions the sequenced path's kernels
// call (ilqr_lane.hip, tvlqr_lane_body.inc, ilqr_loop_logic.h, linesearch_sm.h), in the order the sequenced path runs them
// without speculation, so every result is bit-identical to it (tests/test_gpu_fused.py); the per-(problem, knot point)
// kernels become loops over k inside the lane.  What this path does not have is the speculative evaluation of several
// backtracking steps per launch: the host hands problems that are still running after `max_sweeps` sweeps back to the
// sequenced loop, which has it (capi_ilqr.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ilqr_lane.hip"
#include "ilqr_loop_logic.h"

namespace altro_hip {

template <int KIND, int n, int m, typename T>
__global__ __launch_bounds__(64) void ilqr_fused_sweeps_kernel(IlqrArgs<T> a, IlqrLoopArgs la, LaneArgs<T> ba, IlqrFusedArgs fa) {
  // the same wave -> problems mapping as the sweep kernels (XCD-aware, tvlqr_lane_body.inc)
  const int nwv = (a.batch + 63) / 64, chk = (nwv + 7) / 8;
  const int wv = (int)((blockIdx.x & 7) * chk + (blockIdx.x >> 3));
  if (wv >= nwv) return;
  const int tid = threadIdx.x;
  const int64_t b0 = (int64_t)wv * 64;
  const int64_t b = b0 + tid;
  const int64_t B = a.batch;
  const bool valid = b < B;
  const int N = a.N;
  const uint32_t lane = (uint32_t)tid * (uint32_t)sizeof(T);
  const uint32_t rowB = (uint32_t)B * (uint32_t)sizeof(T);
  const bool al = a.al.enabled != 0;
  IlqrProb* pp = a.prob + (valid ? b : 0);
  int sweeps = 0;

  for (int it = fa.first_iter; it < fa.first_iter + fa.max_sweeps; ++it) {
    const bool run = valid && pp->running != 0;
    if (__ballot(run) == 0ull) break;
    ++sweeps;
    // CalcExpansions: the cost Hessians change only through the constraints' terms (solver.cpp:448)
    if (al)
      for (int k = 0; k <= N; ++k)
        if (run) ilqr_expand_point<KIND, n, m, T>(a, b, k, false, true);
    // BackwardPass (reg = 0 in the reference, solver.cpp:363); extension: repeat with a growing per-problem reg
    {
      bool again = run;
      for (int attempt = 0; attempt <= fa.reg_retry_max; ++attempt) {
        if (__ballot(again) == 0ull) break;
        int status = -1;
        if (again) {
          const T reg = fa.use_reg ? (T)la.reg[b] : T(0);
          status = lane_backward_lane<n, m, T>(ba, b0, tid, reg);
        }
        if (attempt == fa.reg_retry_max) break;
        again = again && ilqr_reg_retry_logic(*pp, la, la.reg[b], status);
      }
    }
    // ForwardPass: phi(0) and phi'(0), then the line search, one merit pass per trial step (solver.cpp:237-271)
    bool need = false;
    {
      T phi = T(0), dphi = T(0);
      if (run) {
        ilqr_merit_lane<KIND, n, m, T>(a, b, b0, lane, rowB, T(0), true, true, a.cand, phi, dphi);
        need = ilqr_ls_begin_logic(*pp, la.ls, la.tol_meritfun_gradient, (double)phi, (double)dphi);
        pp->evaluating = need ? 1 : 0;
      }
      while (__ballot(need) != 0ull) {
        if (need) {
          ilqr_merit_lane<KIND, n, m, T>(a, b, b0, lane, rowB, (T)pp->ls.alpha, true, true, a.cand, phi, dphi);
          need = ls_feed(pp->ls, la.ls, (double)phi, (double)dphi);
          if (!need) { pp->evaluating = 0; ilqr_ls_end_logic(*pp); }
        }
      }
    }
    // convergence criteria on the accepted candidate, then make it the nominal (solver.cpp:459-469)
    {
      T stat = T(0), feas = T(0);
      // (reads only: unrolled so that the operand loads of four knot points are in flight at once -- one lane walks its
      //  own problem's knot points here, where the stand-alone kernel has a thread per (problem, knot point))
#pragma unroll 4
      for (int k = 0; k <= N; ++k)
        if (run) {
          T r, v;
          ilqr_stationarity_point<n, m, T>(a, b, k, r, v);
          stat = fmax(stat, r);
          feas = fmax(feas, v);
        }
#pragma unroll 4
      for (int k = 0; k <= N; ++k)
        if (run) ilqr_accept_point<n, m, T>(a, b, k);
      if (run) {
        pp->stationarity = (double)stat;
        pp->feasibility = (double)feas;
        ilqr_finish_iter_logic(*pp, la, it);
      }
    }
    // DualUpdate, PenaltyUpdate, refreshed cost gradients for the problems that asked (solver.cpp:470-489)
    if (al) {
      const bool dual = run && pp->dual != 0;
      if (__ballot(dual) != 0ull) {
        for (int k = 0; k <= N; ++k)
          if (dual) ilqr_dual_point<n, m, T>(a, b, k);
        if (dual) (void)ilqr_penalty_update_logic(*pp, la);
        for (int k = 0; k <= N; ++k)
          if (dual) ilqr_expand_point<KIND, n, m, T>(a, b, k, true, false);
      }
    }
  }
  // hand-back: how many problems the launch leaves running, how many sweeps its slowest wave took
  const bool still = valid && pp->running != 0;
  const unsigned long long m64 = __ballot(still);
  if (tid == 0) {
    if (m64) atomicAdd(&fa.counters[1], __popcll(m64));
    atomicMax(&fa.counters[3], sweeps);
  }
}

}  // namespace altro_hip
