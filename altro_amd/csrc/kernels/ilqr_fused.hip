// ilqr_fused.hip -- plan LANE: whole sweeps of SolverImpl::Solve (solver.cpp:447-502) in ONE kernel launch.
//
// On plan LANE no kernel of the loop ever reads another problem's data, so the launch boundaries of the
// launch-sequenced solve (capi_ilqr.hip) carry no dependency, only cost: ~20 launches and two or three host read-backs
// per sweep.  Here a WORKGROUP of four wavefronts owns G = 8, 16 or 32 problems (as few as keeps every workgroup
// resident at once, one per CU; lane = (problem slot, knot-point slot), 64 / G knot points per wave at once) and runs
// the very sequence the host loop enqueues, with `__syncthreads()` where the host has a launch boundary and
// `__syncthreads_count()` where it reads a counter back:
//   * the serial chains run in one wave per chain: the backward sweep (for (4, 2) four lanes per problem, 16 problems
//     per wave: tvlqr_quad_body.inc), and of the merit evaluation (three phases, ilqr_lane.hip) the rollouts and the
//     sums -- one wave per line-search step evaluated, exactly the speculative launches of the host loop (gridDim.y
//     there, the wave index here): the first step alpha0 = 1 next to phi(0), alpha beta^0..3 in the backtracking stage;
//   * the per-knot-point terms of a merit evaluation follow the rollouts as a wave front (progress counters in LDS),
//     drawn chunk by chunk by whichever wave is free;
//   * what else is independent in k -- expansions, stationarity / feasibility, accept, dual updates, the copy of a
//     spare candidate -- is dealt round-robin over the four waves, the same point functions the (problem, knot
//     point)-parallel kernels call.
// Every function called is the body of a kernel of the sequenced path (ilqr_lane.hip, tvlqr_lane_body.inc,
// tvlqr_quad_body.inc, ilqr_loop_logic.h), on the same per-batch arrays, in the same order: results are bit-identical
// to it with and without its speculation (tests/test_gpu_fused.py).  A workgroup retires as soon as its problems have
// stopped: a straggler holds its own four waves, not the batch.  All waves of a workgroup sit on one CU and share its
// vector L1, which is what makes a `__syncthreads()` (or a workgroup-scope release / acquire pair) enough between a
// store of one wave and a load of another.
#pragma once
#include "../fp_contract.h"
#include <hip/hip_runtime.h>

#include "ilqr_lane.hip"
#include "ilqr_loop_logic.h"

namespace altro_hip {

// (Calling the heavy phases instead of inlining them was measured and is far worse -- C2 6.4 ms against 3.3 ms: the
//  by-reference argument blocks go through scratch.  What keeps the register file in check instead is that no phase is
//  monolithic any more: the merit evaluation is the three-phase one of ilqr_lane.hip, the (4, 2) backward sweep the
//  four-lanes-per-problem one of tvlqr_quad_body.inc.)
// Waves per workgroup: four = one per SIMD, each with the whole register file; eight (two per SIMD, 256 registers each)
// where hipcc 7.2 can build it -- the 2-state shapes.  For the 4- and 6-state shapes it emits an illegal spill reload
// ("requires even aligned vector registers") as soon as the kernel is held to 256 registers.
template <int n, typename T>
constexpr int ilqr_fused_waves() { return n <= 2 ? 8 : 4; }

// (The kernel body holds no floating-point arithmetic of its own -- every expression lives in the shared device functions,
//  which carry their own contract(on) regions -- but it is put under the same mode explicitly so that an expression added
//  here later cannot round differently from the launch-sequenced path by accident.)
ALTRO_FP_REGION_ON
template <int KIND, int n, int m, typename T, int G>
__global__ __launch_bounds__((64 * ilqr_fused_waves<n, T>())) void ilqr_fused_sweeps_kernel(IlqrArgs<T> a, IlqrLoopArgs la,
                                                                                           LaneArgs<T> ba, IlqrFusedArgs fa) {
  static_assert(G == 64 || G == 32 || G == 16 || G == 8, "problems per workgroup");
  constexpr int W = ilqr_fused_waves<n, T>();
  constexpr int KS = 64 / G;                  // knot points a wave takes at once in the (problem, knot point)-parallel steps
  constexpr bool QUAD = (n == 4 && m == 2) || (n == 2 && m == 1);   // four lanes per problem: tvlqr_quad_body.inc, tvlqr_quad2_body.inc
  // workgroup -> problems: contiguous runs per XCD like the sweep kernels' waves (tvlqr_lane_body.inc)
  // A LISTED launch (fa.list: the problems still running after an earlier launch, capi_solve.hip) serves slot s = problem list[s]
  // (-1: an empty slot; a workgroup's first slot is never empty).  Every address below is `base + lane offset`, and a listed
  // launch takes base = 0, lane offset = the problem's own index -- the same functions on the same per-problem data, so nothing
  // a problem computes depends on which slot it rides in.
  const bool listed = fa.list != nullptr;
  const int slots = listed ? fa.list_count : a.batch;
  const int nwg = (slots + G - 1) / G, chk = (nwg + 7) / 8;
  const int wg = (int)((blockIdx.x & 7) * chk + (blockIdx.x >> 3));
  if (wg >= nwg) return;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), t = threadIdx.x & 63;   // (w: provably wave-uniform)
  const int pt = t % G, ks = t / G;           // lane = (problem slot, knot-point slot)
  const int64_t s0 = (int64_t)wg * G;         // the workgroup's first slot
  const bool in_range = s0 + pt < slots;
  const int listed_b = listed ? fa.list[in_range ? s0 + pt : s0] : 0;
  const bool valid = in_range && listed_b >= 0;
  const int64_t b0 = listed ? 0 : s0;         // base problem of the workgroup's addresses
  // (lanes past the batch / on an empty slot never dereference their problem: they carry the workgroup's first one)
  const int64_t b = listed ? (int64_t)(valid ? listed_b : fa.list[s0]) : (valid ? s0 + pt : s0);
  const int pl = (int)(b - b0);               // this lane's problem relative to the base
  const int64_t B = a.batch;
  const int bi = (int)b;
  const int N = a.N;
  const uint32_t lane = (uint32_t)pl * (uint32_t)sizeof(T);
  const uint32_t rowB = (uint32_t)B * (uint32_t)sizeof(T);
  const bool al = a.al.enabled != 0;
  const bool serial = ks == 0 && valid;           // the lanes of a wave that run a lane-per-problem chain
  const bool lead = w == 0 && serial;             // wave 0 keeps the per-problem books
  int sweeps = 0;
  bool published = false;                         // (lead lanes) this problem's record has gone out to the host (fa.poll)
  // optional phase clock (ALTRO_HIP_FUSED_CLOCK, a tuning aid): 100 MHz ticks per phase and workgroup
  unsigned long long tick = fa.clk ? wall_clock64() : 0ull;
  auto lap = [&](int phase) {
    if (fa.clk && threadIdx.x == 0) {
      const unsigned long long now = wall_clock64();
      fa.clk[(int64_t)wg * ILQR_FUSED_PHASES + phase] += now - tick;
      tick = now;
    }
  };
  __shared__ int s_again[G];                      // regularisation retry: which problems repeat their backward sweep
  // The (problem, knot point)-parallel steps run on the COMPACTED list of the workgroup's running problems: with na of them
  // left, a wave takes 64 / Gd knot points at once, Gd = the power of two >= na.  The tail of a batched solve -- a few
  // stragglers per workgroup for most of the sweeps -- gets the whole wave per problem there (memory accesses are then one
  // cache line per lane: irrelevant for a handful of problems).  The serial chains keep the fixed lane <-> problem mapping.
  __shared__ int s_list[G];
  __shared__ int s_prob[G];                       // slot -> problem
  __shared__ int s_na;
  if (w == 0 && ks == 0) s_prob[pt] = valid ? bi : -1;
  int KSd = KS, ksd = ks, bik = bi;
  int64_t bk = b;
  bool validk = valid;

  // W * KS knot points at once in every (problem, knot point)-parallel step
#define FUSED_FOR_K(MASKED, CALL)                                  \
  for (int k = w * KSd + ksd; k <= N; k += W * KSd)                \
    if (validk && (MASKED)) { CALL; }

  // BackwardPass of the workgroup's problems: lane t of wave 0 <-> problem b0 + t, or -- (4, 2) -- four lanes per problem,
  // 16 problems in each of the waves 0..3.  `pick`: which problems (all the active ones, or the retrying ones).
  auto backward = [&](bool retry) {
    if constexpr (QUAD) {
      if (w < (G + 15) / 16) {
        const int q = 16 * w + (t >> 2);
        const int64_t bq = q < G ? s_prob[q] : -1;
        if (bq >= 0) {
          if (retry ? s_again[q] != 0 : la.active[bq] != 0) {
            const T rg = (retry || fa.use_reg) ? (T)la.reg[bq] : T(0);
            // (unlisted: the wave's base is its first problem, lane offsets 0 .. 15 as in quad_backward_kernel)
            const int64_t bw = listed ? 0 : s0 + 16 * w;
            if constexpr (n == 4) (void)quad_backward_quad<m, T>(ba, bw, (int)(bq - bw), t & 3, rg);
            else (void)quad2_backward_quad<T>(ba, bw, (int)(bq - bw), t & 3, rg);
          }
        }
      }
    } else {
      if (lead && (retry ? s_again[pt] != 0 : la.active[bi] != 0))
        (void)lane_backward_lane<n, m, T>(ba, b0, pl, (retry || fa.use_reg) ? (T)la.reg[bi] : T(0));
    }
  };
  // One merit launch of the sequenced loop (ilqr_launch_kernel, IK_MERIT) inside the workgroup: the rollouts of the
  // `trials` steps in one wave each; their per-knot-point terms follow BEHIND the rollouts as a wave front -- the rolling
  // waves publish how far they are (s_prog, every other ring group), the other waves, and the rolling ones once they
  // are through, draw chunks of KS (knot point, trial) pairs in k order (s_next) and evaluate them as soon as every
  // trial has passed the chunk's last knot point --; then the sums in one wave each.
  __shared__ int s_prog[4];                      // knot points rolled out so far, per trial
  __shared__ int s_next;                         // next chunk of per-knot-point work
  struct Publish {
    int* slot;
    __device__ __forceinline__ void operator()(int count) const {
      // release at workgroup scope: the stores of the knot points counted are visible to the workgroup before the counter is
      __hip_atomic_store(slot, count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  auto merit = [&](const IlqrArgs<T>& am, int trials, int ph) {
    if (threadIdx.x < 4) s_prog[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_next = 0;
    __syncthreads();
    if (w < trials) {
      if (serial && la.active[bi]) {
        const MeritTrial<T> tr = ilqr_merit_trial<T>(am, b, w);
        if (tr.run) ilqr_merit_roll_lane<KIND, n, m, T, Publish>(am, tr, b0, lane, rowB, Publish{&s_prog[w]});
      }
      Publish{&s_prog[w]}(N + 1);               // (also when no problem of this workgroup takes this step)
    }
    lap(ph);
    const int total = trials * (N + 1), nchunks = (total + KSd - 1) / KSd;
    for (;;) {
      int c = 0;
      if (t == 0) c = atomicAdd(&s_next, 1);
      c = __builtin_amdgcn_readfirstlane(c);
      if (c >= nchunks) break;
      const int last_idx = (c + 1) * KSd - 1 < total ? (c + 1) * KSd - 1 : total - 1;
      const int kmax = last_idx / trials;
      for (int tt = 0; tt < trials; ++tt)
        while (__hip_atomic_load(&s_prog[tt], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= kmax) __builtin_amdgcn_s_sleep(2);
      const int idx = c * KSd + ksd;
      if (idx < total && validk && la.active[bik]) {
        const int k = idx / trials, trial = idx - k * trials;
        const MeritTrial<T> tr = ilqr_merit_trial<T>(am, bk, trial);
        if (tr.run) ilqr_merit_point<KIND, n, m, T>(am, bk, k, trial, tr);
      }
    }
    __syncthreads();
    lap(ph + 1);
    if (w < trials && serial && la.active[bi]) {
      const MeritTrial<T> tr = ilqr_merit_trial<T>(am, b, w);
      if (tr.run) ilqr_merit_sum_body<n, m, T>(am, tr, b, b0, lane, rowB, w);
    }
    __syncthreads();
    lap(ph + 2);
  };

  if constexpr (n > 2) if (fa.prologue) {   // the head of Solve (solver.cpp:420-434), in the order the host loop launches it
                                             // (compiled for the 4-state shapes only: see capi_ilqr.hip)
    if (lead) {                                   // ilqr_loop_init_kernel
      ilqr_prob_init(la.prob[bi]);
      la.reg[bi] = la.reg_initial;
      la.active[bi] = 1;
      la.alpha[bi] = 0.0;
    }
    __syncthreads();
    if (w == 0 && serial) ilqr_rollout_lane<KIND, n, m, T>(a, b);          // initial rollout on the candidate trajectory
    __syncthreads();
    FUSED_FOR_K(true, (ilqr_accept_point<n, m, T>(a, bk, k)));             // ... which becomes the nominal one
    __syncthreads();
    // without constraints the cost Hessian is constant and is written once, here; with them the gradient is formed with
    // the penalty the constraints carry so far and SetPenalty comes after it (solver.cpp:424-430)
    FUSED_FOR_K(true, (ilqr_expand_point<KIND, n, m, T>(a, bk, k, true, !al)));
    __syncthreads();
    if (al && lead) la.prob[bi].rho = la.penalty_initial;                   // ilqr_set_penalty_kernel
    __syncthreads();
  }
  for (int it = fa.first_iter; it < fa.first_iter + fa.max_sweeps; ++it) {
    la.iter = it;
    if (__syncthreads_count(lead && la.prob[bi].running != 0) == 0) break;
    ++sweeps;
    if (lead) ilqr_mark_running_body(la, bi);
    __syncthreads();
    {   // the running problems of this workgroup, compacted (a problem that stops during the sweep stays listed: masks apply)
      const bool act = lead && la.active[bi] != 0;
      const unsigned long long mask = __ballot(act);
      if (act) s_list[__popcll(mask & ((1ull << t) - 1ull))] = pt;
      if (threadIdx.x == 0) s_na = __popcll(mask);
      __syncthreads();
      const int na = s_na;
      int Gd = 1;
      while (Gd < na) Gd <<= 1;
      KSd = 64 / Gd;
      const int slot = t & (Gd - 1);
      ksd = t >> __builtin_ctz(Gd);
      validk = slot < na;
      bik = s_prob[validk ? s_list[slot] : 0];
      bk = bik;
    }
    // CalcExpansions: the cost Hessians change only through the constraints' terms (solver.cpp:448)
    if (al) {
      FUSED_FOR_K(la.active[bik], (ilqr_expand_point<KIND, n, m, T>(a, bk, k, false, true)));
      __syncthreads();
    }
    lap(0);
    // BackwardPass (reg = 0 in the reference, solver.cpp:363); extension: repeat with a growing per-problem reg
    backward(false);
    for (int attempt = 0; attempt < fa.reg_retry_max; ++attempt) {
      __syncthreads();
      const bool again = lead && ilqr_reg_retry_body(la, bi);
      if (w == 0 && ks == 0) s_again[pt] = again ? 1 : 0;
      if (__syncthreads_count(again) == 0) break;
      backward(true);
    }
    if (fa.reg_retry_max > 0) {
      __syncthreads();
      if (lead) ilqr_mark_running_body(la, bi);
    }
    __syncthreads();
    lap(1);
    // ForwardPass: phi(0) and, next to it, the first step the search will ask for (alpha0 = 1)
    {
      IlqrArgs<T> am = a;
      // (roles swapped, IlqrArgs::spec_flip: the alpha0 = 1 pass writes the candidate and the expansion, phi(0) -- whose
      //  trajectory is the nominal one and whose expansion is the one already in place -- goes to spare candidate 0)
      am.alpha = la.alpha; am.active = la.active; am.want_derivative = 1; am.spec_trials = 2; am.spec_pre = 1; am.spec_flip = 1;
      merit(am, 2, 2);
    }
    la.spec_pre = 1;
    bool need0 = false;
    if (lead) {
      const bool was_running = la.prob[bi].running != 0;
      need0 = ilqr_ls_begin_body(la, bi);
      // what the swap changes afterwards: a search that ended ON the first step has its trajectory and expansion in place
      // already; one that ended WITHOUT it (phi' too small, not a descent direction: alpha = 0) takes the phi(0)
      // trajectory back from the spare and has its expansion redone
      if (la.spec_sel[bi] == 1) { la.spec_sel[bi] = 0; la.spec_refresh[bi] = 0; }
      else if (was_running && !need0) { la.spec_sel[bi] = 1; la.spec_refresh[bi] = 1; }
    }
    int searching = __syncthreads_count(need0);
    la.spec_pre = 0;
    FUSED_FOR_K(true, (ilqr_spec_select_point<n, m, T>(a, bk, k)));
    lap(5);
    // the line search: one round = one merit evaluation per searching problem; in the backtracking stage the next four
    // steps of the (known) sequence alpha beta^j at once
    while (searching > 0) {
      __syncthreads();
      const int trials = la.ls.use_backtracking ? 4 : 1;
      {
        IlqrArgs<T> am = a;
        am.alpha = la.alpha; am.active = la.active; am.want_derivative = 1; am.spec_trials = trials; am.spec_pre = 0;
        merit(am, trials, 6);
      }
      la.spec_trials = trials;
      searching = __syncthreads_count(lead && ilqr_ls_feed_body(la, bi));
      la.spec_trials = 1;
      if (trials > 1) FUSED_FOR_K(true, (ilqr_spec_select_point<n, m, T>(a, bk, k)));
    }
    __syncthreads();
    // steps accepted from a speculative trial carry no phi' pass: redo their expansion (what the derivative pass of a
    // sequential trial would have left behind)
    lap(5);
    FUSED_FOR_K(la.spec_refresh[bik], (ilqr_expand_point<KIND, n, m, T>(a, bk, k, true, false)));
    __syncthreads();
    lap(9);
    // convergence criteria on the accepted candidate, then make it the nominal (solver.cpp:459-469)
    if (lead) {
      ilqr_mark_running_body(la, bi);
      if (la.prob[bi].running) { la.prob[bi].stationarity = 0.0; la.prob[bi].feasibility = 0.0; }
    }
    __syncthreads();
    {   // (the maxima over this lane's knot points first, then ONE atomic per lane: 101 atomics on one word serialise)
      // (maxima of the bit patterns, exactly what a sequence of atomicMax calls on the same word computes, NaNs included)
      unsigned long long res_max = 0ull, viol_max = 0ull;
      bool any = false;
      for (int k = w * KSd + ksd; k <= N; k += W * KSd)
        if (validk && la.active[bik]) {
          T res, viol;
          ilqr_stationarity_point<n, m, T>(a, bk, k, res, viol);
          const unsigned long long rb = (unsigned long long)__double_as_longlong((double)res);
          const unsigned long long vb = (unsigned long long)__double_as_longlong((double)viol);
          res_max = rb > res_max ? rb : res_max;
          viol_max = vb > viol_max ? vb : viol_max;
          any = true;
          ilqr_accept_point<n, m, T>(a, bk, k);
        }
      if (any) {
        atomicMax(reinterpret_cast<unsigned long long*>(&a.prob[bk].stationarity), res_max);
        if (al) atomicMax(reinterpret_cast<unsigned long long*>(&a.prob[bk].feasibility), viol_max);
      }
    }
    __syncthreads();
    lap(10);
    if (lead) (void)ilqr_finish_iter_body(la, bi);
    __syncthreads();
    // DualUpdate, PenaltyUpdate, refreshed cost gradients for the problems that asked (solver.cpp:470-489)
    if (al) {
      FUSED_FOR_K(la.prob[bik].dual, (ilqr_dual_point<n, m, T>(a, bk, k)));
      __syncthreads();
      if (lead) ilqr_penalty_update_body(la, bi);
      __syncthreads();
      FUSED_FOR_K(la.active[bik], (ilqr_expand_point<KIND, n, m, T>(a, bk, k, true, false)));
      __syncthreads();
    }
    lap(11);
    // altro_hip_ilqr_solve_async: a problem that stopped in this sweep goes out NOW -- its results and the first input of its
    // solution into the caller's pinned record, then the flag, then the count -- while the rest of the batch keeps iterating
    if (fa.poll && lead && !published && !la.prob[bi].running) {
      IlqrPollRec* rec = fa.poll + bi;
      rec->result = ilqr_result_of(la.prob[bi]);
#pragma unroll
      for (int e = 0; e < 4; ++e) rec->u0[e] = e < m ? (double)a.nom[((int64_t)n + e) * B + b] : 0.0;
      __threadfence_system();
      __hip_atomic_store(&rec->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_fetch_add(fa.poll_count, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      published = true;
    }
  }
#undef FUSED_FOR_K
  // hand-back: how many problems the launch leaves running, how many sweeps its slowest workgroup took
  const bool runs_on = lead && la.prob[bi].running != 0;
  if (fa.run_flags && w == 0 && ks == 0 && in_range) fa.run_flags[s0 + pt] = runs_on ? 1 : 0;
  const int still = __syncthreads_count(runs_on);
  if (threadIdx.x == 0) {
    if (still) atomicAdd(&fa.counters[1], still);
    atomicMax(&fa.counters[3], sweeps);
  }
}

ALTRO_FP_REGION_END

}  // namespace altro_hip
