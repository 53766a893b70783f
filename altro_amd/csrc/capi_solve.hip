// capi_solve.hip -- C ABI: altro_hip_ilqr_solve, SolverImpl::Solve (solver.cpp:414-511) for every problem of the batch at once.
// The host only sequences launches; every per-problem decision is taken on the device (kernels/ilqr_loop_kernels.hip).
//
// One solve is a SolveRun: its phases are the member functions below, in the order Solve itself has them
//   configure        options -> the loop kernels' arguments, validation                       (solver.cpp:414-419)
//   choose_path      plan LANE: whole solves in one launch (kernels/ilqr_fused.hip) or the launch-sequenced loop
//   prologue         initial rollout, accept, first expansion, SetPenalty                      (solver.cpp:420-434)
//   run_fused        the one-launch solve kernel (and its optional phase clock)
//   sweep_head       CalcExpansions' Hessians, BackwardPass, the regularisation retries         (solver.cpp:447-449, 360-378)
//   first_evaluation phi(0) + the line search's first step, ILK_LS_BEGIN                        (solver.cpp:237-249)
//   search_rounds    further line-search rounds, speculative steps                             (linesearch.cpp:37-412)
//   finish_sweep     expansion of steps taken from a speculative trial, stationarity, accept, convergence test   (solver.cpp:459-469)
//   outer_update     DualUpdate, PenaltyUpdate, refreshed gradients, the next sweep's Hessians  (solver.cpp:470-489)
//   sequenced_loop   the sweeps, with the host running AHEAD of the device's verdicts
// (Round 5: this was one 470-line function inside capi_ilqr.hip; the forms that lost their A/B comparisons -- the matrix-core and the
//  LDS form of the two-trial pass, the tile-free expansion's off switch -- went with the split.)
#include "capi_internal.h"

#include <cstring>

using namespace altro_hip;
using namespace altro_hip::capi;

namespace {

constexpr int kFusedFirstChunk = 5;   // sweeps of the one-launch solve before its still-running problems are listed (run_fused)

struct SolveRun {
  altro_hip_batch* h;
  altro_hip_solve_options o;
  IlqrLoopArgs la;
  bool async = false, al = false, reg_on = false;
  bool lane_plan = false, generic_plan = false;
  bool fused = false, fused_prologue = false;
  bool dual = false;              // plan MFMA16: phi(0) and the first step in one pass (IK_MERIT2)
  bool spec_all_on = false, spec_on = false, merit_rounds_dpp = true, run_ahead = true;
  bool guard_on = false;          // affine rounds with the decision guard (altro_hip_solve_options::decision_margin > 0)
  int guarded_total = 0;          // trials the guard sent back for a rollout evaluation
  size_t spare_each = 0;
  int trials_cap = 1;
  int64_t spec_capacity = 512;
  // counter slots: every counting launch gets a fresh, zeroed 8-int slot and publishes it to host-mapped memory
  int slot = 0;
  // progress
  int sweeps = 0, total_merit_launches = 0, running = 0, total_reg_retries = 0, iter0 = 0;
  bool hessians_ready = false;    // the previous sweep's last expansion left the cost Hessians of this one
  bool hessian_stored = false;
  int diag_mode = 0;
  int* active0 = nullptr;
  unsigned forms0 = 0;            // the handle's own forms (the solve's are OR-ed in for its duration)

  explicit SolveRun(altro_hip_batch* h_) : h(h_), active0(h_->i_active), forms0(h_->forms) {}
  ~SolveRun() {   // whatever path leaves the solve: no speculation state, mask or swapped pointer survives it
    h->bwd_active = nullptr; h->bwd_reg = nullptr; h->stat_skip = nullptr;
    h->spec_trials = 1; h->spec_pre = 0;
    h->aff_round = false; h->aff_enabled = false; h->aff_store = false;
    h->i_active = active0;
    h->forms = forms0;
  }

  int loop(int which) {
    if (ilqr_launch_loop(h->stream, which, la)) return fail(ALTRO_HIP_ERR_HIP, "iLQR loop kernel launch failed");
    return 0;
  }
  int reset_slots() {
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemsetAsync(h->i_counters, 0, (size_t)kCounterSlots * 8 * sizeof(int), h->stream));
    slot = 0;
    return 0;
  }
  // launch a counting loop kernel on a fresh slot; returns the slot (< 0: failed, the message is set)
  int counted(int which_kernel) {
    if (slot + 1 >= kCounterSlots && reset_slots()) return -1;
    ++slot;
    la.counters = h->i_counters + 8 * slot;
    la.counters_pub = h->cnt_host_dev + 8 * slot;
    if (ilqr_launch_loop(h->stream, which_kernel, la)) { (void)fail(ALTRO_HIP_ERR_HIP, "iLQR loop kernel launch failed"); return -1; }
    if (hipEventRecord(h->cnt_ev[slot & 15], h->stream) != hipSuccess) { (void)fail(ALTRO_HIP_ERR_HIP, "hipEventRecord failed"); return -1; }
    la.counters_pub = nullptr;
    return slot;
  }
  int verdict(int s, int idx, int* out) {   // wait for slot s's launch and read its counter idx
    HIP_TRY(hipEventSynchronize(h->cnt_ev[s & 15]));
    *out = ((volatile int*)h->cnt_host)[8 * s + idx];
    return 0;
  }
  int spec_units(int searching) const {   // wavefronts one merit launch keeps busy
    return lane_plan ? (h->batch + 63) / 64 : searching;   // LANE: the searching lanes are scattered over all waves
  }
  // wavefronts a round of `trials` steps per searching problem launches: a wave per (problem, trial) in the LDS form, a wave per
  // two problems and two trials in the DPP form (which keeps two waves per SIMD, not four: see the capacity)
  int64_t spec_waves(int searching, int trials) const {
    if (lane_plan || !merit_rounds_dpp) return (int64_t)spec_units(searching) * trials;
    return (int64_t)((searching + 1) / 2) * ((trials + 1) / 2);
  }

  int configure(const altro_hip_solve_options* opts);
  int ensure_counters();
  int choose_path();
  int prologue();
  int run_fused();
  int sweep_head(int iter);
  int first_evaluation(int* prev_slot, int* begin_slot, int* launches, bool* refreshed);
  int search_rounds(int* prev_slot, int rounds_last, int* rounds_out, int* launches, bool* refreshed);
  int finish_sweep(bool refreshed, int* fin_slot);
  int outer_update();
  int sequenced_loop();
};

int SolveRun::configure(const altro_hip_solve_options* opts) {
  if (opts) o = *opts;
  else altro_hip_default_solve_options(&o);
  h->forms = forms0 | o.forms;
  la.prob = h->i_prob; la.alpha = h->i_alpha; la.active = h->i_active; la.phi = h->i_phi; la.dphi = h->i_dphi;
  la.counters = h->i_counters; la.batch = h->batch; la.iter = 0; la.iterations_max = o.iterations_max;
  la.tol_stationarity = o.tol_stationarity; la.tol_meritfun_gradient = o.tol_meritfun_gradient;
  la.tol_primal_feasibility = o.tol_primal_feasibility;
  la.penalty_initial = o.penalty_initial; la.penalty_scaling = o.penalty_scaling; la.penalty_max = o.penalty_max;
  al = !h->al_defs.empty();
  la.al_enabled = al ? 1 : 0;
  la.reg = h->i_reg; la.bwd_status = h->status;
  la.spec_trials = 1; la.spec_pre = 0; la.spec_sel = h->i_spec_sel; la.spec_refresh = h->i_spec_refresh;
  la.reg_initial = o.reg_initial; la.reg_scale = o.reg_scale; la.reg_min = o.reg_min; la.reg_max = o.reg_max;
  reg_on = o.reg_retry_max > 0 || o.reg_initial > 0.0;
  if (o.reg_initial < 0.0 || (o.reg_retry_max > 0 && !(o.reg_scale > 1.0 && o.reg_min > 0.0 && o.reg_max >= o.reg_min)))
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "regularisation retry needs reg_initial >= 0, reg_scale > 1, 0 < reg_min <= reg_max");
  if (al && !(o.penalty_initial > 0.0 && o.penalty_scaling > 0.0 && o.penalty_max > 0.0))
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "penalty_initial, penalty_scaling and penalty_max must be positive");
  if (h->plan == ALTRO_HIP_PLAN_GENERIC && o.reg_retry_max > 0)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "the regularisation retry is built for plans LANE and MFMA16 (plan GENERIC: reg_retry_max = 0)");
  la.ls = ls_default_options();
  la.ls.try_cubic_first = 1;                                   // solver.cpp:248
  la.ls.use_backtracking = o.use_backtracking_linesearch;      // solver.cpp:417
  h->spec_beta = la.ls.beta_decrease; h->spec_max_iters = la.ls.max_iters;
  lane_plan = h->plan == ALTRO_HIP_PLAN_LANE;
  generic_plan = h->plan == ALTRO_HIP_PLAN_GENERIC;   // (constraint rows in their one-wave-per-knot-point form: no merged end pass)
  const int64_t cand_elems = (int64_t)h->batch * (h->N + 1) * (lane_plan ? lane_sizes(h->n, h->m).e_xuy : 28);
  spare_each = (size_t)cand_elems * h->esz;   // one spare candidate trajectory
  trials_cap = spec_trials_cap(h);
  // speculative backtracking: how much of the chip the searching problems occupy, and how much there is
  spec_all_on = !form(h, ALTRO_HIP_FORM_NO_SPECULATION) && !generic_plan;   // (plan GENERIC evaluates one step per launch)
  spec_on = o.use_backtracking_linesearch != 0 && spec_all_on;
  merit_rounds_dpp = !form(h, ALTRO_HIP_FORM_MERIT_LDS) || h->cost_dense || h->model_set;   // (a dense cost, a device model: row-layout kernels only)
  spec_capacity = lane_plan ? 512 : (merit_rounds_dpp ? 2048 : 4096);   // two waves per CU (LANE: latency-bound; more slow each other down) / four per SIMD (MFMA16)
  run_ahead = !form(h, ALTRO_HIP_FORM_NO_RUNAHEAD);
  // Plan MFMA16: phi(0) and the line search's first step from one pass over the records, the candidate's stationarity / feasibility
  // from that same pass.  ALTRO_HIP_FORM_NO_MERIT2 keeps the one-evaluation-per-launch sequence (the comparison the tests hold this against).
  // (round 6: also where kernels/ilqr_row32.hip serves the handle -- plan MFMA32's shapes: its merit kernel has a two-trial form)
  dual = (h->plan == ALTRO_HIP_PLAN_MFMA16 || row32_eligible(h) || row32_model_eligible(h)) && !form(h, ALTRO_HIP_FORM_NO_SPECULATION) && !form(h, ALTRO_HIP_FORM_NO_MERIT2);
  // Plan MFMA16, diagonal cost, bound-type blocks only: the Hessian blocks differ from sweep to sweep on their diagonal alone, so
  // after this solve's first (full) Hessian expansion the later ones store 16 values per knot point instead of 158 (EXPAND_DIAG)
  // (plan GENERIC / MFMA32, every block bound-type -- AlTable::gsel --: the constraints touch the Hessian blocks' diagonals only,
  //  whatever the cost's own blocks are: generic_expand_al_kernel then rewrites n + m entries per knot point instead of (n + m)^2)
  diag_mode = ((h->plan == ALTRO_HIP_PLAN_MFMA16 && al && !h->cost_dense && h->al_all_sel) ||
               (h->plan == ALTRO_HIP_PLAN_GENERIC && al && !h->ragged && h->al_all_gsel)) ? EXPAND_DIAG : 0;
  running = h->batch;
  // Affine line-search trials (kernels/ilqr_merit2_dpp.hip, AFF): plan MFMA16, dynamics as data, fp64; the sweep's phi(0) evaluation
  // leaves the base trajectory and its sensitivity behind.  ALTRO_HIP_FORM_ROLLOUT_ROUNDS keeps every trial a rollout.
  h->aff_enabled = h->plan == ALTRO_HIP_PLAN_MFMA16 && h->dtype == ALTRO_HIP_F64 && !h->model_set && merit_rounds_dpp &&
                   !form(h, ALTRO_HIP_FORM_ROLLOUT_ROUNDS) && !h->aff_failed;
  h->aff_round = false; h->aff_store = false;
  if (h->aff_enabled && !(h->i_sens && h->i_sens_alpha && h->i_aff_part && h->i_aff_on)) {
    // all four buffers or none: a partial allocation is released and remembered, so that no later solve of this handle
    // finds a null pointer behind a non-null one or retries a multi-gigabyte hipMalloc; the sensitivity buffer (192 B per knot point
    // and problem) is only taken while it stays below a quarter of what the handle already holds -- an optimisation must not be what
    // runs a large batch out of memory
    const size_t B = h->batch, chunks = (h->N + 15) / 16;
    const size_t sens_bytes = B * (h->N + 1) * 24 * sizeof(double);
    int rc = sens_bytes > h->device_bytes / 4 + (size_t(64) << 20) ? 1 : 0;
    if (!rc) rc = dmalloc(h, &h->i_sens, sens_bytes);
    if (!rc) rc = dmalloc(h, &h->i_sens_alpha, B * sizeof(double));
    if (!rc) rc = dmalloc(h, &h->i_aff_part, chunks * ILQR_SPEC_TRIALS * B * 2 * sizeof(double));
    if (!rc) rc = dmalloc(h, &h->i_aff_on, (size_t)ILQR_SPEC_TRIALS * B * sizeof(int));
    if (!rc && hipMemsetAsync(h->i_aff_on, 0, (size_t)ILQR_SPEC_TRIALS * B * sizeof(int), h->stream) != hipSuccess) rc = 1;
    if (rc) {   // an optimisation only: this handle's rounds stay rollouts
      for (void** q : {&h->i_sens, &h->i_sens_alpha, &h->i_aff_part, &h->i_aff_on}) { if (*q) (void)hipFree(*q); *q = nullptr; }
      h->aff_enabled = false; h->aff_failed = true;
      (void)hipGetLastError();
    }
  }
  guard_on = h->aff_enabled && o.decision_margin > 0.0;
  la.guard = guard_on ? h->i_guard : nullptr;
  la.active_exact = guard_on ? h->i_active_exact : nullptr;
  la.decision_margin = guard_on ? o.decision_margin : 0.0;
  la.aff_fed = 0;
  la.aff_exact = (guard_on && form(h, ALTRO_HIP_FORM_AFFINE_EXACT)) ? 1 : 0;
  return 0;
}

// Counters without traffic on the stream: every counting launch (ILK_LS_BEGIN / _LS_FEED / _FINISH_ITER / _REG_RETRY) gets a fresh,
// zeroed 8-int slot of i_counters and publishes it into host-mapped memory itself (ilqr_publish_counters); an event behind the
// launch tells the host when to look.  No hipMemsetAsync / hipMemcpyAsync between the phases of a solve, and the host can enqueue
// AHEAD of a verdict it has not read yet.
int SolveRun::ensure_counters() {
  if (h->cnt_host) return 0;
  if (hipHostMalloc((void**)&h->cnt_host, (size_t)kCounterSlots * 8 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    (void)hipGetLastError();
    return fail(ALTRO_HIP_ERR_OUT_OF_MEMORY, "pinned host memory for the solve loop's counters");
  }
  void* dp = nullptr;
  HIP_TRY(hipHostGetDevicePointer(&dp, h->cnt_host, 0));
  h->cnt_host_dev = (int*)dp;
  for (hipEvent_t& e : h->cnt_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return 0;
}

// Plan LANE: whole solves run in ONE launch -- a workgroup of four (eight) waves per 8 / 16 / 32 problems sequencing itself with no
// host in between (kernels/ilqr_fused.hip) -- bit-identical to the launch-sequenced loop (tests/test_gpu_fused.py,
// tools/fuzz_fused.py).  ALTRO_HIP_FORM_SEQUENCED forces the latter;
// altro_hip_solve_options::fused_sweeps = n hands the problems still running after n sweeps over to the loop (a test hook: the hand-over is exact at
// any sweep).  POLICY: fused wherever the kernel exists.  Measured on MI355X (tools/solve_batches.py, profiles/r02p_solve_batches.txt;
// bicycle + steering bound, N = 50, median wall ms fused / sequenced): backtracking search 5.5 / 7.4 at 256 problems, 20 / 31 at 2048,
// 24 / 44 at 8192, 76 / 173 at 65536; cubic search 25 / 33, 28 / 61, 36 / 106, 99 / 316; pendulum, 8192 problems: 2.4 / 4.1, 2.6 / 4.3.
int SolveRun::choose_path() {
  // (the kernel works on the buffers of the three-launch merit evaluation; ALTRO_HIP_LANE_FUSED -- FMA-contracted sweeps -- is a
  //  property of the launch-sequenced kernels only)
  if (lane_plan) merit_split_prepare(h);
  const bool fused_can = lane_plan && o.iterations_max > 0 && !h->spec_no_memory && h->merit_split == 1 &&
                         !(h->flags & ALTRO_HIP_LANE_FUSED) && h->model.kind != MODEL_USER &&   // (run-time models: sequenced loop)
                         !h->cost_dense &&   // (the one-launch kernel is instantiated for the diagonal cost: a dense one runs sequenced)
                         o.stop_when_running_at_most <= 0;   // (the batch-level early return is the sequenced loop's)
  fused = fused_can && !form(h, ALTRO_HIP_FORM_SEQUENCED);
  if (fused && !ensure_spares(h, 3, spare_each))   // the waves' speculative steps (at most four per evaluation) need three
    fused = false;                                 // spare trajectories; without them THIS solve runs the sequenced loop
  if (async) {   // results while the solve runs: only the one-launch path can publish them
    if (!fused || o.fused_sweeps > 0)
      return fail(ALTRO_HIP_ERR_UNSUPPORTED, "altro_hip_ilqr_solve_async needs the one-launch solve kernel (plan LANE with a compiled-in "
                                             "device model, default environment); use altro_hip_ilqr_solve");
    const size_t bytes = (size_t)h->batch * sizeof(IlqrPollRec);
    if (!h->poll_host) {
      if (hipHostMalloc(&h->poll_host, bytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
          hipHostMalloc((void**)&h->poll_count_host, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();
        return fail(ALTRO_HIP_ERR_OUT_OF_MEMORY, "pinned host memory for %d poll records", h->batch);
      }
    }
    std::memset(h->poll_host, 0, bytes);
    *h->poll_count_host = 0;
  }
  // the head of Solve inside the fused kernel when that runs (IlqrFusedArgs::prologue), five launches otherwise
  // (not for the 2-state shapes: their eight-wave kernel lives on 256 registers and spills; with the prologue's code in it the
  //  pendulum solve loses 0.11 ms, the bicycle's MPC step gains 0.03 ms)
  fused_prologue = fused && h->n > 2;
  return 0;
}

// initial rollout, make it the nominal trajectory, expand everything (solver.cpp:420-434)
int SolveRun::prologue() {
  if (fused_prologue) return al_upload(h);   // (what ilqr_run does before any launch)
  int rc = loop(ILK_LOOP_INIT);
  if (rc) return rc;
  if (dual && !al && !h->cost_dense && !h->model_set) {   // (ROLLOUT_INIT: linear dynamics as data, the diagonal cost's gradient: one pass)
    rc = ilqr_run(h, IK_ROLLOUT, false, false, 0, 0.0, ROLLOUT_INIT);
  } else if (!al && row32_eligible(h)) {   // plan MFMA32's shapes: rollout, CopyTrajectory and the first expansion in one pass (row32_rollout_init_kernel)
    rc = ilqr_run(h, IK_ROLLOUT, false, false, 0, 0.0, ROLLOUT_INIT);
  } else {
    rc = ilqr_run(h, IK_ROLLOUT, false, false, 0, 0.0);
    if (!rc) rc = ilqr_run(h, IK_ACCEPT, false, false, 0, 0.0);
    // without constraints the cost Hessian is constant and is written once, here; with them the gradient is formed with the
    // penalty the constraints carry so far and SetPenalty comes after it (solver.cpp:424-430)
    if (!rc) rc = ilqr_run(h, IK_EXPAND, false, false, 0, 0.0, al ? EXPAND_GRADIENT : (EXPAND_GRADIENT | EXPAND_HESSIAN));
  }
  if (rc) return rc;
  return al ? loop(ILK_SET_PENALTY) : 0;
}

static void print_fused_clock(const std::vector<unsigned long long>& hc, int clk_groups, int sweeps) {
  static const char* names[ILQR_FUSED_PHASES] = {"hessians", "backward", "pre:roll", "pre:points", "pre:sums", "ls logic+select",
                                                 "ls:roll", "ls:points", "ls:sums", "re-expand", "stationarity+accept", "duals+gradients"};
  int slow = 0;
  unsigned long long slow_t = 0;
  std::vector<double> mean(ILQR_FUSED_PHASES, 0.0);
  for (int g = 0; g < clk_groups; ++g) {
    unsigned long long tot = 0;
    for (int p = 0; p < ILQR_FUSED_PHASES; ++p) { tot += hc[(size_t)g * ILQR_FUSED_PHASES + p]; mean[p] += (double)hc[(size_t)g * ILQR_FUSED_PHASES + p]; }
    if (tot > slow_t) { slow_t = tot; slow = g; }
  }
  std::fprintf(stderr, "[altro_hip] fused solve phase clock, us (mean over %d workgroups | slowest workgroup %d), %d sweeps max\n", clk_groups, slow, sweeps);
  for (int p = 0; p < ILQR_FUSED_PHASES; ++p)
    std::fprintf(stderr, "  %-22s %10.1f | %10.1f\n", names[p], mean[p] / clk_groups * 0.01, (double)hc[(size_t)slow * ILQR_FUSED_PHASES + p] * 0.01);
  std::fprintf(stderr, "  %-22s %10s | %10.1f\n", "total", "", (double)slow_t * 0.01);
}

// Straggler compaction.  The kernel is one launch of (batch / G) workgroups, each of which stays on its CU until
// ITS problems have stopped.  While the batch fits the chip (one workgroup per CU: up to 8192 problems) that is all a straggler costs --
// its own four waves.  A larger batch queues behind them: workgroups are dealt to CUs as these come free, a CU that drew a workgroup
// with a non-converging problem serves nobody else for iterations_max sweeps, and a straggler in a late workgroup starts its
// iterations_max sweeps late.  So a solve of more than four workgroups per CU runs its first kFusedFirstChunk sweeps as one launch,
// which hands back which slots still run (IlqrFusedArgs::run_flags); ilqr_list_running_kernel lists them in order, dealt round-robin
// to as many workgroups as the chip holds (ilqr_list_layout: a workgroup waits for the slowest line search among its problems, so the
// list is spread thin, not packed), and the rest of the sweeps are ONE launch over that list (IlqrFusedArgs::list: slot s serves
// problem list[s]).  Per problem it is the same code on the same data whichever slot it rides in: status, iterations and iterates
// are the single launch's bit for bit (tests/test_gpu_fused.py, tools/fuzz_fused.py).
// MEASURED (tools/compaction_times.py, C3 bicycles, 80 sweeps): 65536 problems 72 -> 60..66 ms; 32768 problems 42 -> 46 (hence the
// four-workgroups-per-CU rule).  What the tail of such a solve costs is NOT idle lanes or queueing but each straggler's own chain:
// 0.25 ms per sweep for the slowest of 17 stragglers (8192 problems), 0.52 for the slowest of 276 (65536) -- and every further
// re-listing is a barrier at which all wait for the slowest (chunks of 8 sweeps: +5 ms at 8192 problems, chunks of 2: +10 ms).
// altro_hip_solve_options::fused_sweeps = -n (a test hook like its positive form): re-list after every n sweeps whatever the batch.
// ALTRO_HIP_FORM_NO_COMPACTION: the single launch.
int SolveRun::run_fused() {
  int budget = o.iterations_max;
  if (o.fused_sweeps > 0) budget = std::max(1, std::min(o.iterations_max, o.fused_sweeps));
  const bool clock_on = form(h, ALTRO_HIP_FORM_FUSED_CLOCK) && !async;   // (an async solve returns before the clock could be read or freed)
  if (!h->fused_resident) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
    h->fused_resident = cus;   // (four waves of up to 512 registers each: one workgroup per CU)
  }
  const bool hook = o.fused_sweeps < 0;
  auto groups_of = [](int count) { const int G = ilqr_fused_group(count); return (count + G - 1) / G; };
  bool compact = !async && !clock_on && !form(h, ALTRO_HIP_FORM_NO_COMPACTION) && (hook || groups_of(h->batch) > 4 * h->fused_resident);
  // (slots of a listed launch: the problems rounded up to whole workgroups, or `resident` thin workgroups of up to 32)
  const size_t list_cap = (size_t)h->batch + 32 * (size_t)h->fused_resident + 32;
  if (compact && !h->i_fused_list && dmalloc(h, (void**)&h->i_fused_list, 3 * list_cap * sizeof(int))) {
    (void)hipGetLastError();
    compact = false;   // an optimisation only
  }
  int* const flags = h->i_fused_list;
  int* const lists[2] = {h->i_fused_list ? h->i_fused_list + list_cap : nullptr, h->i_fused_list ? h->i_fused_list + 2 * list_cap : nullptr};
  IlqrFusedArgs fa{0, budget, o.reg_retry_max, reg_on ? 1 : 0, h->i_counters, nullptr, fused_prologue ? 1 : 0};
  if (async) {
    void *dp = nullptr, *dc = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dp, h->poll_host, 0));
    HIP_TRY(hipHostGetDevicePointer(&dc, h->poll_count_host, 0));
    fa.poll = (IlqrPollRec*)dp; fa.poll_count = (int*)dc;
  }
  const int clk_G = ilqr_fused_group(h->batch);
  const int clk_groups = (h->batch + clk_G - 1) / clk_G;
  unsigned long long* clk = nullptr;     // ALTRO_HIP_FORM_FUSED_CLOCK: per-phase time of the kernel, printed to stderr (a tuning aid)
  if (clock_on) {
    const size_t bytes = (size_t)clk_groups * ILQR_FUSED_PHASES * sizeof(unsigned long long);
    if (hipMalloc((void**)&clk, bytes) == hipSuccess) { (void)hipMemsetAsync(clk, 0, bytes, h->stream); fa.clk = clk; }
  }
  int done = 0, slots = h->batch, groups = groups_of(h->batch), pp = 0;
  sweeps = 0;
  for (;;) {
    int chunk = budget - done;
    if (compact) {
      if (hook) chunk = std::min(chunk, -o.fused_sweeps);
      else if (done == 0) chunk = std::min(chunk, kFusedFirstChunk);   // (then ONE listed launch for the rest)
    }
    const bool last = done + chunk >= budget;
    HIP_TRY(hipMemsetAsync(h->i_counters, 0, 4 * sizeof(int), h->stream));
    fa.first_iter = done; fa.max_sweeps = chunk;
    fa.prologue = (done == 0 && fused_prologue) ? 1 : 0;
    fa.run_flags = (compact && !last) ? flags : nullptr;
    int frc;
    if (h->dtype == ALTRO_HIP_F64) {
      LaneArgs<double> ba{(const double*)h->l_in, (const double*)h->l_term, (double*)h->l_out, (double*)h->l_outn,
                          (const double*)h->l_x0, (double*)h->l_xuy, (double*)h->delta_V, h->status, h->N, h->batch, 0.0,
                          nullptr, nullptr};
      frc = ilqr_launch_fused<double>(h->stream, h->model.kind, h->n, h->m, ilqr_args<double>(h, false, false, 1, 0.0), la, ba, fa);
    } else {
      LaneArgs<float> ba{(const float*)h->l_in, (const float*)h->l_term, (float*)h->l_out, (float*)h->l_outn,
                         (const float*)h->l_x0, (float*)h->l_xuy, (float*)h->delta_V, h->status, h->N, h->batch, 0.0f,
                         nullptr, nullptr};
      frc = ilqr_launch_fused<float>(h->stream, h->model.kind, h->n, h->m, ilqr_args<float>(h, false, false, 1, 0.0), la, ba, fa);
    }
    if (frc) { if (clk) (void)hipFree(clk); return fail(ALTRO_HIP_ERR_HIP, "fused iLQR kernel launch failed"); }
    h->backward_done = true;
    if (async) {   // the caller polls; altro_hip_ilqr_wait finishes the bookkeeping
      h->async_pending = true;
      h->forward_done = true;
      h->solve_done = true;
      return 0;
    }
    if (fa.run_flags && ilqr_launch_list_running(h->stream, flags, fa.list, slots, lists[pp], h->fused_resident)) {
      if (clk) (void)hipFree(clk);
      return fail(ALTRO_HIP_ERR_HIP, "list kernel launch failed");
    }
    int c4[4];
    HIP_TRY(hipMemcpyAsync(c4, h->i_counters, sizeof(c4), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    sweeps += c4[3];
    running = c4[1];
    done += chunk;
    if (running == 0 || last) break;
    ilqr_list_layout(running, h->fused_resident, &fa.group, &groups);   // (what ilqr_list_running_kernel laid the list out for)
    fa.list = lists[pp]; fa.list_count = slots = fa.group * groups;
    pp ^= 1;
  }
  if (clk) {
    std::vector<unsigned long long> hc((size_t)clk_groups * ILQR_FUSED_PHASES);
    (void)hipMemcpy(hc.data(), clk, hc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(clk);
    print_fused_clock(hc, clk_groups, sweeps);
  }
  iter0 = running > 0 ? budget : o.iterations_max;
  return 0;
}

// CalcExpansions' cost Hessians (solver.cpp:448), BackwardPass with reg = 0 (solver.cpp:363), and -- the labelled extension -- the
// repeats of failed problems with more regularisation
int SolveRun::sweep_head(int iter) {
  la.iter = iter;
  int rc = loop(ILK_MARK_RUNNING);
  if (rc) return rc;
  if (al && !hessians_ready) {
    rc = ilqr_run(h, IK_EXPAND, false, true, 0, 0.0, EXPAND_HESSIAN | (hessian_stored ? diag_mode : 0));
    if (rc) return rc;
    hessian_stored = true;
  }
  if ((rc = launch_backward(h, 0.0))) return rc;
  h->backward_done = true;
  for (int attempt = 0; attempt < o.reg_retry_max; ++attempt) {
    const int sr = counted(ILK_REG_RETRY);
    if (sr < 0) return ALTRO_HIP_ERR_HIP;
    int again = 0;
    if ((rc = verdict(sr, 2, &again))) return rc;
    if (again == 0) break;
    total_reg_retries += again;
    if ((rc = launch_backward(h, 0.0))) return rc;
  }
  return o.reg_retry_max > 0 ? loop(ILK_MARK_RUNNING) : 0;
}

// ForwardPass: phi(0), then the line search (solver.cpp:237-271).  While the running problems leave half of the chip idle, the
// first step the search will ask for (alpha0 = 1, known in advance) rides in the same launch as phi(0) -- phi, phi' and the
// trajectory go to spare row / buffer 0 -- and ILK_LS_BEGIN consumes it at once.
int SolveRun::first_evaluation(int* prev_slot, int* begin_slot, int* launches, bool* refreshed) {
  bool pre = !dual && spec_all_on && !h->spec_no_memory && spec_waves(running, 2) <= spec_capacity;
  if (pre && !ensure_spares(h, 1, spare_each)) {   // an optimisation only: carry on one step per launch
    h->spec_no_memory = true;
    pre = false;
  }
  h->spec_trials = pre ? 2 : 1; h->spec_pre = pre ? 1 : 0;
  // (IK_MERIT2, mode 2: the two-trial evaluation with the broadcasts on the VALU's DPP path and two problems per wave,
  //  kernels/ilqr_merit2_dpp.hip -- with or without constraint blocks)
  h->aff_store = true;                               // (phi(0): the base of the sweep's affine trials)
  int rc = dual ? ilqr_run(h, IK_MERIT2, true, true, 1, 0.0, 2) : ilqr_run(h, IK_MERIT, true, true, 1, 0.0);
  h->aff_store = false;
  h->spec_trials = 1; h->spec_pre = 0;
  if (rc) return rc;
  *launches = 1;
  la.spec_pre = (pre || dual) ? 1 : 0;
  la.spec_flip = dual ? 1 : 0; la.stat_done = h->i_stat_done;
  la.stat_inline = (h->plan == ALTRO_HIP_PLAN_MFMA16 && h->dtype == ALTRO_HIP_F64) ? 1 : 0;   // (the tile's two-trial pass also takes the candidate's stationarity)
  int prev = counted(ILK_LS_BEGIN);      // the slot whose [0] says whether another evaluation is needed
  if (prev < 0) return ALTRO_HIP_ERR_HIP;
  la.spec_pre = 0; la.spec_flip = 0;
  if (dual) {
    // searches that ended WITHOUT the first step (phi' too small, not a descent direction): their candidate is the alpha = 0
    // evaluation's, redone by the single-step kernel for them alone -- launched on their mask whether or not there are any
    // (counted only when there were: the verdict is read with the first round's)
    int* keep = h->i_active;
    h->i_active = h->i_spec_refresh;
    rc = ilqr_run(h, IK_MERIT, true, true, 1, 0.0);
    h->i_active = keep;
    if (rc) return rc;
  } else if (pre) {
    if ((rc = ilqr_run(h, IK_SPEC_SELECT, false, false, 0, 0.0))) return rc;
    *refreshed = true;
  } else {
    // The first trial step is launched without asking the device whether any problem needs it: the masks make it a no-op when
    // none does, and it saves one host read-back per sweep (these loops are latency-bound).
    // (a rollout like the two-trial pass's second row: the first step is the same bits whichever way the sweep's head runs)
    if ((rc = ilqr_run(h, IK_MERIT, true, true, 1, 0.0))) return rc;
    ++*launches;
    prev = counted(ILK_LS_FEED);
    if (prev < 0) return ALTRO_HIP_ERR_HIP;
  }
  *prev_slot = prev;
  *begin_slot = dual ? prev : -1;
  return 0;
}

// Further line-search rounds.  Speculative backtracking: once the problems still searching leave most of the chip idle, one launch
// evaluates the next 2, 4 or 8 steps of the (known) sequence alpha beta^j for each of them; the feed kernel consumes them in order,
// so every decision is the sequential one (kernels/ilqr_types.h).
int SolveRun::search_rounds(int* prev_slot, int rounds_last, int* rounds_out, int* launches, bool* refreshed) {
  int prev = *prev_slot, rc = 0;
  int searching = -1;            // the last count read (the speculation width follows it)
  int rounds = 0;
  int guarded_last = 0;          // problems the last feed read sent to the rollout launch (the sweep's first feed: none)
  for (int guard = 0; guard < 64; ++guard) {
    const bool ahead = run_ahead && rounds < rounds_last;    // history says this round will be needed: enqueue it first
    if (!ahead) {
      if ((rc = verdict(prev, 0, &searching))) return rc;
      if (searching == 0) break;
      if (guard_on && rounds > 0) { if ((rc = verdict(prev, 5, &guarded_last))) return rc; guarded_total += guarded_last; }
    }
    const int width_for = searching > 0 ? searching : running;
    int trials = 1;
    if (spec_on && !h->spec_no_memory)
      while (trials < trials_cap && spec_waves(width_for, trials * 2) <= spec_capacity) trials *= 2;   // as wide as leaves the launch within the capacity
    // plan MFMA16's rounds in the DPP form evaluate two trials per problem in the lanes one trial would leave idle
    // (kernels/ilqr_merit2_dpp.hip): the second step of the known sequence rides along whatever the occupancy
    if (spec_on && !lane_plan && !h->spec_no_memory && trials < 2 && merit_rounds_dpp) trials = 2;
    if (trials > 1 && !ensure_spares(h, trials_cap - 1, spare_each)) {
      while (trials > 1 && trials - 1 > h->spare_count) trials /= 2;   // as wide as the spares there are
      if (trials == 1 && h->spare_count == 0) h->spec_no_memory = true;
    }
    const bool spec = trials > 1;
    h->spec_trials = trials;
    la.spec_trials = h->spec_trials;
    h->aff_round = true;                             // (dynamics as data: the trials of a round are sums over independent knot points)
    rc = ilqr_run(h, IK_MERIT, true, true, 1, 0.0);
    h->aff_round = false;
    if (rc) return rc;
    // The decision guard: the trials the last feed would not decide on affine values (IlqrLoopArgs::guard) are evaluated as rollouts,
    // on their own mask.  Not launched when the last verdict read says there are none (the first round of a sweep never has any);
    // launched blind when the host runs ahead of the verdicts -- an empty mask makes it a no-op.
    if (guard_on && rounds > 0 && (ahead || guarded_last != 0)) {
      int* keep = h->i_active;
      h->i_active = h->i_active_exact;
      rc = ilqr_run(h, IK_MERIT, true, true, 1, 0.0);
      h->i_active = keep;
      if (rc) return rc;
    }
    la.aff_fed = h->aff_enabled ? 1 : 0;
    const int sf = counted(ILK_LS_FEED);
    la.aff_fed = 0;
    if (sf < 0) return ALTRO_HIP_ERR_HIP;
    if (spec && (rc = ilqr_run(h, IK_SPEC_SELECT, false, false, 0, 0.0))) return rc;
    h->spec_trials = 1;
    la.spec_trials = 1;
    if (ahead) {
      if ((rc = verdict(prev, 0, &searching))) return rc;
      if (searching == 0) break;      // the round just enqueued was not needed: it ran on empty masks (not counted)
      if (guard_on && rounds > 0) { if ((rc = verdict(prev, 5, &guarded_last))) return rc; guarded_total += guarded_last; }
    }
    ++*launches; ++rounds;
    if (spec) *refreshed = true;
    prev = sf;
  }
  *prev_slot = prev;
  *rounds_out = rounds;
  return 0;
}

// steps accepted from a speculative trial carry no phi' pass: redo their expansion (what the derivative pass of a sequential trial
// would have left behind); then the convergence criteria on the accepted candidate, and make it the nominal (solver.cpp:459-469)
int SolveRun::finish_sweep(bool refreshed, int* fin_slot) {
  int rc = 0;
  if (refreshed) {
    int* keep = h->i_active;
    h->i_active = h->i_spec_refresh;
    rc = ilqr_run(h, IK_EXPAND, false, true, 0, 0.0, EXPAND_GRADIENT);
    h->i_active = keep;
    if (rc) return rc;
  }
  if ((rc = loop(ILK_MARK_RUNNING))) return rc;
  // (dual: only the problems whose step was not the one the two-trial pass evaluated -- the skip mask; most waves leave at once)
  h->stat_skip = dual ? h->i_stat_done : nullptr;
  rc = ilqr_run(h, IK_STATIONARITY, false, true, 0, 0.0);
  h->stat_skip = nullptr;
  if (!rc) rc = ilqr_run(h, IK_ACCEPT, false, true, 0, 0.0);
  if (rc) return rc;
  *fin_slot = counted(ILK_FINISH_ITER);
  return *fin_slot < 0 ? ALTRO_HIP_ERR_HIP : 0;
}

// DualUpdate, PenaltyUpdate, refreshed gradients for the problems that asked (solver.cpp:470-489)
int SolveRun::outer_update() {
  // plan MFMA16, DPP forms: ONE pass over the constraint rows does the dual update, the gradients and the next sweep's Hessians
  // (EXPAND_DUAL | EXPAND_NEXT), and PenaltyUpdate's bookkeeping follows it
  const bool tile = !lane_plan && !generic_plan;
  const bool expand_dpp = !form(h, ALTRO_HIP_FORM_EXPAND_LDS), alrows_dpp = !form(h, ALTRO_HIP_FORM_ALROWS_LDS);
  int rc;
  if (tile && expand_dpp && alrows_dpp) {
    h->expand_penalty_scaling = o.penalty_scaling; h->expand_penalty_max = o.penalty_max;
    rc = ilqr_run(h, IK_EXPAND, false, true, 0, 0.0, EXPAND_GRADIENT | EXPAND_HESSIAN | EXPAND_NEXT | EXPAND_DUAL | (hessian_stored ? diag_mode : 0));
    if (rc) return rc;
    if ((rc = loop(ILK_PENALTY_UPDATE))) return rc;
    hessians_ready = true;
    return 0;
  }
  if ((rc = ilqr_run(h, IK_DUAL, false, false, 0, 0.0))) return rc;
  if ((rc = loop(ILK_PENALTY_UPDATE))) return rc;
  // ... and, in the same pass over the constraint rows, the cost Hessians the NEXT sweep's CalcExpansions would form: nothing they
  // depend on (trajectory, duals, penalties) changes between here and there (plan MFMA16, DPP form: gradient for the problems whose
  // duals changed, Hessians for every problem still running)
  const bool merged = tile && expand_dpp;
  rc = ilqr_run(h, IK_EXPAND, false, true, 0, 0.0, merged ? (EXPAND_GRADIENT | EXPAND_HESSIAN | EXPAND_NEXT | (hessian_stored ? diag_mode : 0)) : EXPAND_GRADIENT);
  if (rc) return rc;
  hessians_ready = merged;
  return 0;
}

// ---- the launch-sequenced loop, with the host running AHEAD of the device's verdicts -------------------------------------------------
// Every kernel of a sweep masks itself (running / active / the line search's own state), so a launch made for problems that turn out
// not to need it is a no-op.  The host therefore does not wait for a count before it enqueues what follows when the solve's own
// history says the work will be needed: the next line-search round goes out before the previous round's count is read (as many
// rounds ahead as the last sweep took), and from the second sweep on the next sweep's head goes out before the count of problems
// still running is read.  A verdict that says "nobody" stops the enqueueing; what is already queued runs on empty masks.  The stream
// never drains inside a solve, and no memset or copy rides on it (counter slots).  Results are those of the wait-then-launch loop bit
// for bit: the same kernels see the same masks in the same order.  (ALTRO_HIP_NO_RUNAHEAD: wait for every verdict before enqueueing
// on -- the comparison the tests hold this against.)
int SolveRun::sequenced_loop() {
  int rc = 0;
  if (iter0 < o.iterations_max)   // the one memset of the solve: every slot but the one-launch kernel's starts from zero
    HIP_TRY(hipMemsetAsync(h->i_counters + 8, 0, (size_t)(kCounterSlots - 1) * 8 * sizeof(int), h->stream));
  const int stop_at = o.stop_when_running_at_most > 0 ? o.stop_when_running_at_most : 0;
  int pend_finish = -1;          // slot of the previous sweep's ILK_FINISH_ITER whose count has not been read yet
  bool multi_sweep = false;      // a second sweep was needed: from now on the next sweep's head is enqueued ahead
  int rounds_last = 0;           // line-search rounds the previous sweep needed (beyond the dual / first evaluation)
  for (int iter = iter0; iter < o.iterations_max; ++iter) {
    if ((rc = sweep_head(iter))) return rc;
    bool refreshed = false;
    int prev = -1, begin_slot = -1, sweep_merit_launches = 0;
    if ((rc = first_evaluation(&prev, &begin_slot, &sweep_merit_launches, &refreshed))) return rc;
    // the previous sweep's verdict, now that this sweep's head keeps the device busy
    if (pend_finish >= 0) {
      int still = 0;
      if ((rc = verdict(pend_finish, 1, &still))) return rc;
      pend_finish = -1;
      ++sweeps;                                   // (the previous sweep)
      if (still == 0) break;                      // nobody was running: what this sweep enqueued ran on empty masks
      running = still;
    }
    int rounds = 0;
    if ((rc = search_rounds(&prev, rounds_last, &rounds, &sweep_merit_launches, &refreshed))) return rc;
    rounds_last = rounds;
    if (begin_slot >= 0) {   // (the verdicts of ILK_LS_BEGIN's launch are in: it was waited for above)
      int redone = 0;
      if ((rc = verdict(begin_slot, 3, &redone))) return rc;
      if (redone > 0) ++sweep_merit_launches;
    }
    total_merit_launches += sweep_merit_launches;
    int fin = -1;
    if ((rc = finish_sweep(refreshed, &fin))) return rc;
    if (al && (rc = outer_update())) return rc;
    // how many problems still run: read now -- or, once the solve has shown that it takes several sweeps, after the next sweep's
    // head has been enqueued
    if (run_ahead && multi_sweep && iter + 1 < o.iterations_max && stop_at == 0) {
      pend_finish = fin;
      continue;
    }
    int still = 0;
    if ((rc = verdict(fin, 1, &still))) return rc;
    ++sweeps;
    if (still <= stop_at) break;   // (stop_at = 0: nobody runs any more; > 0: altro_hip_solve_options::stop_when_running_at_most)
    running = still;
    multi_sweep = true;
  }
  if (pend_finish >= 0) {   // (the last sweep the iteration limit allowed)
    int still = 0;
    if ((rc = verdict(pend_finish, 1, &still))) return rc;
    ++sweeps;
  }
  return 0;
}

}  // namespace

extern "C" int altro_hip_ilqr_solve(altro_hip_batch* h, const altro_hip_solve_options* opts, altro_hip_solve_result* results) {
  int rc = ilqr_check(h, true);
  if (rc) return rc;
  h->expansion_current = false;
  SolveRun run(h);
  run.async = h->async_request;
  h->async_request = false;
  if (h->async_pending) {   // a solve started with altro_hip_ilqr_solve_async is still out: finish it first
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->async_pending = false;
  }
  if ((rc = run.configure(opts)) || (rc = run.ensure_counters()) || (rc = run.choose_path()) || (rc = run.prologue())) return rc;
  h->spec_trials = 1;
  h->bwd_active = h->i_active;      // the backward sweep skips problems that have stopped, only inside this call (~SolveRun)
  h->bwd_reg = run.reg_on ? h->i_reg : nullptr;
  if (run.fused) {
    if ((rc = run.run_fused())) return rc;
    if (run.async) return 0;
  }
  if ((rc = run.sequenced_loop())) return rc;
  h->forward_done = true;
  h->solve_done = true;
  HIP_TRY(hipStreamSynchronize(h->stream));   // (iterations_max <= 0 reaches this point with kernels still in flight)
  if (results && (rc = ilqr_gather_results(h, results))) return rc;
  h->last_sweeps = run.sweeps;
  h->last_merit_launches = run.total_merit_launches;
  return 0;
}
