// ilqr_types.h -- plain argument / state types of the batched iLQR loop (shared by the kernels and by the host
// code of the C ABI).  No kernels here: this header may be included by any translation unit.
#pragma once
#include "../rtc_compat.h"

#include "../linesearch_sm.h"
#include "../models.h"
#include "al_types.h"

namespace altro_hip {

struct IlqrProb {       // per-problem control state of the batched solve
  int running;          // still iterating
  int iterations;       // AltroStats::iterations (solver.cpp:506)
  int status;           // SolveStatus: 0 Success, 1 Unsolved, 2 MaxIterations
  int ls_failed;
  double phi0, dphi0, alpha, stationarity;
  int ls_iters, evaluating;
  LsState ls;
  // augmented Lagrangian: rho = the constraints' penalty (KnotPointData::rho_, identical for every block
  // of a problem), rho_est = the penalty the cached projected duals were formed with, dual = what the
  // outer update has to do after this sweep (0 nothing, 1 duals, 2 duals + penalty)
  double rho, rho_est, feasibility;
  int dual, n_dual_updates;
  int reg_retries;      // extension: backward passes repeated with a larger reg
};

template <typename T>
struct IlqrArgs {
  T* in;
  T* term;
  const T* out;
  const T* outn;
  T* nom;
  T* cand;
  const T* cost;
  const T* x0;          // [n][batch]
  const double* alpha;  // per problem, or nullptr -> alpha_const
  const int* active;    // per problem, or nullptr -> all
  double* phi;
  double* dphi;
  IlqrProb* prob;
  ModelParams mp;
  int N, batch;
  int want_derivative;
  double alpha_const;
  AlTable<T> al;
  int mode;             // expand kernel: bit 0 = dynamics Jacobians + cost gradient, bit 1 = cost Hessian
  // Speculative backtracking (merit kernel, gridDim.y = spec_trials): block row j > 0 evaluates, for the problems
  // that are in the backtracking stage, the step the state machine WILL ask for after j more failures -- the sequence
  // alpha beta^j is known in advance (linesearch.cpp:385-413) -- into phi[j * batch + b] and the j-th spare
  // candidate trajectory; ilqr_ls_feed_kernel consumes the results in order, so decisions are unchanged.
  T* cand_spec;         // [spec_trials - 1][same shape as cand]
  int64_t spec_stride;  // elements between two spare candidate trajectories
  int spec_trials;      // 1 = no speculation
  int spec_pre;         // 1: this is the phi(0) launch and block row 1 evaluates the search's first step alpha0 = 1
                        //    (phi and phi' into row 1, no expansion stores, spare candidate 0)
  int spec_flip;        // with spec_pre (fused solve kernel only): the roles are swapped -- the alpha0 = 1 pass writes the
                        //    candidate trajectory and the expansion, the phi(0) pass goes to spare candidate 0 and stores
                        //    nothing: the step that is nearly always accepted needs no copy and no re-expansion afterwards
  const int* spec_sel;  // [batch] IK_SPEC_SELECT: copy spare candidate spec_sel[b] - 1 over the candidate of problem b
  double ls_beta;       // CubicLineSearch::beta_decrease
  int ls_max_iters;
  // MeritFunction in three launches (kernels/ilqr_lane.hip): per-knot-point costs of every trial, and the A | B | lx | lu
  // block of a derivative pass that must not touch the backward pass's input record.  nullptr: the one-launch kernel.
  T* merit_jk;          // [spec_trials][N + 1][batch]
  T* spec_jac;          // [N][n n + n m + n + m][batch] then [n][batch]
  // 0: `cost` holds the diagonal cost records Qd | Rd | q | r | c (ALTROSolver::SetLQRCost); 1: the dense quadratic ones
  // Q | R | H | q | r | c (ALTROSolver::SetQuadraticCost, altro_solver.cpp:118-136) -- selects the kernels' CK instantiation
  int cost_kind = 0;
};
constexpr int STAT_NO_FEAS = 32;   // IK_STATIONARITY / IK_DUAL on plan MFMA16: the constraint rows in the DPP form (ilqr_merit2_dpp.hip)
enum { EXPAND_GRADIENT = 1, EXPAND_HESSIAN = 2, EXPAND_LDS = 16 /* plan MFMA16: wave_expand_kernel instead of the DPP form (A/B, tests) */,
       EXPAND_NEXT = 64 /* wave_expand_dpp_kernel at the end of a sweep: the gradient for the problems of the active mask (those whose
                           duals changed), the cost Hessians of the NEXT sweep for every problem still running (IlqrProb::running) */,
       EXPAND_DUAL = 128 /* ... and the sweep's DualUpdate in the same pass (IlqrProb::dual instead of the active mask) */,
       EXPAND_DYN = 256 /* plan MFMA16 with a device model: also the dynamics Jacobians Z = [A B] of the stored candidate trajectory
                           (wave_expand_dyn_kernel) -- the head of Solve and the re-expansion after a speculative step; a merit pass
                           with derivative leaves them itself */,
       EXPAND_DIAG = 512 /* wave_expand_dpp_kernel<.., BOUNDS>: only the DIAGONAL of the Hessian blocks is stored -- the rest was
                            stored by a full expansion of this solve and cannot have changed (diagonal cost, bound-type blocks) */ };

struct IlqrLoopArgs {
  IlqrProb* prob;
  double* alpha;      // [batch] next trial step per problem
  int* active;        // [batch] 1 = problem takes part in the next merit evaluation
  const double* phi;
  const double* dphi;
  int* counters;      // [0] = problems that still need a merit evaluation, [1] = problems still running; [3], [4]: spec_flip;
                      // [7] = blocks of the counting kernel that are through (see counters_pub)
  int* counters_pub = nullptr;   // optional, host-mapped pinned memory [7]: the last block of a counting kernel (ILK_LS_BEGIN, _LS_FEED,
                                 // _FINISH_ITER, _REG_RETRY) copies counters[0..6] there, so the host reads them after an event with no
                                 // copy or memset on the stream -- every such launch gets a fresh zeroed slot of counters (capi_ilqr.hip)
  int spec_trials;    // trials evaluated by the last merit launch (phi holds spec_trials x batch values)
  int spec_pre;       // ILK_LS_BEGIN: row 1 of phi / dphi holds the first trial step (see IlqrArgs::spec_pre)
  int* spec_sel;      // [batch] 1 + the spare candidate trajectory a problem that JUST finished its search took (0: none)
  int* spec_refresh;  // [batch] 1: the accepted step came from a speculative trial (no phi' pass): expansion to be redone
  // Plan MFMA16's dual evaluation (wave_merit2_kernel: phi(0) and the step alpha0 = 1 from ONE stream of the records, the
  // alpha0 = 1 pass being the one that writes the candidate trajectory and the expansion).  ILK_LS_BEGIN then consumes
  // row 1 like spec_pre and sorts the running problems three ways:
  //   ended ON the first step   -> nothing to copy or redo; stat_done[b] = stat_inline (the kernel already left the
  //                                candidate's stationarity / feasibility in the control block)
  //   ended WITHOUT it          -> spec_refresh[b] = 1, counters[3]++: the alpha = 0 evaluation is repeated by the
  //                                single-step kernel for these (phi' too small, not a descent direction: rare)
  //   still searching           -> the ordinary rounds follow
  // counters[4] counts the running problems whose stationarity still has to be computed by IK_STATIONARITY.
  int spec_flip = 0;
  int stat_inline = 0;
  int* stat_done = nullptr;   // [batch]
  // The decision guard of the affine line-search rounds (linesearch_sm.h: ls_feed_is_robust).  ILK_LS_FEED with aff_fed = 1 consumes
  // values an affine round produced: a trial whose turn is not robust against `decision_margin` is NOT fed -- guard[b] = 1, the problem
  // leaves `active` and enters `active_exact`, the next round's rollout launch (on that mask) evaluates the same step in the
  // reference's order, and the feed after it takes those values as they are.  counters[5] counts the problems so marked.
  int* guard = nullptr;          // [batch]
  int* active_exact = nullptr;   // [batch]
  int aff_fed = 0;
  int aff_exact = 0;             // ALTRO_HIP_FORM_AFFINE_EXACT: see ilqr_ls_feed_body
  double decision_margin = 0.0;
  int batch;
  int iter;
  int iterations_max;
  double tol_stationarity, tol_meritfun_gradient, tol_primal_feasibility;
  double penalty_initial, penalty_scaling, penalty_max;
  int al_enabled;
  // regularisation retry (extension beyond the reference, which keeps reg = 0 and ignores failures)
  double* reg;          // [batch]
  const int* bwd_status;  // [batch] -1 or the failing knot point of the last backward pass
  double reg_initial, reg_scale, reg_min, reg_max;
  LsOptions ls;
};

// ---- plan MFMA16 ((12, 4), wave per problem, dynamics given as data): kernels/ilqr_mfma16.hip ----------------
constexpr int MF_NOM = 16;     // nominal record: x 12 | u 4
constexpr int MF_COSTP = 36;   // cost-parameter record: Qd 12 | Rd 4 | q 12 | r 4 | c 1 | pad 3

template <typename S>
struct IlqrWaveArgs {
  const S* dyn;   int64_t dyn_bs, dyn_ks;    // DYN records: Z 192 | f 12
  S* cin;         int64_t cin_bs, cin_ks;    // COST records: triu(Q) 78 | pad 2 | [H R] 64 | [q r] 16
  S* term;                                   // [b][156]: Q_N 144 | q_N 12
  const S* out;   int64_t out_bs, out_ks;    // OUT records: Kt 52 | triu(P) 78 | p 12 | pad 2
  const S* outn;                             // [b][156]
  S* nom;                                    // [k][b][16]
  S* cand;        int64_t xuy_bs, xuy_ks;    // [k][b][28]: x 12 | y 12 | u 4
  const S* costp;                            // [k][b][36]
  const S* x0;                               // [b][12]
  const double* alpha;
  const int* active;
  S* cand_spec; int64_t spec_stride; int spec_trials; int spec_pre; double ls_beta; int ls_max_iters;   // see IlqrArgs
  const int* spec_sel;
  double* phi;
  double* dphi;
  IlqrProb* prob;
  int N, batch, want_derivative;
  double alpha_const;
  AlTable<S> al;
  int mode;                                  // expand kernel: EXPAND_GRADIENT | EXPAND_HESSIAN; rollout kernel: ROLLOUT_INIT
  double penalty_scaling = 10.0, penalty_max = 1e8;   // EXPAND_DUAL: PenaltyUpdate's parameters (solver_options.hpp:27-29)
  const int* skip = nullptr;                 // stationarity kernel: problems whose value wave_merit2_kernel already left
  // The dense quadratic cost of ALTROSolver::SetQuadraticCost (altro_solver.cpp:118-136, knotpoint_data.cpp:64-85): the cost's own
  // W = [Q H^T; H R], [q r] and c in the layout of a COST record -- triu(Q) 78 | c | pad | [H R] 64 | [q r] 16 -- for knot points
  // 0 .. N - 1 (record N: only c_N), and Q_N rows | q_N like TERM.  (The COST records themselves cannot serve: the loop rewrites
  // their [q r] slot with the gradient, and with constraint blocks their Q / [H R] with the Gauss-Newton Hessian.)  Read by the
  // row-layout kernels (kernels/ilqr_merit2_dpp.hip) instead of costp when cost_dense != 0.
  const S* costd = nullptr;                  // [k][b][160]
  const S* costd_term = nullptr;             // [b][156]
  int cost_dense = 0;
  // A device model instead of dynamics given as data (altro_hip_set_model on plan MFMA16, kernels/ilqr_tile_model.hip): the
  // rollout and the merit evaluation step the model, the expansion writes Z = [A B] into the DYN records (mp.kind != MODEL_LINEAR)
  ModelParams mp{MODEL_LINEAR, 0.0f, 0, 2.7, 1.5};
  // Affine line-search trials (dynamics as data, kernels/ilqr_merit2_dpp.hip: AFF): the base trajectory and its sensitivity the first
  // pass of a sweep leaves behind, the step it was taken at, the chunks' shares of phi / phi' and which (trial, problem) is evaluated
  double* sens = nullptr;                    // [k][b][24]: x_k(alpha_b) 12 | dx_k / dalpha 12      (k = 0 .. N)
  double* sens_alpha = nullptr;              // [b]: alpha_b
  double* aff_part = nullptr;                // [chunk][ILQR_SPEC_TRIALS][b][2]
  int* aff_on = nullptr;                     // [ILQR_SPEC_TRIALS][b]
  int aff = 0;                               // IK_MERIT: launch the affine form
};
constexpr int MF_COSTD_C = 78;    // the constant term c inside a dense cost record (the first pad slot of the COST layout)
constexpr int ROLLOUT_INIT = 4;   // wave_rollout_kernel also writes the nominal record and the cost gradient (the head of
                                  // Solve for an unconstrained problem: rollout + CopyTrajectory + expansion in one pass)

template <typename S>
int ilqr_wave_launch_kernel(hipStream_t stream, int which, const IlqrWaveArgs<S>& a);   // ilqr_launch_mfma16.hip
template <typename S>
int ilqr_wave_launch_model(hipStream_t stream, int which, const IlqrWaveArgs<S>& a);    // ilqr_launch_mfma16_model.hip (a.mp.kind)
bool ilqr_tile_model_supported(int kind, int n, int m);                                 // device models of the (12, 4) tile plan
// the merit passes of a handle with more than AL_MAXC constraint slots at some knot point (al_types.h: AL_TILE_MAXC; fp64 records)
template <typename S>
int ilqr_wave_launch_wide(hipStream_t stream, int which, const IlqrWaveArgs<S>& a);     // ilqr_launch_mfma16_wide.hip

// Speculative backtracking: most trials one merit launch evaluates (the host picks 1, 2, 4 or 8 by how idle the chip is)
constexpr int ILQR_SPEC_TRIALS = 8;
enum IlqrKernel { IK_ROLLOUT, IK_ACCEPT, IK_EXPAND, IK_MERIT, IK_STATIONARITY, IK_DUAL, IK_SHIFT, IK_SPEC_SELECT,
                  IK_MERIT2 /* plan MFMA16: phi(0) and the first step in one pass */ };
enum IlqrLoopKernel { ILK_LOOP_INIT, ILK_LS_BEGIN, ILK_LS_FEED, ILK_FINISH_ITER, ILK_MARK_RUNNING, ILK_SET_PENALTY,
                      ILK_PENALTY_UPDATE, ILK_REG_RETRY };

// Launchers (ilqr_launch_f64.hip / ilqr_launch_f32.hip hold the kernel instantiations, so that the kernels of
// the two element types compile in parallel with the rest of the library).  Return 0, 1 = no device model for
// (kind, n, m), 2 = launch error (hipGetLastError has the reason).
// plan LANE: whole sweeps in one launch (kernels/ilqr_fused.hip)
template <typename T>
struct LaneArgs;   // kernels/tvlqr_lane.hip
constexpr int ILQR_FUSED_PHASES = 12;   // IlqrFusedArgs::clk
// What a solve reports per problem, gathered on the device in the layout of altro_hip_solve_result (altro_hip.h): the host
// copies 56 bytes per problem instead of the 280-byte control blocks.
struct IlqrResult {
  int status, iterations;
  double stationarity, final_alpha, final_phi, primal_feasibility, penalty;
  int dual_updates, reg_retries;
};
// altro_hip_ilqr_solve_async / _poll: one record per problem in pinned host memory, published by the fused solve kernel the
// moment the problem stops (mirror of altro_hip_poll_record, altro_hip.h)
struct IlqrPollRec {
  IlqrResult result;
  double u0[4];        // the first input of the solution (what a receding-horizon caller applies)
  int done, reserved;
};
struct IlqrFusedArgs {
  int first_iter;     // index of the first sweep this launch runs (0 right after the initial rollout / expansion)
  int max_sweeps;     // sweeps to run at most in this launch
  int reg_retry_max;  // altro_hip_solve_options::reg_retry_max
  int use_reg;        // per-problem regularisation in force (reg_initial > 0 or retries enabled)
  int* counters;      // [1] += problems still running when the launch ends; [3] = max sweeps any wave ran (atomicMax)
  unsigned long long* clk;   // optional [workgroups][ILQR_FUSED_PHASES] phase clock (100 MHz ticks), a tuning aid
  int prologue;       // 1: the launch starts with the head of Solve (solver.cpp:420-434): control blocks, initial rollout,
                      //    accept, first expansion, SetPenalty -- what the host enqueues as five launches otherwise
  IlqrPollRec* poll = nullptr;   // optional [batch], pinned host memory: results published while the launch still runs
  int* poll_count = nullptr;     // optional, pinned: number of records published so far
  // straggler compaction (capi_solve.hip, run_fused): the launch serves the `list_count` problems list[0 .. list_count) -- slot s of
  // the grid is problem list[s] -- instead of the whole batch; run_flags[s] = 1 at hand-back when slot s is still running
  const int* list = nullptr;     // [list_count] problem per slot, -1 = empty (never a workgroup's first slot)
  int list_count = 0;            // slots = workgroups x `group`
  int group = 0;                 // problems per workgroup of a listed launch (8 / 16 / 32)
  int* run_flags = nullptr;
};
template <typename T, int G>   // G problems per workgroup: one translation unit each (ilqr_fused_unit.inc)
int ilqr_launch_fused_g(hipStream_t stream, int kind, int n, int m, const IlqrArgs<T>& a, const IlqrLoopArgs& la,
                        const LaneArgs<T>& ba, const IlqrFusedArgs& fa);
// Problems per workgroup (of four waves): as few as keeps every workgroup resident at once, one per CU -- the fewer, the more
// knot points each wave takes at once in the (problem, knot point)-parallel steps (ilqr_fused.hip, KS).
inline int ilqr_fused_group(int batch) { return batch <= 8 * 256 ? 8 : batch <= 16 * 256 ? 16 : 32; }
template <typename T>
inline int ilqr_launch_fused(hipStream_t stream, int kind, int n, int m, const IlqrArgs<T>& a, const IlqrLoopArgs& la,
                             const LaneArgs<T>& ba, const IlqrFusedArgs& fa) {
  const int G = fa.list ? fa.group : ilqr_fused_group(a.batch);
  return G == 8    ? ilqr_launch_fused_g<T, 8>(stream, kind, n, m, a, la, ba, fa)
         : G == 16 ? ilqr_launch_fused_g<T, 16>(stream, kind, n, m, a, la, ba, fa)
                   : ilqr_launch_fused_g<T, 32>(stream, kind, n, m, a, la, ba, fa);
}

bool ilqr_supported(int kind, int n, int m);
template <typename T>
int ilqr_launch_kernel(hipStream_t stream, int which, int kind, int n, int m, const IlqrArgs<T>& a);
int ilqr_launch_loop(hipStream_t stream, int which, const IlqrLoopArgs& a);
int ilqr_launch_results(hipStream_t stream, const IlqrProb* prob, IlqrResult* out, int batch);
// Straggler compaction of the one-launch solve: `count` problems on a device that holds `resident` workgroups at once -- problems
// per workgroup and the number of workgroups, spread as thin as the device allows (a workgroup waits for the slowest line search
// among its problems, and its knot-point-parallel steps share four waves)
__host__ __device__ inline void ilqr_list_layout(int count, int resident, int* group, int* groups) {
  const int G = count <= 8 * resident ? 8 : count <= 16 * resident ? 16 : 32;
  *group = G;
  *groups = count <= G * resident ? (count < resident ? count : resident) : (count + G - 1) / G;
}
// The problems of list_in[0 .. count_in) (list_in == nullptr: of 0 .. count_in) whose run flag is set, in order, dealt round-robin
// to the workgroups of ilqr_list_layout(their number, resident): the i-th goes to list_out[(i % groups) * group + i / groups],
// the other slots are -1.
int ilqr_launch_list_running(hipStream_t stream, const int* flags, const int* list_in, int count_in, int* list_out, int resident);

}  // namespace altro_hip
