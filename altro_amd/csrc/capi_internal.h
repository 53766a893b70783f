// capi_internal.h -- the handle behind include/altro_hip/altro_hip.h and the host-side helpers its translation units share
// (capi_core.hip: lifetime, setters, getters; capi_tvlqr.hip: the sweep launchers; capi_ilqr.hip: the iLQR loop, constraint
// blocks and MPC operations; capi_stats.hip: statistics reduction and the multi-device runner).  Host-side plumbing only:
// all arithmetic lives in kernels/*.hip.  There is no CPU fallback anywhere behind this header.
#pragma once
#include "altro_hip/altro_hip.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels/pack.hip"
#include "kernels/tvlqr_generic.hip"
#include "kernels/tvlqr_lane.hip"
#include "kernels/ilqr_types.h"
#include "kernels/ilqr_generic.hip"
#include "kernels/mfma16_layout.h"
#include "linesearch_sm.h"

namespace altro_hip {
namespace capi {

// records the message altro_hip_last_error() returns (thread-local, capi_core.hip) and passes `code` through
int fail(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return ::altro_hip::capi::fail(ALTRO_HIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                        \
  } while (0)

constexpr size_t kStageBytes = size_t(256) << 20;

}  // namespace capi
}  // namespace altro_hip

using namespace altro_hip;   // internal header: the handle below names the kernels' argument types directly

struct altro_hip_batch {
  int N = 0, n = 0, m = 0, batch = 0, dtype = 0, plan = 0, device = 0;
  unsigned flags = 0;
  unsigned forms = 0;     // ALTRO_HIP_FORM_* bits of the handle (altro_hip_set_forms); a solve ORs its options' bits in for its duration
  hipStream_t stream = nullptr;
  bool own_stream = false;
  size_t esz = 8;  // element size on the device
  // common
  void* x0 = nullptr;
  void* delta_V = nullptr;
  int* status = nullptr;
  bool dyn_set = false, cost_set = false, x0_set = false, backward_done = false, forward_done = false;
  int has_f = 0, is_diag = 0;
  int host_batch = 0;   // > 0: host arrays of the next set_* calls hold this many problems, tiled over the batch
  bool dev_ptrs = false;   // altro_hip_set_pointer_mode: bulk arrays of set_* / get_* are device pointers
  // plan GENERIC: reference layout on the device
  void* g_arr[G_NUM] = {};
  int64_t g_bstride[G_NUM] = {};
  // affine line-search trials (plan MFMA16, dynamics as data, fp64: kernels/ilqr_merit2_dpp.hip AFF): buffers, and the two switches the
  // solve loop sets -- on for this solve / this launch is a line-search round
  void *i_sens = nullptr, *i_sens_alpha = nullptr, *i_aff_part = nullptr, *i_aff_on = nullptr;
  bool aff_failed = false;                   // the buffers could not be had once: no retry on this handle
  bool aff_enabled = false, aff_round = false, aff_store = false;   // (aff_store: this launch is the sweep's phi(0) evaluation)
  bool g_tile = false;                       // plan MFMA32: plan GENERIC's arrays, the sweeps of kernels/tvlqr_tile32.hip (capi_tile32.hip)
  bool g_mfma = false;                       // plan GENERIC, fp64: the backward sweep's products on the matrix cores (ALTRO_HIP_GENERIC_MATRIX_CORES)
  bool ragged = false;                       // per-knot-point dimensions (altro_hip_batch_create_dims): plan GENERIC, TVLQR sweeps only
  std::vector<int> nxv, nuv;                 // nx[0..N], nu[0..N-1] of a ragged handle (n, m hold the maxima)
  int64_t* g_off = nullptr;
  int* g_nx = nullptr;
  int* g_nu = nullptr;
  // plan MFMA16
  void *m_in = nullptr, *m_cin = nullptr, *m_term = nullptr, *m_out = nullptr, *m_outn = nullptr, *m_xuy = nullptr,
       *m_qblk = nullptr, *m_trash = nullptr;   // element type = handle dtype (fp32 storage allowed)
  Mfma16Strides m_st{};
  // plan LANE: batch structure-of-arrays ([k][element][batch])
  void *l_in = nullptr, *l_term = nullptr, *l_out = nullptr, *l_outn = nullptr, *l_xuy = nullptr,
       *l_x0 = nullptr;
  // iLQR loop state (plan LANE): nominal trajectory, cost parameters, per-problem control blocks
  void *l_nom = nullptr, *l_cost = nullptr;
  IlqrProb* i_prob = nullptr;
  double *i_alpha = nullptr, *i_phi = nullptr, *i_dphi = nullptr;
  int *i_active = nullptr, *i_counters = nullptr;   // i_counters: kCounterSlots slots of 8 ints (slot 0: the one-launch solve kernel's)
  int* cnt_host = nullptr;        // the sequenced loop's view of the counters: host-mapped pinned memory, one slot per counting launch
  int* cnt_host_dev = nullptr;    // ... its device address
  hipEvent_t cnt_ev[16] = {};     // events recorded behind the counting launches (ring: at most a few are ever in flight)
  // speculative backtracking (altro_hip_ilqr_solve): spare candidate trajectories, allocated on first use
  void* i_cand_spec = nullptr;
  // altro_hip_ilqr_solve_async / _poll / _wait: pinned records the fused kernel publishes into while it runs
  void* poll_host = nullptr; int* poll_count_host = nullptr;
  bool async_request = false, async_pending = false;
  bool rtc_row32_ok = false;          // plans GENERIC / MFMA32: the source's row-layout kernels compiled without scratch memory (capi_rtc.hip)
  bool rtc_has_constraints = false;   // ... whose source also defines altro_user_constraint / _jacobian
  void* rtc = nullptr;            // run-time compiled model (capi_rtc.hip: RtcModule, shared through a per-process cache)
  std::string rtc_source;         // ... its source, and the cost kind (IlqrArgs::cost_kind) its cost-reading kernels were instantiated for
  int rtc_ck = 0;
  int x0_stride = 0;              // elements between two problems' x0 on the device (12 on plan MFMA16, else n)
  int spare_count = 0;            // spare candidate trajectories i_cand_spec holds (sized to the path in use, see spec_trials_cap)
  int spare_failed = 0;           // > 0: an allocation of this many spares failed on this handle (no retry at this size or above)
  int *i_spec_sel = nullptr, *i_spec_refresh = nullptr;
  int* i_fused_list = nullptr;   // straggler compaction of the one-launch solve (capi_solve.hip): [3][batch] = run flags, two lists
  int fused_resident = 0;        // workgroups of the one-launch kernel the device holds at once (0: not asked yet)
  int *i_guard = nullptr, *i_active_exact = nullptr;   // the decision guard of the affine rounds (IlqrLoopArgs::guard, ::active_exact)
  int* i_stat_done = nullptr;     // plan MFMA16's dual merit evaluation (IlqrLoopArgs::stat_done)
  const int* stat_skip = nullptr; // set while a solve's IK_STATIONARITY launches may skip those problems
  // MeritFunction in three launches (plan LANE, kernels/ilqr_lane.hip): per-knot-point costs of every trial and the
  // spare A | B | lx | lu block; allocated on the first merit evaluation, merit_split = 0 keeps the one-launch kernel
  void *i_merit_jk = nullptr, *i_spec_jac = nullptr;
  int merit_split = -1;           // -1: not decided yet (ALTRO_HIP_MERIT_SPLIT, default on)
  void *i_results = nullptr, *i_results_host = nullptr;   // per-problem solve results: device gather + pinned staging (first solve)
  void* g_ws = nullptr;           // plan GENERIC, blocks beyond 64 KB of LDS: the backward sweep's per-problem work blocks (first use)
  bool fwd_quad = false;          // the last LANE forward sweep ran four lanes per problem
  bool bwd_quad = false;          // the last LANE backward sweep ran four lanes per problem (kernels/tvlqr_quad_body.inc)
  int spec_trials = 1;            // trials per merit launch of the CURRENT launch (1 = no speculation)
  int spec_pre = 0;               // the current launch is phi(0) fused with the first trial step
  bool spec_no_memory = false;    // the spare trajectories could not be allocated: no speculation on this handle
  double spec_beta = 0.5;         // the running solve's LsOptions::beta_decrease / max_iters (what the merit kernels
  int spec_max_iters = 25;        // need to reproduce the state machine's step sequence)
  ModelParams model{MODEL_LINEAR, 0.0f, 0, 2.7, 1.5};
  bool model_set = false, lqr_cost_set = false, guess_set = false;
  std::vector<double> x0_host;    // the initial state as given, while the handle may still move to another plan (replan_empty_handle)
  int x0_host_bz = 0;
  bool auto_plan = false;         // created with ALTRO_HIP_PLAN_AUTO (the plan may still move to LANE when a LANE-only device model arrives)
  bool user_stream = false;
  // augmented-Lagrangian constraint blocks (plan LANE): host mirrors + device tables, uploaded lazily
  std::vector<AlDef> al_defs;
  std::vector<AlKnotBig> al_knots;       // (the device table is AlKnot for plans LANE / MFMA16, AlKnotBig for plan GENERIC)
  int al_uniform = 0, al_rows_per_knot = 0;  // knot points 0..N-1 carry the same blocks (kernels/al_types.h)
  std::vector<double> al_G;                  // pool of G blocks, column-major p x (n+m)
  std::vector<std::vector<double>> al_g;     // per block: [p] or [batch][p]
  int al_rows = 0;
  bool al_dirty = false;
  AlKnot* al_d_knots = nullptr;
  AlKnotBig* al_d_big = nullptr;
  void *al_d_G = nullptr, *al_d_g = nullptr, *al_d_z = nullptr;
  void* al_d_Gpad = nullptr;                 // AlTable::Gpad (plan MFMA16)
  int al_Gpad_count = 0;
  bool al_all_gsel = false;              // plan GENERIC: every block is bound-type (the Hessian blocks change on their diagonals only)
  double* g_stat_part = nullptr; size_t g_stat_part_bytes = 0;   // row32_stationarity_kernel's per-chunk maxima (capi_ilqr.hip: gen_run)
  int* al_d_gsel = nullptr;              // plan GENERIC: AlTable::gsel (bound-type blocks)
  bool al_row32_ok = false;              // every block fits kernels/ilqr_row32.hip (row-wise cone, <= 32 rows)
  int al_max_ncon = 0;                   // most blocks (plan MFMA16: slots, al_types.h) any knot point has
  int al_G_count = 0;                        // elements of the device G pool
  int al_has_soc = 0;                        // some block is a second-order cone
  int al_all_sel = 0;                        // every block is bound-type (rows +-e_idx: AlKnot::sel)
  double expand_penalty_scaling = 10.0, expand_penalty_max = 1e8;   // what EXPAND_DUAL's look-ahead of PenaltyUpdate needs
  const int* bwd_active = nullptr;           // per-problem mask for the backward sweep inside ilqr_solve
  const double* bwd_reg = nullptr;           // per-problem regularisation inside ilqr_solve (retry extension)
  // iLQR loop on plan MFMA16 (dynamics given as data): nominal trajectory + cost parameters, allocated by
  // altro_hip_set_tracking_cost; ilqr_linear = the backward sweep ignores the affine term (knotpoint_data.cpp:416)
  void *m_nom = nullptr, *m_costp = nullptr;
  bool ilqr_linear = false;
  // the dense quadratic cost of altro_hip_set_quadratic_cost (ALTROSolver::SetQuadraticCost): plan MFMA16 keeps the cost's own
  // blocks in COST-record layout next to the records the loop rewrites (IlqrWaveArgs::costd / costd_term); plan LANE keeps
  // Q | R | H | q | r | c records in l_costq (IlqrArgs::cost with cost_kind = 1)
  void *m_costd = nullptr, *m_costd_term = nullptr, *l_costq = nullptr;
  bool cost_dense = false;
  // plan MFMA16, no constraint blocks: the last thing that touched the candidate trajectory was a merit evaluation WITH derivative
  // over the whole batch -- which leaves lx, lu (and, with a device model, A, B) of that candidate in the records, while the
  // quadratic cost's Hessian blocks never change: altro_hip_expand then has nothing to do.  Cleared by everything that changes
  // the candidate, the cost or the dynamics.
  bool expansion_current = false;
  // iLQR loop on plan GENERIC (kernels/ilqr_generic.hip): nominal trajectory and the cost's own blocks, dense [b][k][block]
  void *g_xn = nullptr, *g_un = nullptr, *g_cQ = nullptr, *g_cR = nullptr, *g_cH = nullptr, *g_cq = nullptr, *g_cr = nullptr, *g_cc = nullptr;
  double* i_reg = nullptr;
  // staging for host <-> device conversion (grown lazily, never inside the hot path)
  void* stage = nullptr;
  size_t stage_bytes = 0;
  size_t device_bytes = 0;
  // statistics reduction (capi_stats.hip): per-block partials and the reduced vector, on the device
  double *st_partial = nullptr, *st_red = nullptr;
  bool solve_done = false;   // altro_hip_ilqr_solve has run: the per-problem control blocks hold AltroStats
  // profiling: 0 off, 1 = events + a synchronisation per launch, 2 = events only (resolved by profile_get), so that
  // the launches of a timed region can be bracketed without stalling the stream between them
  int prof = 0;
  std::vector<hipEvent_t> prof_ev;    // mode 2: pairs of events, one pair per recorded launch
  std::vector<int> prof_slot;
  int prof_n = 0;
  double prof_min[2] = {0, 0}, prof_max[2] = {0, 0};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t launch_ev0 = nullptr, launch_ev1 = nullptr;   // the events the NEXT profiled launch carries (ProfScope, capi_tvlqr.hip)
  int last_sweeps = 0, last_merit_launches = 0;
  int prof_launches[2] = {0, 0};
  int prof_dropped[2] = {0, 0};   // mode 2: launches issued after the event ring was full (not in the averages)
  double prof_ms[2] = {0, 0};
};

namespace altro_hip {
namespace capi {

inline bool form(const altro_hip_batch* h, unsigned bit) { return (h->forms & bit) != 0; }

inline int dmalloc(altro_hip_batch* h, void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes ? bytes : 16);
  if (e != hipSuccess) {
    *p = nullptr;
    return fail(ALTRO_HIP_ERR_OUT_OF_MEMORY, "hipMalloc(%zu bytes) failed: %s", bytes,
                hipGetErrorString(e));
  }
  h->device_bytes += bytes;
  return 0;
}

inline int check(altro_hip_batch* h) {
  if (!h) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "null handle");
  hipError_t e = hipSetDevice(h->device);
  if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "hipSetDevice(%d): %s", h->device, hipGetErrorString(e));
  return 0;
}

// entry of the iLQR-loop family of calls.  A handle with per-knot-point dimensions (altro_hip_batch_create_dims; the reference's
// ALTROSolver::SetDimension(n, m, k_start, k_stop), altro_solver.cpp:26-47) runs the loop of plan GENERIC, whose kernels walk the
// sweep's offset table; the calls whose meaning needs ONE dimension (the receding-horizon shift, device models) say so.
inline int loop_entry(altro_hip_batch* h, bool takes_varying_dims = false) {
  int rc = check(h);
  if (rc) return rc;
  if (h->ragged && !takes_varying_dims)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "this call needs uniform dimensions; a handle with per-knot-point dimensions (altro_hip_batch_create_dims) "
                                           "takes altro_hip_set_dynamics / _set_quadratic_cost / _set_tracking_cost / _update_linear_costs / _set_initial_state / "
                                           "_set_input_guess / _set_state_guess / _add_linear_constraint, the iLQR loop calls and their getters, all with packed "
                                           "[batch][k][block_k] arrays");
  return 0;
}

inline int ensure_stage(altro_hip_batch* h, size_t bytes) {
  if (h->stage_bytes >= bytes) return 0;
  if (h->stage) (void)hipFree(h->stage);
  h->stage = nullptr;
  h->stage_bytes = 0;
  hipError_t e = hipMalloc(&h->stage, bytes);
  if (e != hipSuccess)
    return fail(ALTRO_HIP_ERR_OUT_OF_MEMORY, "staging hipMalloc(%zu) failed: %s", bytes,
                hipGetErrorString(e));
  h->stage_bytes = bytes;
  return 0;
}

inline int grid_for(int64_t total, int block = 256) {
  int64_t g = (total + block - 1) / block;
  return (int)std::min<int64_t>(std::max<int64_t>(g, 1), 256 * 32);
}

// element counts of one knot point's block in the reference layout
struct Dims {
  int n, m;
  int A() const { return n * n; }
  int B() const { return n * m; }
  int Q(int diag) const { return diag ? n : n * n; }
  int R(int diag) const { return diag ? m : m * m; }
  int H() const { return m * n; }
};

// Upload one reference-layout host array chunk by chunk and hand each chunk to `consume`.
// host layout: [batch or 1][nk or 1][block] doubles.
template <typename F>
inline int upload_chunks(altro_hip_batch* h, const double* host, int block, int nk, int k_zero, int b_zero,
                  int nk_host, int src_off, F consume) {
  const int src_nk = nk_host > 0 ? nk_host : (k_zero ? 1 : nk);
  const size_t per_problem = (size_t)src_nk * block * sizeof(double);
  if (h->dev_ptrs) {   // the caller's array already lives in HBM: no staging, one pass over the whole batch
    SrcArr s{host + src_off, b_zero ? 0 : (int64_t)src_nk * block, k_zero ? 0 : (int64_t)block, 0};
    int rc = consume(s, 0, h->batch);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));   // the caller may reuse its buffer on return
    return 0;
  }
  if (b_zero) {
    int rc = ensure_stage(h, per_problem);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->stage, host, per_problem, hipMemcpyHostToDevice, h->stream));
    SrcArr s{(const double*)h->stage + src_off, 0, k_zero ? 0 : (int64_t)block, 0};
    rc = consume(s, 0, h->batch);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
  }
  int chunk = (int)std::max<size_t>(1, std::min<size_t>(h->batch, kStageBytes / std::max<size_t>(per_problem, 1)));
  int rc = ensure_stage(h, per_problem * chunk);
  if (rc) return rc;
  for (int b0 = 0; b0 < h->batch; b0 += chunk) {
    const int nb = std::min(chunk, h->batch - b0);
    HIP_TRY(hipMemcpyAsync(h->stage, host + (size_t)b0 * src_nk * block, per_problem * nb,
                           hipMemcpyHostToDevice, h->stream));
    SrcArr s{(const double*)h->stage + src_off, (int64_t)src_nk * block, k_zero ? 0 : (int64_t)block, 0};
    rc = consume(s, b0, nb);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));  // the staging buffer is reused by the next chunk
  }
  return 0;
}

template <typename T>
inline int generic_set(altro_hip_batch* h, int arr, const double* host, int block, int nk, int k_zero,
                int b_zero, int k0 = 0, int nk_host = -1, int src_off = 0) {
  if (h->host_batch > 0 && h->host_batch < h->batch && !b_zero)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "altro_hip_set_host_batch tiling is not available on plan GENERIC");
  // writes knot points [k0, k0+nk) of the device array from a host array that holds nk_host knot
  // points per problem (default nk, or 1 when k_zero)
  T* dst = (T*)h->g_arr[arr] + (int64_t)k0 * block;
  const int64_t bs = h->g_bstride[arr];
  return upload_chunks(h, host, block, nk, k_zero, b_zero, nk_host, src_off, [&](SrcArr s, int b0, int nb) {
    const int64_t total = (int64_t)nb * nk * block;
    hipLaunchKernelGGL(expand_copy_kernel<T>, dim3(grid_for(total)), dim3(256), 0, h->stream, dst,
                       bs, (int64_t)block, s, block, nk, b0, nb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "expand_copy launch: %s", hipGetErrorString(e));
    return 0;
  });
}

// One reference-layout host array ([batch or 1][nk_host][block]) into `block` consecutive elements of AoS device
// records: dst[b * dst_bs + k * dst_ks + e], k = 0..nk-1.
template <typename T>
inline int aos_set(altro_hip_batch* h, T* dst, int64_t dst_bs, int64_t dst_ks, const double* host, int block, int nk,
            int k_zero, int b_zero, int nk_host = -1, int src_off = 0) {
  return upload_chunks(h, host, block, nk, k_zero, b_zero, nk_host, src_off, [&](SrcArr s, int b0, int nb) {
    const int64_t total = (int64_t)nb * nk * block;
    hipLaunchKernelGGL(expand_copy_kernel<T>, dim3(grid_for(total)), dim3(256), 0, h->stream, dst, dst_bs, dst_ks, s,
                       block, nk, b0, nb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "expand_copy launch: %s", hipGetErrorString(e));
    return 0;
  });
}
// Download: run `produce(dst_device, b0, nb)` chunk by chunk into staging, then copy to the host.
template <typename F>
inline int download_chunks(altro_hip_batch* h, double* host, int block, int nk, F produce) {
  const size_t per_problem = (size_t)nk * block * sizeof(double);
  if (h->dev_ptrs) {   // write straight into the caller's device array
    int rc = produce(host, 0, h->batch);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
  }
  int chunk = (int)std::max<size_t>(1, std::min<size_t>(h->batch, kStageBytes / std::max<size_t>(per_problem, 1)));
  int rc = ensure_stage(h, per_problem * chunk);
  if (rc) return rc;
  for (int b0 = 0; b0 < h->batch; b0 += chunk) {
    const int nb = std::min(chunk, h->batch - b0);
    rc = produce((double*)h->stage, b0, nb);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(host + (size_t)b0 * nk * block, h->stage, per_problem * nb,
                           hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  return 0;
}

template <typename T>
inline int aos_get(altro_hip_batch* h, double* host, const T* src, int64_t src_bs, int64_t src_ks, int block, int nk) {
  return download_chunks(h, host, block, nk, [&](double* dst, int b0, int nb) {
    const int64_t total = (int64_t)nb * nk * block;
    hipLaunchKernelGGL(gather_copy_kernel<T>, dim3(grid_for(total)), dim3(256), 0, h->stream, dst,
                       (int64_t)nk * block, (int64_t)block, src, src_bs, src_ks, block, nk, b0, nb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "gather_copy launch: %s", hipGetErrorString(e));
    return 0;
  });
}

template <typename T>
inline int generic_get(altro_hip_batch* h, int arr, double* host, int block, int nk) {
  const T* src = (const T*)h->g_arr[arr];
  const int64_t bs = h->g_bstride[arr];
  return download_chunks(h, host, block, nk, [&](double* dst, int b0, int nb) {
    const int64_t total = (int64_t)nb * nk * block;
    hipLaunchKernelGGL(gather_copy_kernel<T>, dim3(grid_for(total)), dim3(256), 0, h->stream, dst,
                       (int64_t)nk * block, (int64_t)block, src, bs, (int64_t)block, block, nk, b0, nb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "gather_copy launch: %s", hipGetErrorString(e));
    return 0;
  });
}

inline int mfma16_get(altro_hip_batch* h, int what, double* host, int block, int nk) {
  return download_chunks(h, host, block, nk, [&](double* dst, int b0, int nb) {
    const int64_t total = (int64_t)nb * nk * block;
    if (h->dtype == ALTRO_HIP_F64)
      hipLaunchKernelGGL(mfma16_unpack_kernel<double>, dim3(grid_for(total)), dim3(256), 0, h->stream, dst, what,
                         (const double*)h->m_out, (const double*)h->m_outn, (const double*)h->m_xuy,
                         (const double*)h->m_qblk, h->m_st, h->N, b0, nb, h->n, h->m);
    else
      hipLaunchKernelGGL(mfma16_unpack_kernel<float>, dim3(grid_for(total)), dim3(256), 0, h->stream, dst, what,
                         (const float*)h->m_out, (const float*)h->m_outn, (const float*)h->m_xuy,
                         (const float*)h->m_qblk, h->m_st, h->N, b0, nb, h->n, h->m);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "mfma16_unpack launch: %s", hipGetErrorString(e));
    return 0;
  });
}

// Whole-array upload of one reference-layout source for the MFMA16 pack kernels (setup path).
// `nk_host` = knot points per problem actually present in the host array.
struct DevSrc {
  void* dev = nullptr;
  SrcArr s{nullptr, 0, 0, 0};
  ~DevSrc() { if (dev) (void)hipFree(dev); }
};
inline int put_src(altro_hip_batch* h, const double* src, int blk, int nk_host, int k_zero, int b_zero,
            DevSrc* out) {
  if (!src) return 0;
  const size_t per_b = (size_t)nk_host * blk;
  const int tiled = (!b_zero && h->host_batch > 0 && h->host_batch < h->batch) ? h->host_batch : 0;
  const size_t bytes = (size_t)(b_zero ? 1 : (tiled ? tiled : h->batch)) * per_b * sizeof(double);
  if (h->dev_ptrs) {   // device pointer: use it in place
    out->s = SrcArr{src, b_zero ? 0 : (int64_t)per_b, k_zero ? 0 : (int64_t)blk, tiled};
    return 0;
  }
  if (hipMalloc(&out->dev, bytes) != hipSuccess)
    return fail(ALTRO_HIP_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed", bytes);
  if (hipMemcpyAsync(out->dev, src, bytes, hipMemcpyHostToDevice, h->stream) != hipSuccess)
    return fail(ALTRO_HIP_ERR_HIP, "H2D copy failed");
  out->s = SrcArr{(const double*)out->dev, b_zero ? 0 : (int64_t)per_b, k_zero ? 0 : (int64_t)blk, tiled};
  return 0;
}
// (cin / term: the COST and TERM records to fill -- the handle's own, or the dense cost's copies m_costd / m_costd_term, which
//  have the same layout and strides)
inline int mfma16_pack_launch(altro_hip_batch* h, int seg, SrcArr s0, SrcArr s1, void* cin = nullptr, void* term = nullptr,
                              int is_diag = -1) {
  const int64_t total = (int64_t)h->batch * h->N * 192;
  if (!cin) cin = h->m_cin;
  if (!term) term = h->m_term;
  if (is_diag < 0) is_diag = h->is_diag;
  if (h->dtype == ALTRO_HIP_F64)
    hipLaunchKernelGGL(mfma16_pack_kernel<double>, dim3(grid_for(total)), dim3(256), 0, h->stream, (double*)h->m_in,
                       (double*)cin, (double*)term, h->m_st, seg, s0, s1, is_diag, h->N, 0, h->batch, h->n, h->m);
  else
    hipLaunchKernelGGL(mfma16_pack_kernel<float>, dim3(grid_for(total)), dim3(256), 0, h->stream, (float*)h->m_in,
                       (float*)cin, (float*)term, h->m_st, seg, s0, s1, is_diag, h->N, 0, h->batch, h->n, h->m);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "mfma16_pack launch: %s", hipGetErrorString(e));
  return 0;
}

// ---- plan LANE dispatch ---------------------------------------------------------------------------
// every (n, m) with n <= 6, m <= 3 (the reference is dimension-generic, tvlqr.cpp:92-121: a shape one off a fast one must not
// fall back to plan GENERIC); (2, 1) and (4, 2) additionally have their several-lanes-per-problem sweeps
#define LANE_SHAPES(X)                                                                                   \
  X(1, 1) X(2, 1) X(3, 1) X(4, 1) X(5, 1) X(6, 1) X(1, 2) X(2, 2) X(3, 2) X(4, 2) X(5, 2) X(6, 2) X(1, 3) \
  X(2, 3) X(3, 3) X(4, 3) X(5, 3) X(6, 3)
inline bool lane_supported(int n, int m) {
#define X(N_, M_) if (n == N_ && m == M_) return true;
  LANE_SHAPES(X)
#undef X
  return false;
}
struct LaneSizes { int e_in, e_term, e_out, e_xuy; };
inline LaneSizes lane_sizes(int n, int m) {
  return LaneSizes{2 * n * n + 2 * n * m + m * m + 2 * n + m, n * n + n, m * n + m + n * n + n, 2 * n + m};
}
// one reference-layout source -> a run of elements of the SoA records [k0, k0+nk)
template <typename T>
inline int lane_pack(altro_hip_batch* h, T* dst_base, int E, const double* host, int len, int dst_off,
              int diag_n, int nk, int k_src0, int nk_host, int kz, int bz, int src_off = 0) {
  DevSrc d;
  int rc = put_src(h, host, len, nk_host, kz, bz, &d);
  if (rc) return rc;
  LaneSeg s{d.s.p ? d.s.p + src_off : nullptr, d.s.bs, d.s.ks, len, dst_off, diag_n, d.s.bmod};
  const int dlen = diag_n > 0 ? diag_n * diag_n : len;
  const int64_t total = (int64_t)h->batch * nk * dlen;
  hipLaunchKernelGGL(lane_pack_kernel<T>, dim3(grid_for(total)), dim3(256), 0, h->stream, dst_base, E, s,
                     nk, k_src0, h->batch);
  if (hipGetLastError() != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "lane_pack launch failed");
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}
template <typename T>
inline int lane_get(altro_hip_batch* h, double* host, const void* src, const void* src_term, int E, int off,
             int off_term, int len, int nk, int nk_main) {
  return download_chunks(h, host, len, nk, [&](double* dst, int b0, int nb) {
    const int64_t total = (int64_t)nb * nk * len;
    hipLaunchKernelGGL(lane_unpack_kernel<T>, dim3(grid_for(total)), dim3(256), 0, h->stream, dst,
                       (const T*)src, (const T*)src_term, E, off, 0, off_term, len, nk, nk_main, b0, nb,
                       h->batch);
    if (hipGetLastError() != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "lane_unpack launch failed");
    return 0;
  });
}

constexpr int kCounterSlots = 2048;   // counter slots of one handle (altro_hip_batch::i_counters)
constexpr int kStatsBlocks = 1024, kStatsStride = 16;   // capi_stats.hip: partials [kStatsBlocks][kStatsStride]

// run-time compiled user models (capi_rtc.hip): the kernels of the launch-sequenced loop, in this order in RtcModule::fn
enum RtcKernel { RTC_ROLLOUT = 0, RTC_ACCEPT, RTC_EXPAND, RTC_MERIT, RTC_MERIT_ROLL, RTC_MERIT_POINT, RTC_MERIT_SUM, RTC_SPEC_SELECT,
                 RTC_ZERO_RESIDUALS, RTC_STATIONARITY, RTC_DUAL, RTC_SHIFT, RTC_NUM };
template <typename T>
int rtc_launch(altro_hip_batch* h, int which, const IlqrArgs<T>& a);
int rtc_tile_launch(altro_hip_batch* h, int which, const IlqrWaveArgs<double>& a);   // plan MFMA16: the model kernels of a caller's source
int rtc_gen_launch(altro_hip_batch* h, int which, const IlqrGenArgs<double>& a);     // plans GENERIC / MFMA32: likewise

// the sweep launchers (capi_tvlqr.hip), also used by the iLQR loop
int replan_empty_handle(altro_hip_batch* h, int plan);   // capi_core.hip
// capi_ilqr.hip, for the solve loop in capi_solve.hip
int al_upload(altro_hip_batch* h);
int ilqr_check(altro_hip_batch* h, bool need_guess);
int ilqr_run(altro_hip_batch* h, int which, bool use_alpha, bool use_active, int want_deriv, double alpha_const,
             int mode = EXPAND_GRADIENT | EXPAND_HESSIAN);
template <typename T>
IlqrArgs<T> ilqr_args(altro_hip_batch* h, bool use_alpha, bool use_active, int want_deriv, double alpha_const);
int spec_trials_cap(const altro_hip_batch* h);
bool ensure_spares(altro_hip_batch* h, int count, size_t bytes_each);
void merit_split_prepare(altro_hip_batch* h);
int ilqr_gather_results(altro_hip_batch* h, altro_hip_solve_result* results);
int launch_backward(altro_hip_batch* h, double reg);
bool row32_eligible(const altro_hip_batch* h);             // capi_ilqr.hip: kernels/ilqr_row32.hip serves the handle
bool row32_model_eligible(const altro_hip_batch* h);
bool tile32_supported(int n, int m);                        // capi_tile32.hip: plan MFMA32
int tile32_launch_backward(altro_hip_batch* h, double reg);
int tile32_launch_forward(altro_hip_batch* h);
int launch_forward(altro_hip_batch* h);
bool mfma16_forward_is_x4(const altro_hip_batch* h);   // the forward sweep runs four problems per wave (pure fp32)

}  // namespace capi
}  // namespace altro_hip
