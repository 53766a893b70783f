// capi_core.hip -- C ABI (include/altro_hip/altro_hip.h): library queries, handle lifetime, problem-data setters,
// result getters, profiling slots.  Layout conversion launches only; no arithmetic.
#include "capi_internal.h"

namespace altro_hip {
namespace capi {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

}  // namespace capi
}  // namespace altro_hip

using namespace altro_hip;
using namespace altro_hip::capi;

namespace {

int lane_get_any(altro_hip_batch* h, int what, double* dst) {
  const int n = h->n, m = h->m, N = h->N;
  const LaneSizes z = lane_sizes(n, m);
  const void *src, *term = nullptr;
  int E, off, off_t = 0, len, nk, nk_main;
  switch (what) {
    case MGET_K: src = h->l_out; E = z.e_out; off = 0; len = m * n; nk = nk_main = N; break;
    case MGET_d: src = h->l_out; E = z.e_out; off = m * n; len = m; nk = nk_main = N; break;
    case MGET_P: src = h->l_out; term = h->l_outn; E = z.e_out; off = m * n + m; off_t = 0; len = n * n; nk = N + 1; nk_main = N; break;
    case MGET_p: src = h->l_out; term = h->l_outn; E = z.e_out; off = m * n + m + n * n; off_t = n * n; len = n; nk = N + 1; nk_main = N; break;
    case MGET_x: src = h->l_xuy; E = z.e_xuy; off = 0; len = n; nk = nk_main = N + 1; break;
    case MGET_y: src = h->l_xuy; E = z.e_xuy; off = n; len = n; nk = nk_main = N + 1; break;
    default: src = h->l_xuy; E = z.e_xuy; off = 2 * n; len = m; nk = nk_main = N; break;
  }
  return h->dtype == ALTRO_HIP_F64 ? lane_get<double>(h, dst, src, term, E, off, off_t, len, nk, nk_main)
                                   : lane_get<float>(h, dst, src, term, E, off, off_t, len, nk, nk_main);
}

}  // namespace

extern "C" {

int altro_hip_version(void) { return ALTRO_HIP_VERSION; }
const char* altro_hip_last_error(void) { return g_last_error.c_str(); }

int altro_hip_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

int altro_hip_device_info(int device, char* name, int cap, int* compute_units, int* wave_size) {
  if (device < 0 || device >= altro_hip_device_count())
    return fail(ALTRO_HIP_ERR_NO_DEVICE, "no HIP device %d (count = %d)", device, altro_hip_device_count());
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, device));
  if (name && cap > 0) snprintf(name, cap, "%s (%s)", p.name, p.gcnArchName);
  if (compute_units) *compute_units = p.multiProcessorCount;
  if (wave_size) *wave_size = p.warpSize;
  return 0;
}

int altro_hip_device_pci_bus_id(int device, char* buf, int cap) {
  if (device < 0 || device >= altro_hip_device_count())
    return fail(ALTRO_HIP_ERR_NO_DEVICE, "no HIP device %d (count = %d)", device, altro_hip_device_count());
  if (!buf || cap < 16) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "buf == NULL or cap < 16");
  HIP_TRY(hipDeviceGetPCIBusId(buf, cap, device));
  return 0;
}

// (nx_k, nu_k: per-knot-point dimensions -- altro_hip_batch_create_dims -- or null for the uniform n, m)
static int batch_create_impl(altro_hip_batch** out, int N, int n, int m, int batch, int dtype, int plan,
                             unsigned flags, int device, void* stream, const int* nx_k, const int* nu_k) {
  if (!out) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "out == NULL");
  *out = nullptr;
  if (N <= 0 || n <= 0 || m <= 0 || batch <= 0)
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "N, n, m, batch must be positive (got %d %d %d %d)", N, n, m, batch);
  if (n > kGenericMaxDim || m > kGenericMaxDim) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "n, m <= %d supported (got %d, %d)", kGenericMaxDim, n, m);
  if (dtype != ALTRO_HIP_F64 && dtype != ALTRO_HIP_F32) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "bad dtype %d", dtype);
  if (altro_hip_device_count() <= device || device < 0)
    return fail(ALTRO_HIP_ERR_NO_DEVICE, "no HIP device %d: the altro_hip hot path has no CPU fallback", device);
  // Plan MFMA16 works on the (12, 4) tile; any n <= 12, m <= 4 rides it zero-padded (kernels/pack.hip): no shape next to a
  // fast one falls off a cliff (the reference is dimension-generic, tvlqr.cpp:92-121).  AUTO: the exact tile shape first,
  // then the lane-per-problem plan where it is instantiated (n <= 6, m <= 3: a whole knot point fits one lane's registers),
  // then the padded tile, and GENERIC for everything larger.  (fp32 handles: fp32 storage, fp64 tile arithmetic.)
  const bool mfma_ok = (n <= 12 && m <= 4);
  const bool was_auto = plan == ALTRO_HIP_PLAN_AUTO;
  if (plan == ALTRO_HIP_PLAN_AUTO) {
    // By measured cost of a sweep (tools/shape_cliff.py, profiles/r05i_shape_cliff*.txt; N = 128, fp64, ms at 4096 / 16384 problems):
    //   (6,3) LANE 2.07 / 2.43, tile 0.77 / 2.97    (6,2) 1.40 / 1.73    (5,3) 1.33 / 1.61    (5,2) 1.23 / 1.46    (6,1) 0.87 / 1.13
    //   (5,1) 0.55, (4,3) 0.71, (4,2) 0.26 ... : LANE below the tile's 0.77 at every batch
    // A lane carries the whole 5 x 5 / 6 x 6 blocks, so below one wave per SIMD LANE is a long single-wave chain, while the padded tile
    // costs the (12, 4) sweep whatever the shape and grows linearly past 4096 problems.  Hence: n >= 5 with m >= 2 rides the padded tile
    // up to 6144 problems ((6, 3): 8192), LANE beyond.  A LANE-only compiled-in device model arriving later moves the (still empty)
    // handle to LANE (altro_hip_set_model).
    const bool small_tile = lane_supported(n, m) && dtype == ALTRO_HIP_F64 && n >= 5 && m >= 2 && batch <= ((n == 6 && m == 3) ? 8192 : 6144) &&
                            !(flags & ALTRO_HIP_LANE_FUSED);   // (a flag only plan LANE honours keeps the handle there: ADVICE r5)
    plan = (n == 12 && m == 4) ? ALTRO_HIP_PLAN_MFMA16
           : (lane_supported(n, m) && !small_tile) ? ALTRO_HIP_PLAN_LANE : (mfma_ok ? ALTRO_HIP_PLAN_MFMA16 : ALTRO_HIP_PLAN_GENERIC);
  }
  // Plan MFMA32 (kernels/tvlqr_tile32.hip): the TVLQR pair on 2 x 2 matrix-core tiles for 12 < n <= 31, m <= 8, n + m <= 32 (and
  // n <= 12 with 4 < m <= 8), fp64, uniform dimensions -- on plan GENERIC's arrays, so it is that plan with other sweep kernels.
  bool tile32 = false;
  if (plan == ALTRO_HIP_PLAN_MFMA32) {
    if (!tile32_supported(n, m) || dtype != ALTRO_HIP_F64 || nx_k)
      return fail(ALTRO_HIP_ERR_UNSUPPORTED, "plan MFMA32 takes fp64 problems with n <= 31, m <= 8, n + m <= 32 past the (12, 4) tile and uniform "
                                             "dimensions (got n = %d, m = %d, dtype %d)", n, m, dtype);
    tile32 = true;
    plan = ALTRO_HIP_PLAN_GENERIC;
  } else if (was_auto && plan == ALTRO_HIP_PLAN_GENERIC && !nx_k && dtype == ALTRO_HIP_F64 && tile32_supported(n, m) &&
             !(flags & (ALTRO_HIP_STORE_QBLOCKS | ALTRO_HIP_GENERIC_MATRIX_CORES))) {
    tile32 = true;
  }
  if (plan == ALTRO_HIP_PLAN_MFMA16 && !mfma_ok)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "plan MFMA16 needs n <= 12 and m <= 4 (got %d, %d)", n, m);
  if (plan == ALTRO_HIP_PLAN_LANE && !lane_supported(n, m))
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "plan LANE is instantiated for n <= 6, m <= 3 (got %d, %d)", n, m);
  if (plan != ALTRO_HIP_PLAN_MFMA16 && plan != ALTRO_HIP_PLAN_GENERIC && plan != ALTRO_HIP_PLAN_LANE)
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "bad plan %d", plan);
  HIP_TRY(hipSetDevice(device));
  altro_hip_batch* h = new altro_hip_batch();
  h->N = N; h->n = n; h->m = m; h->batch = batch; h->dtype = dtype; h->plan = plan;
  h->flags = flags; h->device = device; h->auto_plan = was_auto; h->user_stream = stream != nullptr;
  if (nx_k) { h->ragged = true; h->nxv.assign(nx_k, nx_k + N + 1); h->nuv.assign(nu_k, nu_k + N); }
  // plan GENERIC's products on the matrix cores: only when asked for (the plan's default is the CPU path's sums bit for bit: whole
  // AL-iLQR solves then take the oracle's line-search decisions, which sums that differ in the last bits do not always do)
  h->g_mfma = plan == ALTRO_HIP_PLAN_GENERIC && dtype == ALTRO_HIP_F64 && (flags & ALTRO_HIP_GENERIC_MATRIX_CORES) != 0;
  h->g_tile = tile32;
  h->esz = dtype == ALTRO_HIP_F64 ? 8 : 4;
  if (stream) { h->stream = (hipStream_t)stream; }
  else {
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
      delete h;
      return fail(ALTRO_HIP_ERR_HIP, "hipStreamCreate failed");
    }
    h->own_stream = true;
  }
  int rc = 0;
  const size_t B = (size_t)batch, E = h->esz;
#define ALLOC(ptr, bytes) if (!rc) rc = dmalloc(h, (void**)&(ptr), (bytes))
  h->x0_stride = plan == ALTRO_HIP_PLAN_MFMA16 ? MF_N : n;      // plan MFMA16 keeps x0 in tile rows of 12
  ALLOC(h->x0, B * h->x0_stride * E);
  ALLOC(h->delta_V, B * 2 * E);
  ALLOC(h->status, B * sizeof(int));
  ALLOC(h->st_partial, (size_t)kStatsBlocks * kStatsStride * sizeof(double));
  ALLOC(h->st_red, kStatsStride * sizeof(double));
  if (plan == ALTRO_HIP_PLAN_MFMA16) {
    ALLOC(h->m_in, B * N * MF_DYN * E);
    ALLOC(h->m_cin, B * N * MF_COST * E);
    {
      // knot-point-major slabs [k][b][record] (problem-major [b][k] measured the same)
      const int64_t Bq = batch;
      h->m_st = Mfma16Strides{MF_DYN, Bq * MF_DYN, MF_OUT, Bq * MF_OUT, 28, Bq * 28, MF_COST, Bq * MF_COST};
    }
    ALLOC(h->m_term, B * MF_TERM * E);
    ALLOC(h->m_out, B * N * MF_OUT * E);
    ALLOC(h->m_outn, B * MF_TERM * E);
    ALLOC(h->m_xuy, B * (N + 1) * 28 * E);
    ALLOC(h->m_trash, B * MF_OUT * E);
    if (flags & ALTRO_HIP_STORE_QBLOCKS) ALLOC(h->m_qblk, B * N * MF_QB * E);
    if (!rc && (n < MF_N || m < MF_M)) {   // padded shape: whatever no setter writes must read as zero
      (void)hipMemsetAsync(h->x0, 0, B * MF_N * E, h->stream);
      (void)hipMemsetAsync(h->m_xuy, 0, B * (N + 1) * 28 * E, h->stream);
      (void)hipMemsetAsync(h->m_in, 0, B * N * MF_DYN * E, h->stream);
      (void)hipMemsetAsync(h->m_cin, 0, B * N * MF_COST * E, h->stream);
      (void)hipMemsetAsync(h->m_term, 0, B * MF_TERM * E, h->stream);
    }
  } else if (plan == ALTRO_HIP_PLAN_LANE) {
    const LaneSizes z = lane_sizes(n, m);
    // the LANE kernels address one knot point's record through a 2 GiB buffer window with 32-bit offsets
    if (!rc && (uint64_t)B * (uint64_t)std::max(z.e_in, 2 * n + 2 * m + 1 + z.e_out) * E >= (1ull << 31))
      rc = fail(ALTRO_HIP_ERR_UNSUPPORTED, "plan LANE: batch %d too large for one handle (record rows exceed 2 GiB); "
                                           "split the batch over several handles", batch);
    ALLOC(h->l_in, B * N * z.e_in * E);
    ALLOC(h->l_term, B * z.e_term * E);
    ALLOC(h->l_out, B * N * z.e_out * E);
    ALLOC(h->l_outn, B * z.e_term * E);
    ALLOC(h->l_xuy, B * (N + 1) * z.e_xuy * E);
    ALLOC(h->l_x0, B * n * E);
    ALLOC(h->l_nom, B * (N + 1) * (n + m) * E);
    ALLOC(h->l_cost, B * (N + 1) * (2 * n + 2 * m + 1) * E);
    // on the handle's own (non-blocking) stream: a null-stream memset would race the first kernels launched on it
    if (!rc && hipMemsetAsync(h->l_xuy, 0, B * (N + 1) * z.e_xuy * E, h->stream) != hipSuccess) rc = fail(ALTRO_HIP_ERR_HIP, "memset failed");
    if (!rc && hipMemsetAsync(h->l_cost, 0, B * (N + 1) * (2 * n + 2 * m + 1) * E, h->stream) != hipSuccess) rc = fail(ALTRO_HIP_ERR_HIP, "memset failed");
  } else {
    // block sizes of knot point k (tvlqr.cpp:92-121: A_k is nx[k+1] x nx[k], B_k nx[k+1] x nu[k], K_k nu[k] x nx[k], ...)
    std::vector<int> nx(N + 1, n), nu(N + 1, m);
    if (h->ragged) { nx = h->nxv; for (int k = 0; k < N; ++k) nu[k] = h->nuv[k]; }
    auto blk_at = [&](int a, int k) -> int {
      const int nk = nx[k], mk = nu[k], n2 = nx[k < N ? k + 1 : N];
      const int v[G_NUM] = {n2 * nk, n2 * mk, n2, nk * nk, mk * mk, mk * nk, nk, mk, mk * nk, mk, nk * nk, nk,
                            nk * nk, mk * mk, mk * nk, nk, mk, nk * nk, mk * mk, mk * nk, nk, mk, nk, mk, nk};
      return v[a];
    };
    const int nks[G_NUM] = {N, N, N, N + 1, N, N, N + 1, N, N, N, N + 1, N + 1,
                            N, N, N, N, N, N, N, N, N, N, N + 1, N, N + 1};
    std::vector<int64_t> off((size_t)(N + 1) * G_NUM, 0);
    for (int a = 0; a < G_NUM; ++a) {
      const bool qb = (a >= G_Qxx && a <= G_Qu);
      if (a >= G_Qxx_tmp && a <= G_Qu_tmp) continue;   // scratch blocks: only the tvlqr_* drop-in keeps them
      int64_t at = 0;
      for (int k = 0; k <= N; ++k) { off[(size_t)k * G_NUM + a] = at; if (k < nks[a]) at += blk_at(a, k); }
      h->g_bstride[a] = at;
      if (qb && !(flags & ALTRO_HIP_STORE_QBLOCKS)) continue;
      ALLOC(h->g_arr[a], B * (size_t)at * E);
      // everything defined (zero) from the start: the candidate trajectory, so that altro_hip_set_input_guess / _set_state_guess may come
      // in any order with the cost; the gains, so that the forward sweep of a problem whose backward sweep stopped at a failed
      // factorisation (its K, d below the failing knot point are never written) rolls out finite numbers
      if (!rc && hipMemsetAsync(h->g_arr[a], 0, B * (size_t)at * E, h->stream) != hipSuccess)
        rc = fail(ALTRO_HIP_ERR_HIP, "memset failed");
    }
    ALLOC(h->g_off, off.size() * sizeof(int64_t));
    ALLOC(h->g_nx, (size_t)(N + 1) * sizeof(int));
    ALLOC(h->g_nu, (size_t)(N + 1) * sizeof(int));
    if (!rc) {
      if (hipMemcpy(h->g_off, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(h->g_nx, nx.data(), nx.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(h->g_nu, nu.data(), nu.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(ALTRO_HIP_ERR_HIP, "table upload failed");
    }
  }
  {   // per-problem control blocks of the iLQR loop (every plan: GENERIC runs it through kernels/ilqr_generic.hip)
    if (!rc) h->al_knots.assign((size_t)N + 1, AlKnotBig{});
    ALLOC(h->i_prob, B * sizeof(IlqrProb));
    ALLOC(h->i_alpha, B * 8);
    ALLOC(h->i_phi, B * 8 * ILQR_SPEC_TRIALS);
    ALLOC(h->i_spec_sel, B * sizeof(int));
    ALLOC(h->i_spec_refresh, B * sizeof(int));
    ALLOC(h->i_guard, B * sizeof(int));
    ALLOC(h->i_active_exact, B * sizeof(int));
    if (!rc && (hipMemsetAsync(h->i_guard, 0, B * sizeof(int), h->stream) != hipSuccess || hipMemsetAsync(h->i_active_exact, 0, B * sizeof(int), h->stream) != hipSuccess)) rc = fail(ALTRO_HIP_ERR_HIP, "memset failed");
    ALLOC(h->i_stat_done, B * sizeof(int));
    ALLOC(h->i_dphi, B * 8 * 2);
    ALLOC(h->i_active, B * sizeof(int));
    ALLOC(h->i_counters, (size_t)kCounterSlots * 8 * sizeof(int));
    if (!rc && hipMemsetAsync(h->i_counters, 0, (size_t)kCounterSlots * 8 * sizeof(int), h->stream) != hipSuccess) rc = fail(ALTRO_HIP_ERR_HIP, "memset failed");
    ALLOC(h->i_reg, B * 8);
    if (!rc) {   // every constraint starts with penalty 1 (knotpoint_data.cpp:343)
      std::vector<IlqrProb> pr((size_t)B);
      std::memset(pr.data(), 0, pr.size() * sizeof(IlqrProb));
      for (auto& q : pr) { q.rho = 1.0; q.rho_est = 1.0; q.status = 1; }
      if (hipMemcpy(h->i_prob, pr.data(), pr.size() * sizeof(IlqrProb), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(ALTRO_HIP_ERR_HIP, "control block upload failed");
    }
  }
#undef ALLOC
  if (!rc && (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess))
    rc = fail(ALTRO_HIP_ERR_HIP, "hipEventCreate failed");
  if (rc) {
    altro_hip_batch_destroy(h);
    return rc;
  }
  *out = h;
  return 0;
}

int altro_hip_batch_create(altro_hip_batch** out, int N, int n, int m, int batch, int dtype, int plan,
                           unsigned flags, int device, void* stream) {
  return batch_create_impl(out, N, n, m, batch, dtype, plan, flags, device, stream, nullptr, nullptr);
}

extern "C++" {
namespace altro_hip { namespace capi {
// A handle created with ALTRO_HIP_PLAN_AUTO that nothing has been set on yet becomes a handle of `plan` in place (same address):
// a fresh handle is created and the two exchange their contents.
int replan_empty_handle(altro_hip_batch* h, int plan) {
  // "empty": nothing but the initial state (kept on the host for exactly this move) and configuration calls has reached the handle
  if (!h->auto_plan || h->dyn_set || h->cost_set || h->model_set || h->lqr_cost_set || h->guess_set || !h->al_defs.empty() ||
      (h->x0_set && h->x0_host.empty()))
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "this handle runs plan %d; create it with plan %d (or call altro_hip_set_model before the dynamics, the cost, "
                                           "a guess or a constraint block reach a handle created with ALTRO_HIP_PLAN_AUTO)", h->plan, plan);
  altro_hip_batch* fresh = nullptr;
  int rc = batch_create_impl(&fresh, h->N, h->n, h->m, h->batch, h->dtype, plan, h->flags, h->device, h->user_stream ? (void*)h->stream : nullptr,
                             nullptr, nullptr);
  if (rc) return rc;
  // what configuration calls set before the move stays set (pointer mode, host-batch tiling, forms, profiling)
  fresh->dev_ptrs = h->dev_ptrs; fresh->host_batch = h->host_batch; fresh->forms = h->forms;
  const int prof = h->prof;
  std::vector<double> x0 = std::move(h->x0_host);
  const int x0_bz = h->x0_host_bz;
  std::swap(*h, *fresh);
  h->auto_plan = false;
  altro_hip_batch_destroy(fresh);   // (the old buffers; a caller-supplied stream is shared and not destroyed: own_stream is false on both)
  if (prof && (rc = altro_hip_profile_enable(h, prof))) return rc;
  if (!x0.empty()) {
    const bool dp = h->dev_ptrs;
    h->dev_ptrs = false;
    rc = altro_hip_set_initial_state(h, x0.data(), x0_bz);
    h->dev_ptrs = dp;
  }
  return rc;
}
} }
}  // extern "C++"

int altro_hip_batch_create_dims(altro_hip_batch** out, int N, const int* nx, const int* nu, int batch, int dtype,
                                unsigned flags, int device, void* stream) {
  if (!out) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "out == NULL");
  *out = nullptr;
  if (N <= 0 || !nx || !nu) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "N must be positive, nx and nu non-null");
  int nmax = 0, mmax = 0;
  for (int k = 0; k <= N; ++k) {
    if (nx[k] <= 0 || nx[k] > kGenericMaxDim) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "nx[%d] = %d outside [1, %d]", k, nx[k], kGenericMaxDim);
    nmax = std::max(nmax, nx[k]);
    if (k < N) {
      if (nu[k] <= 0 || nu[k] > kGenericMaxDim) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "nu[%d] = %d outside [1, %d]", k, nu[k], kGenericMaxDim);
      mmax = std::max(mmax, nu[k]);
    }
  }
  return batch_create_impl(out, N, nmax, mmax, batch, dtype, ALTRO_HIP_PLAN_GENERIC, flags, device, stream, nx, nu);
}

void altro_hip_batch_destroy(altro_hip_batch* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  void* ptrs[] = {h->x0, h->delta_V, h->status, h->m_in, h->m_cin, h->m_term, h->m_out, h->m_outn, h->m_xuy,
                  h->m_qblk, h->m_trash, h->g_off, h->g_nx, h->g_nu, h->stage,
                  h->l_in, h->l_term, h->l_out, h->l_outn, h->l_xuy, h->l_x0,
                  h->l_nom, h->l_cost, h->g_ws, h->i_prob, h->i_alpha, h->i_phi, h->i_dphi, h->i_active, h->i_counters,
                  h->al_d_knots, h->al_d_big, h->al_d_gsel, h->g_stat_part, h->al_d_G, h->al_d_Gpad, h->al_d_g, h->al_d_z, h->i_reg, h->m_nom, h->m_costp,
                  h->i_cand_spec, h->i_spec_sel, h->i_spec_refresh, h->i_fused_list, h->i_guard, h->i_active_exact, h->i_stat_done, h->st_partial, h->st_red, h->i_merit_jk, h->i_spec_jac, h->i_results,
                  h->m_costd, h->m_costd_term, h->l_costq, h->i_sens, h->i_sens_alpha, h->i_aff_part, h->i_aff_on, h->g_xn, h->g_un, h->g_cQ, h->g_cR, h->g_cH, h->g_cq, h->g_cr, h->g_cc};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h->i_results_host) (void)hipHostFree(h->i_results_host);
  if (h->poll_host) (void)hipHostFree(h->poll_host);
  if (h->poll_count_host) (void)hipHostFree(h->poll_count_host);
  if (h->cnt_host) (void)hipHostFree(h->cnt_host);
  for (hipEvent_t e : h->cnt_ev) if (e) (void)hipEventDestroy(e);
  for (int a = 0; a < G_NUM; ++a) if (h->g_arr[a]) (void)hipFree(h->g_arr[a]);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  for (hipEvent_t e : h->prof_ev) if (e) (void)hipEventDestroy(e);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int altro_hip_batch_plan(const altro_hip_batch* h) { return h ? (h->g_tile ? (int)ALTRO_HIP_PLAN_MFMA32 : h->plan) : ALTRO_HIP_ERR_BAD_ARGUMENT; }
size_t altro_hip_batch_device_bytes(const altro_hip_batch* h) { return h ? h->device_bytes : 0; }

int altro_hip_set_dynamics(altro_hip_batch* h, const double* A, const double* B, const double* f,
                           int kz, int bz) {
  int rc = check(h);
  if (rc) return rc;
  h->expansion_current = false;
  if (!A || !B) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "A and B are required");
  const int n = h->n, m = h->m, N = h->N;
  h->has_f = f ? 1 : 0;
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {
    DevSrc dA, dB, df;
    const int nkh = kz ? 1 : N;
    rc = put_src(h, A, n * n, nkh, kz, bz, &dA);
    if (!rc) rc = put_src(h, B, n * m, nkh, kz, bz, &dB);
    if (!rc) rc = put_src(h, f, n, nkh, kz, bz, &df);
    if (!rc) rc = mfma16_pack_launch(h, MSEG_Z, dA.s, dB.s);
    if (!rc) rc = mfma16_pack_launch(h, MSEG_F, df.s, SrcArr{nullptr, 0, 0, 0});
    if (!rc) HIP_TRY(hipStreamSynchronize(h->stream));
  } else if (h->plan == ALTRO_HIP_PLAN_LANE) {
    const LaneSizes z = lane_sizes(n, m);
    const int nkh = kz ? 1 : N;
    auto pk = [&](const double* src, int len, int off) -> int {
      return h->dtype == ALTRO_HIP_F64
                 ? lane_pack<double>(h, (double*)h->l_in, z.e_in, src, len, off, 0, N, 0, nkh, kz, bz)
                 : lane_pack<float>(h, (float*)h->l_in, z.e_in, src, len, off, 0, N, 0, nkh, kz, bz);
    };
    rc = pk(A, n * n, 0);
    if (!rc) rc = pk(B, n * m, n * n);
    if (!rc) rc = pk(f, n, n * n + n * m);
  } else if (h->ragged) {   // per-knot-point dimensions: the caller's packed [b][k][block_k] IS the device layout, one block per problem
    if (kz) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "k_stride_zero needs uniform dimensions");
    auto whole = [&](int arr, const double* src) -> int {
      if (!src) { HIP_TRY(hipMemsetAsync(h->g_arr[arr], 0, (size_t)h->batch * h->g_bstride[arr] * h->esz, h->stream)); return 0; }
      return h->dtype == ALTRO_HIP_F64 ? generic_set<double>(h, arr, src, (int)h->g_bstride[arr], 1, 0, bz)
                                       : generic_set<float>(h, arr, src, (int)h->g_bstride[arr], 1, 0, bz);
    };
    rc = whole(G_A, A);
    if (!rc) rc = whole(G_B, B);
    if (!rc) rc = whole(G_f, f);
  } else if (h->dtype == ALTRO_HIP_F64) {
    rc = generic_set<double>(h, G_A, A, n * n, N, kz, bz);
    if (!rc) rc = generic_set<double>(h, G_B, B, n * m, N, kz, bz);
    if (!rc) {
      if (f) rc = generic_set<double>(h, G_f, f, n, N, kz, bz);
      else HIP_TRY(hipMemsetAsync(h->g_arr[G_f], 0, (size_t)h->batch * N * n * h->esz, h->stream));
    }
  } else {
    rc = generic_set<float>(h, G_A, A, n * n, N, kz, bz);
    if (!rc) rc = generic_set<float>(h, G_B, B, n * m, N, kz, bz);
    if (!rc) {
      if (f) rc = generic_set<float>(h, G_f, f, n, N, kz, bz);
      else HIP_TRY(hipMemsetAsync(h->g_arr[G_f], 0, (size_t)h->batch * N * n * h->esz, h->stream));
    }
  }
  if (!rc) { HIP_TRY(hipStreamSynchronize(h->stream)); h->dyn_set = true; }
  return rc;
}

int altro_hip_set_cost(altro_hip_batch* h, const double* Q, const double* R, const double* H,
                       const double* q, const double* r, int is_diag, int kz, int bz) {
  int rc = check(h);
  if (rc) return rc;
  h->expansion_current = false;
  if (!Q || !R || !q || !r) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "Q, R, q, r are required");
  if (!is_diag && !H) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "H is required for a dense cost");
  h->ilqr_linear = false;   // explicit cost blocks = TVLQR semantics again (altro_hip_set_tracking_cost sets it back)
  const int n = h->n, m = h->m, N = h->N;
  Dims d{n, m};
  h->is_diag = is_diag ? 1 : 0;
  // Host Q / q hold N+1 knot points per problem; with k_stride_zero they hold TWO: the running block
  // and the terminal block (a shared running cost with its own terminal cost is the common case).
  const int nkQ = kz ? 2 : N + 1;
  const int nkR = kz ? 1 : N;
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {
    DevSrc dQ, dR, dH, dq, dr;
    rc = put_src(h, Q, d.Q(is_diag), nkQ, kz, bz, &dQ);
    if (!rc) rc = put_src(h, R, d.R(is_diag), nkR, kz, bz, &dR);
    if (!rc && !is_diag) rc = put_src(h, H, d.H(), nkR, kz, bz, &dH);
    if (!rc) rc = put_src(h, q, n, nkQ, kz, bz, &dq);
    if (!rc) rc = put_src(h, r, m, nkR, kz, bz, &dr);
    SrcArr none{nullptr, 0, 0, 0};
    SrcArr tQ = dQ.s, tq = dq.s;  // terminal views: knot point N, or block 1 of the broadcast pair
    if (kz) { tQ.p += d.Q(is_diag); tq.p += n; }
    if (!rc) rc = mfma16_pack_launch(h, MSEG_Q, dQ.s, none);
    if (!rc) rc = mfma16_pack_launch(h, MSEG_TERM_Q, tQ, none);
    if (!rc) rc = mfma16_pack_launch(h, MSEG_HR, dH.s, dR.s);
    if (!rc) rc = mfma16_pack_launch(h, MSEG_QR, dq.s, dr.s);
    if (!rc) rc = mfma16_pack_launch(h, MSEG_TERM_q, tq, none);
    if (!rc) HIP_TRY(hipStreamSynchronize(h->stream));
  } else if (h->plan == ALTRO_HIP_PLAN_LANE) {
    const LaneSizes z = lane_sizes(n, m);
    const int oQ = n * n + n * m + n, oR = oQ + n * n, oH = oR + m * m, oq = oH + m * n, or_ = oq + n;
    auto pk = [&](void* dst, int E, const double* src, int len, int off, int diag, int nk, int k_src0,
                  int nk_host, int src_off = 0) -> int {
      return h->dtype == ALTRO_HIP_F64
                 ? lane_pack<double>(h, (double*)dst, E, src, len, off, diag, nk, k_src0, nk_host, kz, bz, src_off)
                 : lane_pack<float>(h, (float*)dst, E, src, len, off, diag, nk, k_src0, nk_host, kz, bz, src_off);
    };
    // terminal source knot point: index N, or block 1 of the broadcast pair (ks == 0 there, so the
    // device source pointer is shifted by one block instead)
    const int kt = kz ? 0 : N;
    rc = pk(h->l_in, z.e_in, Q, d.Q(is_diag), oQ, is_diag ? n : 0, N, 0, nkQ);
    if (!rc) rc = pk(h->l_in, z.e_in, R, d.R(is_diag), oR, is_diag ? m : 0, N, 0, nkR);
    if (!rc) rc = pk(h->l_in, z.e_in, is_diag ? nullptr : H, d.H(), oH, 0, N, 0, nkR);
    if (!rc) rc = pk(h->l_in, z.e_in, q, n, oq, 0, N, 0, nkQ);
    if (!rc) rc = pk(h->l_in, z.e_in, r, m, or_, 0, N, 0, nkR);
    if (!rc) rc = pk(h->l_term, z.e_term, Q, d.Q(is_diag), 0, is_diag ? n : 0, 1, kt, nkQ, kz ? d.Q(is_diag) : 0);
    if (!rc) rc = pk(h->l_term, z.e_term, q, n, n * n, 0, 1, kt, nkQ, kz ? n : 0);
  } else if (h->ragged) {   // per-knot-point dimensions: every array moves as one packed block per problem
    if (kz) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "k_stride_zero needs uniform dimensions");
    {   // where Q_k / R_k start: n_k (m_k) entries for a diagonal cost, n_k^2 (m_k^2) for a dense one, packed in the allocated blocks
      std::vector<int64_t> off((size_t)(N + 1) * G_NUM);
      // a sweep enqueued by an earlier asynchronous altro_hip_backward still reads the table: the handle's stream is non-blocking, so
      // the null-stream copies below would not wait for it
      HIP_TRY(hipStreamSynchronize(h->stream));
      HIP_TRY(hipMemcpy(off.data(), h->g_off, off.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
      int64_t aq = 0, ar = 0;
      for (int k = 0; k <= N; ++k) {
        off[(size_t)k * G_NUM + G_Q] = aq; off[(size_t)k * G_NUM + G_R] = ar;
        aq += is_diag ? h->nxv[k] : h->nxv[k] * h->nxv[k];
        if (k < N) ar += is_diag ? h->nuv[k] : h->nuv[k] * h->nuv[k];
      }
      h->g_bstride[G_Q] = aq; h->g_bstride[G_R] = ar;
      HIP_TRY(hipMemcpy(h->g_off, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    auto whole = [&](int arr, const double* src) -> int {
      if (!src) { HIP_TRY(hipMemsetAsync(h->g_arr[arr], 0, (size_t)h->batch * h->g_bstride[arr] * h->esz, h->stream)); return 0; }
      return h->dtype == ALTRO_HIP_F64 ? generic_set<double>(h, arr, src, (int)h->g_bstride[arr], 1, 0, bz)
                                       : generic_set<float>(h, arr, src, (int)h->g_bstride[arr], 1, 0, bz);
    };
    rc = whole(G_Q, Q);
    if (!rc) rc = whole(G_q, q);
    if (!rc) rc = whole(G_R, R);
    if (!rc) rc = whole(G_H, is_diag ? nullptr : H);
    if (!rc) rc = whole(G_r, r);
  } else {
    auto set = [&](int arr, const double* src, int blk, int nk, int k0, int nk_host, int src_off = 0) -> int {
      if (!src) {
        HIP_TRY(hipMemsetAsync(h->g_arr[arr], 0, (size_t)h->batch * nk * blk * h->esz, h->stream));
        return 0;
      }
      return h->dtype == ALTRO_HIP_F64 ? generic_set<double>(h, arr, src, blk, nk, kz, bz, k0, nk_host, src_off)
                                       : generic_set<float>(h, arr, src, blk, nk, kz, bz, k0, nk_host, src_off);
    };
    // NOTE: the device block of Q / R is always dense-sized (n*n / m*m); a diagonal cost keeps its
    // diagonal in the head of the block, like the reference does (knotpoint_data.cpp:92-95).
    const int qb = d.Q(is_diag), rb = d.R(is_diag);
    h->g_bstride[G_Q] = (int64_t)(N + 1) * qb;
    h->g_bstride[G_R] = (int64_t)N * rb;
    {
      std::vector<int64_t> off((size_t)(N + 1) * G_NUM);
      HIP_TRY(hipMemcpy(off.data(), h->g_off, off.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
      for (int k = 0; k <= N; ++k) {
        off[(size_t)k * G_NUM + G_Q] = (int64_t)k * qb;
        off[(size_t)k * G_NUM + G_R] = (int64_t)k * rb;
      }
      HIP_TRY(hipMemcpy(h->g_off, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    if (kz) {
      rc = set(G_Q, Q, qb, N, 0, 2);
      if (!rc) rc = set(G_Q, Q, qb, 1, N, 2, qb);
      if (!rc) rc = set(G_q, q, n, N, 0, 2);
      if (!rc) rc = set(G_q, q, n, 1, N, 2, n);
    } else {
      rc = set(G_Q, Q, qb, N + 1, 0, -1);
      if (!rc) rc = set(G_q, q, n, N + 1, 0, -1);
    }
    if (!rc) rc = set(G_R, R, rb, N, 0, -1);
    if (!rc) rc = set(G_H, is_diag ? nullptr : H, d.H(), N, 0, -1);
    if (!rc) rc = set(G_r, r, m, N, 0, -1);
  }
  if (!rc) { HIP_TRY(hipStreamSynchronize(h->stream)); h->cost_set = true; }
  return rc;
}

int altro_hip_set_host_batch(altro_hip_batch* h, int host_batch) {
  if (!h || host_batch < 0) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "bad handle or host_batch");
  h->host_batch = host_batch;
  return 0;
}

int altro_hip_set_initial_state(altro_hip_batch* h, const double* x0, int bz) {
  int rc = check(h);
  if (rc) return rc;
  h->expansion_current = false;
  if (!x0) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "x0 == NULL");
  const int n0 = h->ragged ? h->nxv[0] : h->n;   // (per-knot-point dimensions: x0 has nx[0] entries per problem)
  auto consume = [&](SrcArr s, int b0, int nb) -> int {
    const int64_t total = (int64_t)nb * n0;
    if (h->dtype == ALTRO_HIP_F64)
      hipLaunchKernelGGL(expand_copy_kernel<double>, dim3(grid_for(total)), dim3(256), 0, h->stream,
                         (double*)h->x0, (int64_t)h->x0_stride, (int64_t)h->x0_stride, s, n0, 1, b0, nb);
    else
      hipLaunchKernelGGL(expand_copy_kernel<float>, dim3(grid_for(total)), dim3(256), 0, h->stream,
                         (float*)h->x0, (int64_t)h->x0_stride, (int64_t)h->x0_stride, s, n0, 1, b0, nb);
    if (hipGetLastError() != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "x0 copy launch failed");
    return 0;
  };
  rc = upload_chunks(h, x0, n0, 1, 1, bz, -1, 0, consume);
  if (!rc && h->plan == ALTRO_HIP_PLAN_LANE)
    rc = h->dtype == ALTRO_HIP_F64
             ? lane_pack<double>(h, (double*)h->l_x0, h->n, x0, h->n, 0, 0, 1, 0, 1, 1, bz)
             : lane_pack<float>(h, (float*)h->l_x0, h->n, x0, h->n, 0, 0, 1, 0, 1, 1, bz);
  if (!rc) {
    h->x0_set = true;
    // a handle that may still move to another plan (altro_hip_set_model on an ALTRO_HIP_PLAN_AUTO handle) keeps the initial state
    // on the host, so that "x0 first, model second" works like "model first" (host pointers only: a device array is the caller's)
    h->x0_host.clear();
    if (h->auto_plan && !h->dev_ptrs && !h->ragged) {
      h->x0_host.assign(x0, x0 + (size_t)(bz ? 1 : (h->host_batch > 0 && h->host_batch < h->batch ? h->host_batch : h->batch)) * n0);
      h->x0_host_bz = bz;
      if (!bz && h->host_batch > 0 && h->host_batch < h->batch) h->x0_host.clear();   // (tiled upload: not kept)
    }
  }
  return rc;
}

int altro_hip_set_pointer_mode(altro_hip_batch* h, int device_pointers) {
  int rc = check(h);
  if (rc) return rc;
  h->dev_ptrs = device_pointers != 0;
  return 0;
}

int altro_hip_synchronize(altro_hip_batch* h) {
  int rc = check(h);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

#define GETTER(NAME, GARR, MWHAT, BLOCK, NK, NEED_FWD)                                          \
  int altro_hip_get_##NAME(altro_hip_batch* h, double* dst) {                                  \
    int rc = check(h);                                                                          \
    if (rc) return rc;                                                                          \
    if (!dst) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "NULL destination");                      \
    if (!(NEED_FWD ? h->forward_done : h->backward_done))                                       \
      return fail(ALTRO_HIP_ERR_NOT_SET, "nothing computed yet for get_" #NAME);                \
    if (h->plan == ALTRO_HIP_PLAN_MFMA16) return mfma16_get(h, MWHAT, dst, BLOCK, NK);          \
    if (h->plan == ALTRO_HIP_PLAN_LANE) return lane_get_any(h, MWHAT, dst);                     \
    if (h->ragged) /* per-knot-point dimensions: the packed array of a problem as one block */   \
      return h->dtype == ALTRO_HIP_F64 ? generic_get<double>(h, GARR, dst, (int)h->g_bstride[GARR], 1)   \
                                       : generic_get<float>(h, GARR, dst, (int)h->g_bstride[GARR], 1);   \
    return h->dtype == ALTRO_HIP_F64 ? generic_get<double>(h, GARR, dst, BLOCK, NK)            \
                                     : generic_get<float>(h, GARR, dst, BLOCK, NK);            \
  }
GETTER(K, G_K, MGET_K, h->m * h->n, h->N, false)
GETTER(d, G_d, MGET_d, h->m, h->N, false)
GETTER(P, G_P, MGET_P, h->n * h->n, h->N + 1, false)
GETTER(p, G_p, MGET_p, h->n, h->N + 1, false)
GETTER(x, G_x, MGET_x, h->n, h->N + 1, true)
GETTER(u, G_u, MGET_u, h->m, h->N, true)
GETTER(y, G_y, MGET_y, h->n, h->N + 1, true)
#undef GETTER

int altro_hip_get_delta_V(altro_hip_batch* h, double* dV) {
  int rc = check(h);
  if (rc) return rc;
  if (!h->backward_done) return fail(ALTRO_HIP_ERR_NOT_SET, "backward has not run");
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->dtype == ALTRO_HIP_F64) {
    HIP_TRY(hipMemcpy(dV, h->delta_V, (size_t)h->batch * 2 * 8, hipMemcpyDeviceToHost));
  } else {
    std::vector<float> tmp((size_t)h->batch * 2);
    HIP_TRY(hipMemcpy(tmp.data(), h->delta_V, tmp.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tmp.size(); ++i) dV[i] = tmp[i];
  }
  return 0;
}

int altro_hip_get_status(altro_hip_batch* h, int* status) {
  int rc = check(h);
  if (rc) return rc;
  if (!h->backward_done) return fail(ALTRO_HIP_ERR_NOT_SET, "backward has not run");
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(status, h->status, (size_t)h->batch * sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}

int altro_hip_get_qblocks(altro_hip_batch* h, double* dst) {
  int rc = check(h);
  if (rc) return rc;
  if (!(h->flags & ALTRO_HIP_STORE_QBLOCKS)) return fail(ALTRO_HIP_ERR_NOT_SET, "handle was created without ALTRO_HIP_STORE_QBLOCKS");
  if (!h->backward_done) return fail(ALTRO_HIP_ERR_NOT_SET, "backward has not run");
  const int n = h->n, m = h->m, N = h->N;
  const int per = n * n + m * m + m * n + n + m;
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) return mfma16_get(h, MGET_QBLK, dst, per, N);
  if (h->plan == ALTRO_HIP_PLAN_LANE) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "plan LANE does not store the Q-blocks");
  if (h->ragged) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "altro_hip_get_qblocks interleaves uniform blocks; not available with per-knot-point dimensions");
  // generic: five separate reference-layout arrays -> interleave on the host
  std::vector<double> tmp((size_t)h->batch * N * n * n);
  const int arrs[5] = {G_Qxx, G_Quu, G_Qux, G_Qx, G_Qu};
  const int blks[5] = {n * n, m * m, m * n, n, m};
  int offp = 0;
  for (int i = 0; i < 5; ++i) {
    rc = h->dtype == ALTRO_HIP_F64 ? generic_get<double>(h, arrs[i], tmp.data(), blks[i], N)
                                   : generic_get<float>(h, arrs[i], tmp.data(), blks[i], N);
    if (rc) return rc;
    for (size_t bk = 0; bk < (size_t)h->batch * N; ++bk)
      memcpy(dst + bk * per + offp, tmp.data() + bk * blks[i], sizeof(double) * blks[i]);
    offp += blks[i];
  }
  return 0;
}

int altro_hip_profile_enable(altro_hip_batch* h, int enable) {
  int rc = check(h);
  if (rc) return rc;
  if (enable < 0 || enable > 2) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "profile mode %d (0 off, 1 per-launch sync, 2 events only)", enable);
  if (enable == 2 && h->prof_ev.empty()) {
    constexpr int kLaunches = 4096;   // launches one asynchronous window can hold (later ones go unrecorded)
    h->prof_ev.resize(2 * kLaunches, nullptr);
    h->prof_slot.assign(kLaunches, 0);
    for (auto& e : h->prof_ev)
      if (hipEventCreate(&e) != hipSuccess) {
        for (auto& d : h->prof_ev) if (d) (void)hipEventDestroy(d);
        h->prof_ev.clear();
        return fail(ALTRO_HIP_ERR_HIP, "hipEventCreate failed");
      }
  }
  if (h->prof == 2 && enable != 2) {   // leave nothing half-recorded behind
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->prof_n = 0;
  }
  h->prof = enable;
  return 0;
}
int altro_hip_profile_reset(altro_hip_batch* h) {
  if (!h) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "null handle");
  if (h->prof == 2) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); }
  h->prof_n = 0;
  h->prof_launches[0] = h->prof_launches[1] = 0;
  h->prof_dropped[0] = h->prof_dropped[1] = 0;
  h->prof_ms[0] = h->prof_ms[1] = 0.0;
  h->prof_min[0] = h->prof_min[1] = h->prof_max[0] = h->prof_max[1] = 0.0;
  return 0;
}
namespace {
int profile_drain(altro_hip_batch* h) {   // mode 2: turn the recorded event pairs into per-slot totals
  if (h->prof_n == 0) return 0;
  int rc = check(h);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int i = 0; i < h->prof_n; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->prof_ev[2 * i], h->prof_ev[2 * i + 1]) != hipSuccess) continue;
    const int slot = h->prof_slot[i];
    h->prof_min[slot] = h->prof_launches[slot] ? std::min(h->prof_min[slot], (double)ms) : (double)ms;
    h->prof_max[slot] = std::max(h->prof_max[slot], (double)ms);
    h->prof_ms[slot] += ms;
    h->prof_launches[slot] += 1;
  }
  h->prof_n = 0;
  return 0;
}
}  // namespace
int altro_hip_profile_get(altro_hip_batch* h, int slot, int* launches, double* total_ms,
                          const char** kernel_name) {
  if (!h || slot < 0 || slot > 1) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "bad handle or slot");
  int rc = profile_drain(h);
  if (rc) return rc;
  if (launches) *launches = h->prof_launches[slot];
  if (total_ms) *total_ms = h->prof_ms[slot];
  if (kernel_name) {
    static const char* names[3][2] = {{"generic_backward_kernel", "generic_forward_kernel"},
                                      {"mfma16_backward_kernel", "mfma16_forward_kernel"},
                                      {"lane_backward_kernel", "lane_forward_kernel"}};
    *kernel_name = names[h->plan == ALTRO_HIP_PLAN_MFMA16 ? 1 : (h->plan == ALTRO_HIP_PLAN_LANE ? 2 : 0)][slot];
    if (h->g_tile && !(slot == 0 && h->is_diag)) *kernel_name = slot == 0 ? "tile32_backward_kernel" : "tile32_forward_kernel";
    if (slot == 0 && h->plan == ALTRO_HIP_PLAN_LANE && h->bwd_quad) *kernel_name = h->n == 4 ? "quad_backward_kernel" : "quad2_backward_kernel";
    if (slot == 1 && h->plan == ALTRO_HIP_PLAN_LANE && h->fwd_quad) *kernel_name = "quad_forward_kernel";
    if (slot == 1 && mfma16_forward_is_x4(h)) *kernel_name = "mfma16_forward_f32x4_kernel";
    if (slot == 0 && h->plan == ALTRO_HIP_PLAN_MFMA16 && h->dtype == ALTRO_HIP_F32 && (h->flags & ALTRO_HIP_F32_PURE) &&
        !(h->flags & ALTRO_HIP_STORE_QBLOCKS))
      *kernel_name = (h->batch % 4 == 0) ? "mfma16_backward_f32x4_kernel"
                                                                                    : "mfma16_backward_f32_kernel";
  }
  return 0;
}
int altro_hip_profile_dropped(altro_hip_batch* h, int slot) {
  if (!h || slot < 0 || slot > 1) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "bad handle or slot");
  return h->prof_dropped[slot];
}
int altro_hip_profile_get_range(altro_hip_batch* h, int slot, double* min_ms, double* max_ms) {
  if (!h || slot < 0 || slot > 1) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "bad handle or slot");
  int rc = profile_drain(h);
  if (rc) return rc;
  if (min_ms) *min_ms = h->prof_min[slot];
  if (max_ms) *max_ms = h->prof_max[slot];
  return 0;
}

double altro_hip_algorithmic_bytes(const altro_hip_batch* h, int slot) {
  if (!h) return 0.0;
  const double n = h->n, m = h->m, w = (double)h->esz;
  // backward 3n^2+3nm+m^2+3n+2m, forward-LTV 2n^2+2nm+4n+2m elements per kp
  const double bwd = 3 * n * n + 3 * n * m + m * m + 3 * n + 2 * m;
  const double fwd = 2 * n * n + 2 * n * m + 4 * n + 2 * m;
  return (slot == 0 ? bwd : fwd) * w * (double)h->N * (double)h->batch;
}

}  // extern "C"
