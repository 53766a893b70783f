// capi.hip -- implementation of the C ABI declared in include/altro_hip/altro_hip.h.
// Host-side plumbing only: handle lifetime, device buffers, layout conversion launches, plan
// dispatch, error strings.  All arithmetic lives in kernels/*.hip.  There is no CPU fallback: every
// compute entry point needs a HIP device and says so when there is none.
#include "altro_hip/altro_hip.h"

#include <hip/hip_runtime.h>

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
#include "linesearch_sm.h"
#include "kernels/tvlqr_mfma16.hip"
#include "kernels/tvlqr_mfma16_f32.hip"

using namespace altro_hip;

namespace {

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

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(ALTRO_HIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                  __FILE__, __LINE__);                                                        \
  } while (0)

constexpr size_t kStageBytes = size_t(256) << 20;

}  // namespace

struct altro_hip_batch {
  int N = 0, n = 0, m = 0, batch = 0, dtype = 0, plan = 0, device = 0;
  unsigned flags = 0;
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
  int *i_active = nullptr, *i_counters = nullptr;
  // speculative backtracking (altro_hip_ilqr_solve): spare candidate trajectories, allocated on first use
  void* i_cand_spec = nullptr;
  int *i_spec_sel = nullptr, *i_spec_refresh = nullptr;
  int spec_trials = 1;            // trials per merit launch of the CURRENT launch (1 = no speculation)
  int spec_pre = 0;               // the current launch is phi(0) fused with the first trial step
  bool spec_no_memory = false;    // the spare trajectories could not be allocated: no speculation on this handle
  double spec_beta = 0.5;         // the running solve's LsOptions::beta_decrease / max_iters (what the merit kernels
  int spec_max_iters = 25;        // need to reproduce the state machine's step sequence)
  ModelParams model{MODEL_LINEAR, 0.0f, 0, 2.7, 1.5};
  bool model_set = false, lqr_cost_set = false, guess_set = false;
  // augmented-Lagrangian constraint blocks (plan LANE): host mirrors + device tables, uploaded lazily
  std::vector<AlDef> al_defs;
  std::vector<AlKnot> al_knots;
  int al_uniform = 0, al_rows_per_knot = 0;  // knot points 0..N-1 carry the same blocks (kernels/al_types.h)
  std::vector<double> al_G;                  // pool of G blocks, column-major p x (n+m)
  std::vector<std::vector<double>> al_g;     // per block: [p] or [batch][p]
  int al_rows = 0;
  bool al_dirty = false;
  AlKnot* al_d_knots = nullptr;
  void *al_d_G = nullptr, *al_d_g = nullptr, *al_d_z = nullptr;
  const int* bwd_active = nullptr;           // per-problem mask for the backward sweep inside ilqr_solve
  const double* bwd_reg = nullptr;           // per-problem regularisation inside ilqr_solve (retry extension)
  // iLQR loop on plan MFMA16 (dynamics given as data): nominal trajectory + cost parameters, allocated by
  // altro_hip_set_tracking_cost; ilqr_linear = the backward sweep ignores the affine term (knotpoint_data.cpp:416)
  void *m_nom = nullptr, *m_costp = nullptr;
  bool ilqr_linear = false;
  double* i_reg = nullptr;
  // staging for host <-> device conversion (grown lazily, never inside the hot path)
  void* stage = nullptr;
  size_t stage_bytes = 0;
  size_t device_bytes = 0;
  // profiling
  bool prof = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int last_sweeps = 0, last_merit_launches = 0;
  int prof_launches[2] = {0, 0};
  double prof_ms[2] = {0, 0};
};

namespace {

int dmalloc(altro_hip_batch* h, void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes ? bytes : 16);
  if (e != hipSuccess) {
    *p = nullptr;
    return fail(ALTRO_HIP_ERR_OUT_OF_MEMORY, "hipMalloc(%zu bytes) failed: %s", bytes,
                hipGetErrorString(e));
  }
  h->device_bytes += bytes;
  return 0;
}

int check(altro_hip_batch* h) {
  if (!h) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "null handle");
  hipError_t e = hipSetDevice(h->device);
  if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "hipSetDevice(%d): %s", h->device, hipGetErrorString(e));
  return 0;
}

int ensure_stage(altro_hip_batch* h, size_t bytes) {
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
int upload_chunks(altro_hip_batch* h, const double* host, int block, int nk, int k_zero, int b_zero,
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
int generic_set(altro_hip_batch* h, int arr, const double* host, int block, int nk, int k_zero,
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
int aos_set(altro_hip_batch* h, T* dst, int64_t dst_bs, int64_t dst_ks, const double* host, int block, int nk,
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
template <typename T>
int aos_get(altro_hip_batch* h, double* host, const T* src, int64_t src_bs, int64_t src_ks, int block, int nk) {
  return download_chunks(h, host, block, nk, [&](double* dst, int b0, int nb) {
    const int64_t total = (int64_t)nb * nk * block;
    hipLaunchKernelGGL(gather_copy_kernel<T>, dim3(grid_for(total)), dim3(256), 0, h->stream, dst,
                       (int64_t)nk * block, (int64_t)block, src, src_bs, src_ks, block, nk, b0, nb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "gather_copy launch: %s", hipGetErrorString(e));
    return 0;
  });
}

// Download: run `produce(dst_device, b0, nb)` chunk by chunk into staging, then copy to the host.
template <typename F>
int download_chunks(altro_hip_batch* h, double* host, int block, int nk, F produce) {
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
int generic_get(altro_hip_batch* h, int arr, double* host, int block, int nk) {
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

int mfma16_get(altro_hip_batch* h, int what, double* host, int block, int nk) {
  return download_chunks(h, host, block, nk, [&](double* dst, int b0, int nb) {
    const int64_t total = (int64_t)nb * nk * block;
    if (h->dtype == ALTRO_HIP_F64)
      hipLaunchKernelGGL(mfma16_unpack_kernel<double>, dim3(grid_for(total)), dim3(256), 0, h->stream, dst, what,
                         (const double*)h->m_out, (const double*)h->m_outn, (const double*)h->m_xuy,
                         (const double*)h->m_qblk, h->m_st, h->N, b0, nb);
    else
      hipLaunchKernelGGL(mfma16_unpack_kernel<float>, dim3(grid_for(total)), dim3(256), 0, h->stream, dst, what,
                         (const float*)h->m_out, (const float*)h->m_outn, (const float*)h->m_xuy,
                         (const float*)h->m_qblk, h->m_st, h->N, b0, nb);
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
int put_src(altro_hip_batch* h, const double* src, int blk, int nk_host, int k_zero, int b_zero,
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
int mfma16_pack_launch(altro_hip_batch* h, int seg, SrcArr s0, SrcArr s1) {
  const int64_t total = (int64_t)h->batch * h->N * 192;
  if (h->dtype == ALTRO_HIP_F64)
    hipLaunchKernelGGL(mfma16_pack_kernel<double>, dim3(grid_for(total)), dim3(256), 0, h->stream, (double*)h->m_in,
                       (double*)h->m_cin, (double*)h->m_term, h->m_st, seg, s0, s1, h->is_diag, h->N, 0, h->batch);
  else
    hipLaunchKernelGGL(mfma16_pack_kernel<float>, dim3(grid_for(total)), dim3(256), 0, h->stream, (float*)h->m_in,
                       (float*)h->m_cin, (float*)h->m_term, h->m_st, seg, s0, s1, h->is_diag, h->N, 0, h->batch);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "mfma16_pack launch: %s", hipGetErrorString(e));
  return 0;
}


// ---- plan LANE dispatch ---------------------------------------------------------------------------
#define LANE_SHAPES(X) X(2, 1) X(4, 2) X(3, 1) X(6, 3)
bool lane_supported(int n, int m) {
#define X(N_, M_) if (n == N_ && m == M_) return true;
  LANE_SHAPES(X)
#undef X
  return false;
}
struct LaneSizes { int e_in, e_term, e_out, e_xuy; };
LaneSizes lane_sizes(int n, int m) {
  return LaneSizes{2 * n * n + 2 * n * m + m * m + 2 * n + m, n * n + n, m * n + m + n * n + n, 2 * n + m};
}
template <typename T>
int lane_launch(altro_hip_batch* h, bool backward, double reg) {
  LaneArgs<T> a{(const T*)h->l_in, (const T*)h->l_term, (T*)h->l_out, (T*)h->l_outn, (const T*)h->l_x0,
                (T*)h->l_xuy, (T*)h->delta_V, h->status, h->N, h->batch, (T)reg, backward ? h->bwd_active : nullptr,
                backward ? h->bwd_reg : nullptr};
  const dim3 grid(8 * (((h->batch + 63) / 64 + 7) / 8)), block(64);
  const bool fused = (h->flags & ALTRO_HIP_LANE_FUSED) != 0;
#define X(N_, M_)                                                                                              \
  if (h->n == N_ && h->m == M_) {                                                                              \
    if (backward && fused) hipLaunchKernelGGL((lane_backward_kernel_fused<N_, M_, T>), grid, block, 0, h->stream, a); \
    else if (backward) hipLaunchKernelGGL((lane_backward_kernel<N_, M_, T>), grid, block, 0, h->stream, a);    \
    else if (fused) hipLaunchKernelGGL((lane_forward_kernel_fused<N_, M_, T>), grid, block, 0, h->stream, a);  \
    else hipLaunchKernelGGL((lane_forward_kernel<N_, M_, T>), grid, block, 0, h->stream, a);                   \
  }
  LANE_SHAPES(X)
#undef X
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "lane kernel launch: %s", hipGetErrorString(e));
  return 0;
}
// one reference-layout source -> a run of elements of the SoA records [k0, k0+nk)
template <typename T>
int lane_pack(altro_hip_batch* h, T* dst_base, int E, const double* host, int len, int dst_off,
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
int lane_get(altro_hip_batch* h, double* host, const void* src, const void* src_term, int E, int off,
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

template <typename T>
GenericArgs<T> generic_args(altro_hip_batch* h, double reg) {
  GenericArgs<T> a;
  for (int i = 0; i < G_NUM; ++i) {
    a.base[i] = (T*)h->g_arr[i];
    a.bstride[i] = h->g_bstride[i];
  }
  a.off = h->g_off;
  a.nx = h->g_nx;
  a.nu = h->g_nu;
  a.x0 = (const T*)h->x0;
  a.x0_stride = h->n;
  a.delta_V = (T*)h->delta_V;
  a.status = h->status;
  a.N = h->N;
  a.batch = h->batch;
  a.nmax = h->n;
  a.mmax = h->m;
  a.reg = (T)reg;
  a.is_diag = h->is_diag;
  a.store_q = (h->flags & ALTRO_HIP_STORE_QBLOCKS) ? 1 : 0;
  a.want_y = 1;
  return a;
}

struct ProfScope {
  altro_hip_batch* h;
  int slot;
  ProfScope(altro_hip_batch* h_, int slot_) : h(h_), slot(slot_) {
    if (h->prof) (void)hipEventRecord(h->ev0, h->stream);
  }
  ~ProfScope() {
    if (h->prof) {
      (void)hipEventRecord(h->ev1, h->stream);
      (void)hipEventSynchronize(h->ev1);
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) {
        h->prof_ms[slot] += ms;
        h->prof_launches[slot] += 1;
      }
    }
  }
};


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


// ---- iLQR loop (plan LANE) ---------------------------------------------------------------------------
// (re)build the device tables of the constraint blocks; duals restart from zero when the structure changes
template <typename T>
int al_upload_typed(altro_hip_batch* h) {
  const int64_t B = h->batch;
  for (void** p : {(void**)&h->al_d_knots, &h->al_d_G, &h->al_d_g, &h->al_d_z})
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (h->al_defs.empty()) { h->al_rows = 0; return 0; }
  std::vector<T> G(h->al_G.begin(), h->al_G.end());
  std::vector<T> g;
  std::vector<AlDef> defs = h->al_defs;
  for (size_t i = 0; i < defs.size(); ++i) {
    defs[i].g_off = (int64_t)g.size();
    const std::vector<double>& src = h->al_g[i];
    const int p = defs[i].p;
    if (defs[i].g_per_problem) {
      const size_t base = g.size();
      g.resize(base + (size_t)p * B);
      for (int64_t b = 0; b < B; ++b)
        for (int r = 0; r < p; ++r) g[base + (size_t)r * B + b] = (T)src[(size_t)b * p + r];
    } else {
      for (int r = 0; r < p; ++r) g.push_back((T)src[r]);
    }
  }
  int rows = 0;
  std::vector<AlKnot> knots = h->al_knots;
  for (auto& kn : knots)
    for (int j = 0; j < kn.ncon; ++j) {
      const AlDef& d = defs[kn.def[j]];
      kn.z_off[j] = rows; rows += d.p;
      kn.cone[j] = d.cone; kn.p[j] = d.p; kn.g_per_problem[j] = d.g_per_problem; kn.G_off[j] = d.G_off; kn.g_off[j] = d.g_off;
      // bound-type block: every row of G is +-e_idx
      const int w = h->n + h->m;
      bool sel = d.cone != CONE_SOC;
      for (int r = 0; r < d.p && sel; ++r) {
        int nz = 0, at = -1;
        for (int e = 0; e < w; ++e) {
          const double v = h->al_G[(size_t)d.G_off + r + (size_t)e * d.p];
          if (v != 0.0) { ++nz; at = e; if (v != 1.0 && v != -1.0) sel = false; }
        }
        if (nz != 1) sel = false;
        else kn.sidx[j][r] = h->al_G[(size_t)d.G_off + r + (size_t)at * d.p] > 0 ? at + 1 : -(at + 1);
      }
      kn.sel[j] = sel ? 1 : 0;
    }
  h->al_rows = rows;
  if (h->plan == ALTRO_HIP_PLAN_LANE && (uint64_t)rows * (uint64_t)B * sizeof(T) >= (1ull << 31))
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "plan LANE: %d dual rows x batch %d exceed the 2 GiB buffer window; split the batch", rows, h->batch);
  {   // uniform running knot points?  (then the dual rows of knot point k start k * rows_per_knot after those of 0)
    const AlKnot& k0 = knots[0];
    int r0 = 0;
    for (int j = 0; j < k0.ncon; ++j) r0 += defs[k0.def[j]].p;
    bool uni = h->N >= 1 && k0.ncon > 0;
    for (int k = 1; k < h->N && uni; ++k) {
      uni = knots[k].ncon == k0.ncon;
      for (int j = 0; j < k0.ncon && uni; ++j) uni = knots[k].def[j] == k0.def[j] && knots[k].z_off[j] == k0.z_off[j] + k * r0;
    }
    h->al_uniform = uni ? 1 : 0;
    h->al_rows_per_knot = r0;
  }
  int rc = 0;
  if ((rc = dmalloc(h, (void**)&h->al_d_knots, knots.size() * sizeof(AlKnot)))) return rc;
  if ((rc = dmalloc(h, &h->al_d_G, G.size() * sizeof(T)))) return rc;
  if ((rc = dmalloc(h, &h->al_d_g, g.size() * sizeof(T)))) return rc;
  if ((rc = dmalloc(h, &h->al_d_z, (size_t)rows * B * sizeof(T)))) return rc;
  HIP_TRY(hipMemcpy(h->al_d_knots, knots.data(), knots.size() * sizeof(AlKnot), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->al_d_G, G.data(), G.size() * sizeof(T), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->al_d_g, g.data(), g.size() * sizeof(T), hipMemcpyHostToDevice));
  // (memsets go on the handle's own stream: it is non-blocking, so a null-stream memset would race the kernels)
  HIP_TRY(hipMemsetAsync(h->al_d_z, 0, (size_t)rows * B * sizeof(T), h->stream));
  h->al_knots = knots;
  return 0;
}
int al_upload(altro_hip_batch* h) {
  if (!h->al_dirty) return 0;
  int rc = h->dtype == ALTRO_HIP_F64 ? al_upload_typed<double>(h) : al_upload_typed<float>(h);
  if (!rc) h->al_dirty = false;
  return rc;
}

template <typename T>
IlqrArgs<T> ilqr_args(altro_hip_batch* h, bool use_alpha, bool use_active, int want_deriv, double alpha_const) {
  IlqrArgs<T> a;
  a.al.knots = h->al_d_knots; a.al.G = (const T*)h->al_d_G; a.al.g = (const T*)h->al_d_g;
  a.al.z = (T*)h->al_d_z; a.al.enabled = h->al_defs.empty() ? 0 : 1;
  a.al.uniform = h->al_uniform; a.al.rows_per_knot = h->al_rows_per_knot; a.al.N = h->N;
  a.mode = EXPAND_GRADIENT | EXPAND_HESSIAN;
  a.in = (T*)h->l_in; a.term = (T*)h->l_term; a.out = (const T*)h->l_out; a.outn = (const T*)h->l_outn;
  a.nom = (T*)h->l_nom; a.cand = (T*)h->l_xuy; a.cost = (const T*)h->l_cost; a.x0 = (const T*)h->l_x0;
  a.alpha = use_alpha ? h->i_alpha : nullptr;
  a.active = use_active ? h->i_active : nullptr;
  a.phi = h->i_phi; a.dphi = h->i_dphi; a.prob = h->i_prob;
  a.mp = h->model; a.N = h->N; a.batch = h->batch; a.want_derivative = want_deriv; a.alpha_const = alpha_const;
  a.cand_spec = (T*)h->i_cand_spec; a.spec_trials = h->i_cand_spec ? h->spec_trials : 1; a.spec_sel = h->i_spec_sel;
  a.spec_pre = h->i_cand_spec ? h->spec_pre : 0;
  a.spec_stride = (int64_t)h->batch * (h->N + 1) * lane_sizes(h->n, h->m).e_xuy;
  a.ls_beta = h->spec_beta; a.ls_max_iters = h->spec_max_iters;
  return a;
}

template <typename T>
int ilqr_launch(altro_hip_batch* h, int which, IlqrArgs<T> a) {
  const int rc = ilqr_launch_kernel<T>(h->stream, which, h->model.kind, h->n, h->m, a);
  if (rc == 1) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "no device model for (kind, n, m) = (%d, %d, %d)", h->model.kind, h->n, h->m);
  if (rc) return fail(ALTRO_HIP_ERR_HIP, "iLQR kernel launch failed");
  return 0;
}
template <typename S>
int wave_run(altro_hip_batch* h, int which, bool use_alpha, bool use_active, int want_deriv, double alpha_const, int mode) {
  IlqrWaveArgs<S> a;
  a.al.knots = h->al_d_knots; a.al.G = (const S*)h->al_d_G; a.al.g = (const S*)h->al_d_g; a.al.z = (S*)h->al_d_z;
  a.al.enabled = h->al_defs.empty() ? 0 : 1;
  a.al.uniform = h->al_uniform; a.al.rows_per_knot = h->al_rows_per_knot; a.al.N = h->N;
  a.mode = mode;
  a.dyn = (const S*)h->m_in; a.dyn_bs = h->m_st.in_bs; a.dyn_ks = h->m_st.in_ks;
  a.cin = (S*)h->m_cin; a.cin_bs = h->m_st.cin_bs; a.cin_ks = h->m_st.cin_ks;
  a.term = (S*)h->m_term; a.out = (const S*)h->m_out; a.out_bs = h->m_st.out_bs; a.out_ks = h->m_st.out_ks;
  a.outn = (const S*)h->m_outn; a.nom = (S*)h->m_nom; a.cand = (S*)h->m_xuy; a.xuy_bs = h->m_st.xuy_bs;
  a.xuy_ks = h->m_st.xuy_ks; a.costp = (const S*)h->m_costp; a.x0 = (const S*)h->x0;
  a.alpha = use_alpha ? h->i_alpha : nullptr; a.active = use_active ? h->i_active : nullptr;
  a.phi = h->i_phi; a.dphi = h->i_dphi; a.prob = h->i_prob; a.N = h->N; a.batch = h->batch;
  a.want_derivative = want_deriv; a.alpha_const = alpha_const;
  a.cand_spec = (S*)h->i_cand_spec; a.spec_trials = h->i_cand_spec ? h->spec_trials : 1; a.spec_sel = h->i_spec_sel;
  a.spec_pre = h->i_cand_spec ? h->spec_pre : 0;
  a.spec_stride = (int64_t)h->batch * (h->N + 1) * 28;
  a.ls_beta = h->spec_beta; a.ls_max_iters = h->spec_max_iters;
  const int rc = ilqr_wave_launch_kernel<S>(h->stream, which, a);
  if (rc == 1) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "operation %d is not available on plan MFMA16", which);
  if (rc) return fail(ALTRO_HIP_ERR_HIP, "iLQR kernel launch failed");
  return 0;
}
int ilqr_run(altro_hip_batch* h, int which, bool use_alpha, bool use_active, int want_deriv, double alpha_const,
             int mode = EXPAND_GRADIENT | EXPAND_HESSIAN) {
  int rc = al_upload(h);
  if (rc) return rc;
  if (h->plan == ALTRO_HIP_PLAN_MFMA16)   // linear dynamics: "expand" = cost gradient (+ AL Hessian terms when constrained)
    return h->dtype == ALTRO_HIP_F64 ? wave_run<double>(h, which, use_alpha, use_active, want_deriv, alpha_const, mode)
                                     : wave_run<float>(h, which, use_alpha, use_active, want_deriv, alpha_const, mode);
  if (h->dtype == ALTRO_HIP_F64) {
    auto a = ilqr_args<double>(h, use_alpha, use_active, want_deriv, alpha_const);
    a.mode = mode;
    return ilqr_launch<double>(h, which, a);
  }
  auto a = ilqr_args<float>(h, use_alpha, use_active, want_deriv, alpha_const);
  a.mode = mode;
  return ilqr_launch<float>(h, which, a);
}
int ilqr_check(altro_hip_batch* h, bool need_guess) {
  int rc = check(h);
  if (rc) return rc;
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {   // dynamics are data (altro_hip_set_dynamics), no device model
    if (!h->dyn_set) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_dynamics has not been called");
  } else if (h->plan != ALTRO_HIP_PLAN_LANE) {
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "the device iLQR loop is implemented for plans LANE and MFMA16");
  } else if (!h->model_set) {
    return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_model has not been called");
  }
  if (!h->lqr_cost_set) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_tracking_cost has not been called");
  if (!h->x0_set) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_initial_state has not been called");
  if (need_guess && !h->guess_set) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_input_guess has not been called");
  return 0;
}

template <typename S>
Mfma16Args<S> mfma16_args(altro_hip_batch* h, double reg) {
  Mfma16Args<S> a;
  a.in = (const S*)h->m_in;
  a.cin = (const S*)h->m_cin;
  a.cin_bs = h->m_st.cin_bs; a.cin_ks = h->m_st.cin_ks;
  a.in_bs = h->m_st.in_bs; a.in_ks = h->m_st.in_ks; a.out_bs = h->m_st.out_bs; a.out_ks = h->m_st.out_ks;
  a.xuy_bs = h->m_st.xuy_bs; a.xuy_ks = h->m_st.xuy_ks;
  a.term = (const S*)h->m_term; a.out = (S*)h->m_out; a.outn = (S*)h->m_outn; a.qblk = (S*)h->m_qblk;
  a.trash = (S*)h->m_trash; a.x0 = (const S*)h->x0; a.xuy = (S*)h->m_xuy; a.delta_V = (S*)h->delta_V;
  a.status = h->status; a.N = h->N; a.batch = h->batch; a.reg = reg; a.has_f = h->has_f && !h->ilqr_linear;
  return a;
}
template <typename S>
void mfma16_launch_backward(altro_hip_batch* h, double reg, bool sq) {
  auto a = mfma16_args<S>(h, reg);
  const dim3 grid(mf_grid(h->batch)), block(64);
  if (sq && a.has_f) hipLaunchKernelGGL((mfma16_backward_kernel<true, true, S>), grid, block, 0, h->stream, a);
  else if (sq) hipLaunchKernelGGL((mfma16_backward_kernel<true, false, S>), grid, block, 0, h->stream, a);
  else if (a.has_f) hipLaunchKernelGGL((mfma16_backward_kernel<false, true, S>), grid, block, 0, h->stream, a);
  else hipLaunchKernelGGL((mfma16_backward_kernel<false, false, S>), grid, block, 0, h->stream, a);
}

template <typename S>
void mfma16_launch_forward(altro_hip_batch* h, const Mfma16Args<S>& a) {
  // register-ring depth 3: depths 1..4 were measured, 3 is the knee
  hipLaunchKernelGGL((mfma16_forward_kernel<S, 3>), dim3(mf_grid(h->batch)), dim3(64), 0, h->stream, a);
}

int launch_backward(altro_hip_batch* h, double reg) {
  ProfScope ps(h, 0);
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {
    const bool sq = (h->flags & ALTRO_HIP_STORE_QBLOCKS) != 0;
    if (h->dtype == ALTRO_HIP_F64) mfma16_launch_backward<double>(h, reg, sq);
    else if (sq || !(h->flags & ALTRO_HIP_F32_PURE)) mfma16_launch_backward<float>(h, reg, sq);   // fp32 storage, fp64 tiles
    else {   // opt-in: pure fp32 on v_mfma_f32_16x16x4_f32
      auto a = mfma16_args<float>(h, reg);
      if (a.has_f) hipLaunchKernelGGL((mfma16_backward_f32_kernel<true, 3>), dim3(mf_grid(h->batch)), dim3(64), 0, h->stream, a);
      else hipLaunchKernelGGL((mfma16_backward_f32_kernel<false, 3>), dim3(mf_grid(h->batch)), dim3(64), 0, h->stream, a);
    }
  } else if (h->plan == ALTRO_HIP_PLAN_LANE) {
    return h->dtype == ALTRO_HIP_F64 ? lane_launch<double>(h, true, reg) : lane_launch<float>(h, true, reg);
  } else if (h->dtype == ALTRO_HIP_F64) {
    auto a = generic_args<double>(h, reg);
    size_t lds = generic_backward_lds_bytes<double>(h->n, h->m);
    hipLaunchKernelGGL(generic_backward_kernel<double>, dim3(h->batch), dim3(64), lds, h->stream, a);
  } else {
    auto a = generic_args<float>(h, reg);
    size_t lds = generic_backward_lds_bytes<float>(h->n, h->m);
    hipLaunchKernelGGL(generic_backward_kernel<float>, dim3(h->batch), dim3(64), lds, h->stream, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "backward launch: %s", hipGetErrorString(e));
  return 0;
}

int launch_forward(altro_hip_batch* h) {
  ProfScope ps(h, 1);
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {
    if (h->dtype == ALTRO_HIP_F64) {
      auto a = mfma16_args<double>(h, 0.0);
      mfma16_launch_forward<double>(h, a);
    } else {
      auto a = mfma16_args<float>(h, 0.0);
      mfma16_launch_forward<float>(h, a);
    }
  } else if (h->plan == ALTRO_HIP_PLAN_LANE) {
    return h->dtype == ALTRO_HIP_F64 ? lane_launch<double>(h, false, 0.0) : lane_launch<float>(h, false, 0.0);
  } else if (h->dtype == ALTRO_HIP_F64) {
    auto a = generic_args<double>(h, 0.0);
    size_t lds = (size_t)(2 * h->n + h->m) * sizeof(double) + 64;
    hipLaunchKernelGGL(generic_forward_kernel<double>, dim3(h->batch), dim3(64), lds, h->stream, a);
  } else {
    auto a = generic_args<float>(h, 0.0);
    size_t lds = (size_t)(2 * h->n + h->m) * sizeof(float) + 64;
    hipLaunchKernelGGL(generic_forward_kernel<float>, dim3(h->batch), dim3(64), lds, h->stream, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "forward launch: %s", hipGetErrorString(e));
  return 0;
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

int altro_hip_batch_create(altro_hip_batch** out, int N, int n, int m, int batch, int dtype, int plan,
                           unsigned flags, int device, void* stream) {
  if (!out) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "out == NULL");
  *out = nullptr;
  if (N <= 0 || n <= 0 || m <= 0 || batch <= 0)
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "N, n, m, batch must be positive (got %d %d %d %d)", N, n, m, batch);
  if (n > 32 || m > 32) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "n, m <= 32 supported (got %d, %d)", n, m);
  if (dtype != ALTRO_HIP_F64 && dtype != ALTRO_HIP_F32) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "bad dtype %d", dtype);
  if (altro_hip_device_count() <= device || device < 0)
    return fail(ALTRO_HIP_ERR_NO_DEVICE, "no HIP device %d: the altro_hip hot path has no CPU fallback", device);
  const bool mfma_ok = (n == 12 && m == 4);   // fp32 handles: fp32 storage, fp64 tile arithmetic
  if (plan == ALTRO_HIP_PLAN_AUTO)
    plan = mfma_ok ? ALTRO_HIP_PLAN_MFMA16 : (lane_supported(n, m) ? ALTRO_HIP_PLAN_LANE : ALTRO_HIP_PLAN_GENERIC);
  if (plan == ALTRO_HIP_PLAN_MFMA16 && !mfma_ok)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "plan MFMA16 needs (n, m) = (12, 4)");
  if (plan == ALTRO_HIP_PLAN_LANE && !lane_supported(n, m))
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "plan LANE is instantiated for (n, m) in {(2,1), (3,1), (4,2), (6,3)}");
  if (plan != ALTRO_HIP_PLAN_MFMA16 && plan != ALTRO_HIP_PLAN_GENERIC && plan != ALTRO_HIP_PLAN_LANE)
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "bad plan %d", plan);
  HIP_TRY(hipSetDevice(device));
  altro_hip_batch* h = new altro_hip_batch();
  h->N = N; h->n = n; h->m = m; h->batch = batch; h->dtype = dtype; h->plan = plan;
  h->flags = flags; h->device = device;
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
  ALLOC(h->x0, B * n * E);
  ALLOC(h->delta_V, B * 2 * E);
  ALLOC(h->status, B * sizeof(int));
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
    const int blk[G_NUM] = {n * n, n * m, n, n * n, m * m, m * n, n, m, m * n, m, n * n, n,
                            n * n, m * m, m * n, n, m, n * n, m * m, m * n, n, m, n, m, n};
    const int nks[G_NUM] = {N, N, N, N + 1, N, N, N + 1, N, N, N, N + 1, N + 1,
                            N, N, N, N, N, N, N, N, N, N, N + 1, N, N + 1};
    std::vector<int64_t> off((size_t)(N + 1) * G_NUM, 0);
    for (int a = 0; a < G_NUM; ++a) {
      const bool qb = (a >= G_Qxx && a <= G_Qu);
      if (a >= G_Qxx_tmp && a <= G_Qu_tmp) continue;   // scratch blocks: only the tvlqr_* drop-in keeps them
      h->g_bstride[a] = (int64_t)nks[a] * blk[a];
      for (int k = 0; k <= N; ++k) off[(size_t)k * G_NUM + a] = (int64_t)k * blk[a];
      if (qb && !(flags & ALTRO_HIP_STORE_QBLOCKS)) continue;
      ALLOC(h->g_arr[a], B * nks[a] * blk[a] * E);
    }
    ALLOC(h->g_off, off.size() * sizeof(int64_t));
    ALLOC(h->g_nx, (size_t)(N + 1) * sizeof(int));
    ALLOC(h->g_nu, (size_t)(N + 1) * sizeof(int));
    if (!rc) {
      std::vector<int> nx(N + 1, n), nu(N + 1, m);
      if (hipMemcpy(h->g_off, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(h->g_nx, nx.data(), nx.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(h->g_nu, nu.data(), nu.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(ALTRO_HIP_ERR_HIP, "table upload failed");
    }
  }
  if (plan == ALTRO_HIP_PLAN_LANE || plan == ALTRO_HIP_PLAN_MFMA16) {   // per-problem control blocks of the iLQR loop
    if (!rc) h->al_knots.assign((size_t)N + 1, AlKnot{});
    ALLOC(h->i_prob, B * sizeof(IlqrProb));
    ALLOC(h->i_alpha, B * 8);
    ALLOC(h->i_phi, B * 8 * ILQR_SPEC_TRIALS);
    ALLOC(h->i_spec_sel, B * sizeof(int));
    ALLOC(h->i_spec_refresh, B * sizeof(int));
    ALLOC(h->i_dphi, B * 8 * 2);
    ALLOC(h->i_active, B * sizeof(int));
    ALLOC(h->i_counters, 4 * sizeof(int));
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

void altro_hip_batch_destroy(altro_hip_batch* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  void* ptrs[] = {h->x0, h->delta_V, h->status, h->m_in, h->m_cin, h->m_term, h->m_out, h->m_outn, h->m_xuy,
                  h->m_qblk, h->m_trash, h->g_off, h->g_nx, h->g_nu, h->stage,
                  h->l_in, h->l_term, h->l_out, h->l_outn, h->l_xuy, h->l_x0,
                  h->l_nom, h->l_cost, h->i_prob, h->i_alpha, h->i_phi, h->i_dphi, h->i_active, h->i_counters,
                  h->al_d_knots, h->al_d_G, h->al_d_g, h->al_d_z, h->i_reg, h->m_nom, h->m_costp,
                  h->i_cand_spec, h->i_spec_sel, h->i_spec_refresh};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (int a = 0; a < G_NUM; ++a) if (h->g_arr[a]) (void)hipFree(h->g_arr[a]);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int altro_hip_batch_plan(const altro_hip_batch* h) { return h ? h->plan : ALTRO_HIP_ERR_BAD_ARGUMENT; }
size_t altro_hip_batch_device_bytes(const altro_hip_batch* h) { return h ? h->device_bytes : 0; }

int altro_hip_set_dynamics(altro_hip_batch* h, const double* A, const double* B, const double* f,
                           int kz, int bz) {
  int rc = check(h);
  if (rc) return rc;
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
  if (!x0) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "x0 == NULL");
  auto consume = [&](SrcArr s, int b0, int nb) -> int {
    const int64_t total = (int64_t)nb * h->n;
    if (h->dtype == ALTRO_HIP_F64)
      hipLaunchKernelGGL(expand_copy_kernel<double>, dim3(grid_for(total)), dim3(256), 0, h->stream,
                         (double*)h->x0, (int64_t)h->n, (int64_t)h->n, s, h->n, 1, b0, nb);
    else
      hipLaunchKernelGGL(expand_copy_kernel<float>, dim3(grid_for(total)), dim3(256), 0, h->stream,
                         (float*)h->x0, (int64_t)h->n, (int64_t)h->n, s, h->n, 1, b0, nb);
    if (hipGetLastError() != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "x0 copy launch failed");
    return 0;
  };
  rc = upload_chunks(h, x0, h->n, 1, 1, bz, -1, 0, consume);
  if (!rc && h->plan == ALTRO_HIP_PLAN_LANE)
    rc = h->dtype == ALTRO_HIP_F64
             ? lane_pack<double>(h, (double*)h->l_x0, h->n, x0, h->n, 0, 0, 1, 0, 1, 1, bz)
             : lane_pack<float>(h, (float*)h->l_x0, h->n, x0, h->n, 0, 0, 1, 0, 1, 1, bz);
  if (!rc) h->x0_set = true;
  return rc;
}

int altro_hip_set_pointer_mode(altro_hip_batch* h, int device_pointers) {
  int rc = check(h);
  if (rc) return rc;
  h->dev_ptrs = device_pointers != 0;
  return 0;
}

int altro_hip_backward(altro_hip_batch* h, double reg) {
  int rc = check(h);
  if (rc) return rc;
  if (!h->dyn_set || !h->cost_set) return fail(ALTRO_HIP_ERR_NOT_SET, "set_dynamics and set_cost must precede backward");
  rc = launch_backward(h, reg);
  if (!rc) h->backward_done = true;
  return rc;
}

int altro_hip_forward_ltv(altro_hip_batch* h) {
  int rc = check(h);
  if (rc) return rc;
  if (!h->backward_done) return fail(ALTRO_HIP_ERR_NOT_SET, "backward must precede forward_ltv");
  if (!h->x0_set) return fail(ALTRO_HIP_ERR_NOT_SET, "set_initial_state must precede forward_ltv");
  rc = launch_forward(h);
  if (!rc) h->forward_done = true;
  return rc;
}

int altro_hip_sweep(altro_hip_batch* h, double reg) {
  int rc = altro_hip_backward(h, reg);
  if (!rc) rc = altro_hip_forward_ltv(h);
  return rc;
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

int altro_hip_stats_reduce(altro_hip_batch* h, altro_hip_stats* out) {
  int rc = check(h);
  if (rc) return rc;
  if (!out) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "out == NULL");
  if (!h->backward_done) return fail(ALTRO_HIP_ERR_NOT_SET, "backward has not run");
  std::vector<int> st(h->batch);
  std::vector<double> dv((size_t)h->batch * 2);
  rc = altro_hip_get_status(h, st.data());
  if (!rc) rc = altro_hip_get_delta_V(h, dv.data());
  if (rc) return rc;
  out->problems = h->batch;
  out->cholesky_failures = 0;
  out->sum_delta_V0 = out->sum_delta_V1 = 0.0;
  out->max_abs_xN = 0.0;
  for (int b = 0; b < h->batch; ++b) {
    if (st[b] != ALTRO_HIP_TVLQR_SUCCESS) { out->cholesky_failures++; continue; }
    out->sum_delta_V0 += dv[2 * (size_t)b];
    out->sum_delta_V1 += dv[2 * (size_t)b + 1];
  }
  if (h->forward_done) {
    std::vector<double> x((size_t)h->batch * (h->N + 1) * h->n);
    rc = altro_hip_get_x(h, x.data());
    if (rc) return rc;
    for (int b = 0; b < h->batch; ++b)
      for (int i = 0; i < h->n; ++i) {
        double v = std::abs(x[((size_t)b * (h->N + 1) + h->N) * h->n + i]);
        if (v > out->max_abs_xN) out->max_abs_xN = v;
      }
  }
  return 0;
}

int altro_hip_profile_enable(altro_hip_batch* h, int enable) {
  if (!h) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "null handle");
  h->prof = enable != 0;
  return 0;
}
int altro_hip_profile_reset(altro_hip_batch* h) {
  if (!h) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "null handle");
  h->prof_launches[0] = h->prof_launches[1] = 0;
  h->prof_ms[0] = h->prof_ms[1] = 0.0;
  return 0;
}
int altro_hip_profile_get(altro_hip_batch* h, int slot, int* launches, double* total_ms,
                          const char** kernel_name) {
  if (!h || slot < 0 || slot > 1) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "bad handle or slot");
  if (launches) *launches = h->prof_launches[slot];
  if (total_ms) *total_ms = h->prof_ms[slot];
  if (kernel_name) {
    static const char* names[3][2] = {{"generic_backward_kernel", "generic_forward_kernel"},
                                      {"mfma16_backward_kernel", "mfma16_forward_kernel"},
                                      {"lane_backward_kernel", "lane_forward_kernel"}};
    *kernel_name = names[h->plan == ALTRO_HIP_PLAN_MFMA16 ? 1 : (h->plan == ALTRO_HIP_PLAN_LANE ? 2 : 0)][slot];
  }
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


// ---- iLQR loop entry points -----------------------------------------------------------------------------
int altro_hip_set_model(altro_hip_batch* h, int model, float timestep, int bicycle_frame,
                        double bicycle_length, double bicycle_lr) {
  int rc = check(h);
  if (rc) return rc;
  if (!(timestep > 0.0f)) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "time step must be positive (ErrorCodes::TimestepNotPositive)");
  if (h->plan != ALTRO_HIP_PLAN_LANE || !ilqr_supported(model, h->n, h->m))
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "no device model %d for plan %d with (n, m) = (%d, %d)", model, h->plan, h->n, h->m);
  h->model = ModelParams{model, timestep, bicycle_frame, bicycle_length > 0 ? bicycle_length : 2.7,
                         bicycle_lr > 0 ? bicycle_lr : 1.5};
  h->model_set = true;
  return 0;
}

int altro_hip_set_tracking_cost(altro_hip_batch* h, const double* Qd, const double* Rd, const double* xref,
                                const double* uref, int kz, int bz) {
  // ALTROSolver::SetLQRCost (altro_solver.cpp:138-172): q = -Q xref, r = -R uref,
  // c = 1/2 xref'Q xref (+ 1/2 uref'R uref for k < N) -> KnotPointData::SetDiagonalCost
  int rc = check(h);
  if (rc) return rc;
  if (h->plan != ALTRO_HIP_PLAN_LANE && h->plan != ALTRO_HIP_PLAN_MFMA16)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "tracking cost needs plan LANE or MFMA16");
  if (!Qd || !Rd || !xref || !uref) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "Qd, Rd, xref, uref are required");
  if (h->dev_ptrs)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "altro_hip_set_tracking_cost forms q = -Q xref on the host: pass host arrays "
                                           "(altro_hip_set_pointer_mode(h, 0))");
  const int n = h->n, m = h->m, N = h->N;
  const int nb = bz ? 1 : h->batch, nkx = kz ? 2 : N + 1, nku = kz ? 1 : N;
  const int E = 2 * n + 2 * m + 1;
  // host-side arithmetic exactly as the reference's setter does it, then one upload per field
  std::vector<double> q((size_t)nb * nkx * n), r((size_t)nb * nku * m), c((size_t)nb * nkx);
  for (int b = 0; b < nb; ++b)
    for (int k = 0; k < nkx; ++k) {
      const double* Q_ = Qd + ((size_t)b * nkx + k) * n;
      const double* x_ = xref + ((size_t)b * nkx + k) * n;
      double cc = 0.0;
      for (int i = 0; i < n; ++i) { q[((size_t)b * nkx + k) * n + i] = -(Q_[i] * x_[i]); cc += x_[i] * Q_[i] * x_[i]; }
      cc *= 0.5;
      const bool terminal = kz ? (k == 1) : (k == N);
      if (!terminal) {
        const int ku = kz ? 0 : k;
        const double* R_ = Rd + ((size_t)b * nku + ku) * m;
        const double* u_ = uref + ((size_t)b * nku + ku) * m;
        double cu = 0.0;
        for (int i = 0; i < m; ++i) cu += u_[i] * R_[i] * u_[i];
        cc += 0.5 * cu;
      }
      c[(size_t)b * nkx + k] = cc;
    }
  for (int b = 0; b < nb; ++b)
    for (int k = 0; k < nku; ++k)
      for (int i = 0; i < m; ++i)
        r[((size_t)b * nku + k) * m + i] = -(Rd[((size_t)b * nku + k) * m + i] * uref[((size_t)b * nku + k) * m + i]);
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {
    // (a) the backward sweep's blocks: lxx = diag(Qd), luu = diag(Rd), lux = 0; lx, lu are refreshed by the loop
    rc = altro_hip_set_cost(h, Qd, Rd, nullptr, q.data(), r.data(), 1, kz, bz);
    if (rc) return rc;
    // (b) the cost parameters the merit function needs: [k][b][36] = Qd | Rd | q | r | c
    const size_t Ez = h->dtype == ALTRO_HIP_F64 ? 8 : 4;
    const int64_t B = h->batch;
    if (!h->m_costp) {
      if ((rc = dmalloc(h, &h->m_costp, (size_t)B * (N + 1) * MF_COSTP * Ez))) return rc;
      if ((rc = dmalloc(h, &h->m_nom, (size_t)B * (N + 1) * MF_NOM * Ez))) return rc;
      HIP_TRY(hipMemsetAsync(h->m_costp, 0, (size_t)B * (N + 1) * MF_COSTP * Ez, h->stream));
      HIP_TRY(hipMemsetAsync(h->m_nom, 0, (size_t)B * (N + 1) * MF_NOM * Ez, h->stream));
    }
    auto put = [&](const double* src, int len, int off, int k0, int nk, int nk_host, int src_off) -> int {
      // records k0..k0+nk-1; with kz the host holds {running, terminal} and src_off selects which
      if (h->dtype == ALTRO_HIP_F64)
        return aos_set<double>(h, (double*)h->m_costp + (size_t)k0 * B * MF_COSTP + off, MF_COSTP, B * MF_COSTP, src, len, nk,
                               kz, bz, nk_host, src_off);
      return aos_set<float>(h, (float*)h->m_costp + (size_t)k0 * B * MF_COSTP + off, MF_COSTP, B * MF_COSTP, src, len, nk, kz,
                            bz, nk_host, src_off);
    };
    rc = put(Qd, n, 0, 0, N, nkx, 0);
    if (!rc) rc = put(Rd, m, 12, 0, N, nku, 0);
    if (!rc) rc = put(q.data(), n, 16, 0, N, nkx, 0);
    if (!rc) rc = put(r.data(), m, 28, 0, N, nku, 0);
    if (!rc) rc = put(c.data(), 1, 32, 0, N, nkx, 0);
    // terminal record N: element N of a full host array, or the second entry of a {running, terminal} pair
    auto put_term = [&](const double* src, int len, int off) -> int {
      const double* base = kz ? src : src;   // per problem the host holds nkx knot points
      if (h->dtype == ALTRO_HIP_F64)
        return aos_set<double>(h, (double*)h->m_costp + (size_t)N * B * MF_COSTP + off, MF_COSTP, B * MF_COSTP, base, len, 1, 1,
                               bz, nkx, (kz ? 1 : N) * len);
      return aos_set<float>(h, (float*)h->m_costp + (size_t)N * B * MF_COSTP + off, MF_COSTP, B * MF_COSTP, base, len, 1, 1, bz,
                            nkx, (kz ? 1 : N) * len);
    };
    if (!rc) rc = put_term(Qd, n, 0);
    if (!rc) rc = put_term(q.data(), n, 16);
    if (!rc) rc = put_term(c.data(), 1, 32);
    if (!rc) { h->lqr_cost_set = true; h->ilqr_linear = true; }
    return rc;
  }
  auto pk = [&](const double* src, int len, int off, int nk, int k_src0, int nk_host, int src_off) -> int {
    return h->dtype == ALTRO_HIP_F64
               ? lane_pack<double>(h, (double*)h->l_cost + (size_t)0, E, src, len, off, 0, nk, k_src0, nk_host, kz, bz, src_off)
               : lane_pack<float>(h, (float*)h->l_cost + (size_t)0, E, src, len, off, 0, nk, k_src0, nk_host, kz, bz, src_off);
  };
  auto pk_term = [&](const double* src, int len, int off, int nk_host) -> int {   // record N of l_cost
    const size_t base = (size_t)N * E * h->batch;
    return h->dtype == ALTRO_HIP_F64
               ? lane_pack<double>(h, (double*)h->l_cost + base, E, src, len, off, 0, 1, kz ? 0 : N, nk_host, kz, bz, kz ? len : 0)
               : lane_pack<float>(h, (float*)h->l_cost + base, E, src, len, off, 0, 1, kz ? 0 : N, nk_host, kz, bz, kz ? len : 0);
  };
  rc = pk(Qd, n, 0, N, 0, nkx, 0);
  if (!rc) rc = pk(Rd, m, n, N, 0, nku, 0);
  if (!rc) rc = pk(q.data(), n, n + m, N, 0, nkx, 0);
  if (!rc) rc = pk(r.data(), m, 2 * n + m, N, 0, nku, 0);
  if (!rc) rc = pk(c.data(), 1, 2 * n + 2 * m, N, 0, nkx, 0);
  if (!rc) rc = pk_term(Qd, n, 0, nkx);
  if (!rc) rc = pk_term(q.data(), n, n + m, nkx);
  if (!rc) rc = pk_term(c.data(), 1, 2 * n + 2 * m, nkx);
  if (!rc) { h->lqr_cost_set = true; h->cost_set = true; h->dyn_set = true; h->is_diag = 0; h->has_f = 0; }
  return rc;
}

int altro_hip_set_input_guess(altro_hip_batch* h, const double* u, int kz, int bz) {
  // ALTROSolver::SetInput (altro_solver.cpp:242-251): writes the CANDIDATE inputs u_
  int rc = check(h);
  if (rc) return rc;
  if (h->plan != ALTRO_HIP_PLAN_LANE && h->plan != ALTRO_HIP_PLAN_MFMA16)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "input guess needs plan LANE or MFMA16");
  if (!u) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "u == NULL");
  const int n = h->n, m = h->m, N = h->N;
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {   // candidate records [k][b][28] = x | y | u
    rc = h->dtype == ALTRO_HIP_F64
             ? aos_set<double>(h, (double*)h->m_xuy + 24, h->m_st.xuy_bs, h->m_st.xuy_ks, u, m, N, kz, bz)
             : aos_set<float>(h, (float*)h->m_xuy + 24, h->m_st.xuy_bs, h->m_st.xuy_ks, u, m, N, kz, bz);
    if (!rc) h->guess_set = true;
    return rc;
  }
  rc = h->dtype == ALTRO_HIP_F64
           ? lane_pack<double>(h, (double*)h->l_xuy, 2 * n + m, u, m, 2 * n, 0, N, 0, kz ? 1 : N, kz, bz)
           : lane_pack<float>(h, (float*)h->l_xuy, 2 * n + m, u, m, 2 * n, 0, N, 0, kz ? 1 : N, kz, bz);
  if (!rc) h->guess_set = true;
  return rc;
}

int altro_hip_open_loop_rollout(altro_hip_batch* h) {
  int rc = ilqr_check(h, true);
  if (!rc) rc = ilqr_run(h, IK_ROLLOUT, false, false, 0, 0.0);
  if (!rc) h->forward_done = true;
  return rc;
}
int altro_hip_accept(altro_hip_batch* h) {
  int rc = ilqr_check(h, false);
  if (!rc) rc = ilqr_run(h, IK_ACCEPT, false, false, 0, 0.0);
  return rc;
}
int altro_hip_expand(altro_hip_batch* h) {
  int rc = ilqr_check(h, false);
  if (!rc) rc = ilqr_run(h, IK_EXPAND, false, false, 0, 0.0);
  return rc;
}
int altro_hip_merit(altro_hip_batch* h, const double* alpha, int alpha_is_uniform, int want_derivative,
                    double* phi, double* dphi) {
  int rc = ilqr_check(h, false);
  if (rc) return rc;
  if (!h->backward_done) return fail(ALTRO_HIP_ERR_NOT_SET, "backward must precede merit");
  if (!alpha || !phi) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "alpha and phi are required");
  if (!alpha_is_uniform)
    HIP_TRY(hipMemcpyAsync(h->i_alpha, alpha, (size_t)h->batch * 8, hipMemcpyHostToDevice, h->stream));
  rc = ilqr_run(h, IK_MERIT, !alpha_is_uniform, false, want_derivative, alpha[0]);
  if (rc) return rc;
  h->forward_done = true;
  HIP_TRY(hipMemcpyAsync(phi, h->i_phi, (size_t)h->batch * 8, hipMemcpyDeviceToHost, h->stream));
  if (want_derivative && dphi)
    HIP_TRY(hipMemcpyAsync(dphi, h->i_dphi, (size_t)h->batch * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}
int altro_hip_stationarity(altro_hip_batch* h, double* out) {
  int rc = ilqr_check(h, false);
  if (rc) return rc;
  rc = ilqr_run(h, IK_STATIONARITY, false, false, 0, 0.0);
  if (rc) return rc;
  std::vector<IlqrProb> pr(h->batch);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(pr.data(), h->i_prob, pr.size() * sizeof(IlqrProb), hipMemcpyDeviceToHost));
  for (int b = 0; b < h->batch; ++b) out[b] = pr[b].stationarity;
  return 0;
}
int altro_hip_feasibility(altro_hip_batch* h, double* out) {
  // SolverImpl::Feasibility (solver.cpp:224-231) of the candidate trajectory
  int rc = ilqr_check(h, false);
  if (rc) return rc;
  if (!out) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "out == NULL");
  rc = ilqr_run(h, IK_STATIONARITY, false, false, 0, 0.0);
  if (rc) return rc;
  std::vector<IlqrProb> pr(h->batch);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(pr.data(), h->i_prob, pr.size() * sizeof(IlqrProb), hipMemcpyDeviceToHost));
  for (int b = 0; b < h->batch; ++b) out[b] = pr[b].feasibility;
  return 0;
}

// ---- MPC receding-horizon operations on the resident batch -----------------
int altro_hip_shift_trajectory(altro_hip_batch* h) {
  int rc = ilqr_check(h, true);
  if (!rc) rc = ilqr_run(h, IK_SHIFT, false, false, 0, 0.0);
  return rc;
}
int altro_hip_update_linear_costs(altro_hip_batch* h, const double* q, const double* r, const double* c,
                                  int k_first, int k_last, int kz, int bz) {
  // ALTROSolver::UpdateLinearCosts (altro_solver.cpp:266-281) -> KnotPointData::UpdateLinearCosts
  // (knotpoint_data.cpp:193-226) for knot points k_first..k_last (inclusive) of every problem
  int rc = check(h);
  if (rc) return rc;
  if (h->plan != ALTRO_HIP_PLAN_LANE && h->plan != ALTRO_HIP_PLAN_MFMA16)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "linear cost update needs plan LANE or MFMA16");
  if (!h->lqr_cost_set) return fail(ALTRO_HIP_ERR_NOT_SET, "no quadratic cost to update (ErrorCodes::CostNotQuadratic)");
  const int n = h->n, m = h->m, N = h->N;
  if (k_first < 0 || k_last > N || k_first > k_last)
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "knot point range [%d, %d] outside [0, %d] (ErrorCodes::BadIndex)", k_first, k_last, N);
  if (r && k_last == N)
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "cannot update linear input costs at the terminal knot point "
                                            "(ErrorCodes::InvalidOptAtTerminalKnotPoint)");
  const int nk = k_last - k_first + 1;
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {
    const int64_t B = h->batch;
    auto put = [&](const double* src, int len, int off) -> int {
      if (!src) return 0;
      if (h->dtype == ALTRO_HIP_F64)
        return aos_set<double>(h, (double*)h->m_costp + (size_t)k_first * B * MF_COSTP + off, MF_COSTP, B * MF_COSTP, src, len,
                               nk, kz, bz);
      return aos_set<float>(h, (float*)h->m_costp + (size_t)k_first * B * MF_COSTP + off, MF_COSTP, B * MF_COSTP, src, len, nk,
                            kz, bz);
    };
    rc = put(q, n, 16);
    if (!rc) rc = put(r, m, 28);
    if (!rc) rc = put(c, 1, 32);
    return rc;
  }
  const int E = 2 * n + 2 * m + 1;
  const size_t base = (size_t)k_first * E * h->batch;
  auto pk = [&](const double* src, int len, int off) -> int {
    if (!src) return 0;
    return h->dtype == ALTRO_HIP_F64
               ? lane_pack<double>(h, (double*)h->l_cost + base, E, src, len, off, 0, nk, 0, kz ? 1 : nk, kz, bz)
               : lane_pack<float>(h, (float*)h->l_cost + base, E, src, len, off, 0, nk, 0, kz ? 1 : nk, kz, bz);
  };
  rc = pk(q, n, n + m);
  if (!rc) rc = pk(r, m, 2 * n + m);
  if (!rc) rc = pk(c, 1, 2 * n + 2 * m);
  return rc;
}
int altro_hip_get_knot(altro_hip_batch* h, int k, double* x, double* u) {
  // ALTROSolver::GetState / GetInput (altro_solver.cpp:323-347) of one knot point for the whole batch:
  // x [batch][n], u [batch][m] (u must be NULL at k = N)
  int rc = check(h);
  if (rc) return rc;
  const int n = h->n, m = h->m, N = h->N;
  if (k < 0 || k > N) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "knot point %d outside [0, %d] (ErrorCodes::BadIndex)", k, N);
  if (u && k == N) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "no input at the terminal knot point");
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {
    if (!h->m_nom) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_tracking_cost has not been called");
    const int64_t B = h->batch;
    if (x) {
      rc = h->dtype == ALTRO_HIP_F64
               ? aos_get<double>(h, x, (const double*)h->m_nom + (size_t)k * B * MF_NOM, MF_NOM, B * MF_NOM, n, 1)
               : aos_get<float>(h, x, (const float*)h->m_nom + (size_t)k * B * MF_NOM, MF_NOM, B * MF_NOM, n, 1);
      if (rc) return rc;
    }
    if (u)
      rc = h->dtype == ALTRO_HIP_F64
               ? aos_get<double>(h, u, (const double*)h->m_nom + (size_t)k * B * MF_NOM + 12, MF_NOM, B * MF_NOM, m, 1)
               : aos_get<float>(h, u, (const float*)h->m_nom + (size_t)k * B * MF_NOM + 12, MF_NOM, B * MF_NOM, m, 1);
    return rc;
  }
  if (h->plan != ALTRO_HIP_PLAN_LANE) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "plans LANE and MFMA16 only");
  const size_t E = h->dtype == ALTRO_HIP_F64 ? 8 : 4;
  const char* rec = (const char*)h->l_nom + (size_t)k * (n + m) * h->batch * E;
  if (x) {
    rc = h->dtype == ALTRO_HIP_F64 ? lane_get<double>(h, x, rec, nullptr, n + m, 0, 0, n, 1, 1)
                                   : lane_get<float>(h, x, rec, nullptr, n + m, 0, 0, n, 1, 1);
    if (rc) return rc;
  }
  if (u)
    rc = h->dtype == ALTRO_HIP_F64 ? lane_get<double>(h, u, rec, nullptr, n + m, n, 0, m, 1, 1)
                                   : lane_get<float>(h, u, rec, nullptr, n + m, n, 0, m, 1, 1);
  return rc;
}

int altro_hip_add_linear_constraint(altro_hip_batch* h, int k_first, int k_last, int cone, int p, const double* G,
                                    const double* g, int g_per_problem) {
  // ALTROSolver::SetConstraint (altro_solver.cpp:175-215) for c(x,u) = G [x;u] - g
  int rc = check(h);
  if (rc) return rc;
  if (h->plan != ALTRO_HIP_PLAN_LANE && h->plan != ALTRO_HIP_PLAN_MFMA16)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "constraints need plan LANE or MFMA16");
  if (!G || !g) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "G and g are required");
  if (cone < CONE_EQUALITY || cone > CONE_SOC) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "unknown cone %d", cone);
  const int pmax = cone == CONE_SOC ? AL_MAXSOC : AL_MAXP;
  if (p < 1 || p > pmax) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "constraint dimension %d outside [1, %d]", p, pmax);
  if (k_first < 0 || k_last > h->N || k_first > k_last)
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "knot point range [%d, %d] outside [0, %d] (ErrorCodes::BadIndex)", k_first, k_last, h->N);
  if ((int)h->al_defs.size() >= AL_MAXDEF) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "at most %d constraint blocks", AL_MAXDEF);
  for (int k = k_first; k <= k_last; ++k)
    if (h->al_knots[k].ncon >= AL_MAXC)
      return fail(ALTRO_HIP_ERR_UNSUPPORTED, "at most %d constraint blocks per knot point (k = %d)", AL_MAXC, k);
  const int w = h->n + h->m;
  AlDef d{cone, p, g_per_problem ? 1 : 0, (int)h->al_G.size(), 0};
  h->al_G.insert(h->al_G.end(), G, G + (size_t)p * w);
  h->al_g.emplace_back(g, g + (size_t)p * (g_per_problem ? h->batch : 1));
  const int id = (int)h->al_defs.size();
  h->al_defs.push_back(d);
  for (int k = k_first; k <= k_last; ++k) {
    AlKnot& kn = h->al_knots[k];
    kn.def[kn.ncon++] = id;
  }
  h->al_dirty = true;
  return id;
}
int altro_hip_clear_constraints(altro_hip_batch* h) {
  int rc = check(h);
  if (rc) return rc;
  h->al_defs.clear(); h->al_G.clear(); h->al_g.clear();
  h->al_knots.assign((size_t)h->N + 1, AlKnot{});
  h->al_dirty = true;
  return al_upload(h);
}
int altro_hip_reset_duals(altro_hip_batch* h, double penalty) {
  // duals back to zero and every constraint's penalty to `penalty` (what a fresh Initialize leaves: 1)
  int rc = check(h);
  if (rc) return rc;
  if (h->plan != ALTRO_HIP_PLAN_LANE && h->plan != ALTRO_HIP_PLAN_MFMA16) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "constraints need plan LANE or MFMA16");
  if (!(penalty > 0.0)) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "penalty must be positive");
  if ((rc = al_upload(h))) return rc;
  const size_t E = h->dtype == ALTRO_HIP_F64 ? 8 : 4;
  if (h->al_d_z) HIP_TRY(hipMemsetAsync(h->al_d_z, 0, (size_t)h->al_rows * h->batch * E, h->stream));
  std::vector<IlqrProb> pr(h->batch);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(pr.data(), h->i_prob, pr.size() * sizeof(IlqrProb), hipMemcpyDeviceToHost));
  for (auto& q : pr) { q.rho = penalty; q.rho_est = penalty; }
  HIP_TRY(hipMemcpy(h->i_prob, pr.data(), pr.size() * sizeof(IlqrProb), hipMemcpyHostToDevice));
  return 0;
}
int altro_hip_get_duals(altro_hip_batch* h, int k, int slot, double* z) {
  // duals of constraint block `slot` of knot point k, [batch][p]
  int rc = check(h);
  if (rc) return rc;
  if (h->plan != ALTRO_HIP_PLAN_LANE && h->plan != ALTRO_HIP_PLAN_MFMA16) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "constraints need plan LANE or MFMA16");
  if ((rc = al_upload(h))) return rc;
  if (k < 0 || k > h->N || slot < 0 || slot >= h->al_knots[k].ncon || !z)
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "no constraint block %d at knot point %d", slot, k);
  const AlKnot& kn = h->al_knots[k];
  const int p = h->al_defs[kn.def[slot]].p;
  const int64_t B = h->batch;
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->dtype == ALTRO_HIP_F64) {
    std::vector<double> t((size_t)p * B);
    HIP_TRY(hipMemcpy(t.data(), (const double*)h->al_d_z + (size_t)kn.z_off[slot] * B, t.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t b = 0; b < B; ++b)
      for (int r = 0; r < p; ++r) z[(size_t)b * p + r] = t[(size_t)r * B + b];
  } else {
    std::vector<float> t((size_t)p * B);
    HIP_TRY(hipMemcpy(t.data(), (const float*)h->al_d_z + (size_t)kn.z_off[slot] * B, t.size() * 4, hipMemcpyDeviceToHost));
    for (int64_t b = 0; b < B; ++b)
      for (int r = 0; r < p; ++r) z[(size_t)b * p + r] = t[(size_t)r * B + b];
  }
  return 0;
}

int altro_hip_ilqr_solve(altro_hip_batch* h, const altro_hip_solve_options* opts,
                         altro_hip_solve_result* results) {
  // SolverImpl::Solve (solver.cpp:414-511) for every problem of the batch at once.  The host only
  // sequences launches and reads two counters per step; all per-problem decisions are on the device.
  int rc = ilqr_check(h, true);
  if (rc) return rc;
  altro_hip_solve_options o;
  if (opts) o = *opts;
  else altro_hip_default_solve_options(&o);
  IlqrLoopArgs la;
  la.prob = h->i_prob; la.alpha = h->i_alpha; la.active = h->i_active; la.phi = h->i_phi; la.dphi = h->i_dphi;
  la.counters = h->i_counters; la.batch = h->batch; la.iter = 0; la.iterations_max = o.iterations_max;
  la.tol_stationarity = o.tol_stationarity; la.tol_meritfun_gradient = o.tol_meritfun_gradient;
  la.tol_primal_feasibility = o.tol_primal_feasibility;
  la.penalty_initial = o.penalty_initial; la.penalty_scaling = o.penalty_scaling; la.penalty_max = o.penalty_max;
  const bool al = !h->al_defs.empty();
  la.al_enabled = al ? 1 : 0;
  la.reg = h->i_reg; la.bwd_status = h->status;
  la.spec_trials = 1; la.spec_pre = 0; la.spec_sel = h->i_spec_sel; la.spec_refresh = h->i_spec_refresh;
  la.reg_initial = o.reg_initial; la.reg_scale = o.reg_scale; la.reg_min = o.reg_min; la.reg_max = o.reg_max;
  const bool reg_on = o.reg_retry_max > 0 || o.reg_initial > 0.0;
  if (reg_on && h->plan != ALTRO_HIP_PLAN_LANE)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "the regularisation schedule is a plan-LANE feature");
  if (o.reg_initial < 0.0 || (o.reg_retry_max > 0 && !(o.reg_scale > 1.0 && o.reg_min > 0.0 && o.reg_max >= o.reg_min)))
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "regularisation retry needs reg_initial >= 0, reg_scale > 1, 0 < reg_min <= reg_max");
  if (al && !(o.penalty_initial > 0.0 && o.penalty_scaling > 0.0 && o.penalty_max > 0.0))
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "penalty_initial, penalty_scaling and penalty_max must be positive");
  la.ls = ls_default_options();
  la.ls.try_cubic_first = 1;                                   // solver.cpp:248
  la.ls.use_backtracking = o.use_backtracking_linesearch;      // solver.cpp:417
  h->spec_beta = la.ls.beta_decrease; h->spec_max_iters = la.ls.max_iters;
  int counters[3];
  auto read_counters = [&]() -> int {
    HIP_TRY(hipMemcpyAsync(counters, h->i_counters, sizeof(counters), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
  };
  auto zero_counter = [&](int idx) -> int {
    HIP_TRY(hipMemsetAsync(h->i_counters + idx, 0, sizeof(int), h->stream));
    return 0;
  };
  // initial rollout, make it the nominal trajectory, expand everything (solver.cpp:420-434)
  if (ilqr_launch_loop(h->stream, ILK_LOOP_INIT, la)) return fail(ALTRO_HIP_ERR_HIP, "iLQR loop kernel launch failed");
  rc = ilqr_run(h, IK_ROLLOUT, false, false, 0, 0.0);
  if (!rc) rc = ilqr_run(h, IK_ACCEPT, false, false, 0, 0.0);
  // without constraints the cost Hessian is constant and is written once, here; with them the gradient is
  // formed with the penalty the constraints carry so far and SetPenalty comes after it (solver.cpp:424-430)
  if (!rc) rc = ilqr_run(h, IK_EXPAND, false, false, 0, 0.0, al ? EXPAND_GRADIENT : (EXPAND_GRADIENT | EXPAND_HESSIAN));
  if (rc) return rc;
  if (al && ilqr_launch_loop(h->stream, ILK_SET_PENALTY, la)) return fail(ALTRO_HIP_ERR_HIP, "iLQR loop kernel launch failed");
  int total_merit_launches = 0, sweeps = 0;
  // speculative backtracking: how much of the chip the searching problems occupy, and how much there is
  const bool spec_all_on = std::getenv("ALTRO_HIP_NO_SPECULATION") == nullptr;
  const bool spec_on = o.use_backtracking_linesearch != 0 && spec_all_on;
  int running = h->batch;   // problems still iterating (counters[1] of the previous sweep)
  const bool lane_plan = h->plan == ALTRO_HIP_PLAN_LANE;
  auto spec_units = [&](int searching) -> int {   // wavefronts one merit launch keeps busy
    return lane_plan ? (h->batch + 63) / 64 : searching;   // LANE: the searching lanes are scattered over all waves
  };
  const int64_t spec_capacity = lane_plan ? 512 : 4096;    // two waves per CU (LANE: latency-bound; more slow each other down) / four per SIMD (MFMA16)
  const int64_t cand_elems = (int64_t)h->batch * (h->N + 1) * (lane_plan ? lane_sizes(h->n, h->m).e_xuy : 28);
  const size_t spare_bytes = (size_t)(ILQR_SPEC_TRIALS - 1) * cand_elems * h->esz;   // spare candidate trajectories
  h->spec_trials = 1;
  struct MaskGuard {   // the backward sweep skips problems that have stopped, only inside this loop
    altro_hip_batch* h;
    ~MaskGuard() { h->bwd_active = nullptr; h->bwd_reg = nullptr; }
  } mask_guard{h};
  h->bwd_active = h->i_active;
  h->bwd_reg = reg_on ? h->i_reg : nullptr;
  int total_reg_retries = 0;
  for (int iter = 0; iter < o.iterations_max; ++iter) {
    la.iter = iter;
    if (ilqr_launch_loop(h->stream, ILK_MARK_RUNNING, la)) return fail(ALTRO_HIP_ERR_HIP, "iLQR loop kernel launch failed");
    if (al) {                                                   // CalcExpansions: cost Hessians (solver.cpp:448)
      rc = ilqr_run(h, IK_EXPAND, false, true, 0, 0.0, EXPAND_HESSIAN);
      if (rc) return rc;
    }
    rc = launch_backward(h, 0.0);                               // BackwardPass (reg = 0, solver.cpp:363)
    if (rc) return rc;
    h->backward_done = true;
    for (int attempt = 0; attempt < o.reg_retry_max; ++attempt) {   // extension: repeat failed problems with more reg
      if ((rc = zero_counter(2))) return rc;
      if (ilqr_launch_loop(h->stream, ILK_REG_RETRY, la)) return fail(ALTRO_HIP_ERR_HIP, "iLQR loop kernel launch failed");
      if ((rc = read_counters())) return rc;
      if (counters[2] == 0) break;
      total_reg_retries += counters[2];
      rc = launch_backward(h, 0.0);
      if (rc) return rc;
    }
    if (o.reg_retry_max > 0 && ilqr_launch_loop(h->stream, ILK_MARK_RUNNING, la)) return fail(ALTRO_HIP_ERR_HIP, "iLQR loop kernel launch failed");
    // ForwardPass: phi(0), then the line search (solver.cpp:237-271).  While the running problems leave half of the
    // chip idle, the first step the search will ask for (alpha0 = 1, known in advance) rides in the same launch as
    // phi(0) -- phi, phi' and the trajectory go to spare row / buffer 0 -- and ILK_LS_BEGIN consumes it at once.
    bool refreshed = false;
    bool pre = spec_all_on && !h->spec_no_memory && (int64_t)spec_units(running) * 2 <= spec_capacity;
    if (pre && !h->i_cand_spec && dmalloc(h, &h->i_cand_spec, spare_bytes)) {
      (void)hipGetLastError();    // an optimisation only: carry on one step per launch
      h->spec_no_memory = true;
      pre = false;
    }
    h->spec_trials = pre ? 2 : 1; h->spec_pre = pre ? 1 : 0;
    rc = ilqr_run(h, IK_MERIT, true, true, 1, 0.0);
    h->spec_trials = 1; h->spec_pre = 0;
    if (rc) return rc;
    ++total_merit_launches;
    if ((rc = zero_counter(0))) return rc;
    la.spec_pre = pre ? 1 : 0;
    if (ilqr_launch_loop(h->stream, ILK_LS_BEGIN, la)) return fail(ALTRO_HIP_ERR_HIP, "iLQR loop kernel launch failed");
    la.spec_pre = 0;
    if (pre) {
      rc = ilqr_run(h, IK_SPEC_SELECT, false, false, 0, 0.0);
      if (rc) return rc;
      refreshed = true;
    } else {
      // The first trial step is launched without asking the device whether any problem needs it: the masks make it
      // a no-op when none does, and it saves one host read-back per sweep (these loops are latency-bound).
      rc = ilqr_run(h, IK_MERIT, true, true, 1, 0.0);
      if (rc) return rc;
      ++total_merit_launches;
      if ((rc = zero_counter(0))) return rc;
      if (ilqr_launch_loop(h->stream, ILK_LS_FEED, la)) return fail(ALTRO_HIP_ERR_HIP, "iLQR loop kernel launch failed");
    }
    if ((rc = read_counters())) return rc;
    int guard = 0;
    while (counters[0] > 0 && guard++ < 64) {
      // Speculative backtracking: once the problems still searching leave most of the chip idle, one launch
      // evaluates the next 2, 4 or 8 steps of the (known) sequence alpha beta^j for each of them; the feed
      // kernel consumes them in order, so every decision is the sequential one (kernels/ilqr_types.h).
      int trials = 1;
      if (spec_on && !h->spec_no_memory)
        while (trials < ILQR_SPEC_TRIALS && (int64_t)spec_units(counters[0]) * trials * 2 <= spec_capacity) trials *= 2;
      if (trials > 1 && !h->i_cand_spec && dmalloc(h, &h->i_cand_spec, spare_bytes)) {
        (void)hipGetLastError();
        h->spec_no_memory = true;
        trials = 1;
      }
      const bool spec = trials > 1;
      h->spec_trials = trials;
      la.spec_trials = h->spec_trials;
      rc = ilqr_run(h, IK_MERIT, true, true, 1, 0.0);
      if (rc) return rc;
      ++total_merit_launches;
      if ((rc = zero_counter(0))) return rc;
      if (ilqr_launch_loop(h->stream, ILK_LS_FEED, la)) return fail(ALTRO_HIP_ERR_HIP, "iLQR loop kernel launch failed");
      if (spec) {
        rc = ilqr_run(h, IK_SPEC_SELECT, false, false, 0, 0.0);
        if (rc) return rc;
        refreshed = true;
      }
      h->spec_trials = 1;
      la.spec_trials = 1;
      if ((rc = read_counters())) return rc;
    }
    if (refreshed) {   // steps accepted from a speculative trial carry no phi' pass: redo their expansion (what the
                       // derivative pass of a sequential trial would have left behind)
      int* keep = h->i_active;
      h->i_active = h->i_spec_refresh;
      rc = ilqr_run(h, IK_EXPAND, false, true, 0, 0.0, EXPAND_GRADIENT);
      h->i_active = keep;
      if (rc) return rc;
    }
    // convergence criteria on the accepted candidate, then make it the nominal (solver.cpp:459-469)
    if (ilqr_launch_loop(h->stream, ILK_MARK_RUNNING, la)) return fail(ALTRO_HIP_ERR_HIP, "iLQR loop kernel launch failed");
    rc = ilqr_run(h, IK_STATIONARITY, false, true, 0, 0.0);
    if (!rc) rc = ilqr_run(h, IK_ACCEPT, false, true, 0, 0.0);
    if (rc) return rc;
    if ((rc = zero_counter(1))) return rc;
    if (ilqr_launch_loop(h->stream, ILK_FINISH_ITER, la)) return fail(ALTRO_HIP_ERR_HIP, "iLQR loop kernel launch failed");
    if (al) {   // DualUpdate, PenaltyUpdate, refreshed gradients for the problems that asked (solver.cpp:470-489)
      rc = ilqr_run(h, IK_DUAL, false, false, 0, 0.0);
      if (rc) return rc;
      if (ilqr_launch_loop(h->stream, ILK_PENALTY_UPDATE, la)) return fail(ALTRO_HIP_ERR_HIP, "iLQR loop kernel launch failed");
      rc = ilqr_run(h, IK_EXPAND, false, true, 0, 0.0, EXPAND_GRADIENT);
      if (rc) return rc;
    }
    if ((rc = read_counters())) return rc;
    ++sweeps;
    if (counters[1] == 0) break;
    running = counters[1];
  }
  h->forward_done = true;
  if (results) {
    std::vector<IlqrProb> pr(h->batch);
    HIP_TRY(hipMemcpy(pr.data(), h->i_prob, pr.size() * sizeof(IlqrProb), hipMemcpyDeviceToHost));
    for (int b = 0; b < h->batch; ++b) {
      results[b].status = pr[b].status;
      results[b].iterations = pr[b].iterations;
      results[b].stationarity = pr[b].stationarity;
      results[b].final_alpha = pr[b].alpha;
      results[b].final_phi = pr[b].ls_iters > 0 ? pr[b].ls.phi : pr[b].phi0;
      results[b].primal_feasibility = pr[b].feasibility;
      results[b].penalty = pr[b].rho;
      results[b].dual_updates = pr[b].n_dual_updates;
      results[b].reg_retries = pr[b].reg_retries;
    }
  }
  h->last_sweeps = sweeps;
  h->last_merit_launches = total_merit_launches;
  return 0;
}

void altro_hip_default_solve_options(altro_hip_solve_options* o) {
  if (!o) return;
  o->iterations_max = 200;            // solver_options.hpp:16-39
  o->tol_stationarity = 1e-4;
  o->tol_primal_feasibility = 1e-4;
  o->tol_meritfun_gradient = 1e-8;
  o->use_backtracking_linesearch = 0;
  o->penalty_initial = 1.0;
  o->penalty_scaling = 10.0;
  o->penalty_max = 1e8;
  o->reg_initial = 0.0;     // the reference: reg = 0, failures ignored (solver.cpp:363, :449)
  o->reg_retry_max = 0;
  o->reg_scale = 10.0;
  o->reg_min = 1e-6;
  o->reg_max = 1e8;
}
int altro_hip_last_solve_counts(const altro_hip_batch* h, int* sweeps, int* merit_launches) {
  if (!h) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "null handle");
  if (sweeps) *sweeps = h->last_sweeps;
  if (merit_launches) *merit_launches = h->last_merit_launches;
  return 0;
}
int altro_hip_get_nominal(altro_hip_batch* h, double* x, double* u) {
  int rc = check(h);
  if (rc) return rc;
  const int n = h->n, m = h->m, N = h->N;
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {
    if (!h->m_nom) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_tracking_cost has not been called");
    const int64_t B = h->batch;
    if (x) {
      rc = h->dtype == ALTRO_HIP_F64 ? aos_get<double>(h, x, (const double*)h->m_nom, MF_NOM, B * MF_NOM, n, N + 1)
                                     : aos_get<float>(h, x, (const float*)h->m_nom, MF_NOM, B * MF_NOM, n, N + 1);
      if (rc) return rc;
    }
    if (u)
      rc = h->dtype == ALTRO_HIP_F64 ? aos_get<double>(h, u, (const double*)h->m_nom + 12, MF_NOM, B * MF_NOM, m, N)
                                     : aos_get<float>(h, u, (const float*)h->m_nom + 12, MF_NOM, B * MF_NOM, m, N);
    return rc;
  }
  if (h->plan != ALTRO_HIP_PLAN_LANE) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "nominal trajectory exists for plans LANE and MFMA16");
  if (x) {
    rc = h->dtype == ALTRO_HIP_F64 ? lane_get<double>(h, x, h->l_nom, nullptr, n + m, 0, 0, n, N + 1, N + 1)
                                   : lane_get<float>(h, x, h->l_nom, nullptr, n + m, 0, 0, n, N + 1, N + 1);
    if (rc) return rc;
  }
  if (u)
    rc = h->dtype == ALTRO_HIP_F64 ? lane_get<double>(h, u, h->l_nom, nullptr, n + m, n, 0, m, N, N)
                                   : lane_get<float>(h, u, h->l_nom, nullptr, n + m, n, 0, m, N, N);
  return rc;
}
// Expansion the backward pass will consume: A | B | lx | lu (reference layout), for parity tests.
int altro_hip_get_expansion(altro_hip_batch* h, double* A, double* B, double* lx, double* lu) {
  int rc = check(h);
  if (rc) return rc;
  if (h->plan != ALTRO_HIP_PLAN_LANE) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "plan LANE only");
  const int n = h->n, m = h->m, N = h->N;
  const LaneSizes z = lane_sizes(n, m);
  const int oq = 2 * n * n + 2 * n * m + m * m + n, orr = oq + n;
  auto get = [&](double* dst, int off, int len, int nk, bool with_term, int off_term) -> int {
    if (!dst) return 0;
    return h->dtype == ALTRO_HIP_F64
               ? lane_get<double>(h, dst, h->l_in, with_term ? h->l_term : nullptr, z.e_in, off, off_term, len, nk, N)
               : lane_get<float>(h, dst, h->l_in, with_term ? h->l_term : nullptr, z.e_in, off, off_term, len, nk, N);
  };
  rc = get(A, 0, n * n, N, false, 0);
  if (!rc) rc = get(B, n * n, n * m, N, false, 0);
  if (!rc) rc = get(lx, oq, n, N + 1, true, n * n);
  if (!rc) rc = get(lu, orr, m, N, false, 0);
  return rc;
}

// The line-search state machine driven on the host by a callback: lets CPU-only tests pin it against
// the real reference line search (oracle/_ref) without a GPU.
double altro_hip_linesearch_host(altro_hip_merit_fn f, void* ctx, double alpha0, double phi0, double dphi0,
                                 int try_cubic_first, int use_backtracking, double c1, double c2,
                                 int* status, int* iters, double* phi, double* dphi) {
  LsOptions o = ls_default_options();
  o.try_cubic_first = try_cubic_first; o.use_backtracking = use_backtracking; o.c1 = c1; o.c2 = c2;
  LsState s;
  bool need = ls_begin(s, o, alpha0, phi0, dphi0);
  while (need) {
    double p = 0.0, dp = 0.0;
    f(s.alpha, &p, s.want_derivative ? &dp : nullptr, ctx);
    need = ls_feed(s, o, p, dp);
  }
  if (status) *status = s.status;
  if (iters) *iters = s.n_iters;
  if (phi) *phi = s.phi;
  if (dphi) *dphi = s.dphi;
  return s.alpha;
}

// MFMA layout self-test (tests/test_gpu_parity.py): max |D - (A B + C)| for random operands.
double altro_hip_selftest_mfma_f64(int device) {
  if (hipSetDevice(device) != hipSuccess) return -1.0;
  double hA[64], hB[64], hC[256], hD[256];
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1 << 24) - 0.5; };
  for (double& v : hA) v = rnd();
  for (double& v : hB) v = rnd();
  for (double& v : hC) v = rnd();
  double *dA, *dB, *dC, *dD;
  if (hipMalloc(&dA, sizeof(hA)) != hipSuccess || hipMalloc(&dB, sizeof(hB)) != hipSuccess ||
      hipMalloc(&dC, sizeof(hC)) != hipSuccess || hipMalloc(&dD, sizeof(hD)) != hipSuccess)
    return -1.0;
  (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  (void)hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma16_selftest_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
  if (hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost) != hipSuccess) return -1.0;
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dD);
  double worst = 0.0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double ref = hC[i * 16 + j];
      for (int k = 0; k < 4; ++k) ref += hA[i * 4 + k] * hB[k * 16 + j];
      worst = std::max(worst, std::abs(ref - hD[i * 16 + j]));
    }
  return worst;
}

}  // extern "C"
