// capi_tvlqr.hip -- C ABI: the TVLQR sweep (altro_hip_backward / forward_ltv / sweep) and its per-plan kernel launchers.
// This unit instantiates the sweep kernels of all three plans (GENERIC, LANE, MFMA16 in fp64 / fp32 storage / pure fp32).
#include "capi_internal.h"
#include "models.h"

#include "kernels/tvlqr_mfma16.hip"
#include "kernels/tvlqr_mfma16_f32.hip"
#include "kernels/tvlqr_mfma16_f32x4.hip"
#include "kernels/tvlqr_mfma16_fwd_f32x4.hip"

using namespace altro_hip;
using namespace altro_hip::capi;

// a launch that carries the profiling events of the enclosing ProfScope (null events: a plain launch)
#define PROF_LAUNCH(kernel, grid, block, lds, stream, ...) \
  hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, h->launch_ev0, h->launch_ev1, 0, __VA_ARGS__)

namespace {

template <typename T>
int lane_launch(altro_hip_batch* h, bool backward, double reg) {
  LaneArgs<T> a{(const T*)h->l_in, (const T*)h->l_term, (T*)h->l_out, (T*)h->l_outn, (const T*)h->l_x0,
                (T*)h->l_xuy, (T*)h->delta_V, h->status, h->N, h->batch, (T)reg, backward ? h->bwd_active : nullptr,
                backward ? h->bwd_reg : nullptr};
  const dim3 grid(8 * (((h->batch + 63) / 64 + 7) / 8)), block(64);
  const bool fused = (h->flags & ALTRO_HIP_LANE_FUSED) != 0;
  // (4, 2) and (2, 1): four lanes per problem (kernels/tvlqr_quad_body.inc, tvlqr_quad2_body.inc), same records, bit-identical
  // results -- while the batch leaves the chip idle.  Measured on MI355X (tools/c3_quad_ab.sh, profiles/r03i_quad_ab.txt,
  // backward + forward ms, quad / lane-per-problem): (4, 2) 0.12 / 0.21 at 4096 problems, 0.16 / 0.23 at 8192, 0.28 / 0.27 at
  // 16384, 0.56 / 0.44 at 32768, 1.10 / 0.84 at 65536; (2, 1) 0.10 / 0.12, 0.11 / 0.12, 0.16 / 0.14, 0.31 / 0.25, 0.60 / 0.49.
  // The quad form shortens the chain of one wave (what bounds small batches); once every SIMD has its waves the sweep is
  // HBM-bound and the lane-per-problem form's 512-byte runs stream better than the quad's four 128-byte segments.
  // ALTRO_HIP_FORM_LANE_QUAD_OFF / _ON forces one or the other.
  const bool quad_on = form(h, ALTRO_HIP_FORM_LANE_QUAD_ON) ? true : form(h, ALTRO_HIP_FORM_LANE_QUAD_OFF) ? false : h->batch <= 12288;
  const bool q42 = h->n == 4 && h->m == 2, q21 = h->n == 2 && h->m == 1;
  if (backward) h->bwd_quad = quad_on && (q42 || q21);
  if (!backward) h->fwd_quad = quad_on && q42;
  if (!backward && quad_on && q42) {   // forward sweep of (4, 2): four lanes per problem as well
    const dim3 qgrid(8 * (((h->batch + 15) / 16 + 7) / 8));
    if (fused) PROF_LAUNCH((quad_forward_kernel_fused<2, T>), qgrid, block, 0, h->stream, a);
    else PROF_LAUNCH((quad_forward_kernel<2, T>), qgrid, block, 0, h->stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "quad kernel launch: %s", hipGetErrorString(e));
    return 0;
  }
  if (backward && h->bwd_quad) {
    const dim3 qgrid(8 * (((h->batch + 15) / 16 + 7) / 8));
    if (q42 && fused) PROF_LAUNCH((quad_backward_kernel_fused<2, T>), qgrid, block, 0, h->stream, a);
    else if (q42) PROF_LAUNCH((quad_backward_kernel<2, T>), qgrid, block, 0, h->stream, a);
    else if (fused) PROF_LAUNCH((quad2_backward_kernel_fused<T>), qgrid, block, 0, h->stream, a);
    else PROF_LAUNCH((quad2_backward_kernel<T>), qgrid, block, 0, h->stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "quad kernel launch: %s", hipGetErrorString(e));
    return 0;
  }
#define X(N_, M_)                                                                                              \
  if (h->n == N_ && h->m == M_) {                                                                              \
    if (backward && fused) PROF_LAUNCH((lane_backward_kernel_fused<N_, M_, T>), grid, block, 0, h->stream, a); \
    else if (backward) PROF_LAUNCH((lane_backward_kernel<N_, M_, T>), grid, block, 0, h->stream, a);    \
    else if (fused) PROF_LAUNCH((lane_forward_kernel_fused<N_, M_, T>), grid, block, 0, h->stream, a);  \
    else PROF_LAUNCH((lane_forward_kernel<N_, M_, T>), grid, block, 0, h->stream, a);                   \
  }
  LANE_SHAPES(X)
#undef X
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "lane kernel launch: %s", hipGetErrorString(e));
  return 0;
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
  a.no_f = h->ilqr_linear ? 1 : 0;
  return a;
}

// One profiled kernel launch (altro_hip_profile_enable).  The launch itself carries the two events -- hipExtLaunchKernelGGL
// stamps them from the dispatch's own completion signal -- instead of two hipEventRecord calls around it: those are marker
// packets with barriers on the handle's stream, 15 us per sweep of a timed region whose sweeps take 1.45 ms
// (profiles/r03zc_event_overhead.txt), and they time the packets' turnaround along with the kernel.
struct ProfScope {
  altro_hip_batch* h;
  int slot, idx = -1;
  static void add(altro_hip_batch* h, int slot, float ms) {
    h->prof_min[slot] = h->prof_launches[slot] ? std::min(h->prof_min[slot], (double)ms) : (double)ms;
    h->prof_max[slot] = std::max(h->prof_max[slot], (double)ms);
    h->prof_ms[slot] += ms;
    h->prof_launches[slot] += 1;
  }
  ProfScope(altro_hip_batch* h_, int slot_) : h(h_), slot(slot_) {
    h->launch_ev0 = nullptr; h->launch_ev1 = nullptr;
    if (h->prof == 1) { h->launch_ev0 = h->ev0; h->launch_ev1 = h->ev1; }
    else if (h->prof == 2) {
      if (2 * (h->prof_n + 1) <= (int)h->prof_ev.size()) {
        idx = h->prof_n++;
        h->prof_slot[idx] = slot;
        h->launch_ev0 = h->prof_ev[2 * idx]; h->launch_ev1 = h->prof_ev[2 * idx + 1];
      } else {
        h->prof_dropped[slot] += 1;   // the ring is full: say so instead of averaging over a silent prefix
      }
    }
  }
  ~ProfScope() {
    if (h->prof == 1 && h->launch_ev1) {
      (void)hipEventSynchronize(h->ev1);
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) add(h, slot, ms);
    }
    h->launch_ev0 = nullptr; h->launch_ev1 = nullptr;   // (mode 2: altro_hip_profile_get reads the pairs, no wait here)
  }
};


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
  a.active = h->bwd_active; a.reg_pp = h->bwd_reg;   // set only inside altro_hip_ilqr_solve
  return a;
}
template <typename S>
void mfma16_launch_backward(altro_hip_batch* h, double reg, bool sq) {
  auto a = mfma16_args<S>(h, reg);
  const dim3 grid(mf_grid(h->batch)), block(64);
  if (sq && a.has_f) PROF_LAUNCH((mfma16_backward_kernel<true, true, S>), grid, block, 0, h->stream, a);
  else if (sq) PROF_LAUNCH((mfma16_backward_kernel<true, false, S>), grid, block, 0, h->stream, a);
  else if (a.has_f) PROF_LAUNCH((mfma16_backward_kernel<false, true, S>), grid, block, 0, h->stream, a);
  else PROF_LAUNCH((mfma16_backward_kernel<false, false, S>), grid, block, 0, h->stream, a);
}

template <typename S>
void mfma16_launch_forward(altro_hip_batch* h, const Mfma16Args<S>& a) {
  // register-ring depth 3: depths 1..4 were measured, 3 is the knee
  PROF_LAUNCH((mfma16_forward_kernel<S, 3>), dim3(mf_grid(h->batch)), dim3(64), 0, h->stream, a);
}

}  // namespace

namespace altro_hip {
namespace capi {

int launch_backward(altro_hip_batch* h, double reg) {
  ProfScope ps(h, 0);
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {
    const bool sq = (h->flags & ALTRO_HIP_STORE_QBLOCKS) != 0;
    if (h->dtype == ALTRO_HIP_F64) mfma16_launch_backward<double>(h, reg, sq);
    else if (sq || !(h->flags & ALTRO_HIP_F32_PURE)) mfma16_launch_backward<float>(h, reg, sq);   // fp32 storage, fp64 tiles
    else {   // opt-in: pure fp32 on v_mfma_f32_16x16x4_f32
      auto a = mfma16_args<float>(h, reg);
      // four problems per wave (kernels/tvlqr_mfma16_f32x4.hip) whenever the batch allows it; the one-problem kernel stays for
      // batches that are not whole quads.  Ring depth 2, two waves per SIMD: measured on C4 against (1,2) (1,3) (3,1) (3,2) (4,1) --
      // 3.40 ms against 4.1-4.4 (profiles/r02g_c4_quad_variants.txt); the other instantiations left with round 5's clean-up
      if (h->batch % 4 == 0) {
        const dim3 grid(mf_grid(h->batch / 4));
        if (a.has_f) PROF_LAUNCH((mfma16_backward_f32x4_kernel<true, 2, 2>), grid, dim3(64), 0, h->stream, a);
        else PROF_LAUNCH((mfma16_backward_f32x4_kernel<false, 2, 2>), grid, dim3(64), 0, h->stream, a);
      } else if (a.has_f) PROF_LAUNCH((mfma16_backward_f32_kernel<true, 3>), dim3(mf_grid(h->batch)), dim3(64), 0, h->stream, a);
      else PROF_LAUNCH((mfma16_backward_f32_kernel<false, 3>), dim3(mf_grid(h->batch)), dim3(64), 0, h->stream, a);
    }
  } else if (h->plan == ALTRO_HIP_PLAN_LANE) {
    return h->dtype == ALTRO_HIP_F64 ? lane_launch<double>(h, true, reg) : lane_launch<float>(h, true, reg);
  } else if (h->g_tile && !h->is_diag) {   // plan MFMA32 (a diagonal cost keeps its diagonals packed: plan GENERIC's own kernel reads those)
    int rc = tile32_launch_backward(h, reg);
    if (rc) return rc;
  } else {
    // the knot point's blocks in LDS while they fit; the form without a block for Qxx ("late Q", 3 n^2 instead of 4 n^2 elements) where
    // that keeps more problems on a CU than the batch otherwise gets, or keeps the blocks in LDS at all; past that the same kernel on a
    // per-problem work block in global memory (n, m beyond ~37 in fp64)
    const bool f64 = h->dtype == ALTRO_HIP_F64;
    const size_t lds4 = f64 ? generic_backward_lds_bytes<double>(h->n, h->m) : generic_backward_lds_bytes<float>(h->n, h->m);
    const size_t lds3 = f64 ? generic_backward_lds_bytes<double>(h->n, h->m, true) : generic_backward_lds_bytes<float>(h->n, h->m, true);
    const auto per_cu = [](size_t lds) { return (int)std::min<size_t>(16, (160 * 1024) / std::max<size_t>(lds, 1)); };   // (16: four waves per SIMD by registers)
    const int cus = 256;
    bool late_q = lds4 > kGenericLdsLimit ? lds3 <= kGenericLdsLimit
                                          : (per_cu(lds3) > per_cu(lds4) && (int64_t)h->batch > (int64_t)cus * per_cu(lds4));
    if (form(h, ALTRO_HIP_FORM_GENERIC_LATE_Q_ON)) late_q = lds3 <= kGenericLdsLimit;
    else if (form(h, ALTRO_HIP_FORM_GENERIC_LATE_Q_OFF)) late_q = false;
    const size_t lds = late_q ? lds3 : lds4;
    const bool big = lds > kGenericLdsLimit;
    if (big && !h->g_ws) {
      int rc = dmalloc(h, &h->g_ws, (size_t)h->batch * ((lds + 15) / 16 * 16));
      if (rc) return rc;
    }
    if (f64) {
      auto a = generic_args<double>(h, reg);
      a.ws = (double*)h->g_ws; a.ws_stride = (int64_t)((lds + 15) / 16 * 16 / sizeof(double));
      const dim3 grid(h->batch), blk(64);
      if (h->g_mfma) {   // the products on the matrix cores (ALTRO_HIP_GENERIC_MATRIX_CORES)
        if (big) PROF_LAUNCH((generic_backward_kernel<double, true, true>), grid, blk, 0, h->stream, a);
        else if (late_q) PROF_LAUNCH((generic_backward_kernel<double, false, true, true>), grid, blk, lds, h->stream, a);
        else PROF_LAUNCH((generic_backward_kernel<double, false, true>), grid, blk, lds, h->stream, a);
      } else if (big) PROF_LAUNCH((generic_backward_kernel<double, true>), grid, blk, 0, h->stream, a);
      else if (late_q) PROF_LAUNCH((generic_backward_kernel<double, false, false, true>), grid, blk, lds, h->stream, a);
      else PROF_LAUNCH((generic_backward_kernel<double, false>), grid, blk, lds, h->stream, a);
    } else {
      auto a = generic_args<float>(h, reg);
      a.ws = (float*)h->g_ws; a.ws_stride = (int64_t)((lds + 15) / 16 * 16 / sizeof(float));
      if (big) PROF_LAUNCH((generic_backward_kernel<float, true>), dim3(h->batch), dim3(64), 0, h->stream, a);
      else if (late_q) PROF_LAUNCH((generic_backward_kernel<float, false, false, true>), dim3(h->batch), dim3(64), lds, h->stream, a);
      else PROF_LAUNCH((generic_backward_kernel<float, false>), dim3(h->batch), dim3(64), lds, h->stream, a);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "backward launch: %s", hipGetErrorString(e));
  return 0;
}

// the pure-fp32 forward sweep with four problems per wave: fp32 records, fp32 arithmetic asked for, a batch of whole quads
bool mfma16_forward_is_x4(const altro_hip_batch* h) {
  return h->plan == ALTRO_HIP_PLAN_MFMA16 && h->dtype == ALTRO_HIP_F32 && (h->flags & ALTRO_HIP_F32_PURE) && h->batch % 4 == 0;
}

int launch_forward(altro_hip_batch* h) {
  ProfScope ps(h, 1);
  h->expansion_current = false;   // the sweep writes the candidate trajectory: whatever expansion a merit pass left is not of this point
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {
    if (h->dtype == ALTRO_HIP_F64) {
      auto a = mfma16_args<double>(h, 0.0);
      mfma16_launch_forward<double>(h, a);
    } else {
      auto a = mfma16_args<float>(h, 0.0);
      if (mfma16_forward_is_x4(h)) {   // pure fp32, four problems per wave (kernels/tvlqr_mfma16_fwd_f32x4.hip)
        const dim3 grid(mf_grid(h->batch / 4));
        // ring depth 2, two waves per SIMD: measured on C4 on two boxes against seven other (depth, waves) pairs and the one-problem
        // kernel (profiles/r03e_ / r03f_c4_fwd_variants.txt: 2.47 / 2.375 ms; the others 2.39-2.84)
        PROF_LAUNCH((mfma16_forward_f32x4_kernel<2, 2>), grid, dim3(64), 0, h->stream, a);
      } else {
        mfma16_launch_forward<float>(h, a);
      }
    }
  } else if (h->plan == ALTRO_HIP_PLAN_LANE) {
    return h->dtype == ALTRO_HIP_F64 ? lane_launch<double>(h, false, 0.0) : lane_launch<float>(h, false, 0.0);
  } else if (h->g_tile) {
    int rc = tile32_launch_forward(h);
    if (rc) return rc;
  } else if (h->dtype == ALTRO_HIP_F64) {
    // the knot point's blocks staged in LDS and fetched a knot point ahead while sixteen waves still fit a CU (10 KB each: measured at
    // 4096 problems, (13, 4) 1.53 -> 1.21 ms, (16, 4) 1.62 -> 1.27, but (24, 8) at 13 KB 2.37 -> 3.58); else rows from global memory
    auto a = generic_args<double>(h, 0.0);
    const size_t staged = generic_forward_lds_bytes(h->n, h->m, a.want_y, sizeof(double));
    if (staged <= kGenericForwardStageLimit) PROF_LAUNCH((generic_forward_kernel<double, true>), dim3(h->batch), dim3(64), staged, h->stream, a);
    else PROF_LAUNCH((generic_forward_kernel<double, false>), dim3(h->batch), dim3(64), (size_t)(2 * h->n + h->m) * sizeof(double) + 64, h->stream, a);
  } else {
    auto a = generic_args<float>(h, 0.0);
    const size_t staged = generic_forward_lds_bytes(h->n, h->m, a.want_y, sizeof(float));
    if (staged <= kGenericForwardStageLimit) PROF_LAUNCH((generic_forward_kernel<float, true>), dim3(h->batch), dim3(64), staged, h->stream, a);
    else PROF_LAUNCH((generic_forward_kernel<float, false>), dim3(h->batch), dim3(64), (size_t)(2 * h->n + h->m) * sizeof(float) + 64, h->stream, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "forward launch: %s", hipGetErrorString(e));
  return 0;
}

}  // namespace capi
}  // namespace altro_hip

extern "C" {

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

// MFMA layout self-test (tests/test_gpu_parity.py): max |D - (A B + C)| for random operands.
namespace {
__global__ void sincos_selftest_kernel(const double* x, int n, double* s, double* c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sincos_hd<double>(x[i], &s[i], &c[i]);
}
}  // namespace
// the device's fp64 sincos (models.h) on n arguments: the accuracy test of tests/test_gpu_trig.py
int altro_hip_selftest_sincos(int device, const double* x, int n, double* s, double* c) {
  if (!x || !s || !c || n <= 0) return ALTRO_HIP_ERR_BAD_ARGUMENT;
  if (hipSetDevice(device) != hipSuccess) return ALTRO_HIP_ERR_NO_DEVICE;
  double *dx = nullptr, *ds = nullptr, *dc = nullptr;
  const size_t bytes = (size_t)n * sizeof(double);
  int rc = 0;
  if (hipMalloc((void**)&dx, bytes) != hipSuccess || hipMalloc((void**)&ds, bytes) != hipSuccess ||
      hipMalloc((void**)&dc, bytes) != hipSuccess) rc = ALTRO_HIP_ERR_OUT_OF_MEMORY;
  if (!rc && hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ALTRO_HIP_ERR_HIP;
  if (!rc) {
    hipLaunchKernelGGL(sincos_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dx, n, ds, dc);
    if (hipMemcpy(s, ds, bytes, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(c, dc, bytes, hipMemcpyDeviceToHost) != hipSuccess)
      rc = ALTRO_HIP_ERR_HIP;
  }
  if (dx) (void)hipFree(dx);
  if (ds) (void)hipFree(ds);
  if (dc) (void)hipFree(dc);
  return rc;
}
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

// 4-block layout self-test (tests/test_gpu_parity.py): max |D_b - (A_b B_b + C_b)| over the four blocks of
// v_mfma_f32_16x16x1_4b_f32 with the layout kernels/tvlqr_mfma16_f32x4.hip assumes.
double altro_hip_selftest_mfma_f32_4b(int device) {
  if (hipSetDevice(device) != hipSuccess) return -1.0;
  float hA[64], hB[64], hC[1024], hD[1024];
  unsigned s = 2468u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / (1 << 24) - 0.5f; };
  for (float& v : hA) v = rnd();
  for (float& v : hB) v = rnd();
  for (float& v : hC) v = rnd();
  float *dA, *dB, *dC, *dD;
  if (hipMalloc(&dA, sizeof(hA)) != hipSuccess || hipMalloc(&dB, sizeof(hB)) != hipSuccess ||
      hipMalloc(&dC, sizeof(hC)) != hipSuccess || hipMalloc(&dD, sizeof(hD)) != hipSuccess)
    return -1.0;
  (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  (void)hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma16_selftest_4b_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
  if (hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost) != hipSuccess) return -1.0;
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dD);
  double worst = 0.0;
  for (int b = 0; b < 4; ++b)
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        const double ref = (double)hC[b * 256 + i * 16 + j] + (double)hA[16 * b + i] * (double)hB[16 * b + j];
        worst = std::max(worst, std::abs(ref - (double)hD[b * 256 + i * 16 + j]));
      }
  return worst;
}

}  // extern "C"
