// capi_tile32.hip -- plan MFMA32's sweep launchers: which instantiation of kernels/tvlqr_tile32.hip a shape runs.
// (A translation unit of its own: the instantiations compile beside the other plans' kernels.)
#include "capi_tile32.h"

using namespace altro_hip;
using namespace altro_hip::capi;

#define PROF_LAUNCH(kernel, grid, block, lds, stream, ...) \
  hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, h->launch_ev0, h->launch_ev1, 0, __VA_ARGS__)

namespace {

Tile32Args tile32_args(altro_hip_batch* h, double reg) {
  Tile32Args a{};
  auto P = [&](int arr) { return (double*)h->g_arr[arr]; };
  a.A = P(G_A); a.B = P(G_B); a.f = P(G_f); a.Q = P(G_Q); a.R = P(G_R); a.H = P(G_H); a.q = P(G_q); a.r = P(G_r);
  a.K = P(G_K); a.d = P(G_d); a.P = P(G_P); a.p = P(G_p);
  a.bsA = h->g_bstride[G_A]; a.bsB = h->g_bstride[G_B]; a.bsf = h->g_bstride[G_f]; a.bsQ = h->g_bstride[G_Q];
  a.bsR = h->g_bstride[G_R]; a.bsH = h->g_bstride[G_H]; a.bsq = h->g_bstride[G_q]; a.bsr = h->g_bstride[G_r];
  a.bsK = h->g_bstride[G_K]; a.bsd = h->g_bstride[G_d]; a.bsP = h->g_bstride[G_P]; a.bsp = h->g_bstride[G_p];
  a.x = P(G_x); a.u = P(G_u); a.y = P(G_y);
  a.bsx = h->g_bstride[G_x]; a.bsu = h->g_bstride[G_u]; a.bsy = h->g_bstride[G_y];
  a.x0 = (const double*)h->x0;
  a.delta_V = (double*)h->delta_V; a.status = h->status;
  a.N = h->N; a.batch = h->batch; a.n = h->n; a.m = h->m;
  a.reg = reg;
  a.no_f = (h->ilqr_linear || !h->has_f) ? 1 : 0;
  a.active = h->bwd_active; a.reg_pp = h->bwd_reg;
  a.L = tile32_lds_layout(h->n, h->m);
  return a;
}

template <int KC, int MC>
int forward_launch(altro_hip_batch* h, const Tile32Args& a) {
  const Tile32FwdLds L = tile32_fwd_lds_layout(a.n, a.m);
  const size_t lds = (size_t)L.total * sizeof(double);
  PROF_LAUNCH((tile32_forward_kernel<KC, MC, 2>), dim3(mf_grid(h->batch)), dim3(64), lds, h->stream, a);
  return 0;
}

}  // namespace

namespace altro_hip {
namespace capi {

bool tile32_supported(int n, int m) { return tile32_shape_ok(n, m); }

int tile32_backward_dispatch(const Tile32Launch& at, const Tile32Args& a) {
  switch (a.n & 7) {
    case 0: return tile32_backward_unit0(at, a);
    case 1: return tile32_backward_unit1(at, a);
    case 2: return tile32_backward_unit2(at, a);
    case 3: return tile32_backward_unit3(at, a);
    case 4: return tile32_backward_unit4(at, a);
    case 5: return tile32_backward_unit5(at, a);
    case 6: return tile32_backward_unit6(at, a);
    default: return tile32_backward_unit7(at, a);
  }
}

int tile32_launch_backward(altro_hip_batch* h, double reg) {
  return tile32_backward_dispatch(Tile32Launch{h->stream, h->launch_ev0, h->launch_ev1}, tile32_args(h, reg));
}

int tile32_launch_forward(altro_hip_batch* h) {
  const Tile32Args a = tile32_args(h, 0.0);
  const int kc = (h->n + 3) / 4, mc = (h->m + 3) / 4;
#define T32_CASE(KC_, MC_) if (kc == KC_ && mc == MC_) return forward_launch<KC_, MC_>(h, a);
  T32_CASE(2, 2) T32_CASE(3, 2) T32_CASE(4, 1) T32_CASE(4, 2) T32_CASE(5, 1) T32_CASE(5, 2) T32_CASE(6, 1) T32_CASE(6, 2)
  T32_CASE(7, 1) T32_CASE(7, 2) T32_CASE(8, 1)
#undef T32_CASE
  return fail(ALTRO_HIP_ERR_UNSUPPORTED, "plan MFMA32 has no forward kernel for (n, m) = (%d, %d)", h->n, h->m);
}

}  // namespace capi
}  // namespace altro_hip
