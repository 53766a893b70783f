// capi_tile32.h -- plan MFMA32: what the launcher units share.  The backward kernel is instantiated per shape (n, m as compile-time
// constants, kernels/tvlqr_tile32.hip); the 156 shapes are spread over eight translation units by n mod 8 (tile32_bwd_unit.inc) so that
// they compile side by side.
#pragma once
#include "capi_internal.h"

#include "kernels/tvlqr_tile32.hip"

namespace altro_hip {
namespace capi {

constexpr bool tile32_shape_ok(int n, int m) {
  return n >= 5 && n <= T32_MAX_N && m >= 1 && m <= T32_MAX_M && n + m <= 32 && !(n <= 12 && m <= 4);
}
// waves per SIMD an instantiation's registers are budgeted for: 3 lets the compiler use up to 168, and the n <= 15, m <= 4 kernels
// come out at <= 128 without a spill (four waves per SIMD: 4096 problems in one round); 2 = 256 registers for the rest
constexpr int tile32_wps(int n, int m) { return (m <= 4 && n <= 16) ? 3 : 2; }
// the m > 4 shapes keep an 8 x 8 Cholesky factor in registers: their lane-dependent addresses are recomputed every knot point
constexpr bool tile32_launder(int n, int m) { return m > 4; }

#define T32_PROF_LAUNCH(kernel, grid, block, lds, stream, ...) \
  hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, h->launch_ev0, h->launch_ev1, 0, __VA_ARGS__)

template <int N_, int M_>
int tile32_backward_static(altro_hip_batch* h, const Tile32Args& a) {
  if constexpr (tile32_shape_ok(N_, M_)) {
    constexpr Tile32Lds L = tile32_lds_layout(N_, M_);
    constexpr size_t lds = (size_t)L.total * sizeof(double);
    static_assert(lds <= 64 * 1024, "the images of one knot point fit the default LDS window");
    T32_PROF_LAUNCH((tile32_backward_kernel<N_, M_, tile32_wps(N_, M_), tile32_launder(N_, M_)>), dim3(mf_grid(h->batch)), dim3(64), lds,
                    h->stream, a);
    return 0;
  } else {
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "plan MFMA32 has no kernel for (n, m) = (%d, %d)", N_, M_);
  }
}

// one per translation unit tile32_bwd_R.hip: the shapes with n mod 8 == R
int tile32_backward_unit0(altro_hip_batch* h, const Tile32Args& a);
int tile32_backward_unit1(altro_hip_batch* h, const Tile32Args& a);
int tile32_backward_unit2(altro_hip_batch* h, const Tile32Args& a);
int tile32_backward_unit3(altro_hip_batch* h, const Tile32Args& a);
int tile32_backward_unit4(altro_hip_batch* h, const Tile32Args& a);
int tile32_backward_unit5(altro_hip_batch* h, const Tile32Args& a);
int tile32_backward_unit6(altro_hip_batch* h, const Tile32Args& a);
int tile32_backward_unit7(altro_hip_batch* h, const Tile32Args& a);

}  // namespace capi
}  // namespace altro_hip
