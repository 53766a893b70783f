// capi_tile32.h -- plan MFMA32: what the launcher units share.  The backward kernel is instantiated per shape (n, m as compile-time
// constants, kernels/tvlqr_tile32.hip); the 156 shapes are spread over eight translation units by n mod 8 (tile32_bwd_unit.inc) so that
// they compile side by side.
#pragma once
#include "capi_internal.h"

#include "kernels/tvlqr_tile32.hip"

namespace altro_hip {
namespace capi {

// the shapes plan MFMA32 takes (AUTO's rule: the (12, 4) tile of plan MFMA16 keeps n <= 12, m <= 4) ...
constexpr bool tile32_shape_ok(int n, int m) {
  return n >= 5 && n <= T32_MAX_N && m >= 1 && m <= T32_MAX_M && n + m <= 32 && !(n <= 12 && m <= 4);
}
// ... and the shapes a backward kernel is instantiated for: those, plus n = 7 .. 12 with m <= 4 for the single-problem seam
// (tvlqr_dropin.hip: one problem has no use for the four-problems-per-wave tile of plan MFMA16, and n <= 6, m <= 3 rides a lane)
constexpr bool tile32_kernel_ok(int n, int m) {
  return n >= 5 && n <= T32_MAX_N && m >= 1 && m <= T32_MAX_M && n + m <= 32 && !(n <= 6 && m <= 4);
}
// waves per SIMD an instantiation's registers are budgeted for: 3 lets the compiler use up to 168, and the n <= 15, m <= 4 kernels
// come out at <= 128 without a spill (four waves per SIMD: 4096 problems in one round); 2 = 256 registers for the rest
constexpr int tile32_wps(int n, int m) { return (m <= 4 && n <= 16) ? 3 : 2; }
// the m > 4 shapes keep an 8 x 8 Cholesky factor in registers: their lane-dependent addresses are recomputed every knot point
constexpr bool tile32_launder(int n, int m) { return m > 4; }

// where a launch goes: the handle's stream and profiling events, or a bare stream (the seam)
struct Tile32Launch {
  hipStream_t stream;
  hipEvent_t ev0, ev1;
};

template <int N_, int M_>
int tile32_backward_static(const Tile32Launch& at, const Tile32Args& a) {
  if constexpr (tile32_kernel_ok(N_, M_)) {
    constexpr Tile32Lds L = tile32_lds_layout(N_, M_);
    constexpr size_t lds = (size_t)L.total * sizeof(double);
    static_assert(lds <= 64 * 1024, "the images of one knot point fit the default LDS window");
    hipExtLaunchKernelGGL((tile32_backward_kernel<N_, M_, tile32_wps(N_, M_), tile32_launder(N_, M_)>), dim3(mf_grid(a.batch)), dim3(64), lds,
                          at.stream, at.ev0, at.ev1, 0, a);
    return 0;
  } else {
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "plan MFMA32 has no kernel for (n, m) = (%d, %d)", N_, M_);
  }
}

// one per translation unit tile32_bwd_R.hip: the shapes with n mod 8 == R
int tile32_backward_unit0(const Tile32Launch& at, const Tile32Args& a);
int tile32_backward_unit1(const Tile32Launch& at, const Tile32Args& a);
int tile32_backward_unit2(const Tile32Launch& at, const Tile32Args& a);
int tile32_backward_unit3(const Tile32Launch& at, const Tile32Args& a);
int tile32_backward_unit4(const Tile32Launch& at, const Tile32Args& a);
int tile32_backward_unit5(const Tile32Launch& at, const Tile32Args& a);
int tile32_backward_unit6(const Tile32Launch& at, const Tile32Args& a);
int tile32_backward_unit7(const Tile32Launch& at, const Tile32Args& a);

// the unit that holds (a.n, a.m)
int tile32_backward_dispatch(const Tile32Launch& at, const Tile32Args& a);

}  // namespace capi
}  // namespace altro_hip
