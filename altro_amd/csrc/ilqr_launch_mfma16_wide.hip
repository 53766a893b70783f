// Kernel instantiations of plan MFMA16's merit and expansion passes for handles with more than AL_MAXC constraint slots at a knot
// point (kernels/al_types.h: AL_TILE_MAXC slots of AL_MAXP rows -- an input box AND a state box on a (12, 4) problem): the
// NC = AL_TILE_MAXC instantiations of wave_merit_dpp_kernel and wave_expand_dpp_kernel.  A translation unit of their own: the two-slot
// kernels of ilqr_launch_mfma16.hip stay what they were, register for register.  (What keeps the six-slot merit kernel at two waves
// per SIMD without spills -- 251 registers -- is in kernels/ilqr_merit2_dpp.hip: one set of (z, g) registers, asked for after use,
// and the gradient's column sums taken with the rows.  Before those two changes: one wave per SIMD or 64-114 spilled registers,
// the two-trial pass of C1 + input box + state box 2.4-2.5 ms; a four-slot instantiation 1.04 ms)
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels/ilqr_mfma16.hip"
#include "kernels/ilqr_merit2_dpp.hip"

namespace altro_hip {

#define WIDE_MODELS(X) X(MODEL_QUADROTOR)

// 0 ok, 1 = no such kernel, 2 = launch error
template <>
int ilqr_wave_launch_wide<double>(hipStream_t stream, int which, const IlqrWaveArgs<double>& a) {
  using S = double;
  constexpr int W = AL_TILE_MAXC;
  const dim3 b64(64);
  const unsigned gsh = (unsigned)a.al.Gpad_count * 8u;   // the padded constraint Jacobians in dynamic LDS
  const int trials = a.spec_trials > 1 ? a.spec_trials : 1;
  const dim3 pairs(mf_grid((a.batch + 1) / 2), which == IK_MERIT ? (trials + 1) / 2 : 1);
  if (!a.al.enabled) return 1;
  if (which == IK_EXPAND) {   // four (problem, knot point) pairs per wave
    const dim3 grid((unsigned)((int64_t)((a.batch + 3) / 4) * (a.N + 1)));
    if (a.cost_dense) hipLaunchKernelGGL((wave_expand_dpp_kernel<S, true, false, W>), grid, b64, 0, stream, a);
    else if (a.al.all_sel) hipLaunchKernelGGL((wave_expand_dpp_kernel<S, false, true, W>), grid, b64, 0, stream, a);
    else hipLaunchKernelGGL((wave_expand_dpp_kernel<S, false, false, W>), grid, b64, 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
  }
  if (which != IK_MERIT && which != IK_MERIT2) return 1;
  if (a.mp.kind != MODEL_LINEAR) {
    bool done = false;
#define X(K_)                                                                                                                        \
    if (!done && a.mp.kind == K_) {                                                                                                  \
      done = true;                                                                                                                   \
      if (which == IK_MERIT) {                                                                                                       \
        if (a.cost_dense) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, false, true, K_, true, false, W>), pairs, b64, gsh, stream, a);   \
        else hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, false, false, K_, true, false, W>), pairs, b64, gsh, stream, a);               \
      } else {                                                                                                                       \
        if (a.cost_dense) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, true, true, K_, true, false, W>), pairs, b64, gsh, stream, a);    \
        else hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, true, false, K_, true, false, W>), pairs, b64, gsh, stream, a);                \
      }                                                                                                                              \
    }
    WIDE_MODELS(X)
#undef X
    if (!done) return 1;
    return hipGetLastError() == hipSuccess ? 0 : 2;
  }
  // (SOC_: the second-order cone's code only for handles that have one -- it costs the kernel ~20 registers)
#define WIDE_MERIT(DUAL_, DENSE_, AFF_, GRID_)                                                                                              \
  do {                                                                                                                                       \
    if (a.al.has_soc) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, DUAL_, DENSE_, 0, true, AFF_, W>), GRID_, b64, gsh, stream, a);     \
    else hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, DUAL_, DENSE_, 0, false, AFF_, W>), GRID_, b64, gsh, stream, a);                 \
  } while (0)
  if (which == IK_MERIT && a.aff) {   // affine trials: a chunk of knot points per wave, then the chunks' shares added up
    const int chunks = (a.N + MD_AFF_CHUNK - 1) / MD_AFF_CHUNK;
    const dim3 grid(mf_grid((a.batch + 1) / 2), (trials + 1) / 2, chunks);
    if (a.cost_dense) WIDE_MERIT(false, true, true, grid);
    else WIDE_MERIT(false, false, true, grid);
    const int64_t tot = (int64_t)ILQR_SPEC_TRIALS * a.batch;
    hipLaunchKernelGGL(wave_aff_reduce_kernel<S>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, a, chunks);
  } else if (which == IK_MERIT) {
    if (a.cost_dense) WIDE_MERIT(false, true, false, pairs);
    else WIDE_MERIT(false, false, false, pairs);
  } else {
    if (a.cost_dense) WIDE_MERIT(true, true, false, pairs);
    else WIDE_MERIT(true, false, false, pairs);
  }
#undef WIDE_MERIT
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
// (fp32 records: the wide form is not instantiated -- altro_hip_add_linear_constraint says so when a block would need it)
template <>
int ilqr_wave_launch_wide<float>(hipStream_t, int, const IlqrWaveArgs<float>&) { return 1; }

}  // namespace altro_hip
