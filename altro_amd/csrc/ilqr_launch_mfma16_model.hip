// Kernel instantiations of plan MFMA16's iLQR loop for NONLINEAR device models (kernels/ilqr_tile_model.hip and the MK
// instantiations of wave_merit_dpp_kernel): a translation unit of their own so that they compile next to the data-dynamics ones.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels/ilqr_mfma16.hip"
#include "kernels/ilqr_merit2_dpp.hip"

namespace altro_hip {

#define TILE_MODELS(X) X(MODEL_QUADROTOR)

bool ilqr_tile_model_supported(int kind, int n, int m) {
#define X(K_) if (kind == K_ && n == 12 && m == 4) return true;
  TILE_MODELS(X)
#undef X
  return false;
}

// 0 ok, 1 = no such model / element type, 2 = launch error
template <>
int ilqr_wave_launch_model<double>(hipStream_t stream, int which, const IlqrWaveArgs<double>& a) {
  using S = double;
  const dim3 b64(64);
  const unsigned gsh = a.al.enabled ? (unsigned)a.al.Gpad_count * 8u : 0u;   // wave_merit_dpp_kernel's dynamic LDS: the padded constraint Jacobians
  const dim3 rows4((unsigned)((a.batch + 3) / 4));
  const dim3 pairs(mf_grid((a.batch + 1) / 2), which == IK_MERIT ? ((a.spec_trials > 1 ? a.spec_trials : 1) + 1) / 2 : 1);
  bool done = false;
#define X(K_)                                                                                                                   \
  if (!done && a.mp.kind == K_) {                                                                                               \
    done = true;                                                                                                                \
    switch (which) {                                                                                                            \
      case IK_ROLLOUT: hipLaunchKernelGGL((wave_rollout_model_kernel<S, K_>), rows4, b64, 0, stream, a); break;                 \
      case IK_EXPAND:                                                                                                           \
        hipLaunchKernelGGL((wave_expand_dyn_kernel<S, K_>), dim3((unsigned)((int64_t)((a.batch + 3) / 4) * a.N)), b64, 0, stream, a); \
        break;                                                                                                                  \
      case IK_MERIT:                                                                                                            \
        if (a.cost_dense && a.al.enabled) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, false, true, K_>), pairs, b64, gsh, stream, a);   \
        else if (a.cost_dense) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, false, false, true, K_>), pairs, b64, gsh, stream, a);             \
        else if (a.al.enabled) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, false, false, K_>), pairs, b64, gsh, stream, a);             \
        else hipLaunchKernelGGL((wave_merit_dpp_kernel<S, false, false, false, K_>), pairs, b64, gsh, stream, a);                              \
        break;                                                                                                                  \
      case IK_MERIT2:                                                                                                           \
        if (a.cost_dense && a.al.enabled) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, true, true, K_>), pairs, b64, gsh, stream, a);    \
        else if (a.cost_dense) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, false, true, true, K_>), pairs, b64, gsh, stream, a);              \
        else if (a.al.enabled) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, true, false, K_>), pairs, b64, gsh, stream, a);              \
        else hipLaunchKernelGGL((wave_merit_dpp_kernel<S, false, true, false, K_>), pairs, b64, gsh, stream, a);                               \
        break;                                                                                                                  \
      default: return 1;                                                                                                        \
    }                                                                                                                           \
  }
  TILE_MODELS(X)
#undef X
  if (!done) return 1;
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
template <>
int ilqr_wave_launch_model<float>(hipStream_t, int, const IlqrWaveArgs<float>&) { return 1; }   // (fp64 records only)

}  // namespace altro_hip
