// Kernel instantiations of the iLQR loop for plan MFMA16 (see kernels/ilqr_mfma16.hip, kernels/ilqr_types.h).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels/ilqr_mfma16.hip"
#include "kernels/ilqr_merit2_dpp.hip"

namespace altro_hip {

template <typename S>
static int wave_launch(hipStream_t stream, int which, const IlqrWaveArgs<S>& a) {
  const dim3 waves(mf_grid(a.batch)), b64(64), b256(256);   // XCD-aware problem mapping: kernels/mfma16_layout.h
  const unsigned gsh = a.al.enabled ? (unsigned)a.al.Gpad_count * 8u : 0u;   // wave_merit_dpp_kernel's dynamic LDS: the padded constraint Jacobians
  const int64_t flat_n = (int64_t)a.batch * (a.N + 1) * 16;
  const dim3 flat((unsigned)std::min<int64_t>((flat_n + 255) / 256, 1 << 20));
  // more than two constraint slots at some knot point: the merit kernel's wide instantiation (ilqr_launch_mfma16_wide.hip)
  if ((which == IK_MERIT || which == IK_MERIT2) && a.al.enabled && a.al.max_ncon > AL_MAXC)
    return ilqr_wave_launch_wide<S>(stream, which, a);
  if (a.mp.kind != MODEL_LINEAR) {   // a device model instead of dynamics as data: kernels/ilqr_tile_model.hip
    if (which == IK_ROLLOUT || which == IK_MERIT || which == IK_MERIT2) return ilqr_wave_launch_model<S>(stream, which, a);
    if (which == IK_EXPAND && (a.mode & EXPAND_DYN)) {
      const int rc = ilqr_wave_launch_model<S>(stream, IK_EXPAND, a);
      if (rc) return rc;
    }
  }
  switch (which) {
    case IK_ROLLOUT: hipLaunchKernelGGL(wave_rollout_kernel<S>, waves, b64, 0, stream, a); break;
    case IK_ACCEPT: hipLaunchKernelGGL(wave_accept_kernel<S>, flat, b256, 0, stream, a); break;
    case IK_EXPAND:
      if (a.al.enabled && a.al.max_ncon > AL_MAXC)
        return ilqr_wave_launch_wide<S>(stream, IK_EXPAND, a);
      if (a.cost_dense) {   // the dense quadratic cost: row-layout kernels only (capi_ilqr.hip keeps EXPAND_LDS off)
        const int64_t blocks = ((int64_t)a.batch * (a.N + 1) * 16 + 255) / 256;
        if (a.al.enabled) hipLaunchKernelGGL((wave_expand_dpp_kernel<S, true>), dim3((unsigned)((int64_t)((a.batch + 3) / 4) * (a.N + 1))), b64, 0, stream, a);
        else if (a.mode & EXPAND_GRADIENT) hipLaunchKernelGGL(wave_expand_grad_dense_kernel<S>, dim3((unsigned)(blocks < 262144 ? blocks : 262144)), dim3(256), 0, stream, a);
      } else if (a.al.enabled && (a.mode & EXPAND_LDS) == 0) {   // four (problem, knot point) pairs per wave, kernels/ilqr_merit2_dpp.hip
        if (a.al.all_sel) hipLaunchKernelGGL((wave_expand_dpp_kernel<S, false, true>), dim3((unsigned)((int64_t)((a.batch + 3) / 4) * (a.N + 1))), b64, 0, stream, a);
        else hipLaunchKernelGGL((wave_expand_dpp_kernel<S, false>), dim3((unsigned)((int64_t)((a.batch + 3) / 4) * (a.N + 1))), b64, 0, stream, a);
      } else if (a.al.enabled) {
        hipLaunchKernelGGL(wave_expand_kernel<S>, dim3((unsigned)((int64_t)a.batch * (a.N + 1))), b64, 0, stream, a);
      } else if (a.mode & EXPAND_GRADIENT) {   // no constraint blocks: the Hessian is constant, the gradient is 16 entries
        const int64_t blocks = ((int64_t)a.batch * (a.N + 1) * 16 + 255) / 256;
        hipLaunchKernelGGL(wave_expand_grad_kernel<S>, dim3((unsigned)(blocks < 262144 ? blocks : 262144)), dim3(256), 0, stream, a);
      }
      break;
    case IK_DUAL:
      if (a.mode & STAT_NO_FEAS) hipLaunchKernelGGL(wave_dual_update_dpp_kernel<S>, dim3((unsigned)((int64_t)((a.batch + 3) / 4) * (a.N + 1))), b64, 0, stream, a);
      else hipLaunchKernelGGL(wave_dual_update_kernel<S>, dim3((unsigned)((int64_t)a.batch * (a.N + 1))), b64, 0, stream, a);
      break;
    case IK_MERIT:
      if constexpr (sizeof(S) == 8) {
        if (a.aff) {   // affine trials (dynamics as data): a chunk of knot points per wave, then the chunks' shares added up
          const int chunks = (a.N + MD_AFF_CHUNK - 1) / MD_AFF_CHUNK;
          const dim3 grid(mf_grid((a.batch + 1) / 2), ((a.spec_trials > 1 ? a.spec_trials : 1) + 1) / 2, chunks);
          if (a.cost_dense) {
            if (a.al.enabled && !a.al.has_soc) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, false, true, 0, false, true>), grid, b64, gsh, stream, a);
            else if (a.al.enabled) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, false, true, 0, true, true>), grid, b64, gsh, stream, a);
            else hipLaunchKernelGGL((wave_merit_dpp_kernel<S, false, false, true, 0, true, true>), grid, b64, gsh, stream, a);
          } else {
            if (a.al.enabled && !a.al.has_soc) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, false, false, 0, false, true>), grid, b64, gsh, stream, a);
            else if (a.al.enabled) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, false, false, 0, true, true>), grid, b64, gsh, stream, a);
            else hipLaunchKernelGGL((wave_merit_dpp_kernel<S, false, false, false, 0, true, true>), grid, b64, gsh, stream, a);
          }
          const int64_t tot = (int64_t)ILQR_SPEC_TRIALS * a.batch;
          hipLaunchKernelGGL(wave_aff_reduce_kernel<S>, dim3((unsigned)((tot + 255) / 256)), b256, 0, stream, a, chunks);
          break;
        }
      }
      // the DPP form: two problems per wave, two trials per problem (kernels/ilqr_merit2_dpp.hip).  (With constraint blocks and
      // ONE trial per problem its second rows idle; while the constraint Jacobians came from global memory the LDS form won
      // those rounds on long horizons.  With G in LDS and the duals fetched a step ahead the DPP form wins them too -- C1 + input
      // bounds, cubic search, whole solves: N = 192: 69.7 vs 70.9 ms, N = 256: 105.2 vs 108.9.)
      if (a.cost_dense) {   // the dense quadratic cost lives in the row-layout kernel only
        const dim3 grid(mf_grid((a.batch + 1) / 2), ((a.spec_trials > 1 ? a.spec_trials : 1) + 1) / 2);
        if (a.al.enabled && !a.al.has_soc) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, false, true, 0, false>), grid, b64, gsh, stream, a);
        else if (a.al.enabled) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, false, true>), grid, b64, gsh, stream, a);
        else hipLaunchKernelGGL((wave_merit_dpp_kernel<S, false, false, true>), grid, b64, gsh, stream, a);
      } else if (a.mode == 3 || a.mode == 2) {   // (3: ALTRO_HIP_MERIT_DPP=2, kept for the tests that force the form)
        const dim3 grid(mf_grid((a.batch + 1) / 2), ((a.spec_trials > 1 ? a.spec_trials : 1) + 1) / 2);
        if (a.al.enabled && !a.al.has_soc) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, false, false, 0, false>), grid, b64, gsh, stream, a);
        else if (a.al.enabled) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, false>), grid, b64, gsh, stream, a);
        else hipLaunchKernelGGL((wave_merit_dpp_kernel<S, false, false>), grid, b64, gsh, stream, a);
      } else if (a.al.enabled) hipLaunchKernelGGL((wave_merit_kernel<S, true>), dim3(waves.x, a.spec_trials > 1 ? a.spec_trials : 1), b64, 0, stream, a);
      else hipLaunchKernelGGL((wave_merit_kernel<S, false>), dim3(waves.x, a.spec_trials > 1 ? a.spec_trials : 1), b64, 0, stream, a);
      break;
    case IK_MERIT2:
      if (a.cost_dense && a.al.enabled && !a.al.has_soc) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, true, true, 0, false>), dim3(mf_grid((a.batch + 1) / 2)), b64, gsh, stream, a);
      else if (a.cost_dense && a.al.enabled) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, true, true>), dim3(mf_grid((a.batch + 1) / 2)), b64, gsh, stream, a);
      else if (a.cost_dense) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, false, true, true>), dim3(mf_grid((a.batch + 1) / 2)), b64, gsh, stream, a);
      else if (a.al.enabled && !a.al.has_soc) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, true, false, 0, false>), dim3(mf_grid((a.batch + 1) / 2)), b64, gsh, stream, a);
      else if (a.al.enabled) hipLaunchKernelGGL((wave_merit_dpp_kernel<S, true, true>), dim3(mf_grid((a.batch + 1) / 2)), b64, gsh, stream, a);
      else hipLaunchKernelGGL((wave_merit_dpp_kernel<S, false, true>), dim3(mf_grid((a.batch + 1) / 2)), b64, gsh, stream, a);   // two problems per wave
      break;
    case IK_SPEC_SELECT: hipLaunchKernelGGL(wave_spec_select_kernel<S>, dim3(a.batch), b64, 0, stream, a); break;
    case IK_STATIONARITY:
      if (a.al.enabled && (a.mode & STAT_NO_FEAS)) {   // the residual here, the constraint rows four problems per wave
        hipLaunchKernelGGL(wave_stationarity_kernel<S>, waves, b64, 0, stream, a);
        hipLaunchKernelGGL(wave_feasibility_dpp_kernel<S>, dim3((unsigned)((int64_t)((a.batch + 3) / 4) * (a.N + 1))), b64, 0, stream, a);
      } else {
        IlqrWaveArgs<S> a2 = a;
        a2.mode &= ~STAT_NO_FEAS;
        hipLaunchKernelGGL(wave_stationarity_kernel<S>, waves, b64, 0, stream, a2);
      }
      break;
    case IK_SHIFT: hipLaunchKernelGGL(wave_shift_kernel<S>, dim3((a.batch * 16 + 255) / 256), b256, 0, stream, a); break;
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
template <>
int ilqr_wave_launch_kernel<double>(hipStream_t stream, int which, const IlqrWaveArgs<double>& a) { return wave_launch<double>(stream, which, a); }
template <>
int ilqr_wave_launch_kernel<float>(hipStream_t stream, int which, const IlqrWaveArgs<float>& a) { return wave_launch<float>(stream, which, a); }

}  // namespace altro_hip
