// Kernel instantiations of the iLQR loop for plan GENERIC (kernels/ilqr_generic.hip): any (n, m) up to 32, dynamics as data.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels/ilqr_generic.hip"

namespace altro_hip {

template <typename T>
static int gen_launch(hipStream_t stream, int which, const IlqrGenArgs<T>& a) {
  const dim3 waves(a.batch), b64(64), b256(256);
  const int64_t flat_n = (int64_t)a.batch * (a.N + 1) * (a.n + a.m);
  const dim3 flat((unsigned)std::min<int64_t>((flat_n + 255) / 256, 1 << 20));
  switch (which) {
    case IK_ROLLOUT: hipLaunchKernelGGL(generic_rollout_kernel<T>, waves, b64, 0, stream, a); break;
    case IK_ACCEPT: hipLaunchKernelGGL(generic_accept_kernel<T>, flat, b256, 0, stream, a); break;
    case IK_EXPAND:
      if (a.al.enabled) hipLaunchKernelGGL(generic_expand_al_kernel<T>, dim3((unsigned)((int64_t)a.batch * (a.N + 1))), b64, 0, stream, a);
      else hipLaunchKernelGGL(generic_expand_kernel<T>, flat, b256, 0, stream, a);
      break;
    case IK_DUAL: hipLaunchKernelGGL(generic_dual_update_kernel<T>, dim3((unsigned)((int64_t)a.batch * (a.N + 1))), b64, 0, stream, a); break;
    case IK_MERIT: {   // the knot point's matrices staged in LDS while sixteen waves still fit a CU (10 KB each, the kernel's own 2.5 KB included)
      const size_t jv = a.al.enabled ? (size_t)GEN_AL_JV * sizeof(double) : 0;
      const size_t stage = generic_merit_stage_elems(a.n, a.m) * sizeof(T);
      if (2560 + jv + stage <= 10 * 1024) hipLaunchKernelGGL((generic_merit_kernel<T, true>), waves, b64, jv + stage, stream, a);
      else hipLaunchKernelGGL((generic_merit_kernel<T, false>), waves, b64, jv, stream, a);
      break;
    }
    case IK_STATIONARITY: hipLaunchKernelGGL(generic_stationarity_kernel<T>, waves, b64, 0, stream, a); break;
    case IK_SHIFT: hipLaunchKernelGGL(generic_shift_kernel<T>, dim3((unsigned)(((int64_t)a.batch * (a.n + a.m) + 255) / 256)), b256, 0, stream, a); break;
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
template <>
int ilqr_generic_launch<double>(hipStream_t stream, int which, const IlqrGenArgs<double>& a) { return gen_launch<double>(stream, which, a); }
template <>
int ilqr_generic_launch<float>(hipStream_t stream, int which, const IlqrGenArgs<float>& a) { return gen_launch<float>(stream, which, a); }

}  // namespace altro_hip
