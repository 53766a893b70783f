// row32_model.hip -- the row-layout MeritFunction kernels (kernels/ilqr_row32.hip) with a compiled-in device model past the (12, 4)
// tile: MODEL_QUADROTOR13 at (13, 4), the model ilqr_generic_model_supported (ilqr_launch_generic.hip) knows.  A unit of its own: every
// lane of the kernel evaluates the model and its two Jacobians in registers.
#include <hip/hip_runtime.h>

#include "kernels/ilqr_generic.hip"
#include "kernels/ilqr_row32.hip"

namespace altro_hip {

// kind: 0 the merit kernel (IK_MERIT), 3 the two-trial pass (IK_MERIT2), 5 the dynamics expansion of a stored trajectory.  0 launched, 1 = no kernel for this model, 2 = launch error
int row32_model_launch(hipStream_t stream, const IlqrGenArgs<double>& a, int kind) {
  if (a.mp.kind != MODEL_QUADROTOR13 || a.n != 13 || a.m != 4) return 1;
  // One wave per SIMD: the model's Jacobians next to the kernel's ring of fetched blocks want ~420 registers, and a wave that spills
  // them (187 registers at two waves per SIMD) waits for its scratch reloads behind the ring's loads: 1.04 ms per evaluation of 4096
  // vehicles x 30 knot points against 0.31 like this (generic_merit_kernel<.., MK>: 0.96).
  if (kind == 0)
    hipLaunchKernelGGL((row32_merit_kernel<double, 13, 4, 1, false, MODEL_QUADROTOR13>), dim3((unsigned)((a.batch + 1) / 2)), dim3(64), 0, stream, a);
  else if (kind == 5)   // A_k, B_k of the stored candidate trajectory (IK_EXPAND with EXPAND_DYN)
    hipLaunchKernelGGL((row32_expand_dyn_kernel<double, 13, 4, MODEL_QUADROTOR13>), dim3((unsigned)(((int64_t)a.batch * a.N + 1) / 2)), dim3(64), 0, stream, a);
  else if (kind == 3)
    hipLaunchKernelGGL((row32_merit_kernel<double, 13, 4, 1, true, MODEL_QUADROTOR13>), dim3((unsigned)a.batch), dim3(64), 0, stream, a);
  else
    return 1;
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace altro_hip
