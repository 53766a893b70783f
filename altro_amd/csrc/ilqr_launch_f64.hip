// Kernel instantiations of the batched iLQR loop for element type double (see kernels/ilqr_types.h).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels/ilqr_lane.hip"
#include "kernels/ilqr_loop_kernels.hip"

namespace altro_hip {

#define ILQR_MODELS(X)                                                                      \
  X(MODEL_PENDULUM, 2, 1) X(MODEL_BICYCLE, 4, 2) X(MODEL_DOUBLE_INTEGRATOR, 2, 1)           \
  X(MODEL_DOUBLE_INTEGRATOR, 4, 2) X(MODEL_DOUBLE_INTEGRATOR, 6, 3)

template <>
int ilqr_launch_kernel<double>(hipStream_t stream, int which, int kind, int n, int m, const IlqrArgs<double>& a) {
  using T = double;
  const dim3 lanes((a.batch + 63) / 64), b64(64), b256(256);
  const int64_t total = (int64_t)a.batch * (a.N + 1);
  const dim3 flat((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 20));
  const dim3 flat64((unsigned)std::min<int64_t>((total + 63) / 64, 1 << 20));
  bool done = false;
#define X(K_, N_, M_)                                                                                         \
  if (!done && kind == K_ && n == N_ && m == M_) {                                                            \
    done = true;                                                                                              \
    const dim3 shift((unsigned)std::min<int64_t>(((int64_t)a.batch * (N_ + M_) + 255) / 256, 1 << 20));       \
    switch (which) {                                                                                          \
      case IK_ROLLOUT: hipLaunchKernelGGL((ilqr_rollout_kernel<K_, N_, M_, T>), lanes, b64, 0, stream, a); break;   \
      case IK_ACCEPT: hipLaunchKernelGGL((ilqr_accept_kernel<N_, M_, T>), flat, b256, 0, stream, a); break;         \
      case IK_EXPAND:   /* (a.cost_kind: the instantiation for the dense quadratic cost, kernels/ilqr_lane.hip) */  \
        if (a.cost_kind) hipLaunchKernelGGL((ilqr_expand_kernel<K_, N_, M_, T, 1>), flat64, b64, 0, stream, a);       \
        else hipLaunchKernelGGL((ilqr_expand_kernel<K_, N_, M_, T>), flat64, b64, 0, stream, a);                      \
        break;                                                                                                \
      case IK_MERIT: {                                                                                        \
        const unsigned trials = a.spec_trials > 1 ? a.spec_trials : 1;                                        \
        if (a.merit_jk) {   /* rollout | per-knot-point terms on the whole chip | sums and phi' */            \
          hipLaunchKernelGGL((ilqr_merit_roll_kernel<K_, N_, M_, T>), dim3(lanes.x, trials), b64, 0, stream, a);   \
          if (a.cost_kind) hipLaunchKernelGGL((ilqr_merit_point_kernel<K_, N_, M_, T, 1>), dim3(flat64.x, trials), b64, 0, stream, a); \
          else hipLaunchKernelGGL((ilqr_merit_point_kernel<K_, N_, M_, T>), dim3(flat64.x, trials), b64, 0, stream, a); \
          hipLaunchKernelGGL((ilqr_merit_sum_kernel<K_, N_, M_, T>), dim3(lanes.x, trials), b64, 0, stream, a);    \
        } else if (a.cost_kind) {                                                                             \
          hipLaunchKernelGGL((ilqr_merit_kernel<K_, N_, M_, T, 1>), dim3(lanes.x, trials), b64, 0, stream, a); \
        } else {                                                                                              \
          hipLaunchKernelGGL((ilqr_merit_kernel<K_, N_, M_, T>), dim3(lanes.x, trials), b64, 0, stream, a);    \
        }                                                                                                     \
        break;                                                                                                \
      }                                                                                                       \
      case IK_SPEC_SELECT: hipLaunchKernelGGL((ilqr_spec_select_kernel<N_, M_, T>), flat, b256, 0, stream, a); break; \
      case IK_DUAL: hipLaunchKernelGGL((ilqr_dual_update_kernel<N_, M_, T>), flat, b256, 0, stream, a); break;      \
      case IK_SHIFT: hipLaunchKernelGGL((ilqr_shift_kernel<N_, M_, T>), shift, b256, 0, stream, a); break;          \
      default:                                                                                                \
        hipLaunchKernelGGL(ilqr_zero_residuals_kernel<T>, dim3((a.batch + 255) / 256), b256, 0, stream, a);    \
        hipLaunchKernelGGL((ilqr_stationarity_kernel<N_, M_, T>), flat64, b64, 0, stream, a);                   \
        break;                                                                                                \
    }                                                                                                         \
  }
  ILQR_MODELS(X)
#undef X
  if (!done) return 1;
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

bool ilqr_supported(int kind, int n, int m) {
#define X(K_, N_, M_) if (kind == K_ && n == N_ && m == M_) return true;
  ILQR_MODELS(X)
#undef X
  return false;
}

int ilqr_launch_results(hipStream_t stream, const IlqrProb* prob, IlqrResult* out, int batch) {
  hipLaunchKernelGGL(ilqr_results_kernel, dim3((batch + 255) / 256), dim3(256), 0, stream, prob, out, batch);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

int ilqr_launch_list_running(hipStream_t stream, const int* flags, const int* list_in, int count_in, int* list_out, int resident) {
  hipLaunchKernelGGL(ilqr_list_running_kernel, dim3(1), dim3(1024), 0, stream, flags, list_in, count_in, list_out, resident);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

int ilqr_launch_loop(hipStream_t stream, int which, const IlqrLoopArgs& a) {
  const dim3 gb((a.batch + 255) / 256), bb(256);
  switch (which) {
    case ILK_LOOP_INIT: hipLaunchKernelGGL(ilqr_loop_init_kernel, gb, bb, 0, stream, a); break;
    case ILK_LS_BEGIN: hipLaunchKernelGGL(ilqr_ls_begin_kernel, gb, bb, 0, stream, a); break;
    case ILK_LS_FEED: hipLaunchKernelGGL(ilqr_ls_feed_kernel, gb, bb, 0, stream, a); break;
    case ILK_FINISH_ITER: hipLaunchKernelGGL(ilqr_finish_iter_kernel, gb, bb, 0, stream, a); break;
    case ILK_MARK_RUNNING: hipLaunchKernelGGL(ilqr_mark_running_kernel, gb, bb, 0, stream, a); break;
    case ILK_SET_PENALTY: hipLaunchKernelGGL(ilqr_set_penalty_kernel, gb, bb, 0, stream, a); break;
    case ILK_PENALTY_UPDATE: hipLaunchKernelGGL(ilqr_penalty_update_kernel, gb, bb, 0, stream, a); break;
    case ILK_REG_RETRY: hipLaunchKernelGGL(ilqr_reg_retry_kernel, gb, bb, 0, stream, a); break;
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace altro_hip
