// Kernel instantiations of the iLQR loop for plan GENERIC (kernels/ilqr_generic.hip): any (n, m) up to 32, dynamics as data.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels/ilqr_generic.hip"

namespace altro_hip {

// the row-layout merit kernels, eight translation units by n mod 8 (row32_unit.inc)
int row32_merit_unit0(hipStream_t, const IlqrGenArgs<double>&, int); int row32_merit_unit1(hipStream_t, const IlqrGenArgs<double>&, int);
int row32_merit_unit2(hipStream_t, const IlqrGenArgs<double>&, int); int row32_merit_unit3(hipStream_t, const IlqrGenArgs<double>&, int);
int row32_merit_unit4(hipStream_t, const IlqrGenArgs<double>&, int); int row32_merit_unit5(hipStream_t, const IlqrGenArgs<double>&, int);
int row32_merit_unit6(hipStream_t, const IlqrGenArgs<double>&, int); int row32_merit_unit7(hipStream_t, const IlqrGenArgs<double>&, int);
static int row32_merit_dispatch(hipStream_t stream, const IlqrGenArgs<double>& a, int kind = 0) {
  switch (a.n & 7) {
    case 0: return row32_merit_unit0(stream, a, kind);
    case 1: return row32_merit_unit1(stream, a, kind);
    case 2: return row32_merit_unit2(stream, a, kind);
    case 3: return row32_merit_unit3(stream, a, kind);
    case 4: return row32_merit_unit4(stream, a, kind);
    case 5: return row32_merit_unit5(stream, a, kind);
    case 6: return row32_merit_unit6(stream, a, kind);
    default: return row32_merit_unit7(stream, a, kind);
  }
}

int row32_model_launch(hipStream_t, const IlqrGenArgs<double>&, int);   // row32_model.hip

bool ilqr_generic_model_supported(int kind, int n, int m) { return kind == MODEL_QUADROTOR13 && n == 13 && m == 4; }

// the kernels that step a compiled-in device model (fp64 handles); everything else of the loop is the data form's
template <int MK, int MN, int MM>
static int gen_launch_model(hipStream_t stream, int which, const IlqrGenArgs<double>& a) {
  const dim3 waves(a.batch), b64(64), b256(256);
  switch (which) {
    case IK_ROLLOUT:
      hipLaunchKernelGGL((generic_model_rollout_kernel<double, MK, MN, MM>), dim3((a.batch + 63) / 64), b64, 0, stream, a);
      break;
    case IK_MERIT: {
      if (a.row32m) {   // two problems per wave in the row layout, every lane evaluating the model (kernels/ilqr_row32.hip: r32_model_step)
        const int rc = row32_model_launch(stream, a, 0);
        if (rc != 1) return rc;
      }
      const size_t jv = a.al.enabled ? (size_t)GEN_AL_JV * sizeof(double) : 0;
      hipLaunchKernelGGL((generic_merit_kernel<double, false, MK, MN, MM>), waves, b64, jv, stream, a);
      break;
    }
    case IK_EXPAND: {   // after the cost expansion (the caller launched it): A_k, B_k at the candidate trajectory
      if (a.row32m) {
        const int rc = row32_model_launch(stream, a, 5);
        if (rc != 1) return rc;
      }
      const int64_t tot = (int64_t)a.batch * a.N;
      hipLaunchKernelGGL((generic_model_expand_dyn_kernel<double, MK, MN, MM>), dim3((unsigned)((tot + 63) / 64)), b64, 0, stream, a);
      break;
    }
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

template <typename T>
static int gen_launch(hipStream_t stream, int which, const IlqrGenArgs<T>& a) {
  const dim3 waves(a.batch), b64(64), b256(256);
  if constexpr (sizeof(T) == 8) {
    if (a.mp.kind == MODEL_QUADROTOR13 && (which == IK_ROLLOUT || which == IK_MERIT)) return gen_launch_model<MODEL_QUADROTOR13, 13, 4>(stream, which, a);
  }
  const int64_t flat_n = (int64_t)a.batch * (a.N + 1) * (a.n + a.m);
  const dim3 flat((unsigned)std::min<int64_t>((flat_n + 255) / 256, 1 << 20));
  switch (which) {
    case IK_ROLLOUT:
      if constexpr (sizeof(T) == 8) {
        // ROLLOUT_INIT (the host asks for it only where it applies: capi_solve.hip): rollout + CopyTrajectory + the first expansion of
        // an unconstrained problem in one pass of the row layout (kernels/ilqr_row32.hip: row32_rollout_init_kernel)
        if (a.row32 && (a.mode & ROLLOUT_INIT)) return row32_merit_dispatch(stream, a, 2) == 0 ? 0 : 2;
      }
      hipLaunchKernelGGL(generic_rollout_kernel<T>, waves, b64, 0, stream, a);
      break;
    case IK_ACCEPT: hipLaunchKernelGGL(generic_accept_kernel<T>, flat, b256, 0, stream, a); break;
    case IK_EXPAND:
      if constexpr (sizeof(T) == 8) {
        // constraint blocks on plan MFMA32's shapes: the gradient and the DIAGONAL form of the Hessian (EXPAND_DIAG: the host has seen
        // that every block is bound-type and that the full blocks are stored) as a walk of the row layout (row32_expand_kernel)
        if (a.al.enabled && a.row32 && (!(a.mode & EXPAND_HESSIAN) || (a.mode & EXPAND_DIAG)) && !(a.mode & EXPAND_DYN)) {
          const int rc = row32_merit_dispatch(stream, a, 4);
          if (rc != 1) return rc;
        }
      }
      if (a.al.enabled) hipLaunchKernelGGL(generic_expand_al_kernel<T>, dim3((unsigned)((int64_t)a.batch * (a.N + 1))), b64, 0, stream, a);
      else hipLaunchKernelGGL(generic_expand_kernel<T>, flat, b256, 0, stream, a);
      if constexpr (sizeof(T) == 8) {
        if (a.mp.kind == MODEL_QUADROTOR13 && (a.mode & EXPAND_DYN) && hipGetLastError() == hipSuccess)
          return gen_launch_model<MODEL_QUADROTOR13, 13, 4>(stream, IK_EXPAND, a);
      }
      break;
    case IK_DUAL: hipLaunchKernelGGL(generic_dual_update_kernel<T>, dim3((unsigned)((int64_t)a.batch * (a.N + 1))), b64, 0, stream, a); break;
    case IK_MERIT: {   // the knot point's matrices staged in LDS while sixteen waves still fit a CU (10 KB each, the kernel's own 2.5 KB included)
      if constexpr (sizeof(T) == 8) {
        if (a.row32) {   // plan MFMA32's shapes: two problems per wave in the row layout, one kernel per shape (kernels/ilqr_row32.hip)
          const int rc = row32_merit_dispatch(stream, a);
          if (rc != 1) return rc;   // (1: no kernel for this shape -- the LDS form below)
        }
      }
      const size_t jv = a.al.enabled ? (size_t)GEN_AL_JV * sizeof(double) : 0;
      const size_t stage = generic_merit_stage_elems(a.n, a.m) * sizeof(T);
      if (2560 + jv + stage <= 10 * 1024) hipLaunchKernelGGL((generic_merit_kernel<T, true>), waves, b64, jv + stage, stream, a);
      else hipLaunchKernelGGL((generic_merit_kernel<T, false>), waves, b64, jv, stream, a);
      break;
    }
    case IK_MERIT2:   // phi(0) and the line search's first step in one pass: the row layout's shapes only (the host asks where it applies)
      if constexpr (sizeof(T) == 8) {
        if (a.row32) return row32_merit_dispatch(stream, a, 3) == 0 ? 0 : 2;
        if (a.row32m) return row32_model_launch(stream, a, 3) == 0 ? 0 : 2;
      }
      return 1;
    case IK_STATIONARITY:
      if constexpr (sizeof(T) == 8) {
        if (a.row32 || a.row32m) {   // (kernels/ilqr_row32.hip: row32_stationarity_kernel -- it reads A_k, B_k wherever they came from)
          const int rc = row32_merit_dispatch(stream, a, 1);
          if (rc != 1) return rc;
        }
      }
      hipLaunchKernelGGL(generic_stationarity_kernel<T>, waves, b64, 0, stream, a);
      break;
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
