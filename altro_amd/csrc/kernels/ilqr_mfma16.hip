// ilqr_mfma16.hip -- the iLQR loop around the TVLQR sweep for plan MFMA16 ((n, m) = (12, 4), wave per problem),
// for problems whose dynamics are DATA: x+ = A_k x + B_k u + f_k (altro_solver.cpp:68-84 with the reference's
// SetLinearDynamics path, knotpoint_data.cpp:406-419, :710-719) and whose cost is the diagonal tracking cost of
// ALTROSolver::SetLQRCost.  Device-side counterparts of, in the reference:
//   SolverImpl::OpenLoopRollout   solver.cpp:116-131   -> wave_rollout_kernel
//   SolverImpl::CopyTrajectory    solver.cpp:148-157   -> wave_accept_kernel
//   CalcCostGradient              knotpoint_data.cpp:650-681  -> wave_gradient_kernel
//   SolverImpl::MeritFunction     solver.cpp:273-355   -> wave_merit_kernel
//   SolverImpl::Stationarity      solver.cpp:207-222   -> wave_stationarity_kernel
// The line search, convergence logic and sweep sequencing are the plan-independent kernels of
// ilqr_loop_kernels.hip driven by altro_hip_ilqr_solve.  Constraints (al_lane.hip) are a plan-LANE feature.
//
// The linear dynamics make the expansion trivial -- A, B are the data, and the reference zeroes the affine term
// of the EXPANSION (f_.setZero(), knotpoint_data.cpp:416) while the rollout keeps it -- so the backward sweep runs
// with has_f = 0 on the same DYN records whose f slot the rollout reads.  Only lx, lu (the [q r] slot of the COST
// record and the q_N slot of TERM) are rewritten by a derivative evaluation.
//
// Lane roles inside a wave (rows of row-major records, like the forward sweep):
//   lanes  0..11 : row i of Z = [A B]   -> x+[i], dx+/dalpha[i], state cost terms, lx[i]
//   lanes 16..19 : row a of Kt = [K|-d] -> u[a], du/dalpha[a], input cost terms, lu[a]
//   lanes 32..43 : row i of [P | p]     -> y[i]
// Vectors are exchanged through LDS (broadcast reads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ilqr_types.h"
#include "mfma16_layout.h"

namespace altro_hip {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// x_0 = x0 ; x_{k+1} = A x + B u + f on the candidate trajectory (u_ is the guess already stored there)
template <typename S>
__global__ __launch_bounds__(64) void wave_rollout_kernel(IlqrWaveArgs<S> a) {
  __shared__ double xs[12], us[4];
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= a.batch) return;
  if (a.active && !a.active[b]) return;
  const int i = lane < 12 ? lane : 11;
  double x = (double)a.x0[(size_t)b * 12 + i];
  for (int k = 0; k < a.N; ++k) {
    S* c = a.cand + (size_t)b * a.xuy_bs + (size_t)k * a.xuy_ks;
    const S* z = a.dyn + (size_t)b * a.dyn_bs + (size_t)k * a.dyn_ks;
    if (lane < 12) { xs[lane] = x; c[lane] = (S)x; }
    if (lane >= 16 && lane < 20) us[lane - 16] = (double)c[24 + lane - 16];
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 12; ++j) acc += (double)z[i * 16 + j] * xs[j];
    double acc2 = 0.0;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) acc2 += (double)z[i * 16 + 12 + cc] * us[cc];
    const double xn = (acc + acc2) + (double)z[MF_OFF_F + i];
    __syncthreads();
    x = xn;
  }
  if (lane < 12) a.cand[(size_t)b * a.xuy_bs + (size_t)a.N * a.xuy_ks + lane] = (S)x;
}

// nominal <- candidate (x, u)
template <typename S>
__global__ void wave_accept_kernel(IlqrWaveArgs<S> a) {
  const int64_t total = (int64_t)a.batch * (a.N + 1) * 16;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(t % 16);
    const int64_t r = t / 16;
    const int b = (int)(r % a.batch);
    const int k = (int)(r / a.batch);
    if (a.active && !a.active[b]) continue;
    const S* c = a.cand + (size_t)b * a.xuy_bs + (size_t)k * a.xuy_ks;
    S v = S(0);
    if (e < 12) v = c[e];
    else if (k < a.N) v = c[24 + e - 12];
    a.nom[((size_t)k * a.batch + b) * MF_NOM + e] = v;
  }
}

// lx, lu at the candidate point into the backward sweep's [q r] slot (and q_N of TERM)
template <typename S>
__global__ void wave_gradient_kernel(IlqrWaveArgs<S> a) {
  const int64_t total = (int64_t)a.batch * (a.N + 1) * 16;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(t % 16);
    const int64_t r = t / 16;
    const int b = (int)(r % a.batch);
    const int k = (int)(r / a.batch);
    if (a.active && !a.active[b]) continue;
    const S* c = a.cand + (size_t)b * a.xuy_bs + (size_t)k * a.xuy_ks;
    const S* cp = a.costp + ((size_t)k * a.batch + b) * MF_COSTP;
    if (e < 12) {
      const double lx = (double)cp[e] * (double)c[e] + (double)cp[16 + e];
      if (k < a.N) a.cin[(size_t)b * a.cin_bs + (size_t)k * a.cin_ks + MF_OFF_QR + e] = (S)lx;
      else a.term[(size_t)b * MF_TERM + 144 + e] = (S)lx;
    } else if (k < a.N) {
      const int cidx = e - 12;
      const double lu = (double)cp[12 + cidx] * (double)c[24 + cidx] + (double)cp[28 + cidx];
      a.cin[(size_t)b * a.cin_bs + (size_t)k * a.cin_ks + MF_OFF_QR + e] = (S)lu;
    }
  }
}

// MeritFunction (solver.cpp:273-355) for linear dynamics and the diagonal tracking cost
template <typename S>
__global__ __launch_bounds__(64) void wave_merit_kernel(IlqrWaveArgs<S> a) {
  __shared__ double xs[12], dxs[12], das[12], us[4], dus[4];
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= a.batch) return;
  if (a.active && !a.active[b]) return;
  const int N = a.N;
  const double alpha = a.alpha ? a.alpha[b] : a.alpha_const;
  const bool deriv = a.want_derivative != 0;
  const int grp = lane >> 4, sub = lane & 15;
  const bool is_x = lane < 12, is_u = (grp == 1 && sub < 4), is_y = (grp == 2 && sub < 12);
  const int i = sub < 12 ? sub : 11;       // row of Z / [P|p]
  const int ia = sub < 4 ? sub : 3;        // row of Kt
  double x = (double)a.x0[(size_t)b * 12 + i];
  double dxda = 0.0;
  double J = 0.0, dJ = 0.0;                // per-lane partial sums of phi and dphi
  for (int k = 0; k < N; ++k) {
    const S* z = a.dyn + (size_t)b * a.dyn_bs + (size_t)k * a.dyn_ks;
    const S* o = a.out + (size_t)b * a.out_bs + (size_t)k * a.out_ks;
    const S* nm = a.nom + ((size_t)k * a.batch + b) * MF_NOM;
    const S* cp = a.costp + ((size_t)k * a.batch + b) * MF_COSTP;
    S* c = a.cand + (size_t)b * a.xuy_bs + (size_t)k * a.xuy_ks;
    S* ci = a.cin + (size_t)b * a.cin_bs + (size_t)k * a.cin_ks;
    if (is_x) { xs[lane] = x; dxs[lane] = x - (double)nm[lane]; das[lane] = dxda; c[lane] = (S)x; }
    __syncthreads();
    if (is_u) {   // u_ = u + (-K dx + alpha d) ; du_da = -K dx_da + d        (Kt = [K | -d])
      double s = 0.0, s2 = 0.0;
#pragma unroll
      for (int j = 0; j < 12; ++j) { const double kij = (double)o[ia * 13 + j]; s += kij * dxs[j]; s2 += kij * das[j]; }
      const double d = -(double)o[ia * 13 + 12];
      const double u = (double)nm[12 + ia] + (-s + alpha * d);
      const double du = -s2 + d;
      us[ia] = u; dus[ia] = du;
      c[24 + ia] = (S)u;
      const double Rd = (double)cp[12 + ia], rr = (double)cp[28 + ia];
      J += 0.5 * (u * (Rd * u)) + rr * u;
      if (deriv) {
        const double lu = Rd * u + rr;
        ci[MF_OFF_QR + 12 + ia] = (S)lu;
        dJ += lu * du;
      }
    }
    if (is_y) {   // y_ = P dx + p
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 12; ++j) s += (double)o[MF_OFF_P + i * 13 + j] * dxs[j];
      c[12 + i] = (S)(s + (double)o[MF_OFF_P + i * 13 + 12]);
    }
    __syncthreads();
    double xn = 0.0, dxn = 0.0;
    if (is_x) {   // x+ = A x + B u + f ; dx+/da = A dx_da + B du_da ; state cost
      double s = 0.0, s2 = 0.0, t = 0.0, t2 = 0.0;
#pragma unroll
      for (int j = 0; j < 12; ++j) { const double aij = (double)z[i * 16 + j]; s += aij * xs[j]; t += aij * das[j]; }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) { const double bic = (double)z[i * 16 + 12 + cc]; s2 += bic * us[cc]; t2 += bic * dus[cc]; }
      xn = (s + s2) + (double)z[MF_OFF_F + i];
      dxn = t + t2;
      const double Qd = (double)cp[i], q = (double)cp[16 + i];
      J += 0.5 * (x * (Qd * x)) + q * x;
      if (lane == 0) J += (double)cp[32];
      if (deriv) {
        const double lx = Qd * x + q;
        ci[MF_OFF_QR + i] = (S)lx;
        dJ += lx * dxda;
      }
    }
    __syncthreads();
    if (is_x) { x = xn; dxda = dxn; }
  }
  {   // terminal knot point (solver.cpp:319-332)
    const S* nm = a.nom + ((size_t)N * a.batch + b) * MF_NOM;
    const S* cp = a.costp + ((size_t)N * a.batch + b) * MF_COSTP;
    const S* on = a.outn + (size_t)b * MF_TERM;
    S* c = a.cand + (size_t)b * a.xuy_bs + (size_t)N * a.xuy_ks;
    if (is_x) {
      dxs[lane] = x - (double)nm[lane];
      c[lane] = (S)x;
      const double Qd = (double)cp[i], q = (double)cp[16 + i];
      J += 0.5 * (x * (Qd * x)) + q * x;
      if (lane == 0) J += (double)cp[32];
      if (deriv) {
        const double lx = Qd * x + q;
        a.term[(size_t)b * MF_TERM + 144 + i] = (S)lx;
        dJ += lx * dxda;
      }
    }
    __syncthreads();
    if (is_y) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 12; ++j) s += (double)on[i * 13 + j] * dxs[j];
      c[12 + i] = (S)(s + (double)on[i * 13 + 12]);
    }
  }
  const double phi = wave_sum(J), dphi = wave_sum(dJ);
  if (lane == 0) {
    a.phi[b] = phi;
    if (deriv) a.dphi[b] = dphi;
  }
}

// Stationarity (solver.cpp:207-222): max_k |lx + A^T y+ - y|, max_k |lu + B^T y+|
template <typename S>
__global__ __launch_bounds__(64) void wave_stationarity_kernel(IlqrWaveArgs<S> a) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= a.batch) return;
  if (a.active && !a.active[b]) return;
  const int N = a.N;
  const int j = lane & 15;
  double res = 0.0;
  // every lane group of 16 takes every 4th knot point; lane j handles column j of Z = [A B]
  for (int k = lane >> 4; k < N; k += 4) {
    const S* z = a.dyn + (size_t)b * a.dyn_bs + (size_t)k * a.dyn_ks;
    const S* ci = a.cin + (size_t)b * a.cin_bs + (size_t)k * a.cin_ks;
    const S* c = a.cand + (size_t)b * a.xuy_bs + (size_t)k * a.xuy_ks;
    const S* cn = a.cand + (size_t)b * a.xuy_bs + (size_t)(k + 1) * a.xuy_ks;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += (double)z[i * 16 + j] * (double)cn[12 + i];
    const double g = (double)ci[MF_OFF_QR + j] + s;
    res = fmax(res, fabs(j < 12 ? g - (double)c[12 + j] : g));
  }
  if (lane < 12)
    res = fmax(res, fabs((double)a.term[(size_t)b * MF_TERM + 144 + lane] -
                         (double)a.cand[(size_t)b * a.xuy_bs + (size_t)N * a.xuy_ks + 12 + lane]));
  res = wave_max(res);
  if (lane == 0) { a.prob[b].stationarity = res; a.prob[b].feasibility = 0.0; }
}

// ALTROSolver::ShiftTrajectory (altro_solver.cpp:283-293) on the candidate records
template <typename S>
__global__ void wave_shift_kernel(IlqrWaveArgs<S> a) {
  const int64_t total = (int64_t)a.batch * 16;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(t % 16);
    const int b = (int)(t / 16);
    const int el = e < 12 ? e : 24 + (e - 12);
    const int kend = e < 12 ? a.N : a.N - 1;
    S* c = a.cand + (size_t)b * a.xuy_bs + el;
    for (int k = 0; k < kend; ++k) c[(size_t)k * a.xuy_ks] = c[(size_t)(k + 1) * a.xuy_ks];
  }
}

}  // namespace altro_hip
