// ilqr_lane.hip -- the iLQR inner loop around the TVLQR sweep for plan LANE (lane-per-problem SoA).
//
// Device-side counterparts of, in the reference:
//   SolverImpl::OpenLoopRollout      src/altro/solver/solver.cpp:116-131   -> ilqr_rollout_kernel
//   SolverImpl::CopyTrajectory       solver.cpp:148-157                    -> ilqr_accept_kernel
//   CalcDynamicsExpansion / CalcCostGradient / CalcCostHessian (per knot point, independent in k)
//                                    knotpoint_data.cpp:406-419, :650-708  -> ilqr_expand_kernel
//   SolverImpl::MeritFunction        solver.cpp:273-355                    -> ilqr_merit_kernel
//   SolverImpl::Stationarity         solver.cpp:207-222                    -> ilqr_stationarity_kernel
//   SolverImpl::ForwardPass + CubicLineSearch + the sweep loop of Solve (solver.cpp:237-271, :447-502)
//                                    -> ilqr_ls_begin_kernel / ilqr_ls_feed_kernel / ilqr_finish_iter_kernel
// A diagonal quadratic (LQR tracking) cost -- the only cost paths that work in the reference (SURVEY.md
// section 2.1) --, a compiled-in model (models.h), and linear conic constraint blocks handled by the
// augmented Lagrangian of al_lane.hip (duals / penalties per problem; solver.cpp:383-409, :470-489).
//
// Buffers (all [..][element][batch], unit stride across lanes):
//   in   [k][E_IN]   A B f Q R H q r : the backward pass's inputs; A,B,q(=lx),r(=lu) are rewritten by
//                    every derivative evaluation exactly like the reference rewrites A_,B_,lx_,lu_
//   out  [k][E_OUT]  K d P p from the backward pass ; outn = P_N p_N
//   nom  [k][n+m]    nominal trajectory  x | u
//   cand [k][2n+m]   candidate           x_ | y_ | u_
//   cost [k][2n+2m+1] Qd | Rd | q | r | c     (k = N: terminal)   -- the diagonal cost (CK = 0: KnotPointData::SetDiagonalCost)
//        [k][n n + m m + m n + n + m + 1]  Q | R | H | q | r | c (column-major blocks, as given)
//                                                               -- the dense quadratic cost (CK = 1: KnotPointData::SetQuadraticCost,
//                                                                  knotpoint_data.cpp:64-85; value / gradient / Hessian :616-708)
#pragma once
#include "../rtc_compat.h"

#include "../linesearch_sm.h"
#include "../models.h"
#include "ilqr_types.h"
#include "al_lane.hip"
#include "tvlqr_lane.hip"

#include "../fp_contract.h"
ALTRO_FP_REGION_ON   // single-expression a * b + c only: the same rounding in every kernel these functions are inlined into (see models.h)
namespace altro_hip {

// CK: the cost the handle carries (IlqrArgs::cost_kind) -- 0 the diagonal tracking cost of ALTROSolver::SetLQRCost, 1 the dense
// quadratic cost of ALTROSolver::SetQuadraticCost (altro_solver.cpp:118-136).  A template parameter of the kernels that read
// the cost record (expansion, merit evaluation), not a run-time branch: the diagonal instantiations are the ones the fused
// solve kernel is built from, and they stay exactly as they were.
template <int n, int m, int CK = 0>
struct IlqrDims {
  static constexpr int E_NOM = n + m, E_CAND = 2 * n + m;
  static constexpr int E_COST = CK ? n * n + m * m + m * n + n + m + 1 : 2 * n + 2 * m + 1;
  static constexpr int C_Q = 0, C_R = CK ? n * n : n, C_H = n * n + m * m /* CK = 1 only */;
  static constexpr int C_q = CK ? n * n + m * m + m * n : n + m, C_r = C_q + n, C_c = C_r + m;
};

// ---- the cost of one knot point from a register copy of its record ------------------------------------------------------------
// CalcOriginalCost (knotpoint_data.cpp:616-648)
template <int n, int m, int CK, typename T>
__device__ __forceinline__ T ilqr_cost_value(const T* cs, const T* x, const T* u, bool terminal) {
  using I = IlqrDims<n, m, CK>;
  T J;
  if constexpr (CK == 0) {   // diagonal (:636-645)
    T a1 = T(0);
    for (int i = 0; i < n; ++i) a1 += x[i] * (cs[I::C_Q + i] * x[i]);
    J = T(0.5) * a1;
    T b1 = T(0);
    for (int i = 0; i < n; ++i) b1 += cs[I::C_q + i] * x[i];
    J += b1;
    if (!terminal) {
      T a2 = T(0);
      for (int i = 0; i < m; ++i) a2 += u[i] * (cs[I::C_R + i] * u[i]);
      J += T(0.5) * a2;
      T b2 = T(0);
      for (int i = 0; i < m; ++i) b2 += cs[I::C_r + i] * u[i];
      J += b2;
    }
  } else {                   // dense (:624-634): 1/2 x'Qx + q'x + 1/2 u'Ru + r'u + u'Hx
    T a1 = T(0);
    for (int i = 0; i < n; ++i) {
      T t = T(0);
      for (int j = 0; j < n; ++j) t += cs[I::C_Q + i + j * n] * x[j];
      a1 += x[i] * t;
    }
    J = T(0.5) * a1;
    T b1 = T(0);
    for (int i = 0; i < n; ++i) b1 += cs[I::C_q + i] * x[i];
    J += b1;
    if (!terminal) {
      T a2 = T(0);
      for (int i = 0; i < m; ++i) {
        T t = T(0);
        for (int j = 0; j < m; ++j) t += cs[I::C_R + i + j * m] * u[j];
        a2 += u[i] * t;
      }
      J += T(0.5) * a2;
      T b2 = T(0);
      for (int i = 0; i < m; ++i) b2 += cs[I::C_r + i] * u[i];
      J += b2;
      T b3 = T(0);
      for (int i = 0; i < m; ++i) {
        T t = T(0);
        for (int j = 0; j < n; ++j) t += cs[I::C_H + i + j * m] * x[j];
        b3 += u[i] * t;
      }
      J += b3;
    }
  }
  J += cs[I::C_c];
  return J;
}
// CalcOriginalCostGradient (knotpoint_data.cpp:650-681); lu untouched at the terminal knot point
template <int n, int m, int CK, typename T>
__device__ __forceinline__ void ilqr_cost_gradient(const T* cs, const T* x, const T* u, bool terminal, T* lx, T* lu) {
  using I = IlqrDims<n, m, CK>;
  if constexpr (CK == 0) {
    for (int i = 0; i < n; ++i) lx[i] = cs[I::C_Q + i] * x[i] + cs[I::C_q + i];
    if (!terminal)
      for (int i = 0; i < m; ++i) lu[i] = cs[I::C_R + i] * u[i] + cs[I::C_r + i];
  } else {                   // lx = Q x; lx += q; lu = R u; lu += r; lu += H x; lx += H'u   (:659-668)
    for (int i = 0; i < n; ++i) {
      T t = T(0);
      for (int j = 0; j < n; ++j) t += cs[I::C_Q + i + j * n] * x[j];
      lx[i] = t + cs[I::C_q + i];
    }
    if (!terminal) {
      for (int i = 0; i < m; ++i) {
        T t = T(0);
        for (int j = 0; j < m; ++j) t += cs[I::C_R + i + j * m] * u[j];
        T l = t + cs[I::C_r + i];
        T hx = T(0);
        for (int j = 0; j < n; ++j) hx += cs[I::C_H + i + j * m] * x[j];
        lu[i] = l + hx;
      }
      for (int j = 0; j < n; ++j) {
        T t = T(0);
        for (int i = 0; i < m; ++i) t += cs[I::C_H + i + j * m] * u[i];
        lx[j] = lx[j] + t;
      }
    }
  }
}
// CalcOriginalCostHessian (knotpoint_data.cpp:683-708): lxx n x n, luu m x m, lux m x n, column-major
template <int n, int m, int CK, typename T>
__device__ __forceinline__ void ilqr_cost_hessian(const T* cs, bool terminal, T* Qm, T* Rm, T* Hm) {
  using I = IlqrDims<n, m, CK>;
  if constexpr (CK == 0) {
    for (int e = 0; e < n * n; ++e) Qm[e] = (e % n == e / n) ? cs[I::C_Q + e % n] : T(0);
    for (int e = 0; e < m * m; ++e) Rm[e] = (!terminal && e % m == e / m) ? cs[I::C_R + e % m] : T(0);
    for (int e = 0; e < m * n; ++e) Hm[e] = T(0);
  } else {
    for (int e = 0; e < n * n; ++e) Qm[e] = cs[I::C_Q + e];
    for (int e = 0; e < m * m; ++e) Rm[e] = terminal ? T(0) : cs[I::C_R + e];
    for (int e = 0; e < m * n; ++e) Hm[e] = terminal ? T(0) : cs[I::C_H + e];
  }
}

#define ILQR_PROLOGUE                                                    \
  using D = LaneDims<n, m>;                                              \
  using I = IlqrDims<n, m>;                                              \
  using Mdl = DiscreteModel<KIND, n, m, T>;                              \
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;              \
  if (b >= a.batch) return;                                              \
  const int64_t B = a.batch;                                             \
  const int N = a.N;                                                     \
  (void)N; (void)sizeof(D); (void)sizeof(I); (void)sizeof(Mdl)

// x_0 = x0 ; x_{k+1} = f(x_k, u_k) on the candidate trajectory
template <int KIND, int n, int m, typename T>
__device__ __forceinline__ void ilqr_rollout_lane(const IlqrArgs<T>& a, int64_t b) {
  using I = IlqrDims<n, m>;
  using Mdl = DiscreteModel<KIND, n, m, T>;
  const int64_t B = a.batch;
  const int N = a.N;
  T x[n], u[m], xn[n];
  for (int e = 0; e < n; ++e) x[e] = a.x0[(int64_t)e * B + b];
  for (int k = 0; k < N; ++k) {
    T* c = a.cand + (int64_t)k * I::E_CAND * B + b;
    for (int e = 0; e < n; ++e) c[(int64_t)e * B] = x[e];
    for (int e = 0; e < m; ++e) u[e] = c[(int64_t)(2 * n + e) * B];
    Mdl::dynamics(a.mp, x, u, xn);
    for (int e = 0; e < n; ++e) x[e] = xn[e];
  }
  T* c = a.cand + (int64_t)N * I::E_CAND * B + b;
  for (int e = 0; e < n; ++e) c[(int64_t)e * B] = x[e];
}
template <int KIND, int n, int m, typename T>
__global__ __launch_bounds__(64) void ilqr_rollout_kernel(IlqrArgs<T> a) {
  ILQR_PROLOGUE;
  if (a.active && !a.active[b]) return;
  ilqr_rollout_lane<KIND, n, m, T>(a, b);
}

// nominal <- candidate (x, u) of one knot point
template <int n, int m, typename T>
__device__ __forceinline__ void ilqr_accept_point(const IlqrArgs<T>& a, int64_t b, int k) {
  using I = IlqrDims<n, m>;
  const int64_t B = a.batch;
  const T* __restrict__ c = a.cand + (int64_t)k * I::E_CAND * B + b;
  T* __restrict__ o = a.nom + (int64_t)k * I::E_NOM * B + b;
  T x[n], u[m];
#pragma unroll
  for (int e = 0; e < n; ++e) x[e] = c[(int64_t)e * B];
#pragma unroll
  for (int e = 0; e < m; ++e) u[e] = k < a.N ? c[(int64_t)(2 * n + e) * B] : T(0);
#pragma unroll
  for (int e = 0; e < n; ++e) o[(int64_t)e * B] = x[e];
  if (k < a.N) {
#pragma unroll
    for (int e = 0; e < m; ++e) o[(int64_t)(n + e) * B] = u[e];
  }
}
// one thread per (problem, knot point)
template <int n, int m, typename T>
__global__ void ilqr_accept_kernel(IlqrArgs<T> a) {
  const int64_t B = a.batch;
  const int64_t total = B * (a.N + 1);
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = t % B;
    const int k = (int)(t / B);
    if (a.active && !a.active[b]) continue;
    ilqr_accept_point<n, m, T>(a, b, k);
  }
}

// Expansion at the candidate point of ONE knot point: A, B (f = 0), lxx/luu/lux, lx, lu -> the backward pass's input
// record.  Independent in k: the reference's own TODO ("do this in parallel", solver.cpp:190).
template <int KIND, int n, int m, typename T, int CK = 0>
__device__ __forceinline__ void ilqr_expand_point(const IlqrArgs<T>& a, int64_t b, int k, bool grad, bool hess) {
  using D = LaneDims<n, m>;
  using I = IlqrDims<n, m, CK>;
  using Mdl = DiscreteModel<KIND, n, m, T>;
  const int64_t B = a.batch;
  const T* c = a.cand + (int64_t)k * I::E_CAND * B + b;
  const T* cs = a.cost + (int64_t)k * I::E_COST * B + b;
  const bool terminal = k == a.N;
  T x[n], u[m];
  for (int e = 0; e < n; ++e) x[e] = c[(int64_t)e * B];
  for (int e = 0; e < m; ++e) u[e] = terminal ? T(0) : c[(int64_t)(2 * n + e) * B];
  T lx[n], lu[m], Qm[n * n], Rm[m * m], Hm[m * n];
  if constexpr (CK == 0) {
    for (int e = 0; e < n; ++e) lx[e] = cs[(int64_t)(I::C_Q + e) * B] * x[e] + cs[(int64_t)(I::C_q + e) * B];
    for (int e = 0; e < m; ++e)
      lu[e] = terminal ? T(0) : cs[(int64_t)(I::C_R + e) * B] * u[e] + cs[(int64_t)(I::C_r + e) * B];
    for (int e = 0; e < n * n; ++e) Qm[e] = (e % n == e / n) ? cs[(int64_t)(I::C_Q + e % n) * B] : T(0);
    for (int e = 0; e < m * m; ++e)
      Rm[e] = (!terminal && e % m == e / m) ? cs[(int64_t)(I::C_R + e % m) * B] : T(0);
    for (int e = 0; e < m * n; ++e) Hm[e] = T(0);
  } else {   // dense quadratic cost: the record first, then the reference's expressions (knotpoint_data.cpp:650-708)
    T cr[I::E_COST];
    for (int e = 0; e < I::E_COST; ++e) cr[e] = cs[(int64_t)e * B];
    for (int e = 0; e < m; ++e) lu[e] = T(0);
    ilqr_cost_gradient<n, m, CK, T>(cr, x, u, terminal, lx, lu);
    ilqr_cost_hessian<n, m, CK, T>(cr, terminal, Qm, Rm, Hm);
  }
  if (a.al.enabled) {
    const T rho_est = (T)a.prob[b].rho_est, rho = (T)a.prob[b].rho;
    if (grad && hess)
      al_eval<n, m, T, true, true>(a.al, k, b, B, x, u, terminal, rho_est, rho, lx, lu, Qm, Rm, Hm, nullptr, false);
    else if (grad)
      al_eval<n, m, T, true, false>(a.al, k, b, B, x, u, terminal, rho_est, rho, lx, lu, Qm, Rm, Hm, nullptr, false);
    else
      al_eval<n, m, T, false, true>(a.al, k, b, B, x, u, terminal, rho_est, rho, lx, lu, Qm, Rm, Hm, nullptr, false);
  }
  if (terminal) {   // P_N = lxx, p_N = lx
    if (hess)
      for (int e = 0; e < n * n; ++e) a.term[(int64_t)e * B + b] = Qm[e];
    if (grad)
      for (int e = 0; e < n; ++e) a.term[(int64_t)(n * n + e) * B + b] = lx[e];
    return;
  }
  T* in = a.in + (int64_t)k * D::E_IN * B + b;
  if (grad) {
    T Am[n * n], Bm[n * m];
    Mdl::jacobian(a.mp, x, u, Am, Bm);
    for (int e = 0; e < n * n; ++e) in[(int64_t)(D::O_A + e) * B] = Am[e];
    for (int e = 0; e < n * m; ++e) in[(int64_t)(D::O_B + e) * B] = Bm[e];
    for (int e = 0; e < n; ++e) in[(int64_t)(D::O_f + e) * B] = T(0);
    for (int e = 0; e < n; ++e) in[(int64_t)(D::O_q + e) * B] = lx[e];
    for (int e = 0; e < m; ++e) in[(int64_t)(D::O_r + e) * B] = lu[e];
  }
  if (hess) {
    for (int e = 0; e < n * n; ++e) in[(int64_t)(D::O_Q + e) * B] = Qm[e];
    for (int e = 0; e < m * m; ++e) in[(int64_t)(D::O_R + e) * B] = Rm[e];
    for (int e = 0; e < m * n; ++e) in[(int64_t)(D::O_H + e) * B] = Hm[e];
  }
}
// one thread per (problem, knot point)
template <int KIND, int n, int m, typename T, int CK = 0>
__global__ __launch_bounds__(64) void ilqr_expand_kernel(IlqrArgs<T> a) {
  const int64_t B = a.batch;
  const int64_t total = B * (a.N + 1);
  const bool grad = (a.mode & EXPAND_GRADIENT) != 0, hess = (a.mode & EXPAND_HESSIAN) != 0;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = t % B;
    const int k = (int)(t / B);
    if (a.active && !a.active[b]) continue;
    ilqr_expand_point<KIND, n, m, T, CK>(a, b, k, grad, hess);
  }
}

// One knot point's merit-function operands in registers: K | d | P | p, the nominal x | u, the cost record and
// the duals.  None of these addresses depends on the rollout, so record k + 1 is requested before knot point
// k is evaluated (these kernels are latency-bound: one wave per 64 problems, N dependent steps).
template <int n, int m, typename T, int CK = 0>
struct MeritRec {
  using D = LaneDims<n, m>;
  using I = IlqrDims<n, m, CK>;
  T out[D::E_OUT];
  T nom[I::E_NOM];
  T cs[I::E_COST];
  T z[AL_MAXC * AL_MAXP];
};

template <int n, int m, typename T, int CK>
__device__ __forceinline__ void merit_load(MeritRec<n, m, T, CK>& r, const IlqrArgs<T>& a, int k, int64_t b0, uint32_t lane,
                                           uint32_t rowB) {
  using D = LaneDims<n, m>;
  using I = IlqrDims<n, m, CK>;
  const int64_t B = a.batch;
  if (k < a.N) {
    const LaneBuf bo(a.out + b0 + (int64_t)k * D::E_OUT * B);
#pragma unroll
    for (int e = 0; e < D::E_OUT; ++e) r.out[e] = lane_ld<T>(bo, lane, (uint32_t)e * rowB);
  } else {   // terminal: P_N | p_N
    const LaneBuf bo(a.outn + b0);
#pragma unroll
    for (int e = 0; e < n * n + n; ++e) r.out[D::O_P + e] = lane_ld<T>(bo, lane, (uint32_t)e * rowB);
  }
  const LaneBuf bn(a.nom + b0 + (int64_t)k * I::E_NOM * B), bc(a.cost + b0 + (int64_t)k * I::E_COST * B);
#pragma unroll
  for (int e = 0; e < I::E_NOM; ++e) r.nom[e] = lane_ld<T>(bn, lane, (uint32_t)e * rowB);
#pragma unroll
  for (int e = 0; e < I::E_COST; ++e) r.cs[e] = lane_ld<T>(bc, lane, (uint32_t)e * rowB);
  if (a.al.enabled) al_load_z<T>(a.al, k, LaneBuf(a.al.z + b0), lane, rowB, r.z);
}

// one non-terminal knot point of MeritFunction (solver.cpp:286-317)
template <int KIND, int n, int m, typename T, int CK>
__device__ __forceinline__ void merit_step(const MeritRec<n, m, T, CK>& r, const IlqrArgs<T>& a, T* cand, int k, int64_t b, int64_t b0,
                                           uint32_t lane, uint32_t rowB, T alpha, T rho, bool deriv, bool store, bool al, T (&x)[n],
                                           T (&dxda)[n], T& phi, T& dphi) {
  using D = LaneDims<n, m>;
  using I = IlqrDims<n, m, CK>;
  using Mdl = DiscreteModel<KIND, n, m, T>;
  const int64_t B = a.batch;
  T dx[n], u[m], xn[n], y[n];
  for (int i = 0; i < n; ++i) dx[i] = x[i] - r.nom[i];
  for (int i = 0; i < m; ++i) {   // u_ = u + (-K dx + alpha d)
    T s = T(0);
    for (int j = 0; j < n; ++j) s += r.out[D::O_K + i + j * m] * dx[j];
    u[i] = r.nom[n + i] + (-s + alpha * r.out[D::O_d + i]);
  }
  for (int i = 0; i < n; ++i) {   // y_ = P dx + p
    T s = T(0);
    for (int j = 0; j < n; ++j) s += r.out[D::O_P + i + j * n] * dx[j];
    y[i] = s + r.out[D::O_p + i];
  }
  const LaneBuf bc(cand + b0 + (int64_t)k * I::E_CAND * B);
#pragma unroll
  for (int e = 0; e < n; ++e) lane_st<T>(bc, lane, (uint32_t)e * rowB, x[e]);
#pragma unroll
  for (int e = 0; e < n; ++e) lane_st<T>(bc, lane, (uint32_t)(n + e) * rowB, y[e]);
#pragma unroll
  for (int e = 0; e < m; ++e) lane_st<T>(bc, lane, (uint32_t)(2 * n + e) * rowB, u[e]);
  T Am[n * n], Bm[n * m];
  if (deriv) Mdl::step(a.mp, x, u, xn, Am, Bm);
  else Mdl::dynamics(a.mp, x, u, xn);
  T lx[n], lu[m];
  ilqr_cost_gradient<n, m, CK, T>(r.cs, x, u, false, lx, lu);
  T Jk = ilqr_cost_value<n, m, CK, T>(r.cs, x, u, false);
  if (al)   // one instance: the gradient terms are formed even when only phi is wanted (lx, lu are then unused)
    Jk += al_eval<n, m, T, true, false>(a.al, k, b, B, x, u, false, rho, rho, lx, lu, nullptr, nullptr, nullptr, nullptr, false, r.z);
  phi += Jk;
  if (deriv) {
    T duda[m], dxn[n];
    const LaneBuf bi(a.in + b0 + (int64_t)k * D::E_IN * B);
#pragma unroll
    for (int e = 0; e < n * n; ++e) if (store) lane_st<T>(bi, lane, (uint32_t)(D::O_A + e) * rowB, Am[e]);
#pragma unroll
    for (int e = 0; e < n * m; ++e) if (store) lane_st<T>(bi, lane, (uint32_t)(D::O_B + e) * rowB, Bm[e]);
    for (int i = 0; i < m; ++i) {   // du_da = -K dx_da + d
      T s = T(0);
      for (int j = 0; j < n; ++j) s += r.out[D::O_K + i + j * m] * dxda[j];
      duda[i] = -s + r.out[D::O_d + i];
    }
    for (int i = 0; i < n; ++i) {   // dx_da+ = A dx_da + B du_da
      T s = T(0);
      for (int j = 0; j < n; ++j) s += Am[i + j * n] * dxda[j];
      T s2 = T(0);
      for (int j = 0; j < m; ++j) s2 += Bm[i + j * n] * duda[j];
      dxn[i] = s + s2;
    }
#pragma unroll
    for (int e = 0; e < n; ++e) if (store) lane_st<T>(bi, lane, (uint32_t)(D::O_q + e) * rowB, lx[e]);
#pragma unroll
    for (int e = 0; e < m; ++e) if (store) lane_st<T>(bi, lane, (uint32_t)(D::O_r + e) * rowB, lu[e]);
    T s = T(0);
    for (int i = 0; i < n; ++i) s += lx[i] * dxda[i];
    dphi += s;
    s = T(0);
    for (int i = 0; i < m; ++i) s += lu[i] * duda[i];
    dphi += s;
    for (int e = 0; e < n; ++e) dxda[e] = dxn[e];
  }
  for (int e = 0; e < n; ++e) x[e] = xn[e];
}

// MeritFunction (solver.cpp:273-355) for one problem: closed-loop rollout with step alpha, total cost phi and, when
// asked, the directional derivative dphi together with the refreshed A, B, lx, lu (`store`).  Leaves rho_est = rho.
template <int KIND, int n, int m, typename T, int CK = 0>
__device__ __forceinline__ void ilqr_merit_lane(const IlqrArgs<T>& a, int64_t b, int64_t b0, uint32_t lane, uint32_t rowB,
                                                T alpha, bool deriv, bool store, T* cand, T& phi_out, T& dphi_out) {
  using D = LaneDims<n, m>;
  using I = IlqrDims<n, m, CK>;
  const int64_t B = a.batch;
  const int N = a.N;
  const bool al = a.al.enabled != 0;
  const T rho = al ? (T)a.prob[b].rho : T(1);   // CalcCost refreshes the projected duals with the current penalty
  T x[n], dxda[n], phi = T(0), dphi = T(0);
  {
    const LaneBuf bx(a.x0 + b0);
#pragma unroll
    for (int e = 0; e < n; ++e) { x[e] = lane_ld<T>(bx, lane, (uint32_t)e * rowB); dxda[e] = T(0); }
  }
  MeritRec<n, m, T, CK> r0, r1;
  merit_load<n, m, T, CK>(r0, a, 0, b0, lane, rowB);
  for (int k = 0; k < N; ++k) {   // one step instance (code size); the copy waits for record k + 1 after step k
    merit_load<n, m, T, CK>(r1, a, k + 1, b0, lane, rowB);      // k + 1 == N: the terminal record
    merit_step<KIND, n, m, T, CK>(r0, a, cand, k, b, b0, lane, rowB, alpha, rho, deriv, store, al, x, dxda, phi, dphi);
    r0 = r1;
  }
  {   // terminal knot point (solver.cpp:319-332); r0 holds record N
    T lxN[n];
    ilqr_cost_gradient<n, m, CK, T>(r0.cs, x, (const T*)nullptr, true, lxN, (T*)nullptr);
    T Jk = ilqr_cost_value<n, m, CK, T>(r0.cs, x, (const T*)nullptr, true);
    if (al)
      Jk += al_eval<n, m, T, true, false>(a.al, N, b, B, x, (const T*)nullptr, true, rho, rho, lxN, nullptr, nullptr, nullptr, nullptr, nullptr, false, r0.z);
    phi += Jk;
    T dx[n];
    for (int i = 0; i < n; ++i) dx[i] = x[i] - r0.nom[i];
    const LaneBuf bc(cand + b0 + (int64_t)N * I::E_CAND * B), bt(a.term + b0);
    for (int i = 0; i < n; ++i) {
      T s = T(0);
      for (int j = 0; j < n; ++j) s += r0.out[D::O_P + i + j * n] * dx[j];
      lane_st<T>(bc, lane, (uint32_t)(n + i) * rowB, s + r0.out[D::O_p + i]);
    }
#pragma unroll
    for (int e = 0; e < n; ++e) lane_st<T>(bc, lane, (uint32_t)e * rowB, x[e]);
    if (deriv) {
      T s = T(0);
      for (int i = 0; i < n; ++i) {
        if (store) lane_st<T>(bt, lane, (uint32_t)(n * n + i) * rowB, lxN[i]);
        s += lxN[i] * dxda[i];
      }
      dphi += s;
    }
  }
  phi_out = phi;
  dphi_out = dphi;
  if (al) a.prob[b].rho_est = (double)rho;
}

// Which step trial `trial` of problem b evaluates, where its trajectory goes and what it leaves behind: trial 0 is the step
// the search asked for (alpha[b]); trial > 0 a speculative one (IlqrArgs::spec_trials / spec_pre) into its own phi row and
// spare candidate trajectory.
template <typename T>
struct MeritTrial {
  // the same for every problem of a launch row (kept free of per-lane control flow, so that the buffer descriptors built on
  // `cand` stay in scalar registers)
  T* cand;
  bool deriv;   // phi' wanted (backtracking trials never ask for it)
  bool store;   // a derivative pass also leaves A, B, lx, lu behind -- except the fused first trial
  // per problem
  double alpha;
  bool run;     // false: nothing to evaluate for this (problem, trial)
};
template <typename T>
__device__ __forceinline__ MeritTrial<T> ilqr_merit_trial(const IlqrArgs<T>& a, int64_t b, int trial) {
  MeritTrial<T> tr;
  tr.deriv = a.want_derivative != 0 && (trial == 0 || a.spec_pre);
  const bool aside = a.spec_pre && (a.spec_flip ? trial == 0 : trial > 0);   // spare candidate 0, no expansion stores
  tr.store = !aside;
  tr.cand = aside ? a.cand_spec : (trial > 0 && !a.spec_pre) ? a.cand_spec + (int64_t)(trial - 1) * a.spec_stride : a.cand;
  tr.alpha = a.alpha ? a.alpha[b] : a.alpha_const;
  tr.run = false;
  if (trial > 0 && a.spec_pre) {  // next to phi(0): the step alpha0 = 1 the search will ask for first, phi and phi'
    if (trial > 1) return tr;
    tr.alpha = 1.0;
  } else if (trial > 0) {
    const LsState& ls = a.prob[b].ls;
    if (ls.stage == LS_STAGE_BACKTRACK) {          // pending: alpha beta^0 with bt_iter = t; trial j is bt_iter = t + j
      if (ls.bt_iter + trial >= a.ls_max_iters) return tr;
    } else if (ls.stage == LS_STAGE_CUBIC) {       // pending: the cubic guess; if it is rejected the backtracking
      if (trial >= a.ls_max_iters) return tr;      // sequence starts at alpha0 beta with bt_iter = 1 (linesearch.cpp:138)
      tr.alpha = ls.alpha0;
    } else {
      return tr;
    }
    for (int t = 0; t < trial; ++t) tr.alpha = tr.alpha * a.ls_beta;   // the state machine's own sequence of products
  }
  tr.run = true;
  return tr;
}
// One merit evaluation of problem b in one lane: rollout, costs, phi' and the expansion, knot point after knot point.
template <int KIND, int n, int m, typename T, int CK = 0>
__device__ __forceinline__ void ilqr_merit_body(const IlqrArgs<T>& a, int64_t b, int64_t b0, uint32_t lane, uint32_t rowB, int trial) {
  const int64_t B = a.batch;
  const MeritTrial<T> tr = ilqr_merit_trial<T>(a, b, trial);
  if (!tr.run) return;
  T phi, dphi;
  ilqr_merit_lane<KIND, n, m, T, CK>(a, b, b0, lane, rowB, (T)tr.alpha, tr.deriv, tr.store, tr.cand, phi, dphi);
  a.phi[(int64_t)trial * B + b] = (double)phi;
  if (tr.deriv) a.dphi[(int64_t)trial * B + b] = (double)dphi;
}
// one launch = one merit evaluation per problem; gridDim.y > 1: the speculative trials ride along
template <int KIND, int n, int m, typename T, int CK = 0>
__global__ __launch_bounds__(64) void ilqr_merit_kernel(IlqrArgs<T> a) {
  ILQR_PROLOGUE;
  if (a.active && !a.active[b]) return;
  ilqr_merit_body<KIND, n, m, T, CK>(a, b, (int64_t)blockIdx.x * 64, threadIdx.x * (uint32_t)sizeof(T),
                                 (uint32_t)B * (uint32_t)sizeof(T), (int)blockIdx.y);
}

// ---- MeritFunction in three launches ---------------------------------------------------------------------------------
// Of everything ilqr_merit_lane does per knot point only two things are serial in k: the closed-loop rollout
// x+ = f(x, u + alpha d - K dx) and the sensitivity dx/dalpha behind phi'.  Costs, constraint terms, dynamics Jacobians
// and cost gradients depend on k through (x_k, u_k) alone.  So:
//   1. ilqr_merit_roll_kernel   lane per (problem, trial): the rollout, nothing else -- the latency chain of the launch
//   2. ilqr_merit_point_kernel  thread per (problem, knot point, trial), the whole chip: y, J_k (+ constraint terms),
//                               A, B, lx, lu (into the backward pass's input record, or -- a speculative first step --
//                               into the spare block a.spec_jac), J_k into a.merit_jk
//   3. ilqr_merit_sum_kernel    lane per (problem, trial): phi = sum J_k in k order, the dx/dalpha recursion and phi'
// Same expressions, same summation order as the one-launch ilqr_merit_kernel (ALTRO_HIP_MERIT_SPLIT=0 keeps it): bit-identical
// (tests/test_gpu_merit_split.py).  The fused solve kernel (ilqr_fused.hip) runs the same three phases inside a workgroup.
template <int n, int m, typename T>
struct MeritJac {   // where a derivative pass leaves A | B | lx | lu of knot point k and lx of the terminal one
  T* p; int E, oA, oB, oq, orr;
  T* term; int oterm;
};
template <int n, int m, typename T>
__device__ __forceinline__ MeritJac<n, m, T> ilqr_merit_jac(const IlqrArgs<T>& a, bool store) {
  using D = LaneDims<n, m>;
  MeritJac<n, m, T> j;
  if (store) {
    j.p = a.in; j.E = D::E_IN; j.oA = D::O_A; j.oB = D::O_B; j.oq = D::O_q; j.orr = D::O_r; j.term = a.term; j.oterm = n * n;
  } else {
    j.p = a.spec_jac; j.E = n * n + n * m + n + m; j.oA = 0; j.oB = n * n; j.oq = n * n + n * m; j.orr = n * n + n * m + n;
    j.term = a.spec_jac + (int64_t)a.N * j.E * a.batch; j.oterm = 0;
  }
  return j;
}

// depth of the prefetch rings of the serial kernels: about 48 loads in flight per wave, at most 4 records
#define MERIT_RING(LOADS) ((LOADS) <= 12 ? 4 : (LOADS) <= 16 ? 3 : 2)
template <int n, int m, typename T>
struct RollRec {
  T K[n * m], d[m], nom[n + m];
};
template <int n, int m, typename T>
__device__ __forceinline__ void roll_load(RollRec<n, m, T>& r, const IlqrArgs<T>& a, int k, int64_t b0, uint32_t lane, uint32_t rowB) {
  using D = LaneDims<n, m>;
  using I = IlqrDims<n, m>;
  const int64_t B = a.batch;
  const LaneBuf bo(a.out + b0 + (int64_t)k * D::E_OUT * B), bn(a.nom + b0 + (int64_t)k * I::E_NOM * B);
#pragma unroll
  for (int e = 0; e < n * m; ++e) r.K[e] = lane_ld<T>(bo, lane, (uint32_t)(D::O_K + e) * rowB);
#pragma unroll
  for (int e = 0; e < m; ++e) r.d[e] = lane_ld<T>(bo, lane, (uint32_t)(D::O_d + e) * rowB);
#pragma unroll
  for (int e = 0; e < n + m; ++e) r.nom[e] = lane_ld<T>(bn, lane, (uint32_t)e * rowB);
}
// The closed-loop rollout of one (problem, trial) into the trial's candidate trajectory (x_, u_).  `pub(c)`, called every
// other group of the loop and once at the end (c = N + 1), tells a consumer that knot points 0 .. c - 1 are stored (the
// fused solve kernel evaluates their per-knot-point terms in other waves while the rollout goes on).
struct RollNoPublish { __device__ __forceinline__ void operator()(int) const {} };
template <int KIND, int n, int m, typename T, class Pub = RollNoPublish>
__device__ __forceinline__ void ilqr_merit_roll_lane(const IlqrArgs<T>& a, const MeritTrial<T>& tr, int64_t b0, uint32_t lane,
                                                     uint32_t rowB, const Pub& pub = Pub()) {
  using I = IlqrDims<n, m>;
  using Mdl = DiscreteModel<KIND, n, m, T>;
  const int64_t B = a.batch;
  const int N = a.N;
  const T alpha = (T)tr.alpha;
  T x[n], u[m], xn[n], dx[n];
  {
    const LaneBuf bx(a.x0 + b0);
#pragma unroll
    for (int e = 0; e < n; ++e) x[e] = lane_ld<T>(bx, lane, (uint32_t)e * rowB);
  }
  // None of the records' addresses depends on the rollout: they are requested RD knot points ahead (a ring of registers,
  // statically indexed through the unrolled inner loop).  With one wave per SIMD the loop is otherwise paced by the
  // latency of one record's loads (~0.5 us each knot point), not by the arithmetic.  The refills are UNCONDITIONAL (index
  // clamped to the last record): a load under a branch lands in a temporary, and the copy into the ring waits for it.
  constexpr int RD = MERIT_RING(n * m + 2 * m + n);
  RollRec<n, m, T> ring[RD];
  const int last = N > 0 ? N - 1 : 0;
#pragma unroll
  for (int d = 0; d < RD; ++d) roll_load<n, m, T>(ring[d], a, d < last ? d : last, b0, lane, rowB);
  auto step = [&](RollRec<n, m, T>& r0, int k, bool refill) {
    for (int i = 0; i < n; ++i) dx[i] = x[i] - r0.nom[i];
    for (int i = 0; i < m; ++i) {   // u_ = u + (-K dx + alpha d)
      T s = T(0);
      for (int j = 0; j < n; ++j) s += r0.K[i + j * m] * dx[j];
      u[i] = r0.nom[n + i] + (-s + alpha * r0.d[i]);
    }
    if (refill) roll_load<n, m, T>(r0, a, k + RD < last ? k + RD : last, b0, lane, rowB);
    const LaneBuf bc(tr.cand + b0 + (int64_t)k * I::E_CAND * B);
#pragma unroll
    for (int e = 0; e < n; ++e) lane_st<T>(bc, lane, (uint32_t)e * rowB, x[e]);
#pragma unroll
    for (int e = 0; e < m; ++e) lane_st<T>(bc, lane, (uint32_t)(2 * n + e) * rowB, u[e]);
    Mdl::dynamics(a.mp, x, u, xn);
    for (int e = 0; e < n; ++e) x[e] = xn[e];
  };
  int k = 0;
  for (; k + RD <= N; k += RD) {   // whole groups: nothing conditional between the loads
#pragma unroll
    for (int d = 0; d < RD; ++d) step(ring[d], k + d, true);
    if ((k / RD) & 1) pub(k + RD);
  }
#pragma unroll
  for (int d = 0; d < RD; ++d)     // the last N mod RD knot points are in the ring already
    if (k + d < N) step(ring[d], k + d, false);
  const LaneBuf bc(tr.cand + b0 + (int64_t)N * I::E_CAND * B);
#pragma unroll
  for (int e = 0; e < n; ++e) lane_st<T>(bc, lane, (uint32_t)e * rowB, x[e]);
  pub(N + 1);
}
template <int KIND, int n, int m, typename T>
__global__ __launch_bounds__(64) void ilqr_merit_roll_kernel(IlqrArgs<T> a) {
  ILQR_PROLOGUE;
  if (a.active && !a.active[b]) return;
  const MeritTrial<T> tr = ilqr_merit_trial<T>(a, b, (int)blockIdx.y);
  if (!tr.run) return;
  ilqr_merit_roll_lane<KIND, n, m, T>(a, tr, (int64_t)blockIdx.x * 64, threadIdx.x * (uint32_t)sizeof(T),
                                      (uint32_t)B * (uint32_t)sizeof(T));
}

// everything of MeritFunction at ONE knot point of a rolled-out trial (solver.cpp:286-332 without the two recursions)
template <int KIND, int n, int m, typename T, int CK = 0>
__device__ __forceinline__ void ilqr_merit_point(const IlqrArgs<T>& a, int64_t b, int k, int trial, const MeritTrial<T>& tr) {
  using D = LaneDims<n, m>;
  using I = IlqrDims<n, m, CK>;
  using Mdl = DiscreteModel<KIND, n, m, T>;
  const int64_t B = a.batch;
  const int N = a.N;
  const bool terminal = k == N;
  const bool al = a.al.enabled != 0;
  T* c = tr.cand + (int64_t)k * I::E_CAND * B + b;
  const T* cs_ = a.cost + (int64_t)k * I::E_COST * B + b;
  const T* nm = a.nom + (int64_t)k * I::E_NOM * B + b;
  const T* Pp = terminal ? a.outn + b : a.out + (int64_t)k * D::E_OUT * B + (int64_t)D::O_P * B + b;   // P | p
  T x[n], u[m], dx[n], cs[I::E_COST], P[n * n + n];
  for (int e = 0; e < n; ++e) x[e] = c[(int64_t)e * B];
  for (int e = 0; e < m; ++e) u[e] = terminal ? T(0) : c[(int64_t)(2 * n + e) * B];
  for (int e = 0; e < n; ++e) dx[e] = nm[(int64_t)e * B];
  for (int e = 0; e < I::E_COST; ++e) cs[e] = cs_[(int64_t)e * B];
  for (int e = 0; e < n * n + n; ++e) P[e] = Pp[(int64_t)e * B];
  const T rho = al ? (T)a.prob[b].rho : T(1);
  for (int e = 0; e < n; ++e) dx[e] = x[e] - dx[e];
  for (int i = 0; i < n; ++i) {   // y_ = P dx + p
    T s = T(0);
    for (int j = 0; j < n; ++j) s += P[i + j * n] * dx[j];
    c[(int64_t)(n + i) * B] = s + P[n * n + i];
  }
  static_assert(D::O_p == D::O_P + n * n, "P | p are adjacent in the backward pass's output record");
  const MeritJac<n, m, T> jd = ilqr_merit_jac<n, m, T>(a, tr.store);
  T lx[n], lu[m];
  ilqr_cost_gradient<n, m, CK, T>(cs, x, u, terminal, lx, lu);
  T Jk;
  if (!terminal) {
    Jk = ilqr_cost_value<n, m, CK, T>(cs, x, u, false);
    if (al) Jk += al_eval<n, m, T, true, false>(a.al, k, b, B, x, u, false, rho, rho, lx, lu, nullptr, nullptr, nullptr, nullptr, false);
    if (tr.deriv) {
      T Am[n * n], Bm[n * m];
      Mdl::jacobian(a.mp, x, u, Am, Bm);
      T* j = jd.p + (int64_t)k * jd.E * B + b;
      for (int e = 0; e < n * n; ++e) j[(int64_t)(jd.oA + e) * B] = Am[e];
      for (int e = 0; e < n * m; ++e) j[(int64_t)(jd.oB + e) * B] = Bm[e];
      for (int e = 0; e < n; ++e) j[(int64_t)(jd.oq + e) * B] = lx[e];
      for (int e = 0; e < m; ++e) j[(int64_t)(jd.orr + e) * B] = lu[e];
    }
  } else {
    Jk = ilqr_cost_value<n, m, CK, T>(cs, x, (const T*)nullptr, true);
    if (al) Jk += al_eval<n, m, T, true, false>(a.al, N, b, B, x, (const T*)nullptr, true, rho, rho, lx, nullptr, nullptr, nullptr, nullptr, nullptr, false);
    if (tr.deriv)
      for (int e = 0; e < n; ++e) jd.term[(int64_t)(jd.oterm + e) * B + b] = lx[e];
  }
  a.merit_jk[((int64_t)trial * (N + 1) + k) * B + b] = Jk;
}
template <int KIND, int n, int m, typename T, int CK = 0>
__global__ __launch_bounds__(64) void ilqr_merit_point_kernel(IlqrArgs<T> a) {
  const int64_t B = a.batch;
  const int64_t total = B * (a.N + 1);
  const int trial = (int)blockIdx.y;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = t % B;
    const int k = (int)(t / B);
    if (a.active && !a.active[b]) continue;
    const MeritTrial<T> tr = ilqr_merit_trial<T>(a, b, trial);
    if (!tr.run) continue;
    ilqr_merit_point<KIND, n, m, T, CK>(a, b, k, trial, tr);
  }
}

template <int n, int m, typename T>
struct SumRec {
  T K[n * m], d[m], A[n * n], Bm[n * m], lx[n], lu[m], J;
};
template <int n, int m, bool DERIV, typename T>
__device__ __forceinline__ void sum_load(SumRec<n, m, T>& r, const IlqrArgs<T>& a, const MeritJac<n, m, T>& jd, const T* jk, int k,
                                         int64_t b0, uint32_t lane, uint32_t rowB) {
  using D = LaneDims<n, m>;
  const int64_t B = a.batch;
  r.J = lane_ld<T>(LaneBuf(jk + b0 + (int64_t)k * B), lane, 0u);
  if constexpr (DERIV) {
    const LaneBuf bo(a.out + b0 + (int64_t)k * D::E_OUT * B), bj(jd.p + b0 + (int64_t)k * jd.E * B);
#pragma unroll
    for (int e = 0; e < n * m; ++e) r.K[e] = lane_ld<T>(bo, lane, (uint32_t)(D::O_K + e) * rowB);
#pragma unroll
    for (int e = 0; e < m; ++e) r.d[e] = lane_ld<T>(bo, lane, (uint32_t)(D::O_d + e) * rowB);
#pragma unroll
    for (int e = 0; e < n * n; ++e) r.A[e] = lane_ld<T>(bj, lane, (uint32_t)(jd.oA + e) * rowB);
#pragma unroll
    for (int e = 0; e < n * m; ++e) r.Bm[e] = lane_ld<T>(bj, lane, (uint32_t)(jd.oB + e) * rowB);
#pragma unroll
    for (int e = 0; e < n; ++e) r.lx[e] = lane_ld<T>(bj, lane, (uint32_t)(jd.oq + e) * rowB);
#pragma unroll
    for (int e = 0; e < m; ++e) r.lu[e] = lane_ld<T>(bj, lane, (uint32_t)(jd.orr + e) * rowB);
  }
}
// phi (and phi') of one (problem, trial) from the per-knot-point terms: the sums in k order, the dx/dalpha recursion
template <int n, int m, bool DERIV, typename T>
__device__ __forceinline__ void ilqr_merit_sum_lane(const IlqrArgs<T>& a, const MeritJac<n, m, T>& jd, const T* jk, int64_t b0,
                                                    uint32_t lane, uint32_t rowB, T& phi_out, T& dphi_out) {
  const int64_t B = a.batch;
  const int N = a.N;
  T dxda[n], phi = T(0), dphi = T(0);
  for (int e = 0; e < n; ++e) dxda[e] = T(0);
  // terminal record first (it is needed last, its loads are in flight for the whole loop); ring of RD records, refilled
  // unconditionally (see ilqr_merit_roll_kernel)
  T JN = lane_ld<T>(LaneBuf(jk + b0 + (int64_t)N * B), lane, 0u), lxN[n];
  if constexpr (DERIV) {
    const LaneBuf bt(jd.term + b0);
#pragma unroll
    for (int e = 0; e < n; ++e) lxN[e] = lane_ld<T>(bt, lane, (uint32_t)(jd.oterm + e) * rowB);
  }
  constexpr int RD = DERIV ? MERIT_RING(2 * n * m + 2 * m + n * n + n + 1) : 4;
  SumRec<n, m, T> ring[RD];
  const int last = N > 0 ? N - 1 : 0;
#pragma unroll
  for (int d = 0; d < RD; ++d) sum_load<n, m, DERIV, T>(ring[d], a, jd, jk, d < last ? d : last, b0, lane, rowB);
  auto step = [&](SumRec<n, m, T>& r0, int k, bool refill) {
    phi += r0.J;
    if constexpr (DERIV) {
      T duda[m], dxn[n];
      for (int i = 0; i < m; ++i) {   // du_da = -K dx_da + d
        T s = T(0);
        for (int j = 0; j < n; ++j) s += r0.K[i + j * m] * dxda[j];
        duda[i] = -s + r0.d[i];
      }
      for (int i = 0; i < n; ++i) {   // dx_da+ = A dx_da + B du_da
        T s = T(0);
        for (int j = 0; j < n; ++j) s += r0.A[i + j * n] * dxda[j];
        T s2 = T(0);
        for (int j = 0; j < m; ++j) s2 += r0.Bm[i + j * n] * duda[j];
        dxn[i] = s + s2;
      }
      T s = T(0);
      for (int i = 0; i < n; ++i) s += r0.lx[i] * dxda[i];
      dphi += s;
      s = T(0);
      for (int i = 0; i < m; ++i) s += r0.lu[i] * duda[i];
      dphi += s;
      for (int e = 0; e < n; ++e) dxda[e] = dxn[e];
    }
    if (refill) sum_load<n, m, DERIV, T>(r0, a, jd, jk, k + RD < last ? k + RD : last, b0, lane, rowB);
  };
  int k = 0;
  for (; k + RD <= N; k += RD) {
#pragma unroll
    for (int d = 0; d < RD; ++d) step(ring[d], k + d, true);
  }
#pragma unroll
  for (int d = 0; d < RD; ++d)
    if (k + d < N) step(ring[d], k + d, false);
  phi += JN;
  if constexpr (DERIV) {
    T s = T(0);
    for (int i = 0; i < n; ++i) s += lxN[i] * dxda[i];
    dphi += s;
  }
  phi_out = phi;
  dphi_out = dphi;
}
// phi / phi' of one (problem, trial) into its row of a.phi / a.dphi
template <int n, int m, typename T>
__device__ __forceinline__ void ilqr_merit_sum_body(const IlqrArgs<T>& a, const MeritTrial<T>& tr, int64_t b, int64_t b0, uint32_t lane,
                                                    uint32_t rowB, int trial) {
  const int64_t B = a.batch;
  const MeritJac<n, m, T> jd = ilqr_merit_jac<n, m, T>(a, tr.store);
  const T* jk = a.merit_jk + (int64_t)trial * (a.N + 1) * B;
  T phi, dphi;
  if (tr.deriv) ilqr_merit_sum_lane<n, m, true, T>(a, jd, jk, b0, lane, rowB, phi, dphi);
  else ilqr_merit_sum_lane<n, m, false, T>(a, jd, jk, b0, lane, rowB, phi, dphi);
  a.phi[(int64_t)trial * B + b] = (double)phi;
  if (tr.deriv) a.dphi[(int64_t)trial * B + b] = (double)dphi;
  if (a.al.enabled) a.prob[b].rho_est = (double)(T)a.prob[b].rho;
}
template <int KIND, int n, int m, typename T>
__global__ __launch_bounds__(64) void ilqr_merit_sum_kernel(IlqrArgs<T> a) {
  ILQR_PROLOGUE;
  if (a.active && !a.active[b]) return;
  const int trial = (int)blockIdx.y;
  const MeritTrial<T> tr = ilqr_merit_trial<T>(a, b, trial);
  if (!tr.run) return;
  ilqr_merit_sum_body<n, m, T>(a, tr, b, (int64_t)blockIdx.x * 64, threadIdx.x * (uint32_t)sizeof(T),
                               (uint32_t)B * (uint32_t)sizeof(T), trial);
}

// Speculative backtracking: a problem that just ended its search on spare trajectory spec_sel[b] - 1 gets it copied
// over its candidate trajectory (rows x | y | u); this is one knot point of that copy.
template <int n, int m, typename T>
__device__ __forceinline__ void ilqr_spec_select_point(const IlqrArgs<T>& a, int64_t b, int k) {
  using I = IlqrDims<n, m>;
  const int s = a.spec_sel[b];
  if (s <= 0) return;
  const int64_t B = a.batch;
  const int64_t row0 = (int64_t)k * I::E_CAND * B + b;
  const T* __restrict__ src = a.cand_spec + (int64_t)(s - 1) * a.spec_stride + row0;
  T* __restrict__ dst = a.cand + row0;
  T v[I::E_CAND];
#pragma unroll
  for (int e = 0; e < I::E_CAND; ++e) v[e] = src[(int64_t)e * B];
#pragma unroll
  for (int e = 0; e < I::E_CAND; ++e) dst[(int64_t)e * B] = v[e];
}
template <int n, int m, typename T>
__global__ void ilqr_spec_select_kernel(IlqrArgs<T> a) {
  using I = IlqrDims<n, m>;
  const int64_t B = a.batch;
  const int64_t total = (int64_t)(a.N + 1) * I::E_CAND * B;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int s = a.spec_sel[t % B];
    if (s > 0) a.cand[t] = a.cand_spec[(int64_t)(s - 1) * a.spec_stride + t];
  }
}

// Stationarity (solver.cpp:207-222) and Feasibility (solver.cpp:224-231) of the candidate trajectory: maxima over
// knot points of independent per-knot-point residuals, so one thread per (problem, knot point) and an atomic max per
// problem (non-negative doubles order like their bit patterns) -- not an N-step loop per problem, which would be
// one more latency chain in every sweep.  ilqr_zero_residuals_kernel runs first.
template <typename T>
__global__ void ilqr_zero_residuals_kernel(IlqrArgs<T> a) {
  const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  if (a.active && !a.active[b]) return;
  a.prob[b].stationarity = 0.0;
  a.prob[b].feasibility = 0.0;
}
// residuals of ONE knot point: stationarity (solver.cpp:207-222) and, with constraints, the violation (solver.cpp:224-231)
template <int n, int m, typename T>
__device__ __forceinline__ void ilqr_stationarity_point(const IlqrArgs<T>& a, int64_t b, int k, T& res, T& viol) {
  using D = LaneDims<n, m>;
  using I = IlqrDims<n, m>;
  const int64_t B = a.batch;
  const int N = a.N;
  const T* __restrict__ c = a.cand + (int64_t)k * I::E_CAND * B + b;
  res = T(0);
  viol = T(0);
  if (k < N) {
    // every operand first, then the arithmetic: the loads of one knot point (and, in the fused solve kernel's loop over k,
    // of the next ones) are in flight together instead of one exposed round trip per element
    const T* __restrict__ in = a.in + (int64_t)k * D::E_IN * B + b;
    const T* __restrict__ cn = a.cand + (int64_t)(k + 1) * I::E_CAND * B + b;
    T yn[n], Am[n * n], Bm[n * m], lx[n], lu[m], y[n];
#pragma unroll
    for (int e = 0; e < n; ++e) yn[e] = cn[(int64_t)(n + e) * B];
#pragma unroll
    for (int e = 0; e < n * n; ++e) Am[e] = in[(int64_t)(D::O_A + e) * B];
#pragma unroll
    for (int e = 0; e < n * m; ++e) Bm[e] = in[(int64_t)(D::O_B + e) * B];
#pragma unroll
    for (int e = 0; e < n; ++e) lx[e] = in[(int64_t)(D::O_q + e) * B];
#pragma unroll
    for (int e = 0; e < m; ++e) lu[e] = in[(int64_t)(D::O_r + e) * B];
#pragma unroll
    for (int e = 0; e < n; ++e) y[e] = c[(int64_t)(n + e) * B];
#pragma unroll
    for (int j = 0; j < n; ++j) {
      T s = T(0);
#pragma unroll
      for (int i = 0; i < n; ++i) s += Am[i + j * n] * yn[i];
      res = fmax(res, fabs(lx[j] + s - y[j]));
    }
#pragma unroll
    for (int j = 0; j < m; ++j) {
      T s = T(0);
#pragma unroll
      for (int i = 0; i < n; ++i) s += Bm[i + j * n] * yn[i];
      res = fmax(res, fabs(lu[j] + s));
    }
  } else {
    for (int j = 0; j < n; ++j)
      res = fmax(res, fabs(a.term[(int64_t)(n * n + j) * B + b] - c[(int64_t)(n + j) * B]));
  }
  if (a.al.enabled) {
    const T rho = (T)a.prob[b].rho;
    T x[n], u[m];
    for (int e = 0; e < n; ++e) x[e] = c[(int64_t)e * B];
    for (int e = 0; e < m; ++e) u[e] = k < N ? c[(int64_t)(2 * n + e) * B] : T(0);
    al_eval<n, m, T, false, false>(a.al, k, b, B, x, u, k == N, rho, rho, nullptr, nullptr, nullptr, nullptr, nullptr,
                                   &viol, false);
  }
}
template <int n, int m, typename T>
__global__ __launch_bounds__(64) void ilqr_stationarity_kernel(IlqrArgs<T> a) {
  const int64_t B = a.batch;
  const int64_t total = B * (a.N + 1);
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = t % B;
    const int k = (int)(t / B);
    if (a.active && !a.active[b]) continue;
    T res, viol;
    ilqr_stationarity_point<n, m, T>(a, b, k, res, viol);
    atomicMax(reinterpret_cast<unsigned long long*>(&a.prob[b].stationarity),
              (unsigned long long)__double_as_longlong((double)res));
    if (a.al.enabled)
      atomicMax(reinterpret_cast<unsigned long long*>(&a.prob[b].feasibility),
                (unsigned long long)__double_as_longlong((double)viol));
  }
}

// DualUpdate (knotpoint_data.cpp:503-510) of ONE knot point: z <- the projected duals of the accepted trajectory
template <int n, int m, typename T>
__device__ __forceinline__ void ilqr_dual_point(const IlqrArgs<T>& a, int64_t b, int k) {
  using I = IlqrDims<n, m>;
  const int64_t B = a.batch;
  const T* ck = a.cand + (int64_t)k * I::E_CAND * B + b;
  T x[n], u[m];
  for (int e = 0; e < n; ++e) x[e] = ck[(int64_t)e * B];
  for (int e = 0; e < m; ++e) u[e] = k < a.N ? ck[(int64_t)(2 * n + e) * B] : T(0);
  const T rho = (T)a.prob[b].rho_est;
  al_eval<n, m, T, false, false>(a.al, k, b, B, x, u, k == a.N, rho, rho, nullptr, nullptr, nullptr, nullptr,
                                 nullptr, nullptr, true);
}
// one thread per (problem, knot point); only problems whose sweep asked for it (IlqrProb::dual != 0)
template <int n, int m, typename T>
__global__ void ilqr_dual_update_kernel(IlqrArgs<T> a) {
  const int64_t B = a.batch;
  const int64_t total = B * (a.N + 1);
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = t % B;
    const int k = (int)(t / B);
    if (!a.prob[b].dual) continue;
    ilqr_dual_point<n, m, T>(a, b, k);
  }
}

// ALTROSolver::ShiftTrajectory (altro_solver.cpp:283-293): x_[k] <- x_[k+1] for k < N, u_[k] <- u_[k+1] for
// k < N - 1, on the candidate trajectory.  One thread per (problem, element), sequential in k.
template <int n, int m, typename T>
__global__ void ilqr_shift_kernel(IlqrArgs<T> a) {
  using I = IlqrDims<n, m>;
  const int64_t B = a.batch;
  const int64_t total = B * (n + m);
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = t % B;
    const int e = (int)(t / B);
    const bool is_u = e >= n;
    const int el = is_u ? 2 * n + (e - n) : e;          // candidate record: x_ | y_ | u_
    const int kend = is_u ? a.N - 1 : a.N;
    T* c = a.cand + (int64_t)el * B + b;
    const int64_t ks = (int64_t)I::E_CAND * B;
    for (int k = 0; k < kend; ++k) c[(int64_t)k * ks] = c[(int64_t)(k + 1) * ks];
  }
}

}  // namespace altro_hip
ALTRO_FP_REGION_END   // back to the including translation unit's own mode (fp_contract.h)
