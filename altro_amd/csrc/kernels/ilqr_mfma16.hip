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
// ilqr_loop_kernels.hip driven by altro_hip_ilqr_solve.  Linear constraint blocks c = G [x;u] - g
// (knotpoint_data.cpp:489-613) are evaluated by lanes 48..55, one constraint row each (a second-order-cone block
// as a whole by lane 48).
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
#include "../rtc_compat.h"   // (also compiled at run time: a caller's own model on the tile plan, capi_rtc.hip)

#include "ilqr_types.h"
#include "mfma16_layout.h"
#include "al_lane.hip"   // soc_projection / soc_jacobian / soc_hessian

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

// Constraint rows of knot point k for this wave's problem.  Lane 48 + i (i < 8) owns row i of each of the (at most
// two) blocks -- value, estimated / projected dual, AL cost share, violation -- and publishes to LDS, for the lanes
// that own gradient / Hessian entries,
//   jv[c * 8 + i]        = (J^T z_proj)_i                      gradient factor (cones.cpp:153-178)
//   Jm[c * 64 + i * 8 + r] = J_ir  (projection Jacobian, p x p)  Gauss-Newton factor        (Jm may be nullptr)
//   Hm[c * 16 + i * 4 + r] = (d/dz J^T z_proj)_ir               second-order-cone curvature  (Hm may be nullptr)
// For the zero / identity / orthant cones J is diagonal and the curvature vanishes.  A second-order-cone block
// (p <= 4, cones.cpp:13-123) is evaluated as a whole by lane 48.  xs / us: the point, in LDS.  With dual_update the
// projected dual becomes the dual (knotpoint_data.cpp:503-510).  Must be called by all lanes (no barrier inside).
// (ROW0: the lane that owns row 0 -- 48 everywhere but in the two-trial kernel, where each half of the wave evaluates its
//  own trial's rows in its lanes 16..23 and passes lane % 32 with ROW0 = 16.)
template <typename S, int ROW0 = 48>
__device__ __forceinline__ void wave_al_rows(const AlTable<S>& t, int k, int b, int64_t B, const double* xs, const double* us,
                                             bool terminal, double rho_est, int lane, double* jv, double* Jm, double* Hm,
                                             double& cost, double& viol, bool dual_update) {
  int zshift;
  const AlKnot ALTRO_CONST_AS& kn = al_knot<S>(t, k, zshift);
  const int i = lane - ROW0;
  const bool row_lane = (i >= 0 && i < AL_MAXP);
#pragma unroll
  for (int c = 0; c < AL_MAXC; ++c) {
    const bool has = c < kn.ncon;
    const int p = has ? kn.p[c] : 0, cone = has ? kn.cone[c] : CONE_IDENTITY;
    const S* G = t.G + (has ? kn.G_off[c] : 0);
    if (row_lane) {   // clear this lane's row of the published matrices
      jv[c * AL_MAXP + i] = 0.0;
      if (Jm)
        for (int r = 0; r < AL_MAXP; ++r) Jm[c * 64 + i * 8 + r] = 0.0;
      if (Hm && i < AL_MAXSOC)
        for (int r = 0; r < AL_MAXSOC; ++r) Hm[c * 16 + i * 4 + r] = 0.0;
    }
    if (!has) continue;
    if (cone != CONE_SOC) {
      if (row_lane && i < p) {
        double s = 0.0;
#pragma unroll
        for (int e = 0; e < 12; ++e) s += (double)G[i + e * p] * xs[e];
        if (!terminal) {
#pragma unroll
          for (int e = 0; e < 4; ++e) s += (double)G[i + (12 + e) * p] * us[e];
        }
        const double gi = kn.g_per_problem[c] ? (double)t.g[kn.g_off[c] + (int64_t)i * B + b] : (double)t.g[kn.g_off[c] + i];
        const double val = s - gi;
        S* zp_ = t.z + (int64_t)(kn.z_off[c] + zshift + i) * B + b;
        const double ze = (double)*zp_ - rho_est * val;
        double zp = 0.0, mkv = 0.0;
        if (cone == CONE_EQUALITY) { zp = ze; mkv = 1.0; viol = fmax(viol, fabs(val)); }
        else if (cone == CONE_INEQUALITY) { zp = fmin(0.0, ze); mkv = (ze <= 0.0) ? 1.0 : 0.0; viol = fmax(viol, fabs(fmin(0.0, val) - val)); }
        cost += zp * zp / (2.0 * rho_est);
        jv[c * AL_MAXP + i] = mkv * zp;
        if (Jm) Jm[c * 64 + i * 8 + i] = mkv;
        if (dual_update) *zp_ = (S)zp;
      }
    } else if (i == 0) {   // the whole second-order-cone block on one lane
      double val[AL_MAXSOC], ze[AL_MAXSOC], zp[AL_MAXSOC], pv[AL_MAXSOC];
#pragma unroll
      for (int r = 0; r < AL_MAXSOC; ++r) {
        val[r] = 0.0; ze[r] = 0.0;
        if (r < p) {
          double s = 0.0;
          for (int e = 0; e < 12; ++e) s += (double)G[r + e * p] * xs[e];
          if (!terminal)
            for (int e = 0; e < 4; ++e) s += (double)G[r + (12 + e) * p] * us[e];
          const double gi = kn.g_per_problem[c] ? (double)t.g[kn.g_off[c] + (int64_t)r * B + b] : (double)t.g[kn.g_off[c] + r];
          val[r] = s - gi;
          ze[r] = (double)t.z[(int64_t)(kn.z_off[c] + zshift + r) * B + b] - rho_est * val[r];
        }
      }
      soc_projection<double>(p, ze, zp);
      soc_projection<double>(p, val, pv);
      double sq = 0.0;
#pragma unroll
      for (int r = 0; r < AL_MAXSOC; ++r)
        if (r < p) { sq += zp[r] * zp[r]; viol = fmax(viol, fabs(pv[r] - val[r])); }
      cost += sq / (2.0 * rho_est);
      double J[AL_MAXSOC * AL_MAXSOC];
      soc_jacobian<double>(p, ze, J);
#pragma unroll
      for (int r = 0; r < AL_MAXSOC; ++r) {
        if (r >= p) continue;
        double sj = 0.0;
#pragma unroll
        for (int q = 0; q < AL_MAXSOC; ++q) sj += J[q + r * AL_MAXSOC] * zp[q];     // (J^T z_proj)_r
        jv[c * AL_MAXP + r] = sj;
        if (Jm)
          for (int q = 0; q < AL_MAXSOC; ++q) Jm[c * 64 + r * 8 + q] = J[r + q * AL_MAXSOC];
        if (dual_update) t.z[(int64_t)(kn.z_off[c] + zshift + r) * B + b] = (S)zp[r];
      }
      if (Hm) {
        double Hp[AL_MAXSOC * AL_MAXSOC];
        soc_hessian<double>(p, ze, zp, Hp);
#pragma unroll
        for (int r = 0; r < AL_MAXSOC; ++r)
#pragma unroll
          for (int q = 0; q < AL_MAXSOC; ++q) Hm[c * 16 + r * 4 + q] = Hp[r + q * AL_MAXSOC];
      }
    }
  }
}
// sum_c sum_i G_c[i][e] * w[c * 8 + i]  for column e of the constraint Jacobians of knot point k
template <typename S>
__device__ __forceinline__ double wave_al_col(const AlTable<S>& t, int k, int e, const double* w) {
  int zshift;
  const AlKnot ALTRO_CONST_AS& kn = al_knot<S>(t, k, zshift);
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < AL_MAXC; ++c)
    if (c < kn.ncon) {
      const int p = kn.p[c];
      const S* G = t.G + kn.G_off[c];
      for (int i = 0; i < p; ++i) s += (double)G[i + e * p] * w[c * AL_MAXP + i];
    }
  return s;
}

// x_0 = x0 ; x_{k+1} = A x + B u + f on the candidate trajectory (u_ is the guess already stored there).
// The DYN record is loaded coalesced, one knot point ahead, and staged in LDS with its rows padded to 17 (bank
// conflicts, see the forward sweep); lanes 0..11 then read their row.
template <typename S>
__global__ __launch_bounds__(64) void wave_rollout_kernel(IlqrWaveArgs<S> a) {
  constexpr int ZLD = 17;
  __shared__ double zimg[12 * ZLD + 12], xs[12], us[4];
  const int b = mf_problem(blockIdx.x, a.batch), lane = threadIdx.x;
  if (b >= a.batch) return;
  if (a.active && !a.active[b]) return;
  const int N = a.N;
  const int i = lane < 12 ? lane : 11;
  // ROLLOUT_INIT (the head of Solve for an unconstrained problem, solver.cpp:420-434): the same pass also makes the
  // trajectory the nominal one (CopyTrajectory, what wave_accept_kernel does) and forms the cost gradient lx = Qd x + q,
  // lu = Rd u + r at it (what wave_expand_grad_kernel does) -- the values and expressions of those two kernels, from the
  // x and u this wave already holds, instead of two more passes over the trajectory.
  const bool init = (a.mode & ROLLOUT_INIT) != 0;
  const int e16 = lane & 15;
  const S* __restrict__ dynb = a.dyn + (size_t)b * a.dyn_bs;
  S* __restrict__ candb = a.cand + (size_t)b * a.xuy_bs;
  double x = (double)a.x0[(size_t)b * 12 + i];
  double zr[3], fr, ur, cq = 0.0, cl = 0.0;
  auto load = [&](int k) {
    const S* z = dynb + (size_t)k * a.dyn_ks;
#pragma unroll
    for (int c = 0; c < 3; ++c) zr[c] = (double)z[MF_OFF_Z + c * 64 + lane];
    fr = (double)z[MF_OFF_F + i];
    ur = (double)candb[(size_t)k * a.xuy_ks + 24 + (lane & 3)];
    if (init) {
      const S* cp = a.costp + ((size_t)k * a.batch + b) * MF_COSTP;
      cq = (double)cp[e16]; cl = (double)cp[16 + e16];
    }
  };
  load(0);
  for (int k = 0; k < N; ++k) {
    __syncthreads();   // readers of the previous image are done
#pragma unroll
    for (int c = 0; c < 3; ++c) zimg[(4 * c + (lane >> 4)) * ZLD + (lane & 15)] = zr[c];
    zimg[12 * ZLD + i] = fr;
    if (lane < 4) us[lane] = ur;
    if (lane < 12) { xs[lane] = x; candb[(size_t)k * a.xuy_ks + lane] = (S)x; }
    const double cqk = cq, clk = cl;
    load(k + 1 < N ? k + 1 : N - 1);
    __syncthreads();
    if (init && lane < 16) {
      const double z = (double)(S)(lane < 12 ? x : us[lane - 12]);     // as the stored trajectory holds it
      a.nom[((size_t)k * a.batch + b) * MF_NOM + lane] = (S)z;
      a.cin[(size_t)b * a.cin_bs + (size_t)k * a.cin_ks + MF_OFF_QR + lane] = (S)(cqk * z + clk);
    }
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 12; ++j) acc += zimg[i * ZLD + j] * xs[j];
    double acc2 = 0.0;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) acc2 += zimg[i * ZLD + 12 + cc] * us[cc];
    x = (acc + acc2) + zimg[12 * ZLD + i];
  }
  if (lane < 12) a.cand[(size_t)b * a.xuy_bs + (size_t)a.N * a.xuy_ks + lane] = (S)x;
  if (init && lane < 16) {
    const S* cp = a.costp + ((size_t)N * a.batch + b) * MF_COSTP;
    const double z = (double)(S)x;
    a.nom[((size_t)N * a.batch + b) * MF_NOM + lane] = lane < 12 ? (S)z : S(0);
    if (lane < 12) a.term[(size_t)b * MF_TERM + 144 + lane] = (S)((double)cp[lane] * z + (double)cp[16 + lane]);
  }
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

// Speculative backtracking: copy spare candidate trajectory spec_sel[b] - 1 over the candidate of problem b.  A wave per
// problem: most rounds select a spare for few problems or none, and those waves leave after one load.
template <typename S>
__global__ __launch_bounds__(64) void wave_spec_select_kernel(IlqrWaveArgs<S> a) {
  const int b = blockIdx.x;
  if (b >= a.batch) return;
  const int sl = a.spec_sel[b];
  if (sl <= 0) return;
  const int total = (a.N + 1) * 28;
  const S* __restrict__ src = a.cand_spec + (size_t)(sl - 1) * a.spec_stride + (size_t)b * a.xuy_bs;
  S* __restrict__ dst = a.cand + (size_t)b * a.xuy_bs;
  for (int t = threadIdx.x; t < total; t += 64) {
    const int k = t / 28, e = t - 28 * k;
    const size_t off = (size_t)k * a.xuy_ks + e;
    dst[off] = src[off];
  }
}

// Cost gradient of an UNconstrained problem: lx = Qd x + q, lu = Rd u + r, one thread per (knot point, problem,
// entry) in the order the records lie in HBM.  (The wave-per-knot-point kernel below is for the AL terms; without
// them it would spend a million 64-lane workgroups on 16 multiply-adds each.)
template <typename S>
__global__ void wave_expand_grad_kernel(IlqrWaveArgs<S> a) {
  const int64_t total = (int64_t)(a.N + 1) * a.batch * 16;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(t & 15);
    const int b = (int)((t >> 4) % a.batch);
    const int k = (int)((t >> 4) / a.batch);
    if (a.active && !a.active[b]) continue;
    const bool terminal = k == a.N;
    if (terminal && e >= 12) continue;
    const S* c = a.cand + (size_t)b * a.xuy_bs + (size_t)k * a.xuy_ks;
    const S* cp = a.costp + ((size_t)k * a.batch + b) * MF_COSTP;
    const double z = (double)c[e < 12 ? e : 12 + e];   // x | (y) | u inside a candidate record
    const double l = (double)cp[e] * z + (double)cp[16 + e];
    if (terminal) a.term[(size_t)b * MF_TERM + 144 + e] = (S)l;
    else a.cin[(size_t)b * a.cin_bs + (size_t)k * a.cin_ks + MF_OFF_QR + e] = (S)l;
  }
}

// The same for the dense quadratic cost of ALTROSolver::SetQuadraticCost (knotpoint_data.cpp:659-668): lx = Q x + H^T u + q,
// lu = R u + H x + r, i.e. entry e of W [x; u] + [q r] with W = [Q H^T; H R] read from the dense cost record
// (IlqrWaveArgs::costd).  The sum runs over [x; u] in order as a chain of fused multiply-adds from zero, then + [q r]_e: the
// order and operations of the row-layout kernels' DPP chain (kernels/ilqr_merit2_dpp.hip), so the gradient a merit pass
// leaves behind and this kernel's are the same bits.
template <typename S>
__global__ void wave_expand_grad_dense_kernel(IlqrWaveArgs<S> a) {
  const int64_t total = (int64_t)(a.N + 1) * a.batch * 16;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(t & 15);
    const int b = (int)((t >> 4) % a.batch);
    const int k = (int)((t >> 4) / a.batch);
    if (a.active && !a.active[b]) continue;
    const bool terminal = k == a.N;
    if (terminal && e >= 12) continue;
    const S* c = a.cand + (size_t)b * a.xuy_bs + (size_t)k * a.xuy_ks;
    double acc = 0.0, lin;
    if (!terminal) {
      const S* cd = a.costd + ((size_t)k * a.batch + b) * MF_COST;
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) {
        const double wv = (double)cd[e < 12 ? (cc < 12 ? MF_OFF_Q + mf_sym(e, cc) : MF_OFF_HR + (cc - 12) * 16 + e) : MF_OFF_HR + (e - 12) * 16 + cc];
        acc = __builtin_fma((double)c[cc < 12 ? cc : 12 + cc], wv, acc);
      }
      lin = (double)cd[MF_OFF_QR + e];
    } else {
      const S* cT = a.costd_term + (size_t)b * MF_TERM;
#pragma unroll
      for (int cc = 0; cc < 12; ++cc) acc = __builtin_fma((double)c[cc], (double)cT[e * 12 + cc], acc);
      lin = (double)cT[144 + e];
    }
    const double l = acc + lin;
    if (terminal) a.term[(size_t)b * MF_TERM + 144 + e] = (S)l;
    else a.cin[(size_t)b * a.cin_bs + (size_t)k * a.cin_ks + MF_OFF_QR + e] = (S)l;
  }
}

// Expansion at the candidate point, one wave per (problem, knot point).
//   EXPAND_GRADIENT: lx, lu (+ AL terms) into the backward sweep's [q r] slot (q_N of TERM at k = N)
//   EXPAND_HESSIAN : [Q H^T; H R] = diag(Qd, Rd) + rho G^T M G  (M = the projection's diagonal Jacobian) into the
//                    Q rows / [H R] slots -- only launched when constraint blocks exist; without them the blocks
//                    set by altro_hip_set_tracking_cost stay as they are.
template <typename S>
__global__ __launch_bounds__(64) void wave_expand_kernel(IlqrWaveArgs<S> a) {
  __shared__ double xs[12], us[4], jv[AL_MAXC * AL_MAXP], Jm[AL_MAXC * 64], Hm[AL_MAXC * 16];
  const int lane = threadIdx.x;
  const int64_t wk = blockIdx.x;
  const int b = (int)(wk % a.batch), k = (int)(wk / a.batch);
  if (k > a.N) return;
  if (a.active && !a.active[b]) return;
  const bool terminal = k == a.N;
  const bool grad = (a.mode & EXPAND_GRADIENT) != 0, hess = (a.mode & EXPAND_HESSIAN) != 0;
  const S* c = a.cand + (size_t)b * a.xuy_bs + (size_t)k * a.xuy_ks;
  const S* cp = a.costp + ((size_t)k * a.batch + b) * MF_COSTP;
  if (lane < 12) xs[lane] = (double)c[lane];
  if (lane >= 12 && lane < 16) us[lane - 12] = terminal ? 0.0 : (double)c[24 + lane - 12];
  __syncthreads();
  const bool al = a.al.enabled != 0;
  if (al) {
    double cost = 0.0, viol = 0.0;
    wave_al_rows<S>(a.al, k, b, a.batch, xs, us, terminal, a.prob[b].rho_est, lane, jv, Jm, Hm, cost, viol, false);
  }
  __syncthreads();
  if (grad && lane < 16) {
    const int e = lane;
    if (e < 12) {
      double lx = (double)cp[e] * xs[e] + (double)cp[16 + e];
      if (al) lx -= wave_al_col<S>(a.al, k, e, jv);
      if (!terminal) a.cin[(size_t)b * a.cin_bs + (size_t)k * a.cin_ks + MF_OFF_QR + e] = (S)lx;
      else a.term[(size_t)b * MF_TERM + 144 + e] = (S)lx;
    } else if (!terminal) {
      double lu = (double)cp[e] * us[e - 12] + (double)cp[16 + e];
      if (al) lu -= wave_al_col<S>(a.al, k, e, jv);
      a.cin[(size_t)b * a.cin_bs + (size_t)k * a.cin_ks + MF_OFF_QR + e] = (S)lu;
    }
  }
  if (hess && al) {   // 16 x 16 tile, 4 entries per lane: row r = lane / 4 (+ 0), columns 4 (lane % 4) + 0..3 ... plain loop
    const double rho = a.prob[b].rho;
    int zshift;
    const AlKnot ALTRO_CONST_AS& kn = al_knot<S>(a.al, k, zshift);
    for (int t = lane; t < 256; t += 64) {
      const int r = t / 16, cc = t % 16;
      if (terminal && (r >= 12 || cc >= 12)) continue;
      if (r < 12 && cc >= 12) continue;     // the H^T block is not stored
      if (!terminal && r < 12 && cc < r) continue;   // nor is the lower triangle of Q (cost records hold triu(Q))
      double v = (r == cc) ? (double)cp[r] : 0.0;   // costp[0..15] = Qd | Rd
      double s = 0.0;
      for (int cidx = 0; cidx < kn.ncon; ++cidx) {
        const int p = kn.p[cidx];
        const S* G = a.al.G + kn.G_off[cidx];
        const double* Jc = Jm + cidx * 64;
        if (kn.cone[cidx] != CONE_SOC && kn.sel[cidx]) {
          // bound-type block (rows +-e_idx, al_types.h): G^T J^T J G is diagonal, entry (r, r) collects the rows that
          // select variable r.  Everything skipped is an exact zero in the sums below: same bits, O(p) instead of
          // O(p^2) per entry and no loads from G.
          if (r == cc)
            for (int i = 0; i < p; ++i) {
              const int code = kn.sidx[cidx][i];
              if ((code < 0 ? -code : code) - 1 != r) continue;
              const double jg = Jc[i * 8 + i] * (code < 0 ? -1.0 : 1.0);
              s += jg * jg;
            }
        } else if (kn.cone[cidx] != CONE_SOC) {   // diagonal projection Jacobian: (J G)_{i r} = J_ii G_ir -- the terms left out
          for (int i = 0; i < p; ++i) {    // of the full sums below are exact zeros, so the bits are the same
            const double jii = Jc[i * 8 + i];
            double jr = 0.0, jc = 0.0;
            jr += jii * (double)G[i + r * p]; jc += jii * (double)G[i + cc * p];
            s += jr * jc;
          }
        } else {
          for (int i = 0; i < p; ++i) {   // (J G)_{i r} (J G)_{i cc}
            double jr = 0.0, jc = 0.0;
            for (int q = 0; q < p; ++q) { jr += Jc[i * 8 + q] * (double)G[q + r * p]; jc += Jc[i * 8 + q] * (double)G[q + cc * p]; }
            s += jr * jc;
          }
        }
        if (kn.cone[cidx] == CONE_SOC) {   // + G^T (d/dz J^T z_proj) G   (knotpoint_data.cpp:561-567)
          const double* Hc = Hm + cidx * 16;
          for (int i = 0; i < p; ++i) {
            double hc = 0.0;
            for (int q = 0; q < p; ++q) hc += Hc[i * 4 + q] * (double)G[q + cc * p];
            s += (double)G[i + r * p] * hc;
          }
        }
      }
      v += rho * s;
      if (terminal) a.term[(size_t)b * MF_TERM + r * 12 + cc] = (S)v;
      else if (r < 12) a.cin[(size_t)b * a.cin_bs + (size_t)k * a.cin_ks + MF_OFF_Q + mf_sym(r, cc)] = (S)v;
      else a.cin[(size_t)b * a.cin_bs + (size_t)k * a.cin_ks + MF_OFF_HR + (r - 12) * 16 + cc] = (S)v;
    }
  }
}

// DualUpdate (knotpoint_data.cpp:503-510) for the problems whose sweep asked for it, one wave per (problem, knot point)
template <typename S>
__global__ __launch_bounds__(64) void wave_dual_update_kernel(IlqrWaveArgs<S> a) {
  __shared__ double xs[12], us[4], jv[AL_MAXC * AL_MAXP], Jm[AL_MAXC * 64], Hm[AL_MAXC * 16];
  const int lane = threadIdx.x;
  const int64_t wk = blockIdx.x;
  const int b = (int)(wk % a.batch), k = (int)(wk / a.batch);
  if (k > a.N || !a.al.enabled) return;
  if (!a.prob[b].dual) return;
  const bool terminal = k == a.N;
  const S* c = a.cand + (size_t)b * a.xuy_bs + (size_t)k * a.xuy_ks;
  if (lane < 12) xs[lane] = (double)c[lane];
  if (lane >= 12 && lane < 16) us[lane - 12] = terminal ? 0.0 : (double)c[24 + lane - 12];
  __syncthreads();
  double cost = 0.0, viol = 0.0;
  wave_al_rows<S>(a.al, k, b, a.batch, xs, us, terminal, a.prob[b].rho_est, lane, jv, nullptr, nullptr, cost, viol, true);
}

// One knot point's merit-function operands, loaded coalesced (lane -> consecutive element) and staged into an LDS
// image: Z rows padded to 17 (bank conflicts, see the forward sweep) | OUT record | f | nominal | cost parameters.
constexpr int MW_ZLD = 17;
constexpr int MW_OUT0 = 12 * MW_ZLD;       // 204
constexpr int MW_F0 = MW_OUT0 + MF_OUT;    // 348
constexpr int MW_NOM0 = MW_F0 + 12;        // 360
constexpr int MW_CP0 = MW_NOM0 + 16;       // 376
constexpr int MW_IMG = MW_CP0 + MF_COSTP;  // 412
struct MeritWaveRegs { double z[3], o[3], f, nm, cp; };
template <typename S>
__device__ __forceinline__ void merit_wave_load(MeritWaveRegs& r, const S* __restrict__ z, const S* __restrict__ o,
                                                const S* __restrict__ nm, const S* __restrict__ cp, int lane) {
#pragma unroll
  for (int c = 0; c < 3; ++c) r.z[c] = (double)z[MF_OFF_Z + c * 64 + lane];
#pragma unroll
  for (int c = 0; c < 2; ++c) r.o[c] = (double)o[c * 64 + lane];
  r.o[2] = (double)o[128 + (lane & 15)];
  r.f = (double)z[MF_OFF_F + (lane < 12 ? lane : 11)];
  r.nm = (double)nm[lane & 15];
  r.cp = (double)cp[lane < MF_COSTP ? lane : MF_COSTP - 1];
}
__device__ __forceinline__ void merit_wave_stage(const MeritWaveRegs& r, double* __restrict__ L, int lane) {
#pragma unroll
  for (int c = 0; c < 3; ++c) L[c * 4 * MW_ZLD + (lane >> 4) * MW_ZLD + (lane & 15)] = r.z[c];
#pragma unroll
  for (int c = 0; c < 2; ++c) L[MW_OUT0 + c * 64 + lane] = r.o[c];
  L[MW_OUT0 + 128 + (lane & 15)] = r.o[2];
  L[MW_F0 + (lane < 12 ? lane : 11)] = r.f;
  L[MW_NOM0 + (lane & 15)] = r.nm;
  L[MW_CP0 + (lane < MF_COSTP ? lane : MF_COSTP - 1)] = r.cp;
}

// MeritFunction (solver.cpp:273-355) for linear dynamics and the diagonal tracking cost (+ AL terms).
// Records are requested DEPTH knot points ahead (register ring, loop unrolled DEPTH times), every lane's loads and
// stores in the loop are unconditional (replica lanes re-read / re-write the same element): see the forward sweep.
// (AL: constraint blocks exist.  A separate instantiation, not a run-time branch: the unconstrained kernel then fits 128
//  registers -- four waves per SIMD, C1's 4096 waves all resident -- where the common one needed 177.)
template <typename S, bool AL>
__global__ __launch_bounds__(64) void wave_merit_kernel(IlqrWaveArgs<S> a) {
  constexpr int DEPTH = 2;
  __shared__ double img[MW_IMG + 4];
  __shared__ double vec[24], das[12], us[4], dus[4], jv[AL_MAXC * AL_MAXP];   // vec = x | dx
  double* const xs = vec;
  double* const dxs = vec + 12;
  __shared__ double crec[28], qrec[16];     // candidate record x | y | u and [lx lu], gathered for one coalesced store
  const int b = mf_problem(blockIdx.x, a.batch), lane = threadIdx.x;
  if (b >= a.batch) return;
  if (a.active && !a.active[b]) return;
  const int N = a.N;
  double alpha = a.alpha ? a.alpha[b] : a.alpha_const;
  S* __restrict__ candb = a.cand + (size_t)b * a.xuy_bs;
  const int trial = blockIdx.y;   // > 0: a speculative backtracking trial (IlqrArgs::spec_trials)
  bool store = true;              // see ilqr_merit_kernel
  if (trial > 0 && a.spec_pre) {
    if (trial > 1) return;
    alpha = 1.0;
    candb = a.cand_spec + (size_t)b * a.xuy_bs;
    store = false;
  } else if (trial > 0) {
    const LsState& ls = a.prob[b].ls;
    if (ls.stage == LS_STAGE_BACKTRACK) {
      if (ls.bt_iter + trial >= a.ls_max_iters) return;
    } else if (ls.stage == LS_STAGE_CUBIC) {       // see ilqr_merit_kernel
      if (trial >= a.ls_max_iters) return;
      alpha = ls.alpha0;
    } else {
      return;
    }
    for (int t = 0; t < trial; ++t) alpha = alpha * a.ls_beta;
    candb = a.cand_spec + (size_t)(trial - 1) * a.spec_stride + (size_t)b * a.xuy_bs;
  }
  const bool deriv = a.want_derivative != 0 && (trial == 0 || a.spec_pre);
  constexpr bool al = AL;
  const double rho = al ? a.prob[b].rho : 1.0;   // CalcCost refreshes the projected duals with the current penalty
  const int grp = lane >> 4, sub = lane & 15;
  const bool is_x = lane < 12, is_u = (grp == 1 && sub < 4), is_y = (grp == 2 && sub < 12);
  const int i = sub < 12 ? sub : 11;       // row of Z / [P|p]
  const int ia = sub < 4 ? sub : 3;        // row of Kt
  // image offsets of this lane's row: a row of Z (lanes 0..15), of Kt (16..31) or of [P | p] (32..63; P gathered from
  // its packed upper triangle), plus the affine column
  int ra[13];
#pragma unroll
  for (int j = 0; j < 12; ++j)
    ra[j] = (grp == 0) ? i * MW_ZLD + j : (grp == 1) ? MW_OUT0 + ia * 13 + j : MW_OUT0 + MF_OFF_P + mf_sym(i, j);
  ra[12] = (grp == 0) ? i * MW_ZLD + 12 : (grp == 1) ? MW_OUT0 + ia * 13 + 12 : MW_OUT0 + MF_OFF_p + i;
  const int vbase = (grp == 0) ? 0 : 12;   // rows of Z multiply x, the others dx: vec = xs | dxs
  const int l27 = lane < 28 ? lane : 27;
  const S* __restrict__ dynb = a.dyn + (size_t)b * a.dyn_bs;
  const S* __restrict__ outb = a.out + (size_t)b * a.out_bs;
  const S* __restrict__ nomb = a.nom + (size_t)b * MF_NOM;
  const S* __restrict__ cpb = a.costp + (size_t)b * MF_COSTP;
  const size_t nom_ks = (size_t)a.batch * MF_NOM, cp_ks = (size_t)a.batch * MF_COSTP;
  double x = (double)a.x0[(size_t)b * 12 + i];
  double dxda = 0.0;
  double J = 0.0, dJ = 0.0, viol = 0.0;    // per-lane partial sums of phi and dphi
  MeritWaveRegs ring[DEPTH];
#pragma unroll
  for (int dd = 0; dd < DEPTH; ++dd) {
    const size_t kk = dd < N ? dd : N - 1;
    merit_wave_load<S>(ring[dd], dynb + kk * a.dyn_ks, outb + kk * a.out_ks, nomb + kk * nom_ks, cpb + kk * cp_ks, lane);
  }
  const int Npad = ((N + DEPTH - 1) / DEPTH) * DEPTH;
  for (int k0 = 0; k0 < Npad; k0 += DEPTH) {
#pragma unroll
   for (int dd = 0; dd < DEPTH; ++dd) {
    const int k = k0 + dd;
    const bool live = k < N;     // padding steps (N odd) run on a clamped record and change nothing
    const int kc = live ? k : N - 1;
    __syncthreads();             // readers of the previous image are done
    merit_wave_stage(ring[dd], img, lane);
    // the three vectors every row product below needs: x, dx = x - x_nominal (its record entry is already in this
    // lane's registers) and dx/dalpha
    if (is_x) { xs[lane] = x; dxs[lane] = x - ring[dd].nm; das[lane] = dxda; crec[lane] = x; }
    {
      const size_t kn = (k + DEPTH < N) ? k + DEPTH : N - 1;
      merit_wave_load<S>(ring[dd], dynb + kn * a.dyn_ks, outb + kn * a.out_ks, nomb + kn * nom_ks, cpb + kn * cp_ks, lane);
    }
    __syncthreads();
    // Phase A, all 64 lanes at once: this lane's row (of Z, of Kt or of [P | p]) times x (rows of Z) or dx (the
    // others), and times dx/dalpha.  Same accumulation order as the per-role loops this replaces.
    double acc = 0.0, acc2 = 0.0;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const double rj = img[ra[j]];
      acc += rj * vec[vbase + j];
      acc2 += rj * das[j];
    }
    const double aff = img[ra[12]];          // -d (rows of Kt), p (rows of [P | p]), B[.][0] (rows of Z)
    double uval = 0.0, duval = 0.0;
    if (is_u) {   // u_ = u + (-K dx + alpha d) ; du_da = -K dx_da + d        (Kt = [K | -d])
      const double d = -aff;
      uval = img[MW_NOM0 + 12 + ia] + (-acc + alpha * d);
      duval = -acc2 + d;
      us[ia] = uval; dus[ia] = duval;
      crec[24 + ia] = uval;
      const double Rd = img[MW_CP0 + 12 + ia], rr = img[MW_CP0 + 28 + ia];
      if (live) J += 0.5 * (uval * (Rd * uval)) + rr * uval;
    }
    if (is_y) crec[12 + i] = acc + aff;      // y_ = P dx + p
    __syncthreads();
    double xn = 0.0, dxn = 0.0;
    if (is_x) {   // x+ = A x + B u + f ; dx+/da = A dx_da + B du_da ; state cost
      double s2 = 0.0, t2 = 0.0;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) { const double bic = img[i * MW_ZLD + 12 + cc]; s2 += bic * us[cc]; t2 += bic * dus[cc]; }
      xn = (acc + s2) + img[MW_F0 + i];
      dxn = acc2 + t2;
      const double Qd = img[MW_CP0 + i], q = img[MW_CP0 + 16 + i];
      if (live) {
        J += 0.5 * (x * (Qd * x)) + q * x;
        if (lane == 0) J += img[MW_CP0 + 32];
      }
    }
    if (al) {
      double Jal = 0.0;
      wave_al_rows<S>(a.al, kc, b, a.batch, xs, us, false, rho, lane, jv, nullptr, nullptr, Jal, viol, false);
      if (live) J += Jal;
    }
    __syncthreads();
    if (deriv && lane < 16) {   // lx (lanes 0..11) and lu (lanes 12..15) with the AL terms; dphi
      const int e = lane;
      const double pt = e < 12 ? x : us[e - 12];
      double l = img[MW_CP0 + e] * pt + img[MW_CP0 + 16 + e];       // costp: Qd | Rd | q | r line up with [x; u]
      if (al) l -= wave_al_col<S>(a.al, kc, e, jv);
      qrec[e] = l;
      if (live) dJ += l * (e < 12 ? dxda : dus[e - 12]);
    }
    __syncthreads();
    {   // one coalesced store of the candidate record (and of [lx lu]).  A padding step's record goes to the terminal
        // slot, which the terminal block below rewrites (x is already x_N there)
      S* c = candb + (size_t)(live ? k : N) * a.xuy_ks;
      c[l27] = (S)crec[l27];
      if (deriv) {
        S* ci = a.cin + (size_t)b * a.cin_bs + (size_t)kc * a.cin_ks;
        const double qv = qrec[sub];
        if (live && store) ci[MF_OFF_QR + sub] = (S)qv;
      }
    }
    if (is_x && live) { x = xn; dxda = dxn; }
   }
  }
  __syncthreads();
  {   // terminal knot point (solver.cpp:319-332)
    const S* nm = a.nom + ((size_t)N * a.batch + b) * MF_NOM;
    const S* cp = a.costp + ((size_t)N * a.batch + b) * MF_COSTP;
    const S* on = a.outn + (size_t)b * MF_TERM;
    S* c = candb + (size_t)N * a.xuy_ks;
    if (is_u) c[24 + ia] = S(0);
    if (is_x) {
      xs[lane] = x;
      dxs[lane] = x - (double)nm[lane];
      c[lane] = (S)x;
      const double Qd = (double)cp[i], q = (double)cp[16 + i];
      J += 0.5 * (x * (Qd * x)) + q * x;
      if (lane == 0) J += (double)cp[32];
    }
    __syncthreads();
    if (al) wave_al_rows<S>(a.al, N, b, a.batch, xs, us, true, rho, lane, jv, nullptr, nullptr, J, viol, false);
    if (is_y) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 12; ++j) s += (double)on[i * 13 + j] * dxs[j];
      c[12 + i] = (S)(s + (double)on[i * 13 + 12]);
    }
    __syncthreads();
    if (deriv && is_x) {
      double lx = (double)cp[i] * x + (double)cp[16 + i];
      if (al) lx -= wave_al_col<S>(a.al, N, i, jv);
      if (store) a.term[(size_t)b * MF_TERM + 144 + i] = (S)lx;
      dJ += lx * dxda;
    }
  }
  const double phi = wave_sum(J), dphi = wave_sum(dJ);
  if (lane == 0) {
    a.phi[(size_t)trial * a.batch + b] = phi;
    if (deriv) a.dphi[(size_t)trial * a.batch + b] = dphi;
    if (al) a.prob[b].rho_est = rho;
  }
}

// phi(0) AND the line search's first step alpha0 = 1 (linesearch.cpp: the search always starts there) in ONE pass over
// the records: the two evaluations of SolverImpl::MeritFunction (solver.cpp:273-355) that every sweep makes stream the
// same DYN / OUT / nominal / cost-parameter records, so "trial 0" (alpha = a.alpha[b], 0 in the solve loop) and "trial 1"
// (alpha = 1) ride the same loads and the same LDS image -- each lane forms its row product against both trials' vectors
// (interleaved in LDS: one 16-byte broadcast read per pair).  Expressions and accumulation order per trial are those of
// wave_merit_kernel.  Outputs: phi / dphi rows 0 and 1; the candidate trajectory x_, u_, y_ and the expansion lx, lu
// are TRIAL 1's (IlqrLoopArgs::spec_flip: the step that is nearly always accepted needs no copy and no second pass).
// For fp64 storage the kernel also leaves trial 1's stationarity (solver.cpp:207-222, with a one-knot-point lag:
// |lx_k + A_k^T y_{k+1} - y_k| needs the NEXT step's y, so the previous record image, gradient and y stay in LDS)
// and feasibility (solver.cpp:224-231) in the control block -- the values wave_stationarity_kernel would compute from
// the stored candidate, without its two further passes over DYN.
// sum over the 32 lanes of this lane's half of the wave: offsets 16 .. 1 of the butterfly wave_sum runs over 64 lanes (whose
// first step, offset 32, only ever adds the exact zeros of the other half: the order of the additions that matter is kept)
__device__ __forceinline__ double half_sum(double v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Lane roles: ONE TRIAL PER HALF WAVE.  Half h = lane / 32 evaluates trial h (0: alpha = a.alpha[b], 1: alpha = 1); within a
// half, r = lane % 32:
//   r  0..11 : row r of Z = [A B]        -> x+[r], dx+/dalpha[r], state cost terms, lx[r]
//   r 16..19 : row r - 16 of Kt = [K|-d] -> u, du/dalpha, input cost terms                     (lu: lanes r 12..15, like the
//   r 12..15, 20..27 : rows 0..3, 4..11 of [P | p] -> y                                          single-trial kernel's 12..15)
// i.e. the single-trial kernel's lanes 0..11 / 16..19 keep their places inside a half, so the per-trial sums are taken in
// its order.  Each lane reads ITS trial's vectors (8-byte broadcast reads): 36 LDS reads per lane and knot point for both
// trials together, where the first version of this kernel (every lane against both trials' interleaved vectors) needed 36
// at twice the width -- the kernel is bound by the LDS pipe, not by HBM.
template <typename S, bool AL>
__global__ __launch_bounds__(64) void wave_merit2_kernel(IlqrWaveArgs<S> a) {
  constexpr int DEPTH = 2;                          // also the image ping-pong: parity of k == dd
  constexpr bool kStat = sizeof(S) == 8;            // stored values == computed values only without a rounding store
  __shared__ double img[2][MW_IMG + 4];
  __shared__ double vec[2][24], das[2][12], us[2][4], dus[2][4];   // [trial]: x | dx, dx/dalpha, u, du/dalpha
  __shared__ double jv[2][AL_MAXC * AL_MAXP];
  __shared__ double crec[2][28], qrec[2][16], yN[12], lxN[12];
  const int b = mf_problem(blockIdx.x, a.batch), lane = threadIdx.x;
  if (b >= a.batch) return;
  if (a.active && !a.active[b]) return;
  const int N = a.N;
  const int h = lane >> 5, r = lane & 31;           // trial, role
  const double alpha = h ? 1.0 : (a.alpha ? a.alpha[b] : a.alpha_const);
  S* __restrict__ candb = a.cand + (size_t)b * a.xuy_bs;
  constexpr bool al = AL;
  const double rho = al ? a.prob[b].rho : 1.0;
  const bool is_x = r < 12, is_u = (r >= 16 && r < 20), is_y = (r >= 12 && r < 16) || (r >= 20 && r < 28);
  const bool cand = h == 1;                         // trial 1 writes the candidate trajectory and the expansion
  const int i = is_x ? r : (r < 16 ? r - 12 : (r < 20 ? 11 : (r < 28 ? r - 16 : 11)));   // row of Z / of [P | p]
  const int ia = is_u ? r - 16 : 3;                 // row of Kt
  int ra[13];
#pragma unroll
  for (int j = 0; j < 12; ++j)
    ra[j] = is_x ? i * MW_ZLD + j : is_u ? MW_OUT0 + ia * 13 + j : MW_OUT0 + MF_OFF_P + mf_sym(i, j);
  ra[12] = is_x ? i * MW_ZLD + 12 : is_u ? MW_OUT0 + ia * 13 + 12 : MW_OUT0 + MF_OFF_p + i;
  const double* const vrow = is_x ? &vec[h][0] : &vec[h][12];   // rows of Z multiply x, the others dx
  const int l27 = lane < 28 ? lane : 27;
  const int sub = lane & 15;
  const S* __restrict__ dynb = a.dyn + (size_t)b * a.dyn_bs;
  const S* __restrict__ outb = a.out + (size_t)b * a.out_bs;
  const S* __restrict__ nomb = a.nom + (size_t)b * MF_NOM;
  const S* __restrict__ cpb = a.costp + (size_t)b * MF_COSTP;
  const size_t nom_ks = (size_t)a.batch * MF_NOM, cp_ks = (size_t)a.batch * MF_COSTP;
  double x = (double)a.x0[(size_t)b * 12 + (is_x ? r : 11)];
  double dxda = 0.0;
  double J = 0.0, Jal = 0.0, dJ = 0.0, viol = 0.0, res = 0.0;   // (Jal: the AL rows' cost shares, summed apart like the
                                                                 //  single-trial kernel's lanes 48..55 do)
  MeritWaveRegs ring[DEPTH];
#pragma unroll
  for (int dd = 0; dd < DEPTH; ++dd) {
    const size_t kk = dd < N ? dd : N - 1;
    merit_wave_load<S>(ring[dd], dynb + kk * a.dyn_ks, outb + kk * a.out_ks, nomb + kk * nom_ks, cpb + kk * cp_ks, lane);
  }
  const int Npad = ((N + DEPTH - 1) / DEPTH) * DEPTH;
  for (int k0 = 0; k0 < Npad; k0 += DEPTH) {
#pragma unroll
   for (int dd = 0; dd < DEPTH; ++dd) {
    const int k = k0 + dd;
    const bool live = k < N;
    const int kc = live ? k : N - 1;
    double* const L = img[dd];
    // (no barrier here: the image, the candidate record and the gradient are double-buffered, and every lane has passed
    //  the last barrier of the previous step before anything single-buffered is rewritten)
    merit_wave_stage(ring[dd], L, lane);
    if (is_x) {   // (ring.nm is the nominal record's element lane % 16 = r)
      vec[h][r] = x; vec[h][12 + r] = x - ring[dd].nm; das[h][r] = dxda;
      if (cand) crec[dd][r] = x;
    }
    {
      const size_t kn = (k + DEPTH < N) ? k + DEPTH : N - 1;
      merit_wave_load<S>(ring[dd], dynb + kn * a.dyn_ks, outb + kn * a.out_ks, nomb + kn * nom_ks, cpb + kn * cp_ks, lane);
    }
    __syncthreads();
    // this lane's row (of Z, of Kt or of [P | p]) times x (rows of Z) or dx (the others), and times dx/dalpha
    double acc = 0.0, acc2 = 0.0;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const double rj = L[ra[j]];
      acc += rj * vrow[j];
      acc2 += rj * das[h][j];
    }
    const double aff = L[ra[12]];          // -d (rows of Kt), p (rows of [P | p]), B[.][0] (rows of Z)
    if (is_u) {   // u_ = u + (-K dx + alpha d) ; du_da = -K dx_da + d        (Kt = [K | -d])
      const double d = -aff;
      const double uval = L[MW_NOM0 + 12 + ia] + (-acc + alpha * d);
      const double duval = -acc2 + d;
      us[h][ia] = uval; dus[h][ia] = duval;
      if (cand) crec[dd][24 + ia] = uval;
      const double Rd = L[MW_CP0 + 12 + ia], rr = L[MW_CP0 + 28 + ia];
      if (live) J += 0.5 * (uval * (Rd * uval)) + rr * uval;
    }
    if (is_y && cand) crec[dd][12 + i] = acc + aff;      // y_ = P dx + p
    __syncthreads();
    double xn = 0.0, dxn = 0.0;
    if (is_x) {   // x+ = A x + B u + f ; dx+/da = A dx_da + B du_da ; state cost
      double s2 = 0.0, t2 = 0.0;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) { const double bic = L[i * MW_ZLD + 12 + cc]; s2 += bic * us[h][cc]; t2 += bic * dus[h][cc]; }
      xn = (acc + s2) + L[MW_F0 + i];
      dxn = acc2 + t2;
      const double Qd = L[MW_CP0 + i], q = L[MW_CP0 + 16 + i];
      if (live) {
        J += 0.5 * (x * (Qd * x)) + q * x;
        if (r == 0) J += L[MW_CP0 + 32];
      }
    }
    if (kStat && lane >= 48 && live && k >= 1) {   // stationarity at knot point k - 1 now that y_k is known (trial 1's lanes 48..63)
      const double* const Lp = img[dd ^ 1];
      double sy = 0.0;
#pragma unroll
      for (int rr = 0; rr < 12; ++rr) sy += Lp[rr * MW_ZLD + sub] * crec[dd][12 + rr];
      const double g = qrec[dd ^ 1][sub] + sy;
      res = fmax(res, fabs(sub < 12 ? g - crec[dd ^ 1][12 + sub] : g));
    }
    if (al) {
      // both trials' constraint rows at once: lanes r = 16..23 of each half own the rows of their trial (trial 1's are
      // lanes 48..55, where the single-trial kernel has them); the feasibility that counts is the candidate's (trial 1)
      double Ja = 0.0, vv = 0.0;
      wave_al_rows<S, 16>(a.al, kc, b, a.batch, &vec[h][0], &us[h][0], false, rho, r, jv[h], nullptr, nullptr, Ja, vv, false);
      if (live) Jal += Ja;
      if (cand && live) viol = fmax(viol, vv);   // (live: a padding step's point is not on the trajectory)
    }
    __syncthreads();
    if (r < 16) {   // lx (r 0..11) and lu (r 12..15) with the AL terms; dphi
      const int e = r;
      const double pt = e < 12 ? x : us[h][e - 12];
      double l = L[MW_CP0 + e] * pt + L[MW_CP0 + 16 + e];       // costp: Qd | Rd | q | r line up with [x; u]
      if (al) l -= wave_al_col<S>(a.al, kc, e, jv[h]);
      if (cand) qrec[dd][e] = l;
      if (live) dJ += l * (e < 12 ? dxda : dus[h][e - 12]);
    }
    __syncthreads();
    {   // one coalesced store of trial 1's candidate record and of its [lx lu]
      S* c = candb + (size_t)(live ? k : N) * a.xuy_ks;
      c[l27] = (S)crec[dd][l27];
      S* ci = a.cin + (size_t)b * a.cin_bs + (size_t)kc * a.cin_ks;
      const double qv = qrec[dd][sub];
      if (live) ci[MF_OFF_QR + sub] = (S)qv;
    }
    if (is_x && live) { x = xn; dxda = dxn; }
   }
  }
  __syncthreads();
  {   // terminal knot point (solver.cpp:319-332), both trials
    const S* nm = a.nom + ((size_t)N * a.batch + b) * MF_NOM;
    const S* cp = a.costp + ((size_t)N * a.batch + b) * MF_COSTP;
    const S* on = a.outn + (size_t)b * MF_TERM;
    S* c = candb + (size_t)N * a.xuy_ks;
    if (is_u && cand) c[24 + ia] = S(0);
    if (is_x) {
      vec[h][r] = x; vec[h][12 + r] = x - (double)nm[r];
      if (cand) c[r] = (S)x;
      const double Qd = (double)cp[i], q = (double)cp[16 + i];
      J += 0.5 * (x * (Qd * x)) + q * x;
      if (r == 0) J += (double)cp[32];
    }
    __syncthreads();
    if (al) {
      double Ja = 0.0, vv = 0.0;
      wave_al_rows<S, 16>(a.al, N, b, a.batch, &vec[h][0], &us[h][0], true, rho, r, jv[h], nullptr, nullptr, Ja, vv, false);
      Jal += Ja;
      if (cand) viol = fmax(viol, vv);
    }
    if (is_y && cand) {
      double sacc = 0.0;
#pragma unroll
      for (int j = 0; j < 12; ++j) sacc += (double)on[i * 13 + j] * vec[1][12 + j];
      const double y = sacc + (double)on[i * 13 + 12];
      c[12 + i] = (S)y;
      yN[i] = y;
    }
    __syncthreads();
    if (is_x) {
      double lx = (double)cp[i] * x + (double)cp[16 + i];
      if (al) lx -= wave_al_col<S>(a.al, N, i, jv[h]);
      if (cand) { a.term[(size_t)b * MF_TERM + 144 + i] = (S)lx; lxN[i] = lx; }
      dJ += lx * dxda;
    }
    if (kStat) {
      __syncthreads();
      const int pl = (N - 1) & 1;      // the last LIVE step's buffers (a padding step writes the other parity)
      if (lane >= 48) {
        double sy = 0.0;
#pragma unroll
        for (int rr = 0; rr < 12; ++rr) sy += img[pl][rr * MW_ZLD + sub] * yN[rr];
        const double g = qrec[pl][sub] + sy;
        res = fmax(res, fabs(sub < 12 ? g - crec[pl][12 + sub] : g));
      }
      if (cand && is_x) res = fmax(res, fabs(lxN[r] - yN[r]));
    }
  }
  const double phi = half_sum(J + Jal), dphi = half_sum(dJ);   // (J + Jal: the 64-lane butterfly's first step, lane l + lane l + 32)
  if (kStat) { res = wave_max(res); if (al) viol = wave_max(viol); }
  if (r == 0) {
    a.phi[(size_t)h * a.batch + b] = phi;
    a.dphi[(size_t)h * a.batch + b] = dphi;
  }
  if (lane == 0) {
    if (al) a.prob[b].rho_est = rho;
    if (kStat) { a.prob[b].stationarity = res; a.prob[b].feasibility = viol; }
  }
}

// ---- the same two-trial evaluation on the matrix cores ------------------------------------------------------------------------
// wave_merit2_kernel is bound by the LDS pipe, and what it moves through LDS is operand broadcast: every lane re-reads the
// vectors x, dx, dx/dalpha its row multiplies.  MFMA removes exactly that (the backward sweep's reason to use it).  Here the
// state of both trials and their sensitivities are four COLUMNS of the 16 x 16 B operand of v_mfma_f64_16x16x4 --
//     column 0: [x; u] of trial 0      column 2: d[x; u]/dalpha of trial 0
//     column 1: [x; u] of trial 1      column 3: d[x; u]/dalpha of trial 1          (columns 4..15: zero)
// -- with [x; u] in the rows: lane (g, t) = (lane / 16, lane % 16) holds row 4 c + g of column t in register c (c = 0..2: x,
// c = 3: u).  The accumulator layout of the instruction (register r <-> row g + 4 r, column t) IS this B-operand layout, so
//     K dx           (A operand: the rows of Kt)                                            3 MFMA
//     Z [x; u]       (A operand: Z read transposed from the LDS image)  -> the next state   4 MFMA
//     P dx           (A operand: P gathered from its packed triangle)   -> y                3 MFMA
//     Z^T y+         (A operand: Z's own fragment registers, one knot point late)  -> the stationarity residual   3 MFMA
// hand their results to the next product in registers: the recursion never goes through LDS.  Sums are taken in the MFMA's
// order, not in wave_merit_kernel's: results agree to rounding (1e-12 relative asserted), not bit for bit.  Unconstrained
// problems, fp64 records (the default for those); everything else runs wave_merit2_kernel.
typedef double mw_f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mw_f64x4 mw_mfma(double a, double b, mw_f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double mw_from_lane_plus2(double v) { return __shfl_down(v, 2, 64); }

template <typename S>
__global__ __launch_bounds__(64) void wave_merit2_mfma_kernel(IlqrWaveArgs<S> a) {
  constexpr int DEPTH = 2;
  __shared__ double img[2][MW_IMG + 4];
  __shared__ double crec[2][28], qrec[2][16];
  const int b = mf_problem(blockIdx.x, a.batch), lane = threadIdx.x;
  if (b >= a.batch) return;
  if (a.active && !a.active[b]) return;
  const int N = a.N;
  const int g = lane >> 4, t = lane & 15;
  const bool xcol = t < 2;                      // a column that holds a trial's [x; u] (2, 3: its sensitivities)
  const bool live_col = t < 4;
  const bool cand = t == 1;                     // trial 1 writes the candidate trajectory and the expansion
  const double alpha = (t == 0) ? (a.alpha ? a.alpha[b] : a.alpha_const) : 1.0;
  S* __restrict__ candb = a.cand + (size_t)b * a.xuy_bs;
  const int tr = t < 12 ? t : 11;               // row of Z / of [P | p] this lane supplies as A operand (rows 12..15: zeroed)
  const int ta = t < 4 ? t : 3;                 // row of Kt
  int aZ[4], aK[3], aP[3];
#pragma unroll
  for (int c = 0; c < 4; ++c) aZ[c] = tr * MW_ZLD + 4 * c + g;
#pragma unroll
  for (int c = 0; c < 3; ++c) { aK[c] = MW_OUT0 + ta * 13 + 4 * c + g; aP[c] = MW_OUT0 + MF_OFF_P + mf_sym(tr, 4 * c + g); }
  const int l27 = lane < 28 ? lane : 27;
  const S* __restrict__ dynb = a.dyn + (size_t)b * a.dyn_bs;
  const S* __restrict__ outb = a.out + (size_t)b * a.out_bs;
  const S* __restrict__ nomb = a.nom + (size_t)b * MF_NOM;
  const S* __restrict__ cpb = a.costp + (size_t)b * MF_COSTP;
  const size_t nom_ks = (size_t)a.batch * MF_NOM, cp_ks = (size_t)a.batch * MF_COSTP;
  // the state: V[c] = row 4 c + g of this lane's column; columns 0, 1 start at x0, the sensitivities at 0
  double V[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) V[c] = xcol ? (double)a.x0[(size_t)b * 12 + 4 * c + g] : 0.0;
  double J = 0.0, dJ = 0.0, res = 0.0;
  double lxp[3] = {0.0, 0.0, 0.0}, lup = 0.0, yp[3] = {0.0, 0.0, 0.0}, zprev[3] = {0.0, 0.0, 0.0};   // knot point k - 1 (the lag)
  MeritWaveRegs ring[DEPTH];
#pragma unroll
  for (int dd = 0; dd < DEPTH; ++dd) {
    const size_t kk = dd < N ? dd : N - 1;
    merit_wave_load<S>(ring[dd], dynb + kk * a.dyn_ks, outb + kk * a.out_ks, nomb + kk * nom_ks, cpb + kk * cp_ks, lane);
  }
  const int Npad = ((N + DEPTH - 1) / DEPTH) * DEPTH;
  for (int k0 = 0; k0 < Npad; k0 += DEPTH) {
#pragma unroll
   for (int dd = 0; dd < DEPTH; ++dd) {
    const int k = k0 + dd;
    const bool live = k < N;
    const int kc = live ? k : N - 1;
    double* const L = img[dd];
    merit_wave_stage(ring[dd], L, lane);
    double zf[3];                               // Z_k in fragment order: lane (g, t) holds Z[4 c + g][t] = (Z^T)[t][4 c + g]
#pragma unroll
    for (int c = 0; c < 3; ++c) zf[c] = ring[dd].z[c];
    {
      const size_t kn = (k + DEPTH < N) ? k + DEPTH : N - 1;
      merit_wave_load<S>(ring[dd], dynb + kn * a.dyn_ks, outb + kn * a.out_ks, nomb + kn * nom_ks, cpb + kn * cp_ks, lane);
    }
    __syncthreads();
    // dx = x - x_nominal in the trial columns; the sensitivity columns go through unchanged
    double W[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) W[c] = V[c] - (xcol ? L[MW_NOM0 + 4 * c + g] : 0.0);
    // K dx | K dx/dalpha: rows 0..3 of the product = register 0 of lane (g, t)
    mw_f64x4 DK = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 3; ++c) DK = mw_mfma(t < 4 ? L[aK[c]] : 0.0, W[c], DK);
    const double d = L[MW_OUT0 + g * 13 + 12] * -1.0;      // d[g] = -Kt[g][12]
    // u_ = u + (-K dx + alpha d) ; du_da = -K dx_da + d
    const double U = xcol ? L[MW_NOM0 + 12 + g] + (-DK[0] + alpha * d) : (live_col ? -DK[0] + d : 0.0);
    // y_ = P dx + p  (P dx/dalpha in the sensitivity columns is not needed, it rides along)
    mw_f64x4 DY = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 3; ++c) DY = mw_mfma(t < 12 ? L[aP[c]] : 0.0, W[c], DY);
    double y[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = DY[c] + L[MW_OUT0 + MF_OFF_p + 4 * c + g];
    // costs and their gradient at (x_k, u_k), in the trial columns
    double lx[3], lu;
    {
      double Jk = 0.0;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double Qd = L[MW_CP0 + 4 * c + g], q = L[MW_CP0 + 16 + 4 * c + g];
        Jk += 0.5 * (V[c] * (Qd * V[c])) + q * V[c];
        lx[c] = Qd * V[c] + q;
      }
      const double Rd = L[MW_CP0 + 12 + g], rr = L[MW_CP0 + 28 + g];
      Jk += 0.5 * (U * (Rd * U)) + rr * U;
      lu = Rd * U + rr;
      if (g == 0) Jk += L[MW_CP0 + 32];
      if (live && xcol) J += Jk;
      // dphi: the sensitivities of trial t sit two columns to the right
      double dk = lu * mw_from_lane_plus2(U);
#pragma unroll
      for (int c = 0; c < 3; ++c) dk += lx[c] * mw_from_lane_plus2(V[c]);
      if (live && xcol) dJ += dk;
    }
    // stationarity at knot point k - 1 now that y_k is known: Z_{k-1}^T y_k in fragment registers, trial 1's column
    if (k >= 1 && live) {
      mw_f64x4 DS = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int c = 0; c < 3; ++c) DS = mw_mfma(zprev[c], y[c], DS);
      if (cand) {
#pragma unroll
        for (int c = 0; c < 3; ++c) res = fmax(res, fabs((lxp[c] + DS[c]) - yp[c]));
        res = fmax(res, fabs(lup + DS[3]));
      }
    }
    // the candidate record x_ | y_ | u_ and [lx lu] of trial 1, gathered for one coalesced store each
    if (cand) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { crec[dd][4 * c + g] = V[c]; crec[dd][12 + 4 * c + g] = y[c]; qrec[dd][4 * c + g] = lx[c]; }
      crec[dd][24 + g] = U;
      qrec[dd][12 + g] = lu;
    }
    if (live) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { lxp[c] = lx[c]; yp[c] = y[c]; zprev[c] = zf[c]; }
      lup = lu;
    }
    // the next state: Z [x; u] (+ f in the trial columns)
    mw_f64x4 DX = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 3; ++c) DX = mw_mfma(t < 12 ? L[aZ[c]] : 0.0, V[c], DX);
    DX = mw_mfma(t < 12 ? L[aZ[3]] : 0.0, U, DX);
    if (live) {
#pragma unroll
      for (int c = 0; c < 3; ++c) V[c] = DX[c] + (xcol ? L[MW_F0 + 4 * c + g] : 0.0);
    }
    __syncthreads();
    {
      S* c = candb + (size_t)(live ? k : N) * a.xuy_ks;
      c[l27] = (S)crec[dd][l27];
      S* ci = a.cin + (size_t)b * a.cin_bs + (size_t)kc * a.cin_ks;
      const double qv = qrec[dd][t];
      if (live) ci[MF_OFF_QR + t] = (S)qv;
    }
   }
  }
  {   // terminal knot point (solver.cpp:319-332), both trials
    const S* nm = a.nom + ((size_t)N * a.batch + b) * MF_NOM;
    const S* cp = a.costp + ((size_t)N * a.batch + b) * MF_COSTP;
    const S* on = a.outn + (size_t)b * MF_TERM;
    S* cN = candb + (size_t)N * a.xuy_ks;
    double W[3], lxN[3];
    {
      double Jk = 0.0, dk = 0.0;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        W[c] = V[c] - (xcol ? (double)nm[4 * c + g] : 0.0);
        const double Qd = (double)cp[4 * c + g], q = (double)cp[16 + 4 * c + g];
        Jk += 0.5 * (V[c] * (Qd * V[c])) + q * V[c];
        lxN[c] = Qd * V[c] + q;
        dk += lxN[c] * mw_from_lane_plus2(V[c]);
      }
      if (g == 0) Jk += (double)cp[32];
      if (xcol) { J += Jk; dJ += dk; }
    }
    mw_f64x4 DY = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 3; ++c) DY = mw_mfma(t < 12 ? (double)on[tr * 13 + 4 * c + g] : 0.0, W[c], DY);
    double yN[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) yN[c] = DY[c] + (double)on[(4 * c + g) * 13 + 12];
    mw_f64x4 DS = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 3; ++c) DS = mw_mfma(zprev[c], yN[c], DS);
    if (cand) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        res = fmax(res, fabs((lxp[c] + DS[c]) - yp[c]));
        res = fmax(res, fabs(lxN[c] - yN[c]));
        cN[4 * c + g] = (S)V[c];
        cN[12 + 4 * c + g] = (S)yN[c];
        a.term[(size_t)b * MF_TERM + 144 + 4 * c + g] = (S)lxN[c];
      }
      res = fmax(res, fabs(lup + DS[3]));
      cN[24 + g] = S(0);
    }
  }
  // per-column sums over the four lane groups; columns 0 and 1 are the trials
  J += __shfl_xor(J, 16, 64); J += __shfl_xor(J, 32, 64);
  dJ += __shfl_xor(dJ, 16, 64); dJ += __shfl_xor(dJ, 32, 64);
  res = wave_max(res);
  if (lane < 2) {
    a.phi[(size_t)lane * a.batch + b] = J;
    a.dphi[(size_t)lane * a.batch + b] = dJ;
  }
  if (lane == 0) { a.prob[b].stationarity = res; a.prob[b].feasibility = 0.0; }
}

// Stationarity (solver.cpp:207-222): max_k |lx + A^T y+ - y|, max_k |lu + B^T y+|
template <typename S>
__global__ __launch_bounds__(64) void wave_stationarity_kernel(IlqrWaveArgs<S> a) {
  const int b = mf_problem(blockIdx.x, a.batch), lane = threadIdx.x;
  if (b >= a.batch) return;
  if (a.active && !a.active[b]) return;
  if (a.skip && a.skip[b]) return;   // wave_merit2_kernel left this candidate's values in the control block already
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
  // Feasibility (solver.cpp:224-231) of the candidate trajectory (STAT_NO_FEAS: wave_feasibility_dpp_kernel computes it)
  double viol = 0.0;
  const bool feas_here = (a.mode & STAT_NO_FEAS) == 0;
  if (a.al.enabled && feas_here) {
    __shared__ double xs[12], us[4], jv[AL_MAXC * AL_MAXP], Jm[AL_MAXC * 64], Hm[AL_MAXC * 16];
    const double rho = a.prob[b].rho;
    for (int k = 0; k <= N; ++k) {
      const S* c = a.cand + (size_t)b * a.xuy_bs + (size_t)k * a.xuy_ks;
      __syncthreads();
      if (lane < 12) xs[lane] = (double)c[lane];
      if (lane >= 12 && lane < 16) us[lane - 12] = k < N ? (double)c[24 + lane - 12] : 0.0;
      __syncthreads();
      double cost = 0.0;
      wave_al_rows<S>(a.al, k, b, a.batch, xs, us, k == N, rho, lane, jv, nullptr, nullptr, cost, viol, false);
    }
    viol = wave_max(viol);
  }
  if (lane == 0) { a.prob[b].stationarity = res; a.prob[b].feasibility = viol; }   // (STAT_NO_FEAS: 0, the start of the atomic maximum)
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
