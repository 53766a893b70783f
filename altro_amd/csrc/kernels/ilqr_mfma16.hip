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
  for (int c = 0; c < AL_TILE_MAXC; ++c) {
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
  for (int c = 0; c < AL_TILE_MAXC; ++c)
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
  __shared__ double xs[12], us[4], jv[AL_TILE_MAXC * AL_MAXP], Jm[AL_TILE_MAXC * 64], Hm[AL_TILE_MAXC * 16];
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
  __shared__ double xs[12], us[4], jv[AL_TILE_MAXC * AL_MAXP], Jm[AL_TILE_MAXC * 64], Hm[AL_TILE_MAXC * 16];
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
  __shared__ double vec[24], das[12], us[4], dus[4], jv[AL_TILE_MAXC * AL_MAXP];   // vec = x | dx
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

// (The two-trial evaluation -- phi(0) and the line search's first step from one pass over the records -- lives in
// kernels/ilqr_merit2_dpp.hip: wave_merit_dpp_kernel<.., DUAL>.  Its LDS form and its matrix-core form, which round 3 built on the way
// there and kept as comparisons (0.86 / 0.99 ms against 0.70-0.75 on C1), were removed in round 5; HISTORY.md has their story.)
// sum over the 32 lanes of this lane's half of the wave: offsets 16 .. 1 of the butterfly wave_sum runs over 64 lanes (whose first
// step, offset 32, only ever adds the exact zeros of the other half: the order of the additions that matter is kept)
__device__ __forceinline__ double half_sum(double v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
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
    __shared__ double xs[12], us[4], jv[AL_TILE_MAXC * AL_MAXP], Jm[AL_TILE_MAXC * 64], Hm[AL_TILE_MAXC * 16];
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
