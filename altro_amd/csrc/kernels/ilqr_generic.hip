// ilqr_generic.hip -- the iLQR loop around the TVLQR sweep for plan GENERIC: any (n_k, m_k) up to 64 -- per-knot-point dimensions
// included, like ALTROSolver::SetDimension(n, m, k_start, k_stop) (altro_solver.cpp:26-47): every block is found through the sweep's
// offset table (generic_arrays.h) and sized by nx[k], nu[k] -- dynamics given as DATA
// (x+ = A_k x + B_k u + f_k: the reference's SetLinearDynamics path, knotpoint_data.cpp:123-142, :406-419, :710-719) and a
// quadratic cost (tracking or dense).  The correctness-first companion of kernels/ilqr_mfma16.hip for the shapes the tile plan
// does not cover (n > 12 or m > 4): one wavefront per problem, vectors exchanged through LDS, the blocks read from the
// reference layout on the device (column-major, lane i <-> row i, so every load is a unit-stride run).  Device-side counterparts of
//   SolverImpl::OpenLoopRollout   solver.cpp:116-131        -> generic_rollout_kernel
//   SolverImpl::CopyTrajectory    solver.cpp:148-157        -> generic_accept_kernel
//   CalcCostGradient / Hessian    knotpoint_data.cpp:650-708 -> generic_expand_kernel
//   SolverImpl::MeritFunction     solver.cpp:273-355        -> generic_merit_kernel
//   SolverImpl::Stationarity      solver.cpp:207-222        -> generic_stationarity_kernel
// The line search, convergence logic and sweep sequencing are the plan-independent kernels of ilqr_loop_kernels.hip.
// Sums are taken per lane and reduced over the wave: results agree with the oracle to rounding (1e-12 relative), not bit for bit.
#pragma once
#include "../rtc_compat.h"   // (also compiled at run time around a caller's model: capi_rtc.hip)

#include "generic_arrays.h"
#include "ilqr_types.h"
#include "al_lane.hip"   // soc_projection / soc_jacobian / soc_hessian

namespace altro_hip {

constexpr int GEN_MAX = 64;   // n, m <= 64: one lane per state row, one per input row
// Which lanes own the input rows: lanes 32.. while every dimension is <= 32 (state rows in lanes 0..31: the two kinds of row run side
// by side), lanes 0.. beyond (a lane then owns a state row AND an input row and does one after the other).
#define GEN_UBASE(a) (((a).n <= 32 && (a).m <= 32) ? 32 : 0)

template <typename T>
struct IlqrGenArgs {
  // dynamics (data) and the backward sweep's inputs / outputs, reference layout [b][k][block]
  const T *A, *B, *f;  int64_t A_bs, B_bs, f_bs;
  T *Q, *R, *H, *q, *r; int64_t Q_bs, R_bs, H_bs, q_bs, r_bs;      // lxx [N+1] | luu | lux | lx [N+1] | lu
  const T *K, *d, *P, *p; int64_t K_bs, d_bs, P_bs, p_bs;
  T *x, *u, *y;        int64_t x_bs, u_bs, y_bs;                    // candidate trajectory x_ [N+1] | u_ [N] | y_ [N+1]
  T *xn, *un;                                                       // nominal, dense: [b][N+1][n], [b][N][m]
  const T *cQ, *cR, *cH, *cq, *cr, *cc;                             // the cost's own blocks, dense: Q [b][N+1][n n], R, H [b][N][..], q, r, c
  const T* x0; int64_t x0_stride;
  const double* alpha; const int* active; double alpha_const;
  double* phi; double* dphi; IlqrProb* prob;
  int N, n, m, batch, want_derivative, mode;                        // n, m: the LARGEST state / input dimension along the horizon
  const int64_t* off;                                               // [(N + 1) * G_NUM]: the sweep's offset table (generic_arrays.h)
  const int *nx, *nu;                                               // [N + 1] dimensions of knot point k
  int64_t sx, su, sQ, sR, sH;                                       // one problem's length of xn | cq, un | cr, cQ, cR, cH (cc: N + 1)
  AlTable<T> al;                                                    // constraint blocks (G: p x (n + m), column-major)
  // a device model instead of dynamics given as data (altro_hip_set_model on plans GENERIC / MFMA32): the rollout and the merit
  // evaluation step the model (explicit midpoint, test_utils.cpp:84-132), the expansion writes A_k, B_k into the sweep's arrays
  ModelParams mp{MODEL_LINEAR, 0.0f, 0, 2.7, 1.5};
  // uniform n <= 31, m <= 8, n + m <= 32, fp64, dynamics as data, constraint blocks (if any) of at most 32 rows in the row-wise cones:
  // MeritFunction runs kernels/ilqr_row32.hip's kernel (the host decides: capi_ilqr.hip)
  int row32 = 0;
  int row32m = 0;                                       // a device model's handle on those shapes: MeritFunction in the row layout too (row32_model.hip)
  double* stat_part = nullptr;                          // row32_stationarity_kernel's per-chunk maxima [chunk][b][2], or null (one chunk)
};

#define GOFF(arr, k) (a.off[(int64_t)(k) * G_NUM + (arr)])

// sum_j M[j ld] v[j], j = 0 .. cnt-1, in index order (M: a row or a column of a block in global memory, v in LDS); EIGHT matrix entries
// are fetched before the first product -- with one load per term a lane waited a global round trip per term, and a knot point's rows
// are chains of a dozen (the merit pass of plan GENERIC at (13, 4), 4096 problems: 3.6 ms, longer than the backward sweep).
template <typename T>
__device__ __forceinline__ double gen_gdot(const T* M, int ld, const double* v, int cnt) {
  double s = 0.0;
  int j = 0;
  for (; j + 8 <= cnt; j += 8) {
    T mv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) mv[q] = M[(j + q) * ld];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += (double)mv[q] * v[j + q];
  }
  if (j + 4 <= cnt) {
    T mv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) mv[q] = M[(j + q) * ld];
#pragma unroll
    for (int q = 0; q < 4; ++q) s += (double)mv[q] * v[j + q];
    j += 4;
  }
  for (; j < cnt; ++j) s += (double)M[j * ld] * v[j];
  return s;
}
// the same with the vector in global memory too (the expansion's one-thread-per-entry kernel)
template <typename T>
__device__ __forceinline__ double gen_gdot_gg(const T* M, int ld, const T* v, int cnt) {
  double s = 0.0;
  int j = 0;
  for (; j + 8 <= cnt; j += 8) {
    T mv[8], vv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { mv[q] = M[(j + q) * ld]; vv[q] = v[j + q]; }
#pragma unroll
    for (int q = 0; q < 8; ++q) s += (double)mv[q] * (double)vv[q];
  }
  for (; j < cnt; ++j) s += (double)M[j * ld] * (double)v[j];
  return s;
}
// the same for two vectors at once: s = sum_j M[j ld] v[j], t = sum_j M[j ld] w[j]
template <typename T>
__device__ __forceinline__ void gen_gdot2(const T* M, int ld, const double* v, const double* w, int cnt, double& s, double& t) {
  s = 0.0; t = 0.0;
  int j = 0;
  for (; j + 8 <= cnt; j += 8) {
    T mv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) mv[q] = M[(j + q) * ld];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const double e = (double)mv[q]; s += e * v[j + q]; t += e * w[j + q]; }
  }
  if (j + 4 <= cnt) {
    T mv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) mv[q] = M[(j + q) * ld];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const double e = (double)mv[q]; s += e * v[j + q]; t += e * w[j + q]; }
    j += 4;
  }
  for (; j < cnt; ++j) { const double e = (double)M[j * ld]; s += e * v[j]; t += e * w[j]; }
}

__device__ __forceinline__ double gen_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double gen_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}


// ---- constraint blocks (augmented-Lagrangian terms, knotpoint_data.cpp:489-613; cones.cpp:13-202) for this plan ---------------------
// Up to GEN_MAXC blocks per knot point, a zero / identity / orthant block of up to GEN_MAXP = 64 rows (the reference takes any number of
// any dimension, knotpoint_data.cpp:155-161).  Lane i owns row i of EVERY block of knot point k -- value c_i = G_i [x; u] - g_i,
// estimated / projected dual, AL cost share, violation -- and publishes to LDS, for the lanes that own gradient / Hessian entries,
//   jv[c * GEN_MAXP + i] = (J^T z_proj)_i     jd[c * GEN_MAXP + i] = J_ii  (the projection Jacobian of those cones is diagonal)
// A second-order-cone block of p <= AL_MAXSOC rows is evaluated whole by lane 0 (one of more rows, up to GEN_MAXSOC, lane per row: below): Jm[c * 16 + i * 4 + r] = J_ir, Hm[c * 16 + i * 4 + r] its curvature.
// xs / us: the point, in LDS (us ignored at the terminal knot point).  Must be called by all lanes; no barrier inside.
constexpr int GEN_AL_JV = GEN_MAXC * GEN_MAXP, GEN_AL_SOC = GEN_MAXC * AL_MAXSOC * AL_MAXSOC;
template <typename T>
__device__ __forceinline__ const AlKnotBig ALTRO_CONST_AS& gen_knot(const AlTable<T>& t, int k, int& zshift) {
  const bool u = t.uniform != 0 && k < t.N;
  zshift = u ? k * t.rows_per_knot : 0;
  return *(const AlKnotBig ALTRO_CONST_AS*)(t.big + (u ? 0 : k));
}
template <typename T>
__device__ __forceinline__ void gen_al_rows(const AlTable<T>& t, int k, int b, int64_t B, int n, int m, const double* xs, const double* us,
                                            bool terminal, double rho_est, int lane, double* jv, double* jd, double* Jm, double* Hm,
                                            double& cost, double& viol, bool dual_update) {
  int zshift;
  const AlKnotBig ALTRO_CONST_AS& kn = gen_knot<T>(t, k, zshift);
  const int i = lane;
  const int ncon = kn.ncon;
  for (int c = 0; c < ncon; ++c) {
    const int p = kn.p[c], cone = kn.cone[c];
    const T* G = t.G + kn.G_off[c];
    jv[c * GEN_MAXP + i] = 0.0;
    if (jd) jd[c * GEN_MAXP + i] = 0.0;
    if (Jm && i < AL_MAXSOC * AL_MAXSOC) { Jm[c * 16 + i] = 0.0; Hm[c * 16 + i] = 0.0; }
    // (a bound-type block -- AlTable::gsel: every row +-e_idx -- adds one product to exact zeros: that product alone, no walk over G)
    const int* sd = t.gsel ? t.gsel + (int64_t)kn.def[c] * (1 + GEN_MAXP) : nullptr;
    const bool bsel = sd && sd[0];
    auto value = [&](int r) -> double {
      double s = 0.0;
      if (bsel) {
        const int se = sd[1 + r], e = (se < 0 ? -se : se) - 1;
        if (e < n) s += (double)G[r + e * p] * xs[e];
        else if (!terminal) s += (double)G[r + e * p] * us[e - n];
      } else {
        for (int e = 0; e < n; ++e) s += (double)G[r + e * p] * xs[e];
        if (!terminal)
          for (int e = 0; e < m; ++e) s += (double)G[r + (n + e) * p] * us[e];
      }
      const double gi = kn.g_per_problem[c] ? (double)t.g[kn.g_off[c] + (int64_t)r * B + b] : (double)t.g[kn.g_off[c] + r];
      return s - gi;
    };
    if (cone != CONE_SOC) {
      if (i < p) {
        const double val = value(i);
        T* zp_ = t.z + (int64_t)(kn.z_off[c] + zshift + i) * B + b;
        const double ze = (double)*zp_ - rho_est * val;
        double zp = 0.0, mkv = 0.0;
        if (cone == CONE_EQUALITY) { zp = ze; mkv = 1.0; viol = fmax(viol, fabs(val)); }
        else if (cone == CONE_INEQUALITY) { zp = fmin(0.0, ze); mkv = (ze <= 0.0) ? 1.0 : 0.0; viol = fmax(viol, fabs(fmin(0.0, val) - val)); }
        cost += zp * zp / (2.0 * rho_est);
        jv[c * GEN_MAXP + i] = mkv * zp;
        if (jd) jd[c * GEN_MAXP + i] = mkv;
        if (dual_update) *zp_ = (T)zp;
      }
    } else if (p > AL_MAXSOC) {
      // A second-order cone of more than AL_MAXSOC rows (p <= GEN_MAXSOC; cones.cpp:13-123 in any dimension): lane i owns row i, the
      // norms and inner products are wave sums (every lane of the wave is here: no branch around the shuffles).  Outside the cone, with
      // a = |v|, u = v / a, t = s / a:  P(v, s) = (a + s)/2 (u, 1),  J = 1/2 [(1 + t) I - t u u^T, u; u^T, 1]  (symmetric),
      // d2/dx2 [b^T P] = 1/(2a) [kappa (I - u u^T) - t (w u^T + u w^T), w; w^T, 0] with gamma = u^T b_v, w = b_v - gamma u,
      // kappa = b_s - t gamma (al_lane.hip: soc_jacobian / soc_hessian, the same closed forms written out per entry).  For the expansion
      // the structure is published instead of the p x p matrices: jd[c GEN_MAXP + i] = u_i, jd[c GEN_MAXP + GEN_MAXSOC + i] = w_i,
      // Jm[c 16 + 0..3] = {region (0 below, 1 inside, 2 outside), t, kappa, 1/(2a)}.
      const int nn = p - 1;
      const bool row = i < p, vrow = i < nn;
      const double val = row ? value(i) : 0.0;
      T* zp_ = t.z + (int64_t)(kn.z_off[c] + zshift + (row ? i : 0)) * B + b;
      const double ze = row ? (double)*zp_ - rho_est * val : 0.0;
      const double az = sqrt(gen_wave_sum(vrow ? ze * ze : 0.0)), sz = __shfl(ze, nn, 64);
      const double av = sqrt(gen_wave_sum(vrow ? val * val : 0.0)), sv = __shfl(val, nn, 64);
      const int region = az <= -sz ? 0 : (az <= sz ? 1 : 2);
      double zp = 0.0, pv = 0.0;
      if (region == 1) zp = ze;
      else if (region == 2) { const double cf = 0.5 * (1.0 + sz / az); zp = vrow ? cf * ze : (i == nn ? cf * az : 0.0); }
      if (av <= -sv) pv = 0.0;
      else if (av <= sv) pv = val;
      else { const double cf = 0.5 * (1.0 + sv / av); pv = vrow ? cf * val : (i == nn ? cf * av : 0.0); }
      if (row) { cost += zp * zp / (2.0 * rho_est); viol = fmax(viol, fabs(pv - val)); }
      // (J^T z_proj)_i
      const double inv_a = region == 2 ? 1.0 / az : 0.0, tt = sz * inv_a;
      const double u = vrow ? ze * inv_a : 0.0;
      const double gamma = gen_wave_sum(u * zp), zps = __shfl(zp, nn, 64);
      double sj = 0.0;
      if (region == 1) sj = zp;
      else if (region == 2) sj = vrow ? 0.5 * (((1.0 + tt) * zp - tt * u * gamma) + u * zps) : (i == nn ? 0.5 * (gamma + zps) : 0.0);
      if (row) {
        jv[c * GEN_MAXP + i] = sj;
        if (dual_update) *zp_ = (T)zp;
      }
      if (Jm) {
        if (i < GEN_MAXSOC) { jd[c * GEN_MAXP + i] = u; jd[c * GEN_MAXP + GEN_MAXSOC + i] = vrow ? zp - gamma * u : 0.0; }
        if (i == 0) { Jm[c * 16 + 0] = (double)region; Jm[c * 16 + 1] = tt; Jm[c * 16 + 2] = zps - tt * gamma; Jm[c * 16 + 3] = 0.5 * inv_a; }
      }
    } else if (i == 0) {
      double val[AL_MAXSOC], ze[AL_MAXSOC], zp[AL_MAXSOC], pv[AL_MAXSOC];
#pragma unroll
      for (int r = 0; r < AL_MAXSOC; ++r) {
        val[r] = 0.0; ze[r] = 0.0;
        if (r < p) {
          val[r] = value(r);
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
        for (int q = 0; q < AL_MAXSOC; ++q) sj += J[q + r * AL_MAXSOC] * zp[q];
        jv[c * GEN_MAXP + r] = sj;
        if (Jm)
          for (int q = 0; q < AL_MAXSOC; ++q) Jm[c * 16 + r * 4 + q] = J[r + q * AL_MAXSOC];
        if (dual_update) t.z[(int64_t)(kn.z_off[c] + zshift + r) * B + b] = (T)zp[r];
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
// sum_c sum_i G_c[i][e] * w[c * GEN_MAXP + i]  for column e of [x; u] of the constraint Jacobians of knot point k
template <typename T>
__device__ __forceinline__ double gen_al_col(const AlTable<T>& t, int k, int e, const double* w) {
  int zshift;
  const AlKnotBig ALTRO_CONST_AS& kn = gen_knot<T>(t, k, zshift);
  double s = 0.0;
  const int ncon = kn.ncon;
  for (int c = 0; c < ncon; ++c) {
    const int p = kn.p[c];
    const T* G = t.G + kn.G_off[c];
    const int* sd = t.gsel ? t.gsel + (int64_t)kn.def[c] * (1 + GEN_MAXP) : nullptr;
    if (sd && sd[0]) {   // bound-type: the rows that select column e (the others would add exact zeros)
      for (int i = 0; i < p; ++i)
        if (sd[1 + i] == e + 1 || sd[1 + i] == -(e + 1)) s += (double)G[i + e * p] * w[c * GEN_MAXP + i];
    } else {
      for (int i = 0; i < p; ++i) s += (double)G[i + e * p] * w[c * GEN_MAXP + i];
    }
  }
  return s;
}

// x_0 = x0 ; x_{k+1} = A x + B u + f on the candidate trajectory (u_ is the guess already stored there)
template <typename T>
__global__ __launch_bounds__(64) void generic_rollout_kernel(IlqrGenArgs<T> a) {
  __shared__ double xs[GEN_MAX], us[GEN_MAX];
  const int b = blockIdx.x, lane = threadIdx.x;
  if (a.active && !a.active[b]) return;
  const int N = a.N;
  double x = (double)a.x0[(int64_t)b * a.x0_stride + (lane < a.nx[0] ? lane : 0)];
  for (int k = 0; k < N; ++k) {
    const int n = a.nx[k], m = a.nu[k], n2 = a.nx[k + 1];
    __syncthreads();
    if (lane < n) { xs[lane] = x; a.x[(int64_t)b * a.x_bs + GOFF(G_x, k) + lane] = (T)x; }
    if (lane < m) us[lane] = (double)a.u[(int64_t)b * a.u_bs + GOFF(G_u, k) + lane];
    __syncthreads();
    const int i = lane < n2 ? lane : 0;                  // row of x_{k+1} (A_k is n2 x n, B_k n2 x m)
    const T* Ak = a.A + (int64_t)b * a.A_bs + GOFF(G_A, k);
    const T* Bk = a.B + (int64_t)b * a.B_bs + GOFF(G_B, k);
    const double s = gen_gdot<T>(Ak + i, n2, xs, n), s2 = gen_gdot<T>(Bk + i, n2, us, m);
    x = (s + s2) + (double)a.f[(int64_t)b * a.f_bs + GOFF(G_f, k) + i];
  }
  if (lane < a.nx[N]) a.x[(int64_t)b * a.x_bs + GOFF(G_x, N) + lane] = (T)x;
}

// nominal <- candidate (x, u)
template <typename T>
__global__ void generic_accept_kernel(IlqrGenArgs<T> a) {
  const int64_t per = a.sx + a.su, total = per * a.batch;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(t / per);
    const int64_t e = t % per;
    if (a.active && !a.active[b]) continue;
    if (e < a.sx) a.xn[(int64_t)b * a.sx + e] = a.x[(int64_t)b * a.x_bs + e];
    else a.un[(int64_t)b * a.su + (e - a.sx)] = a.u[(int64_t)b * a.u_bs + (e - a.sx)];
  }
}

// Expansion at the candidate point, one thread per (problem, knot point, entry of [x; u]):
//   EXPAND_GRADIENT: lx = Q x + H^T u + q, lu = R u + H x + r  (knotpoint_data.cpp:659-668) into the backward sweep's q / r
//   EXPAND_HESSIAN : lxx = Q, luu = R, lux = H (knotpoint_data.cpp:691-698) into its Q / R / H -- a copy of the cost's own blocks
template <typename T>
__global__ void generic_expand_kernel(IlqrGenArgs<T> a) {
  const int N = a.N, w = a.n + a.m;                      // threads per knot point: the largest [x; u]; e < a.n: state row e, else input row e - a.n
  const bool grad = (a.mode & EXPAND_GRADIENT) != 0, hess = (a.mode & EXPAND_HESSIAN) != 0;
  const int64_t total = (int64_t)a.batch * (N + 1) * w;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(t % w);
    const int k = (int)((t / w) % (N + 1));
    const int b = (int)(t / ((int64_t)w * (N + 1)));
    if (a.active && !a.active[b]) continue;
    const bool terminal = k == N;
    const int n = a.nx[k], m = terminal ? 0 : a.nu[k];
    const bool isx = e < a.n;
    const int i = isx ? e : e - a.n;
    if (isx ? i >= n : i >= m) continue;
    const T* xk = a.x + (int64_t)b * a.x_bs + GOFF(G_x, k);
    const T* uk = a.u + (int64_t)b * a.u_bs + GOFF(G_u, k);
    const T* Qk = a.cQ + (int64_t)b * a.sQ + GOFF(G_Q, k);
    const T* Rk = a.cR + (int64_t)b * a.sR + GOFF(G_R, k);
    const T* Hk = a.cH + (int64_t)b * a.sH + GOFF(G_H, k);
    if (grad) {
      double s;
      if (isx) {
        s = gen_gdot_gg<T>(Qk + i, n, xk, n);
        s += (double)a.cq[(int64_t)b * a.sx + GOFF(G_q, k) + i];
        if (!terminal) s += gen_gdot_gg<T>(Hk + i * m, 1, uk, m);
        a.q[(int64_t)b * a.q_bs + GOFF(G_q, k) + i] = (T)s;
      } else {
        s = gen_gdot_gg<T>(Rk + i, m, uk, m);
        s += (double)a.cr[(int64_t)b * a.su + GOFF(G_r, k) + i];
        s += gen_gdot_gg<T>(Hk + i, m, xk, n);
        a.r[(int64_t)b * a.r_bs + GOFF(G_r, k) + i] = (T)s;
      }
    }
    if (hess) {
      if (isx) {
        for (int j = 0; j < n; ++j) a.Q[(int64_t)b * a.Q_bs + GOFF(G_Q, k) + i + j * n] = Qk[i + j * n];
      } else {
        for (int j = 0; j < m; ++j) a.R[(int64_t)b * a.R_bs + GOFF(G_R, k) + i + j * m] = Rk[i + j * m];
        for (int j = 0; j < n; ++j) a.H[(int64_t)b * a.H_bs + GOFF(G_H, k) + i + j * m] = Hk[i + j * m];
      }
    }
  }
}

// MeritFunction (solver.cpp:273-355): closed-loop rollout with step alpha, total cost phi and -- when asked -- the directional
// derivative phi' with the refreshed lx, lu.  Lanes 0..31: state rows, lanes 32..63: input rows.
// STAGE: the knot point's seven matrices (K, P, Q, R, H, A, B) are copied into LDS first -- coalesced, all loads of a batch issued before its
// stores, two batches -- and the rows read from there: two global round trips per knot point instead of one per eight terms of every row
// (eight to ten per knot point at (13, 4)).  The launch stages while sixteen waves still fit a CU.  Dynamic LDS: [jv (with constraint
// blocks)] [the staged matrices].
template <typename T, int CAP>
__device__ __forceinline__ void gen_fetch(T (&v)[CAP], const T* src, int count, int lane) {
#pragma unroll
  for (int c = 0; c < CAP; ++c) { const int e = lane + 64 * c; v[c] = e < count ? src[e] : T(0); }
}
template <typename T, int CAP>
__device__ __forceinline__ void gen_put(T* dst, const T (&v)[CAP], const T* src, int count, int lane) {
#pragma unroll
  for (int c = 0; c < CAP; ++c) { const int e = lane + 64 * c; if (e < count) dst[e] = v[c]; }
  for (int e = 64 * CAP + lane; e < count; e += 64) dst[e] = src[e];
}
inline size_t generic_merit_stage_elems(int n, int m) { return (size_t)3 * n * n + (size_t)3 * n * m + (size_t)m * m; }

// MK != 0: a compiled-in device model of MN states and MM inputs (models.h) in the place of x+ = A x + B u + f: lane 0 evaluates the
// continuous model and its Jacobian at (x, u) and at the midpoint, the state rows form their rows of A_k = I + h Am (I + h/2 A0),
// B_k = h (Am h/2 B0 + Bm) (the chain rule of test_utils.cpp:113-129, in oracle/models_oracle.c's order) from LDS, store them into the
// sweep's arrays (MeritFunction with derivative refreshes the dynamics expansion: solver.cpp:300-305) and take phi' through them.
template <typename T, bool STAGE, int MK = 0, int MN = 1, int MM = 1>
__global__ __launch_bounds__(64, 4) void generic_merit_kernel(IlqrGenArgs<T> a) {
  __shared__ double xs[GEN_MAX], dxs[GEN_MAX], das[GEN_MAX], us[GEN_MAX], dus[GEN_MAX];
  __shared__ double md_xn[MK ? MN : 1], md_J0[MK ? MN * (MN + MM) : 1], md_Jm[MK ? MN * (MN + MM) : 1];
  extern __shared__ __attribute__((aligned(16))) unsigned char gen_dyn[];
  double* const jv = reinterpret_cast<double*>(gen_dyn);                                   // [GEN_AL_JV] when there are constraint blocks
  T* const stg = reinterpret_cast<T*>(gen_dyn + (a.al.enabled ? GEN_AL_JV * sizeof(double) : 0));
  const int b = blockIdx.x, lane = threadIdx.x;
  if (a.active && !a.active[b]) return;
  const int N = a.N;
  const bool al = a.al.enabled != 0;
  const double rho = al ? a.prob[b].rho : 1.0;
  double viol = 0.0;
  const double alpha = a.alpha ? a.alpha[b] : a.alpha_const;
  const bool deriv = a.want_derivative != 0;
  double x = (double)a.x0[(int64_t)b * a.x0_stride + (lane < a.nx[0] ? lane : 0)], dxda = 0.0;
  double J = 0.0, dJ = 0.0;
  for (int k = 0; k < N; ++k) {
    const int n = a.nx[k], m = a.nu[k], n2 = a.nx[k + 1];
    const int ub = GEN_UBASE(a);
    const bool isx = lane < n, isu = lane >= ub && lane - ub < m;
    const int i = isx ? lane : 0, iu = isu ? lane - ub : 0;
    const T* gK = a.K + (int64_t)b * a.K_bs + GOFF(G_K, k);
    const T* gP = a.P + (int64_t)b * a.P_bs + GOFF(G_P, k);
    const T* gQ = a.cQ + (int64_t)b * a.sQ + GOFF(G_Q, k);
    const T* gR = a.cR + (int64_t)b * a.sR + GOFF(G_R, k);
    const T* gH = a.cH + (int64_t)b * a.sH + GOFF(G_H, k);
    const T* gA = a.A + (int64_t)b * a.A_bs + GOFF(G_A, k);
    const T* gB = a.B + (int64_t)b * a.B_bs + GOFF(G_B, k);
    T* const sK = stg; T* const sP = sK + m * n; T* const sQ = sP + n * n; T* const sR = sQ + n * n; T* const sH = sR + m * m;
    T* const sA = sH + m * n; T* const sB = sA + n2 * n;
    __syncthreads();                                     // (the previous knot point's rows are read: the staged blocks may be overwritten)
    if (STAGE) {
      { T rK[2], rP[4], rQ[4];
        gen_fetch<T, 2>(rK, gK, m * n, lane); gen_fetch<T, 4>(rP, gP, n * n, lane); gen_fetch<T, 4>(rQ, gQ, n * n, lane);
        gen_put<T, 2>(sK, rK, gK, m * n, lane); gen_put<T, 4>(sP, rP, gP, n * n, lane); gen_put<T, 4>(sQ, rQ, gQ, n * n, lane); }
      { T rR[1], rH[2], rA[4], rB[2];
        gen_fetch<T, 1>(rR, gR, m * m, lane); gen_fetch<T, 2>(rH, gH, m * n, lane); gen_fetch<T, 4>(rA, gA, n2 * n, lane); gen_fetch<T, 2>(rB, gB, n2 * m, lane);
        gen_put<T, 1>(sR, rR, gR, m * m, lane); gen_put<T, 2>(sH, rH, gH, m * n, lane); gen_put<T, 4>(sA, rA, gA, n2 * n, lane); gen_put<T, 2>(sB, rB, gB, n2 * m, lane); }
    }
    // the knot point's rows (both call sites inline it: the staged blocks keep ds_read, the global ones global_load)
    auto knot = [&](const T* Kk, const T* Pk, const T* Qk, const T* Rk, const T* Hk, const T* Ak, const T* Bk) {
    if (isx) {
      xs[lane] = x; dxs[lane] = x - (double)a.xn[(int64_t)b * a.sx + GOFF(G_x, k) + lane]; das[lane] = dxda;
      a.x[(int64_t)b * a.x_bs + GOFF(G_x, k) + lane] = (T)x;
    }
    __syncthreads();
    if (isu) {   // u_ = u + (-K dx + alpha d) ; du_da = -K dx_da + d
      double s, s2;
      gen_gdot2<T>(Kk + iu, m, dxs, das, n, s, s2);
      const double dk = (double)a.d[(int64_t)b * a.d_bs + GOFF(G_d, k) + iu];
      const double uv = (double)a.un[(int64_t)b * a.su + GOFF(G_u, k) + iu] + (-s + alpha * dk);
      us[iu] = uv; dus[iu] = -s2 + dk;
      a.u[(int64_t)b * a.u_bs + GOFF(G_u, k) + iu] = (T)uv;
    }
    if (isx) {   // y_ = P dx + p
      const double s = gen_gdot<T>(Pk + i, n, dxs, n);
      a.y[(int64_t)b * a.y_bs + GOFF(G_y, k) + i] = (T)(s + (double)a.p[(int64_t)b * a.p_bs + GOFF(G_p, k) + i]);
    }
    __syncthreads();
    if (al) {   // the constraint rows' cost shares at the candidate point; (J^T z_proj) for the gradient below
      double Jal = 0.0;
      gen_al_rows<T>(a.al, k, b, a.batch, n, m, xs, us, false, rho, lane, jv, nullptr, nullptr, nullptr, Jal, viol, false);
      J += Jal;
      __syncthreads();
    }
    if (isx) {   // state row: cost share, lx
      const double qx = gen_gdot<T>(Qk + i, n, xs, n), htu = gen_gdot<T>(Hk + i * m, 1, us, m);
      const double ql = (double)a.cq[(int64_t)b * a.sx + GOFF(G_q, k) + i];
      J += x * (0.5 * qx + ql);
      if (lane == 0) J += (double)a.cc[(int64_t)b * (N + 1) + k];
      double lx = (qx + htu) + ql;
      if (al && deriv) lx -= gen_al_col<T>(a.al, k, i, jv);
      if (deriv) { dJ += lx * dxda; a.q[(int64_t)b * a.q_bs + GOFF(G_q, k) + i] = (T)lx; }
    }
    double xn = 0.0, dxn = 0.0;
    if constexpr (MK == 0) {
    if (lane < n2) {   // row of the next state: A_k is n2 x n, B_k n2 x m
      double s, s2, t, t2;
      gen_gdot2<T>(Ak + lane, n2, xs, das, n, s, t);
      gen_gdot2<T>(Bk + lane, n2, us, dus, m, s2, t2);
      xn = (s + s2) + (double)a.f[(int64_t)b * a.f_bs + GOFF(G_f, k) + lane];
      dxn = t + t2;
    }
    } else {
      using M = DiscreteModel<MK, MN, MM, double>;
      const float h = a.mp.h;
      if (deriv) {   // the two Jacobians zeroed by the whole wave (the evaluating lane then stores what its model knows to be nonzero)
        for (int e = lane; e < MN * (MN + MM); e += 64) { md_J0[e] = 0.0; md_Jm[e] = 0.0; }
        __syncthreads();
      }
      if (lane == 0) {   // the model at (x, u) and at the midpoint
        double xl[MN], ul[MM], k1[MN], xm[MN], k2[MN];
        for (int e = 0; e < MN; ++e) xl[e] = xs[e];
        for (int e = 0; e < MM; ++e) ul[e] = us[e];
        if (deriv) {   // (the Jacobians are written where the rows are formed from -- two local arrays of n (n + m) doubles lived in scratch memory)
          M::cont_fJ_zeroed(a.mp, xl, ul, k1, md_J0);
          for (int e = 0; e < MN; ++e) xm[e] = xl[e] + (double)(h / 2) * k1[e];
          M::cont_fJ_zeroed(a.mp, xm, ul, k2, md_Jm);
        } else {
          M::cont_f(a.mp, xl, ul, k1);
          for (int e = 0; e < MN; ++e) xm[e] = xl[e] + (double)(h / 2) * k1[e];
          M::cont_f(a.mp, xm, ul, k2);
        }
        for (int e = 0; e < MN; ++e) md_xn[e] = xl[e] + (double)h * k2[e];
      }
      __syncthreads();
      if (lane < MN) {
        xn = md_xn[lane];
        if (deriv) {   // row `lane` of A_k and B_k, stored and applied
          T* Aw = const_cast<T*>(Ak); T* Bw = const_cast<T*>(Bk);
          const double* A0 = md_J0; const double* B0 = md_J0 + MN * MN; const double* Am = md_Jm; const double* Bm = md_Jm + MN * MN;
          double t = 0.0;
          for (int jc = 0; jc < MN; ++jc) {
            double sm = 0.0;
            for (int kk = 0; kk < MN; ++kk) sm += ((double)h * Am[lane + kk * MN]) * ((kk == jc ? 1.0 : 0.0) + (double)(h / 2) * A0[kk + jc * MN]);
            const double av = (lane == jc ? 1.0 : 0.0) + sm;
            Aw[lane + jc * MN] = (T)av;
            t += av * das[jc];
          }
          for (int jc = 0; jc < MM; ++jc) {
            double sm = 0.0;
            for (int kk = 0; kk < MN; ++kk) sm += (Am[lane + kk * MN] * (double)(h / 2)) * B0[kk + jc * MN];
            const double bv = (double)h * (sm + Bm[lane + jc * MN]);
            Bw[lane + jc * MN] = (T)bv;
            t += bv * dus[jc];
          }
          dxn = t;
        }
      }
    }
    if (isu) {   // input row: cost share (with the cross term u'Hx), lu
      const double ru = gen_gdot<T>(Rk + iu, m, us, m), hx = gen_gdot<T>(Hk + iu, m, xs, n);
      const double rl = (double)a.cr[(int64_t)b * a.su + GOFF(G_r, k) + iu];
      const double uv = us[iu];
      J += uv * ((0.5 * ru + rl) + hx);
      double lu = (ru + hx) + rl;
      if (al && deriv) lu -= gen_al_col<T>(a.al, k, n + iu, jv);
      if (deriv) { dJ += lu * dus[iu]; a.r[(int64_t)b * a.r_bs + GOFF(G_r, k) + iu] = (T)lu; }
    }
    x = xn; dxda = dxn;                                  // (lanes past n2 carry zeros)
    };
    if (STAGE) knot(sK, sP, sQ, sR, sH, sA, sB);
    else knot(gK, gP, gQ, gR, gH, gA, gB);
  }
  __syncthreads();
  {   // terminal knot point (solver.cpp:319-332)
    const int n = a.nx[N];
    const bool isx = lane < n;
    const int i = isx ? lane : 0;
    if (isx) { xs[lane] = x; dxs[lane] = x - (double)a.xn[(int64_t)b * a.sx + GOFF(G_x, N) + lane]; a.x[(int64_t)b * a.x_bs + GOFF(G_x, N) + lane] = (T)x; }
    __syncthreads();
    if (al) {
      gen_al_rows<T>(a.al, N, b, a.batch, n, 0, xs, us, true, rho, lane, jv, nullptr, nullptr, nullptr, J, viol, false);
      __syncthreads();
    }
    if (isx) {
      const T* Qk = a.cQ + (int64_t)b * a.sQ + GOFF(G_Q, N);
      const T* Pk = a.P + (int64_t)b * a.P_bs + GOFF(G_P, N);
      const double qx = gen_gdot<T>(Qk + i, n, xs, n), s = gen_gdot<T>(Pk + i, n, dxs, n);
      const double ql = (double)a.cq[(int64_t)b * a.sx + GOFF(G_q, N) + i];
      J += x * (0.5 * qx + ql);
      if (lane == 0) J += (double)a.cc[(int64_t)b * (N + 1) + N];
      a.y[(int64_t)b * a.y_bs + GOFF(G_y, N) + i] = (T)(s + (double)a.p[(int64_t)b * a.p_bs + GOFF(G_p, N) + i]);
      double lx = qx + ql;
      if (al && deriv) lx -= gen_al_col<T>(a.al, N, i, jv);
      if (deriv) { dJ += lx * dxda; a.q[(int64_t)b * a.q_bs + GOFF(G_q, N) + i] = (T)lx; }
    }
  }
  const double phi = gen_wave_sum(J), dphi = gen_wave_sum(dJ);
  if (lane == 0) {
    a.phi[b] = phi;
    if (deriv) a.dphi[b] = dphi;
    if (al) a.prob[b].rho_est = rho;
  }
}

// Stationarity (solver.cpp:207-222): max_k |lx + A^T y+ - y|, max_k |lu + B^T y+| on the candidate trajectory, and its feasibility
template <typename T>
__global__ __launch_bounds__(64) void generic_stationarity_kernel(IlqrGenArgs<T> a) {
  __shared__ double yn[GEN_MAX];
  const int b = blockIdx.x, lane = threadIdx.x;
  if (a.active && !a.active[b]) return;
  const int N = a.N;
  double res = 0.0;
  for (int k = 0; k < N; ++k) {
    const int n = a.nx[k], m = a.nu[k], n2 = a.nx[k + 1];
    const int ub = GEN_UBASE(a);
    const bool isx = lane < n, isu = lane >= ub && lane - ub < m;
    const int j = lane, ju = isu ? lane - ub : 0;
    __syncthreads();
    if (lane < n2) yn[lane] = (double)a.y[(int64_t)b * a.y_bs + GOFF(G_y, k + 1) + lane];
    __syncthreads();
    if (isx) {
      const T* Ak = a.A + (int64_t)b * a.A_bs + GOFF(G_A, k);
      const double s = gen_gdot<T>(Ak + j * n2, 1, yn, n2);
      res = fmax(res, fabs(((double)a.q[(int64_t)b * a.q_bs + GOFF(G_q, k) + j] + s) - (double)a.y[(int64_t)b * a.y_bs + GOFF(G_y, k) + j]));
    }
    if (isu) {
      const T* Bk = a.B + (int64_t)b * a.B_bs + GOFF(G_B, k);
      const double s = gen_gdot<T>(Bk + ju * n2, 1, yn, n2);
      res = fmax(res, fabs((double)a.r[(int64_t)b * a.r_bs + GOFF(G_r, k) + ju] + s));
    }
  }
  if (lane < a.nx[N]) res = fmax(res, fabs((double)a.q[(int64_t)b * a.q_bs + GOFF(G_q, N) + lane] - (double)a.y[(int64_t)b * a.y_bs + GOFF(G_y, N) + lane]));
  res = gen_wave_max(res);
  double viol = 0.0;   // Feasibility (solver.cpp:224-231) of the candidate trajectory
  if (a.al.enabled) {
    __shared__ double xs[GEN_MAX], us[GEN_MAX], jv[GEN_AL_JV];
    const double rho = a.prob[b].rho;
    for (int k = 0; k <= N; ++k) {
      const int n = a.nx[k], m = k < N ? a.nu[k] : 0;
      __syncthreads();
      if (lane < n) xs[lane] = (double)a.x[(int64_t)b * a.x_bs + GOFF(G_x, k) + lane];
      if (lane < m) us[lane] = (double)a.u[(int64_t)b * a.u_bs + GOFF(G_u, k) + lane];
      __syncthreads();
      double cost = 0.0;
      gen_al_rows<T>(a.al, k, b, a.batch, n, m, xs, us, k == N, rho, lane, jv, nullptr, nullptr, nullptr, cost, viol, false);
    }
    viol = gen_wave_max(viol);
  }
  if (lane == 0) { a.prob[b].stationarity = res; a.prob[b].feasibility = viol; }
}

// The expansion with constraint blocks, one wave per (problem, knot point) (wave_expand_kernel of kernels/ilqr_mfma16.hip with the
// problem's own dimensions):  EXPAND_GRADIENT  lx, lu with the AL terms (knotpoint_data.cpp:583-595) into the sweep's q / r;
// EXPAND_HESSIAN  [lxx lux^T; lux luu] = the cost's blocks + rho G^T (J^T J + curvature) G (knotpoint_data.cpp:597-613) into Q / R / H.
template <typename T>
__global__ __launch_bounds__(64) void generic_expand_al_kernel(IlqrGenArgs<T> a) {
  __shared__ double xs[GEN_MAX], us[GEN_MAX], jv[GEN_AL_JV], jd[GEN_AL_JV], Jm[GEN_AL_SOC], Hm[GEN_AL_SOC];
  const int lane = threadIdx.x;
  const int64_t wk = blockIdx.x;
  const int b = (int)(wk % a.batch), k = (int)(wk / a.batch);
  const int N = a.N;
  if (k > N) return;
  if (a.active && !a.active[b]) return;
  const bool terminal = k == N;
  const int n = a.nx[k], m = terminal ? 0 : a.nu[k], w = n + m;
  const bool grad = (a.mode & EXPAND_GRADIENT) != 0, hess = (a.mode & EXPAND_HESSIAN) != 0;
  if (lane < n) xs[lane] = (double)a.x[(int64_t)b * a.x_bs + GOFF(G_x, k) + lane];
  if (lane < m) us[lane] = (double)a.u[(int64_t)b * a.u_bs + GOFF(G_u, k) + lane];
  __syncthreads();
  {
    double cost = 0.0, viol = 0.0;
    gen_al_rows<T>(a.al, k, b, a.batch, n, m, xs, us, terminal, a.prob[b].rho_est, lane, jv, jd, Jm, Hm, cost, viol, false);
  }
  __syncthreads();
  const T* Qk = a.cQ + (int64_t)b * a.sQ + GOFF(G_Q, k);
  const T* Rk = terminal ? nullptr : a.cR + (int64_t)b * a.sR + GOFF(G_R, k);
  const T* Hk = terminal ? nullptr : a.cH + (int64_t)b * a.sH + GOFF(G_H, k);
  if (grad) {
    if (lane < n) {
      const int e = lane;
      double s = gen_gdot<T>(Qk + e, n, xs, n);
      s += (double)a.cq[(int64_t)b * a.sx + GOFF(G_q, k) + e];
      if (!terminal) s += gen_gdot<T>(Hk + e * m, 1, us, m);
      s -= gen_al_col<T>(a.al, k, e, jv);
      a.q[(int64_t)b * a.q_bs + GOFF(G_q, k) + e] = (T)s;
    }
    const int ub = GEN_UBASE(a);
    if (!terminal && lane >= ub && lane - ub < m) {
      const int i = lane - ub;
      double s = gen_gdot<T>(Rk + i, m, us, m);
      s += (double)a.cr[(int64_t)b * a.su + GOFF(G_r, k) + i];
      s += gen_gdot<T>(Hk + i, m, xs, n);
      s -= gen_al_col<T>(a.al, k, n + i, jv);
      a.r[(int64_t)b * a.r_bs + GOFF(G_r, k) + i] = (T)s;
    }
  }
  if (hess) {
    const double rho = a.prob[b].rho;
    int zshift;
    const AlKnotBig ALTRO_CONST_AS& kn = gen_knot<T>(a.al, k, zshift);
    const int wt = terminal ? n : w;
    // EXPAND_DIAG (the host: every block of the handle is bound-type, and this solve has stored the full blocks once): what is off the
    // diagonal cannot have changed -- the diagonal's wt entries only
    const bool diag_only = (a.mode & EXPAND_DIAG) != 0;
    for (int t = diag_only ? lane * (wt + 1) : lane; t < wt * wt; t += diag_only ? 64 * (wt + 1) : 64) {
      const int r = t % wt, cc = t / wt;          // entry (r, cc) of the (n + m) x (n + m) block, column-major walk
      if (r < n && cc >= n) continue;             // the lux^T block is not stored
      double v = r < n ? (double)Qk[r + cc * n] : (cc < n ? (double)Hk[(r - n) + cc * m] : (double)Rk[(r - n) + (cc - n) * m]);
      double s = 0.0;
      for (int cidx = 0; cidx < kn.ncon; ++cidx) {
        const int p = kn.p[cidx];
        const T* G = a.al.G + kn.G_off[cidx];
        const double* Jc = Jm + cidx * 16;
        const int* sd = a.al.gsel ? a.al.gsel + (int64_t)kn.def[cidx] * (1 + GEN_MAXP) : nullptr;
        if (sd && sd[0]) {
          // a bound-type block (every row +-e_idx: input and state boxes, pins -- most constraints of an MPC problem): (J G)^T (J G) is
          // diagonal, entry idx = the sum of J_ii^2 over the rows that select idx -- the general loop below adds exactly these terms and
          // exact zeros, at two loads of G per row and entry (16 loads for each of the 289 entries of a (13, 4) knot point with an input
          // box: the texture path bound this kernel at 1.3 - 3.1 ms per launch, profiles/r06h_solve_13_4_bounds_before_row32.txt)
          if (r == cc)
            for (int i = 0; i < p; ++i)
              if (sd[1 + i] == r + 1 || sd[1 + i] == -(r + 1)) { const double jii = jd[cidx * GEN_MAXP + i]; s += (jii * 1.0) * (jii * 1.0); }
        } else if (kn.cone[cidx] != CONE_SOC) {   // diagonal projection Jacobian: (J G)_(i r) = J_ii G_ir
          for (int i = 0; i < p; ++i) {
            const double jii = jd[cidx * GEN_MAXP + i];
            s += (jii * (double)G[i + r * p]) * (jii * (double)G[i + cc * p]);
          }
        } else if (p > AL_MAXSOC) {   // a cone of many rows: the structure gen_al_rows published (region, t, kappa, 1/(2a); u, w)
          const int region = (int)Jc[0];
          if (region == 1) {                        // inside: J = I, no curvature
            for (int i = 0; i < p; ++i) s += (double)G[i + r * p] * (double)G[i + cc * p];
          } else if (region == 2) {
            const double tt = Jc[1], kappa = Jc[2], half = Jc[3];
            const double* uu = jd + cidx * GEN_MAXP;
            const double* ww = uu + GEN_MAXSOC;
            const int nn = p - 1;
            double du_r = 0.0, du_c = 0.0, dw_c = 0.0;
            for (int i = 0; i < nn; ++i) {
              const double gr = (double)G[i + r * p], gc = (double)G[i + cc * p];
              du_r += uu[i] * gr; du_c += uu[i] * gc; dw_c += ww[i] * gc;
            }
            const double gs_r = (double)G[nn + r * p], gs_c = (double)G[nn + cc * p];
            double t1 = 0.0, t2 = 0.0;
            for (int i = 0; i < nn; ++i) {
              const double gr = (double)G[i + r * p], gc = (double)G[i + cc * p];
              const double jr = 0.5 * (((1.0 + tt) * gr - tt * uu[i] * du_r) + uu[i] * gs_r);     // (J G)_(i r)
              const double jc = 0.5 * (((1.0 + tt) * gc - tt * uu[i] * du_c) + uu[i] * gs_c);
              t1 += jr * jc;
              const double hc = half * ((kappa * (gc - uu[i] * du_c) - tt * (ww[i] * du_c + uu[i] * dw_c)) + ww[i] * gs_c);   // (H G)_(i cc)
              t2 += gr * hc;
            }
            t1 += (0.5 * (du_r + gs_r)) * (0.5 * (du_c + gs_c));
            t2 += gs_r * (half * dw_c);
            s += t1 + t2;
          }
        } else {
          for (int i = 0; i < p; ++i) {
            double jr = 0.0, jc = 0.0;
            for (int q = 0; q < p; ++q) { jr += Jc[i * 4 + q] * (double)G[q + r * p]; jc += Jc[i * 4 + q] * (double)G[q + cc * p]; }
            s += jr * jc;
          }
          const double* Hc = Hm + cidx * 16;       // + G^T (d/dz J^T z_proj) G   (knotpoint_data.cpp:561-567)
          for (int i = 0; i < p; ++i) {
            double hc = 0.0;
            for (int q = 0; q < p; ++q) hc += Hc[i * 4 + q] * (double)G[q + cc * p];
            s += (double)G[i + r * p] * hc;
          }
        }
      }
      v += rho * s;
      if (r < n) a.Q[(int64_t)b * a.Q_bs + GOFF(G_Q, k) + r + cc * n] = (T)v;
      else if (cc < n) a.H[(int64_t)b * a.H_bs + GOFF(G_H, k) + (r - n) + cc * m] = (T)v;
      else a.R[(int64_t)b * a.R_bs + GOFF(G_R, k) + (r - n) + (cc - n) * m] = (T)v;
    }
  }
}

// DualUpdate (knotpoint_data.cpp:503-510) for the problems whose sweep asked for it, one wave per (problem, knot point)
template <typename T>
__global__ __launch_bounds__(64) void generic_dual_update_kernel(IlqrGenArgs<T> a) {
  __shared__ double xs[GEN_MAX], us[GEN_MAX], jv[GEN_AL_JV];
  const int lane = threadIdx.x;
  const int64_t wk = blockIdx.x;
  const int b = (int)(wk % a.batch), k = (int)(wk / a.batch);
  const int N = a.N;
  if (k > N || !a.al.enabled) return;
  if (!a.prob[b].dual) return;
  const bool terminal = k == N;
  const int n = a.nx[k], m = terminal ? 0 : a.nu[k];
  if (lane < n) xs[lane] = (double)a.x[(int64_t)b * a.x_bs + GOFF(G_x, k) + lane];
  if (lane < m) us[lane] = (double)a.u[(int64_t)b * a.u_bs + GOFF(G_u, k) + lane];
  __syncthreads();
  double cost = 0.0, viol = 0.0;
  gen_al_rows<T>(a.al, k, b, a.batch, n, m, xs, us, terminal, a.prob[b].rho_est, lane, jv, nullptr, nullptr, nullptr, cost, viol, true);
}

// ---- device models on plan GENERIC: open-loop rollout and dynamics expansion -------------------------------------------------------
// SolverImpl::OpenLoopRollout (solver.cpp:116-131) with a model: one thread per problem steps x_{k+1} = model(x_k, u_k)
template <typename T, int MK, int MN, int MM>
__global__ void generic_model_rollout_kernel(IlqrGenArgs<T> a) {
  using M = DiscreteModel<MK, MN, MM, double>;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  if (a.active && !a.active[b]) return;
  double x[MN], u[MM], xn[MN];
  for (int e = 0; e < MN; ++e) x[e] = (double)a.x0[(int64_t)b * a.x0_stride + e];
  T* xo = a.x + (int64_t)b * a.x_bs;
  const T* ui = a.u + (int64_t)b * a.u_bs;
  for (int k = 0; k < a.N; ++k) {
    for (int e = 0; e < MN; ++e) xo[(int64_t)k * MN + e] = (T)x[e];
    for (int e = 0; e < MM; ++e) u[e] = (double)ui[(int64_t)k * MM + e];
    M::dynamics(a.mp, x, u, xn);
    for (int e = 0; e < MN; ++e) x[e] = xn[e];
  }
  for (int e = 0; e < MN; ++e) xo[(int64_t)a.N * MN + e] = (T)x[e];
}
// KnotPointData::CalcDynamicsExpansion (knotpoint_data.cpp:406-419) at the candidate trajectory: one thread per (problem, knot point)
template <typename T, int MK, int MN, int MM>
__global__ void generic_model_expand_dyn_kernel(IlqrGenArgs<T> a) {
  using M = DiscreteModel<MK, MN, MM, double>;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= (int64_t)a.batch * a.N) return;
  const int b = (int)(t / a.N), k = (int)(t % a.N);
  if (a.active && !a.active[b]) return;
  double x[MN], u[MM], A[MN * MN], B[MN * MM];
  for (int e = 0; e < MN; ++e) x[e] = (double)a.x[(int64_t)b * a.x_bs + (int64_t)k * MN + e];
  for (int e = 0; e < MM; ++e) u[e] = (double)a.u[(int64_t)b * a.u_bs + (int64_t)k * MM + e];
  M::jacobian(a.mp, x, u, A, B);
  T* Ao = const_cast<T*>(a.A) + (int64_t)b * a.A_bs + (int64_t)k * MN * MN;
  T* Bo = const_cast<T*>(a.B) + (int64_t)b * a.B_bs + (int64_t)k * MN * MM;
  for (int e = 0; e < MN * MN; ++e) Ao[e] = (T)A[e];
  for (int e = 0; e < MN * MM; ++e) Bo[e] = (T)B[e];
}

// ALTROSolver::ShiftTrajectory (altro_solver.cpp:283-293) on the candidate trajectory
template <typename T>
__global__ void generic_shift_kernel(IlqrGenArgs<T> a) {
  const int n = a.n, m = a.m, N = a.N;
  const int64_t total = (int64_t)a.batch * (n + m);
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(t % (n + m));
    const int b = (int)(t / (n + m));
    if (e < n) {
      T* c = a.x + (int64_t)b * a.x_bs + e;
      for (int k = 0; k < N; ++k) c[(int64_t)k * n] = c[(int64_t)(k + 1) * n];
    } else {
      T* c = a.u + (int64_t)b * a.u_bs + (e - n);
      for (int k = 0; k < N - 1; ++k) c[(int64_t)k * m] = c[(int64_t)(k + 1) * m];
    }
  }
}

#undef GOFF
#undef GEN_UBASE

template <typename T>
int ilqr_generic_launch(hipStream_t stream, int which, const IlqrGenArgs<T>& a);   // ilqr_launch_generic.hip
bool ilqr_generic_model_supported(int kind, int n, int m);                           // compiled-in device models of plans GENERIC / MFMA32

}  // namespace altro_hip
