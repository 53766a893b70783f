// al_lane.hip -- augmented-Lagrangian / conic terms for the batched iLQR loop (plan LANE).
//
// Device-side counterparts of, in the reference:
//   SecondOrderConeProjection / Jacobian / Hessian   src/altro/solver/cones.cpp:13-123
//   ConicProjection*, DualCone                       cones.cpp:125-202, cones.hpp:13-30
//   KnotPointData::CalcViolations                    knotpoint_data.cpp:489-501
//   DualUpdate / PenaltyUpdate                       knotpoint_data.cpp:503-517, solver.cpp:383-409
//   CalcProjectedDuals / CalcConicJacobians / CalcConicHessians    knotpoint_data.cpp:523-570
//   CalcConstraintCosts / ..CostGradients / ..CostHessians         knotpoint_data.cpp:572-613
//
// Constraints are linear blocks c(x,u) = G [x;u] - g in a cone K (every constraint of the reference's
// tests has this form).  G (p x (n+m), column-major) is shared by all problems of the batch, so its
// entries are wave-uniform scalar loads; g is shared or per problem.  The only per-(problem, knot point)
// state is the dual z: constraint values, estimated/projected duals, conic Jacobians and Hessians are
// recomputed from (x, u, z, rho) wherever they are needed instead of being stored and re-read.
#pragma once
#include "../rtc_compat.h"

#include "al_types.h"
#include "tvlqr_lane.hip"   // LaneBuf / lane_ld row access

#include "../fp_contract.h"
ALTRO_FP_REGION_ON   // single-expression a * b + c only: the same rounding in every kernel these functions are inlined into (see models.h)
namespace altro_hip {

// cones.cpp:13-38 (p <= AL_MAXSOC, fully unrolled so the arrays stay in registers)
template <typename T>
__device__ __forceinline__ void soc_projection(int p, const T* x, T* px) {
  const int nn = p - 1;
  T s = T(0), a = T(0);
#pragma unroll
  for (int i = 0; i < AL_MAXSOC; ++i) {
    if (i < nn) a += x[i] * x[i];
    if (i == nn) s = x[i];
  }
  a = sqrt(a);
  if (a <= -s) {
#pragma unroll
    for (int i = 0; i < AL_MAXSOC; ++i) px[i] = T(0);
  } else if (a <= s) {
#pragma unroll
    for (int i = 0; i < AL_MAXSOC; ++i) px[i] = x[i];
  } else {
    const T c = T(0.5) * (T(1) + s / a);
#pragma unroll
    for (int i = 0; i < AL_MAXSOC; ++i) px[i] = (i < nn) ? c * x[i] : (i == nn ? c * a : T(0));
  }
}

// cones.cpp:40-77; J is column-major with leading dimension AL_MAXSOC
template <typename T>
__device__ __forceinline__ void soc_jacobian(int p, const T* x, T* J) {
  const int nn = p - 1;
  T s = T(0), a = T(0);
#pragma unroll
  for (int i = 0; i < AL_MAXSOC; ++i) {
    if (i < nn) a += x[i] * x[i];
    if (i == nn) s = x[i];
  }
  a = sqrt(a);
#pragma unroll
  for (int i = 0; i < AL_MAXSOC * AL_MAXSOC; ++i) J[i] = T(0);
  if (a <= -s) return;
  if (a <= s) {
#pragma unroll
    for (int i = 0; i < AL_MAXSOC; ++i)
      if (i < p) J[i + i * AL_MAXSOC] = T(1);
    return;
  }
  // outside: P(v, s) = ((a + s) / 2) (u, 1), u = v / a  =>  J = 1/2 [ I + (s/a)(I - u u^T), u ; u^T, 1 ]
  const T inv_a = T(1) / a, t = s * inv_a;
  T u[AL_MAXSOC];
#pragma unroll
  for (int i = 0; i < AL_MAXSOC; ++i) u[i] = (i < nn) ? x[i] * inv_a : T(0);
#pragma unroll
  for (int j = 0; j < AL_MAXSOC; ++j)
#pragma unroll
    for (int i = 0; i < AL_MAXSOC; ++i) {
      T v = T(0);
      if (i < nn && j < nn) v = T(0.5) * (((i == j) ? T(1) + t : T(0)) - t * u[i] * u[j]);
      else if (i < nn && j == nn) v = T(0.5) * u[i];
      else if (i == nn && j < nn) v = T(0.5) * u[j];
      else if (i == nn && j == nn) v = T(0.5);
      J[i + j * AL_MAXSOC] = v;
    }
}

// H = d2/dx2 [b^T P(x)] (what cones.cpp:79-123 returns), from the closed form: outside the cone, with a = |v|, u = v / a,
// Pi = I - u u^T, gamma = u^T b_v, w = Pi b_v:
//   Hvv = 1/(2a) [ (b_s - (s/a) gamma) Pi - (s/a)(w u^T + u w^T) ],   Hvs = w / (2a),   Hss = 0;
// zero inside and below the cone (P is linear there).  Column-major, leading dimension AL_MAXSOC.
template <typename T>
__device__ __forceinline__ void soc_hessian(int p, const T* x, const T* bb, T* H) {
  const int nn = p - 1;
  T s = T(0), bs = T(0), a = T(0);
#pragma unroll
  for (int i = 0; i < AL_MAXSOC; ++i) {
    if (i < nn) a += x[i] * x[i];
    if (i == nn) { s = x[i]; bs = bb[i]; }
  }
  a = sqrt(a);
#pragma unroll
  for (int i = 0; i < AL_MAXSOC * AL_MAXSOC; ++i) H[i] = T(0);
  if (a <= -s || a <= s) return;
  const T inv_a = T(1) / a, half = T(0.5) * inv_a, t = s * inv_a;
  T u[AL_MAXSOC], w[AL_MAXSOC];
  T gamma = T(0);
#pragma unroll
  for (int i = 0; i < AL_MAXSOC; ++i) {
    u[i] = (i < nn) ? x[i] * inv_a : T(0);
    gamma += u[i] * ((i < nn) ? bb[i] : T(0));
  }
#pragma unroll
  for (int i = 0; i < AL_MAXSOC; ++i) w[i] = (i < nn) ? bb[i] - gamma * u[i] : T(0);
  const T kappa = bs - t * gamma;
#pragma unroll
  for (int j = 0; j < AL_MAXSOC; ++j)
#pragma unroll
    for (int i = 0; i < AL_MAXSOC; ++i) {
      T v = T(0);
      if (i < nn && j < nn) v = half * (kappa * (((i == j) ? T(1) : T(0)) - u[i] * u[j]) - t * (w[i] * u[j] + u[i] * w[j]));
      else if (i < nn && j == nn) v = half * w[i];
      else if (i == nn && j < nn) v = half * w[j];
      H[i + j * AL_MAXSOC] = v;
    }
}

// A block's Jacobian entry (row i, column col) and value: the shared G / the product G [x;u], or -- only in the unit hiprtc
// compiles around a caller's source that defines them (capi_rtc.hip) -- the caller's own c(x, u) and dc/d[x;u]
// (knotpoint_data.cpp:489-567 treats a general constraint exactly so: Gauss-Newton in its Jacobian).
#if defined(ALTRO_HIP_USER_MODEL) && defined(ALTRO_HIP_USER_CONSTRAINTS)
#define AL_G(i, col) (uid ? Guser[(i) + (col) * p] : (T)G[(i) + (col) * p])
#define AL_CVAL(i, lin) (uid ? cuser[i] : (lin))
#else
#define AL_G(i, col) G[(i) + (col) * p]
#define AL_CVAL(i, lin) (lin)
#endif

// All AL terms of one knot point of one problem.  Returns the AL cost; subtracts the gradient terms from
// lx / lu (GRAD), adds the Gauss-Newton (+ SOC curvature) terms to lxx / luu / lux (HESS), and tracks the
// largest constraint violation (viol != nullptr).  rho_est is the penalty the estimated duals were formed
// with, rho the one the Hessian is scaled by -- they differ only in the first sweep of a solve, where the
// reference evaluates the gradient before SetPenalty (solver.cpp:424-430).  When `znew` is set the
// projected duals are written back as the new duals (DualUpdate).
template <int n, int m, typename T, bool GRAD, bool HESS>
__device__ __forceinline__ T al_eval(const AlTable<T>& t, int k, int64_t b, int64_t B, const T* x, const T* u,
                                     bool terminal, T rho_est, T rho, T* lx, T* lu, T* lxx, T* luu, T* lux,
                                     T* viol, bool dual_update, const T* zpre = nullptr) {
  constexpr int w = n + m;
  int zshift;
  const AlKnot ALTRO_CONST_AS& kn = al_knot<T>(t, k, zshift);
  T cost = T(0);
  static_assert(AL_MAXC == 2, "the preloaded duals are selected with j == 0 ? first : second");
  for (int j = 0; j < kn.ncon; ++j) {   // runtime loop: one copy of the block code
    const int p = kn.p[j];
    const int cone = kn.cone[j];
    const bool gpp = kn.g_per_problem[j] != 0;
    const T ALTRO_CONST_AS* G = (const T ALTRO_CONST_AS*)(t.G + kn.G_off[j]);
    const T ALTRO_CONST_AS* gsh = (const T ALTRO_CONST_AS*)(t.g + kn.g_off[j]);   // shared right-hand side
    const T* gpb = t.g + kn.g_off[j] + b;                                         // per-problem right-hand side
    T* z = t.z + (int64_t)(kn.z_off[j] + zshift) * B + b;
#if defined(ALTRO_HIP_USER_MODEL) && defined(ALTRO_HIP_USER_CONSTRAINTS)
    const int uid = kn.user[j];
    T Guser[AL_MAXP * w], cuser[AL_MAXP];
    if (uid) {
      T uu[m];
#pragma unroll
      for (int e = 0; e < m; ++e) uu[e] = terminal ? T(0) : u[e];
      altro_user_constraint<T>(uid - 1, x, uu, cuser);
      altro_user_constraint_jacobian<T>(uid - 1, x, uu, Guser);
    }
#endif
    if (cone != CONE_SOC && kn.sel[j]) {
      // Bound-type block (rows +-e_idx): the same arithmetic as the general branch below with the zero terms of
      // the dot products left out -- adding +-0 never changes a sum, so the results are bit-identical -- and
      // without a single load from G.
      // Per-variable sums of the gradient / Gauss-Newton terms, accumulated row by row in the order the full
      // products take them (rows past p contribute nothing and are skipped by a wave-uniform branch).
      T sx[n], su[m], hx[n], hu[m];
#pragma unroll
      for (int e = 0; e < n; ++e) { sx[e] = T(0); hx[e] = T(0); }
#pragma unroll
      for (int e = 0; e < m; ++e) { su[e] = T(0); hu[e] = T(0); }
      T sq = T(0);
#pragma unroll
      for (int i = 0; i < AL_MAXP; ++i) {
        if (i < p) {
          const int code = kn.sidx[j][i];
          const int idx = (code < 0 ? -code : code) - 1;
          const T sgn = code < 0 ? T(-1) : T(1);
          T v = T(0);
#pragma unroll
          for (int e = 0; e < n; ++e) v = (idx == e) ? x[e] : v;
          if (!terminal) {
#pragma unroll
            for (int e = 0; e < m; ++e) v = (idx == n + e) ? u[e] : v;
          }
          const T val = sgn * v - (gpp ? gpb[(int64_t)i * B] : gsh[i]);
          const T ze = (zpre ? (j == 0 ? zpre[i] : zpre[AL_MAXP + i]) : z[(int64_t)i * B]) - rho_est * val;
          T zpi = T(0), mski = T(0);
          if (cone == CONE_EQUALITY) { zpi = ze; mski = T(1); }
          else if (cone == CONE_INEQUALITY) { zpi = fmin(T(0), ze); mski = (ze <= T(0)) ? T(1) : T(0); }
          sq += zpi * zpi;
          if (viol) {
            T vv = T(0);
            if (cone == CONE_EQUALITY) vv = fabs(val);
            else if (cone == CONE_INEQUALITY) vv = fabs(fmin(T(0), val) - val);
            *viol = fmax(*viol, vv);
          }
          if (dual_update) z[(int64_t)i * B] = zpi;
          if (GRAD) {
            const T t = sgn * (mski * zpi);
#pragma unroll
            for (int e = 0; e < n; ++e) sx[e] += (idx == e) ? t : T(0);
            if (!terminal) {
#pragma unroll
              for (int e = 0; e < m; ++e) su[e] += (idx == n + e) ? t : T(0);
            }
          }
          if (HESS) {
            const T hh = (mski * sgn) * (mski * sgn);
#pragma unroll
            for (int e = 0; e < n; ++e) hx[e] += (idx == e) ? hh : T(0);
            if (!terminal) {
#pragma unroll
              for (int e = 0; e < m; ++e) hu[e] += (idx == n + e) ? hh : T(0);
            }
          }
        }
      }
      cost += sq / (T(2) * rho_est);
      if (GRAD) {
#pragma unroll
        for (int e = 0; e < n; ++e) lx[e] -= sx[e];
        if (!terminal) {
#pragma unroll
          for (int e = 0; e < m; ++e) lu[e] -= su[e];
        }
      }
      if (HESS) {
#pragma unroll
        for (int e = 0; e < n; ++e) lxx[e + e * n] += rho * hx[e];
        if (!terminal) {
#pragma unroll
          for (int e = 0; e < m; ++e) luu[e + e * m] += rho * hu[e];
        }
      }
    } else if (cone != CONE_SOC) {
      // zero / identity / orthant: projection and its Jacobian are diagonal (cones.cpp:125-178)
      T zp[AL_MAXP], msk[AL_MAXP];
      T sq = T(0);
#pragma unroll
      for (int i = 0; i < AL_MAXP; ++i) {
        zp[i] = T(0); msk[i] = T(0);
        if (i < p) {
          T s = T(0);
          for (int e = 0; e < n; ++e) s += AL_G(i, e) * x[e];
          if (!terminal)
            for (int e = 0; e < m; ++e) s += AL_G(i, n + e) * u[e];
          const T val = AL_CVAL(i, s) - (gpp ? gpb[(int64_t)i * B] : gsh[i]);
          const T ze = (zpre ? (j == 0 ? zpre[i] : zpre[AL_MAXP + i]) : z[(int64_t)i * B]) - rho_est * val;
          if (cone == CONE_EQUALITY) { zp[i] = ze; msk[i] = T(1); }                     // dual cone: identity
          else if (cone == CONE_INEQUALITY) { zp[i] = fmin(T(0), ze); msk[i] = (ze <= T(0)) ? T(1) : T(0); }
          // CONE_IDENTITY: dual cone is the zero cone, projection 0, Jacobian 0
          sq += zp[i] * zp[i];
          if (viol) {
            T v = T(0);
            if (cone == CONE_EQUALITY) v = fabs(val);
            else if (cone == CONE_INEQUALITY) v = fabs(fmin(T(0), val) - val);
            *viol = fmax(*viol, v);
          }
          if (dual_update) z[(int64_t)i * B] = zp[i];
        }
      }
      cost += sq / (T(2) * rho_est);
      if (GRAD) {
        for (int e = 0; e < n; ++e) {
          T s = T(0);
#pragma unroll
          for (int i = 0; i < AL_MAXP; ++i)
            if (i < p) s += AL_G(i, e) * (msk[i] * zp[i]);
          lx[e] -= s;
        }
        if (!terminal)
          for (int e = 0; e < m; ++e) {
            T s = T(0);
#pragma unroll
            for (int i = 0; i < AL_MAXP; ++i)
              if (i < p) s += AL_G(i, n + e) * (msk[i] * zp[i]);
            lu[e] -= s;
          }
      }
      if (HESS) {
#pragma unroll
        for (int cb = 0; cb < w; ++cb)
#pragma unroll
          for (int ca = 0; ca < w; ++ca) {
            if (ca < n && cb >= n) continue;            // the reference keeps lux only
            if (terminal && (ca >= n || cb >= n)) continue;
            T s = T(0);
#pragma unroll
            for (int i = 0; i < AL_MAXP; ++i)
              if (i < p) s += (msk[i] * AL_G(i, ca)) * (msk[i] * AL_G(i, cb));
            s = rho * s;
            if (ca < n) lxx[ca + cb * n] += s;
            else if (cb >= n) luu[(ca - n) + (cb - n) * m] += s;
            else lux[(ca - n) + cb * m] += s;
          }
      }
    } else {
      T val[AL_MAXSOC], ze[AL_MAXSOC], zp[AL_MAXSOC];
#pragma unroll
      for (int i = 0; i < AL_MAXSOC; ++i) {
        val[i] = T(0); ze[i] = T(0);
        if (i < p) {
          T s = T(0);
          for (int e = 0; e < n; ++e) s += AL_G(i, e) * x[e];
          if (!terminal)
            for (int e = 0; e < m; ++e) s += AL_G(i, n + e) * u[e];
          val[i] = AL_CVAL(i, s) - (gpp ? gpb[(int64_t)i * B] : gsh[i]);
          ze[i] = (zpre ? (j == 0 ? zpre[i] : zpre[AL_MAXP + i]) : z[(int64_t)i * B]) - rho_est * val[i];
        }
      }
      soc_projection<T>(p, ze, zp);   // the SOC is self-dual (cones.hpp:13-30)
      T sq = T(0);
#pragma unroll
      for (int i = 0; i < AL_MAXSOC; ++i)
        if (i < p) sq += zp[i] * zp[i];
      cost += sq / (T(2) * rho_est);
      if (viol) {
        T pv[AL_MAXSOC];
        soc_projection<T>(p, val, pv);
#pragma unroll
        for (int i = 0; i < AL_MAXSOC; ++i)
          if (i < p) *viol = fmax(*viol, fabs(pv[i] - val[i]));
      }
      if (dual_update) {
#pragma unroll
        for (int i = 0; i < AL_MAXSOC; ++i)
          if (i < p) z[(int64_t)i * B] = zp[i];
      }
      if (GRAD || HESS) {
        T J[AL_MAXSOC * AL_MAXSOC];
        soc_jacobian<T>(p, ze, J);
        if (GRAD) {
          T jvp[AL_MAXSOC];
#pragma unroll
          for (int i = 0; i < AL_MAXSOC; ++i) {
            T s = T(0);
#pragma unroll
            for (int r = 0; r < AL_MAXSOC; ++r) s += J[r + i * AL_MAXSOC] * zp[r];
            jvp[i] = s;
          }
          for (int e = 0; e < n; ++e) {
            T s = T(0);
#pragma unroll
            for (int i = 0; i < AL_MAXSOC; ++i)
              if (i < p) s += AL_G(i, e) * jvp[i];
            lx[e] -= s;
          }
          if (!terminal)
            for (int e = 0; e < m; ++e) {
              T s = T(0);
#pragma unroll
              for (int i = 0; i < AL_MAXSOC; ++i)
                if (i < p) s += AL_G(i, n + e) * jvp[i];
              lu[e] -= s;
            }
        }
        if (HESS) {
          T Hp[AL_MAXSOC * AL_MAXSOC];
          soc_hessian<T>(p, ze, zp, Hp);
          T JG[AL_MAXSOC * w], HG[AL_MAXSOC * w];   // J G and Hp G, p x w
#pragma unroll
          for (int e = 0; e < w; ++e)
#pragma unroll
            for (int i = 0; i < AL_MAXSOC; ++i) {
              T s1 = T(0), s2 = T(0);
#pragma unroll
              for (int r = 0; r < AL_MAXSOC; ++r)
                if (r < p) {
                  const T gre = AL_G(r, e);
                  s1 += J[i + r * AL_MAXSOC] * gre;
                  s2 += Hp[i + r * AL_MAXSOC] * gre;
                }
              JG[i + e * AL_MAXSOC] = s1;
              HG[i + e * AL_MAXSOC] = s2;
            }
#pragma unroll
          for (int cb = 0; cb < w; ++cb)
#pragma unroll
            for (int ca = 0; ca < w; ++ca) {
              if (ca < n && cb >= n) continue;
              if (terminal && (ca >= n || cb >= n)) continue;
              T s1 = T(0), s2 = T(0);
#pragma unroll
              for (int r = 0; r < AL_MAXSOC; ++r)
                if (r < p) {
                  s1 += JG[r + ca * AL_MAXSOC] * JG[r + cb * AL_MAXSOC];
                  s2 += AL_G(r, ca) * HG[r + cb * AL_MAXSOC];
                }
              const T s = rho * s1 + rho * s2;
              if (ca < n) lxx[ca + cb * n] += s;
              else if (cb >= n) luu[(ca - n) + (cb - n) * m] += s;
              else lux[(ca - n) + cb * m] += s;
            }
        }
      }
    }
  }
  return cost;
}

// The duals of knot point k of this lane's problem into registers (zero where no row exists), so that a
// sequential kernel can request them one knot point ahead.  `zrow` = t.z + first problem of the wave.
template <typename T, typename BUF>
__device__ __forceinline__ void al_load_z(const AlTable<T>& t, int k, const BUF& bz, uint32_t lane, uint32_t rowB,
                                          T (&zv)[AL_MAXC * AL_MAXP]) {
  int zshift;
  const AlKnot ALTRO_CONST_AS& kn = al_knot<T>(t, k, zshift);
#pragma unroll
  for (int j = 0; j < AL_MAXC; ++j) {
    const int p = j < kn.ncon ? kn.p[j] : 0;
#pragma unroll
    for (int i = 0; i < AL_MAXP; ++i) {
      zv[j * AL_MAXP + i] = T(0);
      if (i < p) zv[j * AL_MAXP + i] = lane_ld<T>(bz, lane, (uint32_t)(kn.z_off[j] + zshift + i) * rowB);
    }
  }
}

}  // namespace altro_hip
ALTRO_FP_REGION_END   // back to the including translation unit's own mode (fp_contract.h)
