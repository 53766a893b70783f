// models.h -- compiled-in dynamics models for the device forward pass (host + device code).
//
// The reference takes dynamics as opaque host std::function callbacks (typedefs.hpp:31-35), which
// cannot run on the GPU.  These are the device-side equivalents of the models the reference's own
// tests use, written from the model equations (test/test_utils.cpp:18-41 double integrator, :43-82
// pendulum, :134-238 kinematic bicycle) and discretised with the same explicit midpoint rule and
// chain-rule Jacobian (test/test_utils.cpp:84-132):
//      x+ = x + h f(x + h/2 f(x, u), u)
//      A  = I + h Am (I + h/2 A0),   B = h (Am h/2 B0 + Bm)
// `float h`: the reference passes the step as a C float; h/2 and (for the double integrator) h*h/2
// are formed in float arithmetic before widening, which is reproduced here.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define ALTRO_HD __host__ __device__ __forceinline__
#else
#define ALTRO_HD inline
#endif

namespace altro_hip {

enum ModelKind { MODEL_LINEAR = 0, MODEL_DOUBLE_INTEGRATOR = 1, MODEL_PENDULUM = 2, MODEL_BICYCLE = 3 };

struct ModelParams {
  int kind;
  float h;         // time step (uniform)
  int frame;       // bicycle: 0 CoG, 1 rear, 2 front
  double length;   // bicycle wheel base (2.7)
  double lr;       // bicycle CoG -> rear axle (1.5)
};

// ---- continuous models: xdot = f(x, u), J = [df/dx df/du] column-major (n x (n+m)) ----------------
template <typename T>
ALTRO_HD void pendulum_f(const T* x, const T* u, T* xdot) {
  const T l = T(0.5), g = T(9.81), b = T(0.1), mm = T(1.0) * l * l;
  xdot[0] = x[1];
  xdot[1] = u[0] / mm - g * sin(x[0]) / l - b * x[1] / mm;
}
template <typename T>
ALTRO_HD void pendulum_J(const T* x, const T* u, T* J) {
  (void)u;
  const T l = T(0.5), g = T(9.81), b = T(0.1), mm = T(1.0) * l * l;
  J[0] = T(0);
  J[1] = -g * cos(x[0]) / l;
  J[2] = T(1);
  J[3] = -b / mm;
  J[4] = T(0);
  J[5] = T(1) / mm;
}

template <typename T>
ALTRO_HD void bicycle_f(const ModelParams& mp, const T* x, const T* u, T* xdot) {
  const T v = u[0], delta_dot = u[1], theta = x[2], delta = x[3];
  const T L = (T)mp.length, lr = (T)mp.lr;
  T omega, st, ct;
  if (mp.frame == 0) {
    const T beta = atan2(lr * delta, L);
    omega = v * cos(beta) * tan(delta) / L;
    st = sin(theta + beta);
    ct = cos(theta + beta);
  } else if (mp.frame == 1) {
    omega = v * tan(delta) / L;
    st = sin(theta);
    ct = cos(theta);
  } else {
    omega = v * sin(delta) / L;
    st = sin(theta + delta);
    ct = cos(theta + delta);
  }
  xdot[0] = v * ct;
  xdot[1] = v * st;
  xdot[2] = omega;
  xdot[3] = delta_dot;
}
template <typename T>
ALTRO_HD void bicycle_J(const ModelParams& mp, const T* x, const T* u, T* J) {
  const T v = u[0], theta = x[2], delta = x[3];
  const T L = (T)mp.length, lr = (T)mp.lr;
  T dbeta = T(0), domega_ddelta, domega_dv, st, ct, ds_dth, dc_dth, ds_dde = T(0), dc_dde = T(0);
  if (mp.frame == 0) {
    const T by = lr * delta, bx = L;
    const T beta = atan2(by, bx);
    dbeta = bx / (bx * bx + by * by) * lr;
    domega_ddelta = v / L * (-sin(beta) * tan(delta) * dbeta + cos(beta) / (cos(delta) * cos(delta)));
    domega_dv = cos(beta) * tan(delta) / L;
    st = sin(theta + beta);
    ct = cos(theta + beta);
    ds_dth = cos(theta + beta);
    dc_dth = -sin(theta + beta);
    ds_dde = cos(theta + beta) * dbeta;
    dc_dde = -sin(theta + beta) * dbeta;
  } else if (mp.frame == 1) {
    domega_ddelta = v / L / (cos(delta) * cos(delta));
    domega_dv = tan(delta) / L;
    st = sin(theta);
    ct = cos(theta);
    ds_dth = cos(theta);
    dc_dth = -sin(theta);
  } else {
    domega_ddelta = v / L * cos(delta);
    domega_dv = sin(delta) / L;
    st = sin(theta + delta);
    ct = cos(theta + delta);
    ds_dth = cos(theta + delta);
    dc_dth = -sin(theta + delta);
    ds_dde = ds_dth;
    dc_dde = dc_dth;
  }
  for (int e = 0; e < 24; ++e) J[e] = T(0);
  J[0 + 2 * 4] = v * dc_dth;
  J[0 + 3 * 4] = v * dc_dde;
  J[0 + 4 * 4] = ct;
  J[1 + 2 * 4] = v * ds_dth;
  J[1 + 3 * 4] = v * ds_dde;
  J[1 + 4 * 4] = st;
  J[2 + 3 * 4] = domega_ddelta;
  J[2 + 4 * 4] = domega_dv;
  J[3 + 5 * 4] = T(1);
}

// f and J of the bicycle at one point with every transcendental evaluated once (the separate functions call
// sin / cos / tan / atan2 on the same arguments up to three times each); same calls on the same arguments,
// so the values are the ones bicycle_f / bicycle_J produce.
template <typename T>
ALTRO_HD void bicycle_fJ(const ModelParams& mp, const T* x, const T* u, T* xdot, T* J) {
  const T v = u[0], delta_dot = u[1], theta = x[2], delta = x[3];
  const T L = (T)mp.length, lr = (T)mp.lr;
  T omega, st, ct, dbeta = T(0), domega_ddelta, domega_dv, ds_dde = T(0), dc_dde = T(0);
  if (mp.frame == 0) {
    const T by = lr * delta, bx = L;
    const T beta = atan2(by, bx);
    const T sb = sin(beta), cb = cos(beta), td = tan(delta), cd = cos(delta);
    st = sin(theta + beta);
    ct = cos(theta + beta);
    omega = v * cb * td / L;
    dbeta = bx / (bx * bx + by * by) * lr;
    domega_ddelta = v / L * (-sb * td * dbeta + cb / (cd * cd));
    domega_dv = cb * td / L;
    ds_dde = ct * dbeta;
    dc_dde = -st * dbeta;
  } else if (mp.frame == 1) {
    const T td = tan(delta), cd = cos(delta);
    omega = v * td / L;
    st = sin(theta);
    ct = cos(theta);
    domega_ddelta = v / L / (cd * cd);
    domega_dv = td / L;
  } else {
    const T sd = sin(delta), cd = cos(delta);
    omega = v * sd / L;
    st = sin(theta + delta);
    ct = cos(theta + delta);
    domega_ddelta = v / L * cd;
    domega_dv = sd / L;
    ds_dde = ct;
    dc_dde = -st;
  }
  xdot[0] = v * ct;
  xdot[1] = v * st;
  xdot[2] = omega;
  xdot[3] = delta_dot;
  for (int e = 0; e < 24; ++e) J[e] = T(0);
  J[0 + 2 * 4] = v * -st;
  J[0 + 3 * 4] = v * dc_dde;
  J[0 + 4 * 4] = ct;
  J[1 + 2 * 4] = v * ct;
  J[1 + 3 * 4] = v * ds_dde;
  J[1 + 4 * 4] = st;
  J[2 + 3 * 4] = domega_ddelta;
  J[2 + 4 * 4] = domega_dv;
  J[3 + 5 * 4] = T(1);
}

// ---- discrete models -------------------------------------------------------------------------------
// KIND is a compile-time ModelKind; n, m the dimensions (double integrator: dim = n/2, the first m
// axes are actuated -- m == dim is the reference's model, m < dim the C1 variant of SURVEY.md 8d).
template <int KIND, int n, int m, typename T>
struct DiscreteModel {
  static ALTRO_HD void cont_f(const ModelParams& mp, const T* x, const T* u, T* xdot) {
    if (KIND == MODEL_PENDULUM) pendulum_f<T>(x, u, xdot);
    else bicycle_f<T>(mp, x, u, xdot);
  }
  static ALTRO_HD void cont_J(const ModelParams& mp, const T* x, const T* u, T* J) {
    if (KIND == MODEL_PENDULUM) pendulum_J<T>(x, u, J);
    else bicycle_J<T>(mp, x, u, J);
  }

  static ALTRO_HD void cont_fJ(const ModelParams& mp, const T* x, const T* u, T* xdot, T* J) {
    if (KIND == MODEL_PENDULUM) { pendulum_f<T>(x, u, xdot); pendulum_J<T>(x, u, J); }
    else bicycle_fJ<T>(mp, x, u, xdot, J);
  }

  // dynamics() and jacobian() in one pass: f(x, u) and the midpoint are formed once and each point's
  // transcendental functions are shared between f and J.  Values are those of the separate functions.
  static ALTRO_HD void step(const ModelParams& mp, const T* x, const T* u, T* xn, T* A, T* B) {
    if (KIND == MODEL_DOUBLE_INTEGRATOR) {
      dynamics(mp, x, u, xn);
      jacobian(mp, x, u, A, B);
    } else {
      const float h = mp.h;
      T k1[n], xm[n], k2[n], J0[n * (n + m)], Jm[n * (n + m)], Tm[n * n];
      cont_fJ(mp, x, u, k1, J0);
      for (int i = 0; i < n; ++i) xm[i] = x[i] + (T)(h / 2) * k1[i];
      cont_fJ(mp, xm, u, k2, Jm);
      for (int i = 0; i < n; ++i) xn[i] = x[i] + (T)h * k2[i];
      const T* A0 = J0;
      const T* B0 = J0 + n * n;
      const T* Am = Jm;
      const T* Bm = Jm + n * n;
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) Tm[i + j * n] = (i == j ? T(1) : T(0)) + (T)(h / 2) * A0[i + j * n];
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
          T s = T(0);
          for (int k = 0; k < n; ++k) s += ((T)h * Am[i + k * n]) * Tm[k + j * n];
          A[i + j * n] = (i == j ? T(1) : T(0)) + s;
        }
      for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i) {
          T s = T(0);
          for (int k = 0; k < n; ++k) s += (Am[i + k * n] * (T)(h / 2)) * B0[k + j * n];
          B[i + j * n] = (T)h * (s + Bm[i + j * n]);
        }
    }
  }

  static ALTRO_HD void dynamics(const ModelParams& mp, const T* x, const T* u, T* xn) {
    const float h = mp.h;
    if (KIND == MODEL_DOUBLE_INTEGRATOR) {
      constexpr int dim = n / 2;
      const T b = (T)(h * h / 2);   // float arithmetic, then widened (test_utils.cpp:20)
      for (int i = 0; i < dim; ++i) {
        const T ui = (i < m) ? u[i] : T(0);
        xn[i] = x[i] + x[i + dim] * (T)h + ui * b;
        xn[i + dim] = x[i + dim] + ui * (T)h;
      }
    } else {
      T xm[n];
      cont_f(mp, x, u, xm);
      for (int i = 0; i < n; ++i) xm[i] *= (T)(h / 2);
      for (int i = 0; i < n; ++i) xm[i] += x[i];
      cont_f(mp, xm, u, xn);
      for (int i = 0; i < n; ++i) xn[i] = x[i] + (T)h * xn[i];
    }
  }

  // A (n x n) and B (n x m), column-major
  static ALTRO_HD void jacobian(const ModelParams& mp, const T* x, const T* u, T* A, T* B) {
    const float h = mp.h;
    if (KIND == MODEL_DOUBLE_INTEGRATOR) {
      constexpr int dim = n / 2;
      const T b = (T)(h * h / 2);
      for (int e = 0; e < n * n; ++e) A[e] = T(0);
      for (int e = 0; e < n * m; ++e) B[e] = T(0);
      for (int i = 0; i < dim; ++i) {
        A[i + i * n] = T(1);
        A[(i + dim) + (i + dim) * n] = T(1);
        A[i + (i + dim) * n] = (T)h;
        if (i < m) {
          B[i + i * n] = b;
          B[(i + dim) + i * n] = (T)h;
        }
      }
    } else {
      T xm[n], J0[n * (n + m)], Jm[n * (n + m)], Tm[n * n];
      cont_f(mp, x, u, xm);
      for (int i = 0; i < n; ++i) xm[i] = x[i] + (T)(h / 2) * xm[i];
      cont_J(mp, x, u, J0);
      cont_J(mp, xm, u, Jm);
      const T* A0 = J0;
      const T* B0 = J0 + n * n;
      const T* Am = Jm;
      const T* Bm = Jm + n * n;
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) Tm[i + j * n] = (i == j ? T(1) : T(0)) + (T)(h / 2) * A0[i + j * n];
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
          T s = T(0);
          for (int k = 0; k < n; ++k) s += ((T)h * Am[i + k * n]) * Tm[k + j * n];
          A[i + j * n] = (i == j ? T(1) : T(0)) + s;
        }
      for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i) {
          T s = T(0);
          for (int k = 0; k < n; ++k) s += (Am[i + k * n] * (T)(h / 2)) * B0[k + j * n];
          B[i + j * n] = (T)h * (s + Bm[i + j * n]);
        }
    }
  }
};

}  // namespace altro_hip
