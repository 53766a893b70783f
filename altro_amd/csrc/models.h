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
#include "rtc_compat.h"

#if defined(__HIPCC__)
#define ALTRO_HD __host__ __device__ __forceinline__
#else
#define ALTRO_HD inline
#endif

#include "fp_contract.h"
// Contraction: `on` = a * b + c is fused where it is ONE source expression, and nowhere else.  Unlike `fast`, which lets
// the optimiser fuse across statements depending on what surrounds the code, this does not depend on the kernel a
// function is inlined into -- the launch-sequenced solve and the fused solve kernel (ilqr_fused.hip) therefore round
// identically, bit for bit.
ALTRO_FP_REGION_ON

namespace altro_hip {

enum ModelKind { MODEL_LINEAR = 0, MODEL_DOUBLE_INTEGRATOR = 1, MODEL_PENDULUM = 2, MODEL_BICYCLE = 3,
                 MODEL_USER = 4 /* altro_hip_set_model_source: the caller's own continuous dynamics, compiled at run time */,
                 MODEL_QUADROTOR = 5 /* 12 states, 4 inputs: the nonlinear model of the (12, 4) tile plan (not a reference model) */,
                 MODEL_QUADROTOR13 = 6 /* 13 states (quaternion attitude), 4 inputs: the compiled-in model of plans GENERIC / MFMA32 */ };

struct ModelParams {
  int kind;
  float h;         // time step (uniform)
  int frame;       // bicycle: 0 CoG, 1 rear, 2 front
  double length;   // bicycle wheel base (2.7)
  double lr;       // bicycle CoG -> rear axle (1.5)
};

// sin and cos of one argument with a single range reduction on the device.  (A hand-written fp64 evaluation -- Cody-Waite
// reduction as a double-double, fdlibm kernels, 0.78 ulp worst case over [-2^20, 2^20] -- was measured against the library
// routine in round 2 and is not faster: the solves moved by less than their run-to-run spread.  tests/test_gpu_trig.py pins
// the accuracy of whatever this calls.)  Every device path calls this one function (sin_hd / cos_hd are its halves).
template <typename T>
ALTRO_HD void sincos_hd(T a, T* s, T* c) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (sizeof(T) == 4) sincosf(a, s, c);
  else sincos(a, s, c);
#else
  *s = sin(a);
  *c = cos(a);
#endif
}
template <typename T>
ALTRO_HD T sin_hd(T a) {
#if defined(__HIP_DEVICE_COMPILE__)
  T s, c;
  sincos_hd<T>(a, &s, &c);
  return s;
#else
  return sin(a);
#endif
}
template <typename T>
ALTRO_HD T cos_hd(T a) {
#if defined(__HIP_DEVICE_COMPILE__)
  T s, c;
  sincos_hd<T>(a, &s, &c);
  return c;
#else
  return cos(a);
#endif
}

// ---- continuous models: xdot = f(x, u), J = [df/dx df/du] column-major (n x (n+m)) ----------------
template <typename T>
ALTRO_HD void pendulum_f(const T* x, const T* u, T* xdot) {
  const T l = T(0.5), g = T(9.81), b = T(0.1), mm = T(1.0) * l * l;
  xdot[0] = x[1];
  xdot[1] = u[0] / mm - g * sin_hd<T>(x[0]) / l - b * x[1] / mm;
}
template <typename T>
ALTRO_HD void pendulum_J(const T* x, const T* u, T* J) {
  (void)u;
  const T l = T(0.5), g = T(9.81), b = T(0.1), mm = T(1.0) * l * l;
  J[0] = T(0);
  J[1] = -g * cos_hd<T>(x[0]) / l;
  J[2] = T(1);
  J[3] = -b / mm;
  J[4] = T(0);
  J[5] = T(1) / mm;
}

// Trigonometry of the kinematic bicycle (test/test_utils.cpp:134-238) at one point.  The reference's test model
// calls atan2 / sin / cos / tan on beta = atan2(lr delta, L), delta and theta + beta; on the GPU the iLQR kernels
// are bound by exactly these fp64 library calls, so they are formed from ONE sincos(theta), ONE sincos(delta) and
// a square root:  cos(beta) = L / rho, sin(beta) = lr delta / rho, rho = sqrt(L^2 + (lr delta)^2)   (L > 0),
// sin(theta + beta) and cos(theta + beta) by the addition theorems, tan(delta) = sin / cos.  Same functions of
// (theta, delta); results differ from the library-call form in the last ulp or two (parity tolerance 1e-10).
template <typename T>
struct BicycleTrig {
  T st, ct;        // sin / cos of the heading of the velocity vector (theta + beta, theta or theta + delta)
  T sb, cb;        // sin / cos of the side-slip angle beta (frame 0)
  T sd, cd, td;    // sin / cos / tan of the steering angle
  T dbeta;         // d beta / d delta (frame 0)
};
template <typename T>
ALTRO_HD BicycleTrig<T> bicycle_trig(const ModelParams& mp, T theta, T delta) {
  BicycleTrig<T> t;
  const T L = (T)mp.length, lr = (T)mp.lr;
  T sth, cth;
  sincos_hd<T>(theta, &sth, &cth);
  sincos_hd<T>(delta, &t.sd, &t.cd);
  t.td = t.sd / t.cd;
  t.sb = T(0); t.cb = T(1); t.dbeta = T(0);
  if (mp.frame == 0) {
    const T by = lr * delta, bx = L;
    const T r2 = bx * bx + by * by;
    const T ir = T(1) / sqrt(r2);
    t.cb = bx * ir;
    t.sb = by * ir;
    t.dbeta = bx / r2 * lr;
    t.st = sth * t.cb + cth * t.sb;
    t.ct = cth * t.cb - sth * t.sb;
  } else if (mp.frame == 1) {
    t.st = sth;
    t.ct = cth;
  } else {
    t.st = sth * t.cd + cth * t.sd;
    t.ct = cth * t.cd - sth * t.sd;
  }
  return t;
}

template <typename T>
ALTRO_HD void bicycle_f_from(const ModelParams& mp, const BicycleTrig<T>& t, const T* u, T* xdot) {
  const T v = u[0], L = (T)mp.length;
  T omega;
  if (mp.frame == 0) omega = v * t.cb * t.td / L;
  else if (mp.frame == 1) omega = v * t.td / L;
  else omega = v * t.sd / L;
  xdot[0] = v * t.ct;
  xdot[1] = v * t.st;
  xdot[2] = omega;
  xdot[3] = u[1];
}
template <typename T>
ALTRO_HD void bicycle_J_from(const ModelParams& mp, const BicycleTrig<T>& t, const T* u, T* J) {
  const T v = u[0], L = (T)mp.length;
  T domega_ddelta, domega_dv, ds_dde = T(0), dc_dde = T(0);
  if (mp.frame == 0) {
    domega_ddelta = v / L * (-t.sb * t.td * t.dbeta + t.cb / (t.cd * t.cd));
    domega_dv = t.cb * t.td / L;
    ds_dde = t.ct * t.dbeta;
    dc_dde = -t.st * t.dbeta;
  } else if (mp.frame == 1) {
    domega_ddelta = v / L / (t.cd * t.cd);
    domega_dv = t.td / L;
  } else {
    domega_ddelta = v / L * t.cd;
    domega_dv = t.sd / L;
    ds_dde = t.ct;
    dc_dde = -t.st;
  }
  for (int e = 0; e < 24; ++e) J[e] = T(0);
  J[0 + 2 * 4] = v * -t.st;
  J[0 + 3 * 4] = v * dc_dde;
  J[0 + 4 * 4] = t.ct;
  J[1 + 2 * 4] = v * t.ct;
  J[1 + 3 * 4] = v * ds_dde;
  J[1 + 4 * 4] = t.st;
  J[2 + 3 * 4] = domega_ddelta;
  J[2 + 4 * 4] = domega_dv;
  J[3 + 5 * 4] = T(1);
}
template <typename T>
ALTRO_HD void bicycle_f(const ModelParams& mp, const T* x, const T* u, T* xdot) {
  bicycle_f_from<T>(mp, bicycle_trig<T>(mp, x[2], x[3]), u, xdot);
}
template <typename T>
ALTRO_HD void bicycle_J(const ModelParams& mp, const T* x, const T* u, T* J) {
  bicycle_J_from<T>(mp, bicycle_trig<T>(mp, x[2], x[3]), u, J);
}
// f and J at one point from one evaluation of the trigonometry
template <typename T>
ALTRO_HD void bicycle_fJ(const ModelParams& mp, const T* x, const T* u, T* xdot, T* J) {
  const BicycleTrig<T> t = bicycle_trig<T>(mp, x[2], x[3]);
  bicycle_f_from<T>(mp, t, u, xdot);
  bicycle_J_from<T>(mp, t, u, J);
}

// ---- quadrotor, 12 states, 4 inputs ------------------------------------------------------------------
// NOT one of the reference's test models (those stop at 4 states): a quaternion-free rigid-body quadrotor, the nonlinear model
// the (12, 4) tile plan ships so that ALTROSolver::SetExplicitDynamics-style dynamics (altro_solver.cpp:68-81) can be iterated
// on the device at the shape BASELINE.json calls "quadrotor-sized".  Same equations as oracle/models_oracle.c (written apart):
//   x = [p (3) | roll phi, pitch theta, yaw psi | v (3, world) | omega (3, body)],  u = [thrust F | torques tau (3)]
//   pdot = v ;  [phi; theta; psi]' = W(phi, theta) omega ;  vdot = -g e3 + F / mass R e3 ;  omegadot = I^-1 (tau - omega x I omega)
// sin / cos of the three Euler angles and 1 / cos(theta): the only transcendental calls and the only division of one evaluation
// (the inertias divide as multiplications by their reciprocals).  The tile plan forms this once per row of 16 lanes
// (kernels/ilqr_tile_model.hip: lane 3, 4, 5 take phi, theta, psi and the row shares the results by DPP).
template <typename T>
struct QuadTrig { T sp, cp, st, ct, ss, cs, ict; };
template <typename T>
ALTRO_HD QuadTrig<T> quadrotor_trig(const T* x) {
  QuadTrig<T> t;
  sincos_hd<T>(x[3], &t.sp, &t.cp);
  sincos_hd<T>(x[4], &t.st, &t.ct);
  sincos_hd<T>(x[5], &t.ss, &t.cs);
  t.ict = T(1) / t.ct;
  return t;
}
constexpr double kQuadMass = 0.5, kQuadG = 9.81, kQuadIx = 0.0023, kQuadIy = 0.0023, kQuadIz = 0.004;
constexpr double kQuadRMass = 1.0 / kQuadMass, kQuadRIx = 1.0 / kQuadIx, kQuadRIy = 1.0 / kQuadIy, kQuadRIz = 1.0 / kQuadIz;
template <typename T>
ALTRO_HD void quadrotor_f_from(const QuadTrig<T>& t, const T* x, const T* u, T* xd) {
  const T tt = t.st * t.ict;
  const T wx = x[9], wy = x[10], wz = x[11];
  xd[0] = x[6]; xd[1] = x[7]; xd[2] = x[8];
  xd[3] = wx + t.sp * tt * wy + t.cp * tt * wz;
  xd[4] = t.cp * wy - t.sp * wz;
  xd[5] = (t.sp * wy + t.cp * wz) * t.ict;
  const T a = u[0] * T(kQuadRMass);
  xd[6] = a * (t.cp * t.st * t.cs + t.sp * t.ss);
  xd[7] = a * (t.cp * t.st * t.ss - t.sp * t.cs);
  xd[8] = a * (t.cp * t.ct) - T(kQuadG);
  xd[9] = (u[1] - T(kQuadIz - kQuadIy) * wy * wz) * T(kQuadRIx);
  xd[10] = (u[2] - T(kQuadIx - kQuadIz) * wz * wx) * T(kQuadRIy);
  xd[11] = (u[3] - T(kQuadIy - kQuadIx) * wx * wy) * T(kQuadRIz);
}
template <typename T>
ALTRO_HD void quadrotor_J_from(const QuadTrig<T>& t, const T* x, const T* u, T* J) {   // 12 x 16, column-major
#pragma unroll
  for (int e = 0; e < 192; ++e) J[e] = T(0);   // (unrolled: the array must dissolve into registers / constants where it is used)
  const T tt = t.st * t.ict, sec2 = t.ict * t.ict;
  const T wx = x[9], wy = x[10], wz = x[11];
  J[0 + 6 * 12] = T(1); J[1 + 7 * 12] = T(1); J[2 + 8 * 12] = T(1);
  J[3 + 3 * 12] = t.cp * tt * wy - t.sp * tt * wz;
  J[3 + 4 * 12] = (t.sp * wy + t.cp * wz) * sec2;
  J[3 + 9 * 12] = T(1); J[3 + 10 * 12] = t.sp * tt; J[3 + 11 * 12] = t.cp * tt;
  J[4 + 3 * 12] = -t.sp * wy - t.cp * wz;
  J[4 + 10 * 12] = t.cp; J[4 + 11 * 12] = -t.sp;
  J[5 + 3 * 12] = (t.cp * wy - t.sp * wz) * t.ict;
  J[5 + 4 * 12] = (t.sp * wy + t.cp * wz) * t.st * sec2;
  J[5 + 10 * 12] = t.sp * t.ict; J[5 + 11 * 12] = t.cp * t.ict;
  const T a = u[0] * T(kQuadRMass);
  J[6 + 3 * 12] = a * (-t.sp * t.st * t.cs + t.cp * t.ss);
  J[6 + 4 * 12] = a * (t.cp * t.ct * t.cs);
  J[6 + 5 * 12] = a * (-t.cp * t.st * t.ss + t.sp * t.cs);
  J[6 + 12 * 12] = (t.cp * t.st * t.cs + t.sp * t.ss) * T(kQuadRMass);
  J[7 + 3 * 12] = a * (-t.sp * t.st * t.ss - t.cp * t.cs);
  J[7 + 4 * 12] = a * (t.cp * t.ct * t.ss);
  J[7 + 5 * 12] = a * (t.cp * t.st * t.cs + t.sp * t.ss);
  J[7 + 12 * 12] = (t.cp * t.st * t.ss - t.sp * t.cs) * T(kQuadRMass);
  J[8 + 3 * 12] = a * (-t.sp * t.ct);
  J[8 + 4 * 12] = a * (-t.cp * t.st);
  J[8 + 12 * 12] = (t.cp * t.ct) * T(kQuadRMass);
  J[9 + 10 * 12] = T(-(kQuadIz - kQuadIy) * kQuadRIx) * wz; J[9 + 11 * 12] = T(-(kQuadIz - kQuadIy) * kQuadRIx) * wy; J[9 + 13 * 12] = T(kQuadRIx);
  J[10 + 9 * 12] = T(-(kQuadIx - kQuadIz) * kQuadRIy) * wz; J[10 + 11 * 12] = T(-(kQuadIx - kQuadIz) * kQuadRIy) * wx; J[10 + 14 * 12] = T(kQuadRIy);
  J[11 + 9 * 12] = T(-(kQuadIy - kQuadIx) * kQuadRIz) * wy; J[11 + 10 * 12] = T(-(kQuadIy - kQuadIx) * kQuadRIz) * wx; J[11 + 15 * 12] = T(kQuadRIz);
}

#if defined(ALTRO_HIP_USER_MODEL) && defined(ALTRO_HIP_TILE_N)
// A caller's model on the (12, 4) tile plan (capi_rtc.hip compiles this unit with ALTRO_HIP_TILE_N / _M = the problem's own
// dimensions n <= 12, m <= 4): the tile's vectors are [x; 0] (12) and [u; 0] (4), its Jacobian 12 x 16 column-major with the
// inputs in columns 12..15.  Padding states have xdot = 0 (they stay at 0; their rows of A are the identity's), padding inputs
// no effect.
template <typename T>
ALTRO_HD void altro_tile_user_f(const T* x, const T* u, T* xdot) {
  T xd[ALTRO_HIP_TILE_N];
  altro_user_dynamics<T>(x, u, xd);
#pragma unroll
  for (int i = 0; i < 12; ++i) xdot[i] = T(0);
#pragma unroll
  for (int i = 0; i < ALTRO_HIP_TILE_N; ++i) xdot[i] = xd[i];
}
template <typename T>
ALTRO_HD void altro_tile_user_J(const T* x, const T* u, T* J) {
  T Ju[ALTRO_HIP_TILE_N * (ALTRO_HIP_TILE_N + ALTRO_HIP_TILE_M)];
#pragma unroll
  for (int e = 0; e < ALTRO_HIP_TILE_N * (ALTRO_HIP_TILE_N + ALTRO_HIP_TILE_M); ++e) Ju[e] = T(0);
  altro_user_jacobian<T>(x, u, Ju);
#pragma unroll
  for (int e = 0; e < 192; ++e) J[e] = T(0);
#pragma unroll
  for (int c = 0; c < ALTRO_HIP_TILE_N + ALTRO_HIP_TILE_M; ++c)
#pragma unroll
    for (int i = 0; i < ALTRO_HIP_TILE_N; ++i)
      J[i + 12 * (c < ALTRO_HIP_TILE_N ? c : 12 + (c - ALTRO_HIP_TILE_N))] = Ju[i + ALTRO_HIP_TILE_N * c];
}
#endif

// ---- quadrotor, 13 states (unit-quaternion attitude), 4 inputs ---------------------------------------------------------------
// The same rigid body with the attitude as a quaternion q = (qw, qx, qy, qz): the state dimension one past the (12, 4) tile, the
// compiled-in device model of plan GENERIC's iLQR loop (and plan MFMA32's, which shares it).  Same equations as
// oracle/models_oracle.c (written apart; pinned there by central differences):
//   x = [p (3) | q (4) | v (3, world) | omega (3, body)],  u = [thrust F | torques tau (3)]
//   pdot = v ;  qdot = 1/2 q (x) [0; omega] ;  vdot = -g e3 + F / mass R(q) e3 ;  omegadot = I^-1 (tau - omega x I omega)
// Polynomial: no transcendental call and no division in an evaluation.
template <typename T>
ALTRO_HD void quadrotor13_f(const T* x, const T* u, T* xd) {
  const T qw = x[3], qx = x[4], qy = x[5], qz = x[6];
  const T wx = x[10], wy = x[11], wz = x[12];
  xd[0] = x[7]; xd[1] = x[8]; xd[2] = x[9];
  xd[3] = T(0.5) * (-qx * wx - qy * wy - qz * wz);
  xd[4] = T(0.5) * (qw * wx + qy * wz - qz * wy);
  xd[5] = T(0.5) * (qw * wy - qx * wz + qz * wx);
  xd[6] = T(0.5) * (qw * wz + qx * wy - qy * wx);
  const T a = u[0] * T(kQuadRMass);
  xd[7] = a * (T(2) * (qx * qz + qw * qy));
  xd[8] = a * (T(2) * (qy * qz - qw * qx));
  xd[9] = a * (T(1) - T(2) * (qx * qx + qy * qy)) - T(kQuadG);
  xd[10] = (u[1] - T(kQuadIz - kQuadIy) * wy * wz) * T(kQuadRIx);
  xd[11] = (u[2] - T(kQuadIx - kQuadIz) * wz * wx) * T(kQuadRIy);
  xd[12] = (u[3] - T(kQuadIy - kQuadIx) * wx * wy) * T(kQuadRIz);
}
// (ZERO = false: the caller has zeroed J -- plan GENERIC's merit kernel does it with all its lanes, the evaluating lane stores the 46 nonzeros)
template <typename T, bool ZERO = true>
ALTRO_HD void quadrotor13_J(const T* x, const T* u, T* J) {   // 13 x 17, column-major
  constexpr int n = 13;
  if (ZERO) {
#pragma unroll
    for (int e = 0; e < 13 * 17; ++e) J[e] = T(0);
  }
  const T qw = x[3], qx = x[4], qy = x[5], qz = x[6];
  const T wx = x[10], wy = x[11], wz = x[12];
#define QJ(i, j) J[(i) + (j) * n]
  QJ(0, 7) = T(1); QJ(1, 8) = T(1); QJ(2, 9) = T(1);
  QJ(3, 4) = T(-0.5) * wx; QJ(3, 5) = T(-0.5) * wy; QJ(3, 6) = T(-0.5) * wz; QJ(3, 10) = T(-0.5) * qx; QJ(3, 11) = T(-0.5) * qy; QJ(3, 12) = T(-0.5) * qz;
  QJ(4, 3) = T(0.5) * wx; QJ(4, 5) = T(0.5) * wz; QJ(4, 6) = T(-0.5) * wy; QJ(4, 10) = T(0.5) * qw; QJ(4, 11) = T(-0.5) * qz; QJ(4, 12) = T(0.5) * qy;
  QJ(5, 3) = T(0.5) * wy; QJ(5, 4) = T(-0.5) * wz; QJ(5, 6) = T(0.5) * wx; QJ(5, 10) = T(0.5) * qz; QJ(5, 11) = T(0.5) * qw; QJ(5, 12) = T(-0.5) * qx;
  QJ(6, 3) = T(0.5) * wz; QJ(6, 4) = T(0.5) * wy; QJ(6, 5) = T(-0.5) * wx; QJ(6, 10) = T(-0.5) * qy; QJ(6, 11) = T(0.5) * qx; QJ(6, 12) = T(0.5) * qw;
  const T a = u[0] * T(kQuadRMass);
  QJ(7, 3) = T(2) * a * qy; QJ(7, 4) = T(2) * a * qz; QJ(7, 5) = T(2) * a * qw; QJ(7, 6) = T(2) * a * qx;
  QJ(7, 13) = T(2) * (qx * qz + qw * qy) * T(kQuadRMass);
  QJ(8, 3) = T(-2) * a * qx; QJ(8, 4) = T(-2) * a * qw; QJ(8, 5) = T(2) * a * qz; QJ(8, 6) = T(2) * a * qy;
  QJ(8, 13) = T(2) * (qy * qz - qw * qx) * T(kQuadRMass);
  QJ(9, 4) = T(-4) * a * qx; QJ(9, 5) = T(-4) * a * qy;
  QJ(9, 13) = (T(1) - T(2) * (qx * qx + qy * qy)) * T(kQuadRMass);
  QJ(10, 11) = -T(kQuadIz - kQuadIy) * wz * T(kQuadRIx); QJ(10, 12) = -T(kQuadIz - kQuadIy) * wy * T(kQuadRIx); QJ(10, 14) = T(kQuadRIx);
  QJ(11, 10) = -T(kQuadIx - kQuadIz) * wz * T(kQuadRIy); QJ(11, 12) = -T(kQuadIx - kQuadIz) * wx * T(kQuadRIy); QJ(11, 15) = T(kQuadRIy);
  QJ(12, 10) = -T(kQuadIy - kQuadIx) * wy * T(kQuadRIz); QJ(12, 11) = -T(kQuadIy - kQuadIx) * wx * T(kQuadRIz); QJ(12, 16) = T(kQuadRIz);
#undef QJ
}

// ---- discrete models -------------------------------------------------------------------------------
// KIND is a compile-time ModelKind; n, m the dimensions (double integrator: dim = n/2, the first m
// axes are actuated -- m == dim is the reference's model, m < dim the C1 variant of SURVEY.md 8d).
template <int KIND, int n, int m, typename T>
struct DiscreteModel {
  // MODEL_USER exists only in the unit hiprtc compiles for altro_hip_set_model_source (capi_rtc.hip): the caller's source --
  // ahead of this header there -- defines, for T = double and float,
  //     template <typename T> __device__ void altro_user_dynamics(const T* x, const T* u, T* xdot);     xdot = f(x, u)
  //     template <typename T> __device__ void altro_user_jacobian(const T* x, const T* u, T* J);        J = [df/dx df/du],
  // J column-major n x (n + m): what ALTROSolver::SetExplicitDynamics' two callbacks compute (altro_solver.cpp:68-81,
  // typedefs.hpp:31-53) for a continuous model; the explicit-midpoint discretisation and its chain rule below are the
  // reference's test harness (test/test_utils.cpp:84-132).
  static ALTRO_HD void cont_f(const ModelParams& mp, const T* x, const T* u, T* xdot) {
    if (KIND == MODEL_PENDULUM) pendulum_f<T>(x, u, xdot);
#if defined(ALTRO_HIP_USER_MODEL) && defined(ALTRO_HIP_TILE_N)
    else if (KIND == MODEL_USER) altro_tile_user_f<T>(x, u, xdot);
#elif defined(ALTRO_HIP_USER_MODEL)
    else if (KIND == MODEL_USER) altro_user_dynamics<T>(x, u, xdot);
#endif
    else if (KIND == MODEL_QUADROTOR) quadrotor_f_from<T>(quadrotor_trig<T>(x), x, u, xdot);
    else if (KIND == MODEL_QUADROTOR13) quadrotor13_f<T>(x, u, xdot);
    else bicycle_f<T>(mp, x, u, xdot);
  }
  static ALTRO_HD void cont_J(const ModelParams& mp, const T* x, const T* u, T* J) {
    if (KIND == MODEL_PENDULUM) pendulum_J<T>(x, u, J);
#if defined(ALTRO_HIP_USER_MODEL) && defined(ALTRO_HIP_TILE_N)
    else if (KIND == MODEL_USER) altro_tile_user_J<T>(x, u, J);
#elif defined(ALTRO_HIP_USER_MODEL)
    else if (KIND == MODEL_USER) altro_user_jacobian<T>(x, u, J);
#endif
    else if (KIND == MODEL_QUADROTOR) quadrotor_J_from<T>(quadrotor_trig<T>(x), x, u, J);
    else if (KIND == MODEL_QUADROTOR13) quadrotor13_J<T>(x, u, J);
    else bicycle_J<T>(mp, x, u, J);
  }

  // the same into a J the caller has zeroed (models that know their zeros skip them; the others fill every entry anyway)
  static ALTRO_HD void cont_fJ_zeroed(const ModelParams& mp, const T* x, const T* u, T* xdot, T* J) {
    if (KIND == MODEL_QUADROTOR13) { quadrotor13_f<T>(x, u, xdot); quadrotor13_J<T, false>(x, u, J); return; }
    cont_fJ(mp, x, u, xdot, J);
  }
  static ALTRO_HD void cont_fJ(const ModelParams& mp, const T* x, const T* u, T* xdot, T* J) {
#if defined(ALTRO_HIP_USER_MODEL) && defined(ALTRO_HIP_TILE_N)
    if (KIND == MODEL_USER) { altro_tile_user_f<T>(x, u, xdot); altro_tile_user_J<T>(x, u, J); return; }
#elif defined(ALTRO_HIP_USER_MODEL)
    if (KIND == MODEL_USER) { altro_user_dynamics<T>(x, u, xdot); altro_user_jacobian<T>(x, u, J); return; }
#endif
    if (KIND == MODEL_QUADROTOR) {   // one set of sincos for f and J
      const QuadTrig<T> t = quadrotor_trig<T>(x);
      quadrotor_f_from<T>(t, x, u, xdot);
      quadrotor_J_from<T>(t, x, u, J);
      return;
    }
    if (KIND == MODEL_QUADROTOR13) { quadrotor13_f<T>(x, u, xdot); quadrotor13_J<T>(x, u, J); return; }
    if (KIND == MODEL_PENDULUM) {   // one sincos for f (sin) and J (cos)
      const T l = T(0.5), g = T(9.81), b = T(0.1), mm = T(1.0) * l * l;
      T sn, cs;
      sincos_hd<T>(x[0], &sn, &cs);
      xdot[0] = x[1];
      xdot[1] = u[0] / mm - g * sn / l - b * x[1] / mm;
      J[0] = T(0); J[1] = -g * cs / l; J[2] = T(1); J[3] = -b / mm; J[4] = T(0); J[5] = T(1) / mm;
    }
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
      // (the SAME expressions as step() below: under `fp contract(on)` identical source rounds identically, which is what
      //  makes a derivative-free merit pass -- a speculative line-search trial -- land on the bits of a full one)
      T k1[n], xm[n], k2[n];
      cont_f(mp, x, u, k1);
      for (int i = 0; i < n; ++i) xm[i] = x[i] + (T)(h / 2) * k1[i];
      cont_f(mp, xm, u, k2);
      for (int i = 0; i < n; ++i) xn[i] = x[i] + (T)h * k2[i];
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

ALTRO_FP_REGION_END   // back to the including translation unit's own mode (fp_contract.h)
