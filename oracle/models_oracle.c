/*
 * oracle/models_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C restatement of the model library the reference's tests use:
 *     /root/reference/test/test_utils.cpp:18-41    discrete double integrator + Jacobian
 *     /root/reference/test/test_utils.cpp:43-82    pendulum continuous dynamics + Jacobian
 *     /root/reference/test/test_utils.cpp:84-132   midpoint discretisation + chain-rule Jacobian
 *     /root/reference/test/test_utils.cpp:134-238  kinematic bicycle (3 reference frames)
 * Pinned by the reference's own model goldens (tests/golden/reference_kats.json):
 *     test/double_integrator_test.cpp:52,59-63, test/pendulum_test.cpp:32,40,
 *     test/bicycle_test.cpp:39,46-48.
 *
 * `float h`: the reference passes the time step as a C float (typedefs.hpp:31-35); the
 * double integrator computes b = h*h/2 in float arithmetic before widening
 * (test_utils.cpp:20).  That is reproduced here because it is visible at 1e-7.
 */
#include <math.h>
#include <string.h>

#define ORACLE_MODEL_DOUBLE_INTEGRATOR 0
#define ORACLE_MODEL_PENDULUM 1
#define ORACLE_MODEL_BICYCLE 2
#define ORACLE_MODEL_QUADROTOR 3   /* NOT from the reference: this repo's own 12-state test model, see below */
#define ORACLE_MODEL_QUADROTOR13 4 /* ... and its 13-state quaternion form (round 6: a shape past the (12, 4) tile) */

/* ---- double integrator (test_utils.cpp:18-41) ---------------------------------- */
void oracle_di_dynamics(double* xnext, const double* x, const double* u, float h, int dim) {
  double b = h * h / 2; /* float arithmetic, then widened -- as in the reference */
  for (int i = 0; i < dim; ++i) {
    xnext[i] = x[i] + x[i + dim] * h + u[i] * b;
    xnext[i + dim] = x[i + dim] + u[i] * h;
  }
}

/* jac is (2dim x 3dim) column-major = [A B] */
void oracle_di_jacobian(double* jac, const double* x, const double* u, float h, int dim) {
  (void)x; (void)u;
  const int n = 2 * dim;
  memset(jac, 0, sizeof(double) * n * 3 * dim);
  double b = h * h / 2;
#define J(i, j) jac[(i) + (j) * n]
  for (int i = 0; i < dim; ++i) {
    J(i, i) = 1.0;
    J(i + dim, i + dim) = 1.0;
    J(i, i + dim) = h;
    J(i, 2 * dim + i) = b;
    J(i + dim, 2 * dim + i) = h;
  }
#undef J
}

/* Variant with a true double time step (used to reproduce the golden constants, which
 * were generated for h = 0.01 as a double: SURVEY.md section 0 item 5).              */
void oracle_di_dynamics_hd(double* xnext, const double* x, const double* u, double h, int dim) {
  double b = h * h / 2;
  for (int i = 0; i < dim; ++i) {
    xnext[i] = x[i] + x[i + dim] * h + u[i] * b;
    xnext[i + dim] = x[i + dim] + u[i] * h;
  }
}
void oracle_di_jacobian_hd(double* jac, double h, int dim) {
  const int n = 2 * dim;
  memset(jac, 0, sizeof(double) * n * 3 * dim);
  double b = h * h / 2;
  for (int i = 0; i < dim; ++i) {
    jac[i + i * n] = 1.0;
    jac[(i + dim) + (i + dim) * n] = 1.0;
    jac[i + (i + dim) * n] = h;
    jac[i + (2 * dim + i) * n] = b;
    jac[(i + dim) + (2 * dim + i) * n] = h;
  }
}

/* ---- pendulum (test_utils.cpp:43-82) ------------------------------------------- */
static const double kPendulumMass = 1.0;
static const double kPendulumLength = 0.5;
static const double kPendulumFrictionCoeff = 0.1;
static const double kPendulumGravity = 9.81;

void oracle_pendulum_dynamics(double* xdot, const double* x, const double* u) {
  double l = kPendulumLength, g = kPendulumGravity, b = kPendulumFrictionCoeff;
  double m = kPendulumMass * l * l;
  double theta = x[0], omega = x[1];
  double omega_dot = u[0] / m - g * sin(theta) / l - b * omega / m;
  xdot[0] = omega;
  xdot[1] = omega_dot;
}

void oracle_pendulum_jacobian(double* jac, const double* x, const double* u) {
  (void)u;
  double l = kPendulumLength, g = kPendulumGravity, b = kPendulumFrictionCoeff;
  double m = kPendulumMass * l * l;
  jac[0] = 0.0;
  jac[1] = -g * cos(x[0]) / l;
  jac[2] = 1.0;
  jac[3] = -b / m;
  jac[4] = 0.0;
  jac[5] = 1 / m;
}

/* ---- bicycle (test_utils.cpp:134-238; defaults test_utils.hpp:142-143) ---------- */
typedef struct {
  int frame;       /* 0 = CenterOfGravity, 1 = Rear, 2 = Front */
  double length;   /* 2.7 */
  double lr;       /* 1.5 */
} oracle_bicycle;

void oracle_bicycle_dynamics(const oracle_bicycle* mdl, double* x_dot, const double* x,
                             const double* u) {
  double v = u[0], delta_dot = u[1], theta = x[2], delta = x[3];
  double beta = 0, omega = 0, stheta = 0, ctheta = 0;
  switch (mdl->frame) {
    case 0:
      beta = atan2(mdl->lr * delta, mdl->length);
      omega = v * cos(beta) * tan(delta) / mdl->length;
      stheta = sin(theta + beta);
      ctheta = cos(theta + beta);
      break;
    case 1:
      omega = v * tan(delta) / mdl->length;
      stheta = sin(theta);
      ctheta = cos(theta);
      break;
    default:
      omega = v * sin(delta) / mdl->length;
      stheta = sin(theta + delta);
      ctheta = cos(theta + delta);
      break;
  }
  x_dot[0] = v * ctheta;
  x_dot[1] = v * stheta;
  x_dot[2] = omega;
  x_dot[3] = delta_dot;
}

void oracle_bicycle_jacobian(const oracle_bicycle* mdl, double* jac, const double* x,
                             const double* u) {
  double v = u[0], theta = x[2], delta = x[3];
  double beta = 0, dbeta_ddelta = 0, by = 0, bx = 0, domega_ddelta = 0, domega_dv = 0;
  double stheta = 0, ctheta = 0, ds_dtheta = 0, dc_dtheta = 0, ds_ddelta = 0, dc_ddelta = 0;
  switch (mdl->frame) {
    case 0:
      by = mdl->lr * delta;
      bx = mdl->length;
      beta = atan2(by, bx);
      dbeta_ddelta = bx / (bx * bx + by * by) * mdl->lr;
      domega_ddelta = v / mdl->length * (-sin(beta) * tan(delta) * dbeta_ddelta +
                                         cos(beta) / (cos(delta) * cos(delta)));
      domega_dv = cos(beta) * tan(delta) / mdl->length;
      stheta = sin(theta + beta);
      ctheta = cos(theta + beta);
      ds_dtheta = +cos(theta + beta);
      dc_dtheta = -sin(theta + beta);
      ds_ddelta = +cos(theta + beta) * dbeta_ddelta;
      dc_ddelta = -sin(theta + beta) * dbeta_ddelta;
      break;
    case 1:
      domega_ddelta = v / mdl->length / (cos(delta) * cos(delta));
      domega_dv = tan(delta) / mdl->length;
      stheta = sin(theta);
      ctheta = cos(theta);
      ds_dtheta = +cos(theta);
      dc_dtheta = -sin(theta);
      break;
    default:
      domega_ddelta = v / mdl->length * cos(delta);
      domega_dv = sin(delta) / mdl->length;
      stheta = sin(theta + delta);
      ctheta = cos(theta + delta);
      ds_dtheta = +cos(theta + delta);
      dc_dtheta = -sin(theta + delta);
      ds_ddelta = ds_dtheta;
      dc_ddelta = dc_dtheta;
      break;
  }
  memset(jac, 0, sizeof(double) * 24);
#define J(i, j) jac[(i) + (j) * 4]
  J(0, 2) = v * dc_dtheta;
  J(0, 3) = v * dc_ddelta;
  J(0, 4) = ctheta;
  J(1, 2) = v * ds_dtheta;
  J(1, 3) = v * ds_ddelta;
  J(1, 4) = stheta;
  J(2, 3) = domega_ddelta;
  J(2, 4) = domega_dv;
  J(3, 5) = 1.0;
#undef J
}

/* ---- quadrotor, 12 states, 4 inputs -----------------------------------------------------------
 * NOT a model of the reference (its tests stop at 4 states): a quaternion-free rigid-body quadrotor used to exercise
 * ALTROSolver::SetExplicitDynamics-style NONLINEAR dynamics at the (12, 4) shape BASELINE.json calls "quadrotor-sized".
 * It enters the solver exactly as the reference's models do -- continuous f and its Jacobian through the midpoint rule
 * below -- and its Jacobian is checked against central differences in tests/test_oracle_kat.py.
 *   x = [p (3) | roll phi, pitch theta, yaw psi | v (3, world) | omega (3, body)],  u = [thrust F | torques tau (3)]
 *   pdot = v ;  [phi; theta; psi]' = W(phi, theta) omega ;  vdot = -g e3 + F/mass R(phi, theta, psi) e3 ;
 *   omegadot = I^-1 (tau - omega x I omega),  I = diag(Ix, Iy, Iz)                                            */
#define QUAD_MASS 0.5
#define QUAD_G 9.81
#define QUAD_IX 0.0023
#define QUAD_IY 0.0023
#define QUAD_IZ 0.004
void oracle_quadrotor_dynamics(double* xd, const double* x, const double* u) {
  const double sp = sin(x[3]), cp = cos(x[3]), st = sin(x[4]), ct = cos(x[4]), ss = sin(x[5]), cs = cos(x[5]);
  const double tt = st / ct;
  const double wx = x[9], wy = x[10], wz = x[11];
  xd[0] = x[6]; xd[1] = x[7]; xd[2] = x[8];
  xd[3] = wx + sp * tt * wy + cp * tt * wz;
  xd[4] = cp * wy - sp * wz;
  xd[5] = (sp * wy + cp * wz) / ct;
  const double a = u[0] / QUAD_MASS;
  xd[6] = a * (cp * st * cs + sp * ss);
  xd[7] = a * (cp * st * ss - sp * cs);
  xd[8] = a * (cp * ct) - QUAD_G;
  xd[9] = (u[1] - (QUAD_IZ - QUAD_IY) * wy * wz) / QUAD_IX;
  xd[10] = (u[2] - (QUAD_IX - QUAD_IZ) * wz * wx) / QUAD_IY;
  xd[11] = (u[3] - (QUAD_IY - QUAD_IX) * wx * wy) / QUAD_IZ;
}
/* jac (12 x 16) column-major = [df/dx df/du] */
void oracle_quadrotor_jacobian(double* jac, const double* x, const double* u) {
  const int n = 12;
  memset(jac, 0, sizeof(double) * 12 * 16);
#define J(i, j) jac[(i) + (j) * n]
  const double sp = sin(x[3]), cp = cos(x[3]), st = sin(x[4]), ct = cos(x[4]), ss = sin(x[5]), cs = cos(x[5]);
  const double tt = st / ct, sec2 = 1.0 / (ct * ct);
  const double wx = x[9], wy = x[10], wz = x[11];
  J(0, 6) = 1.0; J(1, 7) = 1.0; J(2, 8) = 1.0;
  /* Euler-angle rates */
  J(3, 3) = cp * tt * wy - sp * tt * wz;
  J(3, 4) = (sp * wy + cp * wz) * sec2;
  J(3, 9) = 1.0; J(3, 10) = sp * tt; J(3, 11) = cp * tt;
  J(4, 3) = -sp * wy - cp * wz;
  J(4, 10) = cp; J(4, 11) = -sp;
  J(5, 3) = (cp * wy - sp * wz) / ct;
  J(5, 4) = (sp * wy + cp * wz) * st * sec2;
  J(5, 10) = sp / ct; J(5, 11) = cp / ct;
  /* translational acceleration */
  const double a = u[0] / QUAD_MASS;
  J(6, 3) = a * (-sp * st * cs + cp * ss);
  J(6, 4) = a * (cp * ct * cs);
  J(6, 5) = a * (-cp * st * ss + sp * cs);
  J(6, 12) = (cp * st * cs + sp * ss) / QUAD_MASS;
  J(7, 3) = a * (-sp * st * ss - cp * cs);
  J(7, 4) = a * (cp * ct * ss);
  J(7, 5) = a * (cp * st * cs + sp * ss);
  J(7, 12) = (cp * st * ss - sp * cs) / QUAD_MASS;
  J(8, 3) = a * (-sp * ct);
  J(8, 4) = a * (-cp * st);
  J(8, 12) = (cp * ct) / QUAD_MASS;
  /* body rates */
  J(9, 10) = -(QUAD_IZ - QUAD_IY) * wz / QUAD_IX; J(9, 11) = -(QUAD_IZ - QUAD_IY) * wy / QUAD_IX; J(9, 13) = 1.0 / QUAD_IX;
  J(10, 9) = -(QUAD_IX - QUAD_IZ) * wz / QUAD_IY; J(10, 11) = -(QUAD_IX - QUAD_IZ) * wx / QUAD_IY; J(10, 14) = 1.0 / QUAD_IY;
  J(11, 9) = -(QUAD_IY - QUAD_IX) * wy / QUAD_IZ; J(11, 10) = -(QUAD_IY - QUAD_IX) * wx / QUAD_IZ; J(11, 15) = 1.0 / QUAD_IZ;
#undef J
}

/* ---- quadrotor, 13 states (unit-quaternion attitude), 4 inputs -------------------------------------------------------
 * NOT a model of the reference either: the same rigid body with the attitude as a quaternion q = (qw, qx, qy, qz) -- the state
 * dimension one past the (12, 4) tile that round 6's plan MFMA32 and the device models of plan GENERIC are exercised on.
 *   x = [p (3) | q (4) | v (3, world) | omega (3, body)],  u = [thrust F | torques tau (3)]
 *   pdot = v ;  qdot = 1/2 q (x) [0; omega] ;  vdot = -g e3 + F/mass R(q) e3 ;  omegadot = I^-1 (tau - omega x I omega)
 * R(q) e3 in its polynomial form (no normalisation: f is smooth everywhere, and its Jacobian is checked against central
 * differences in tests/test_oracle_kat.py).                                                                              */
void oracle_quadrotor13_dynamics(double* xd, const double* x, const double* u) {
  const double qw = x[3], qx = x[4], qy = x[5], qz = x[6];
  const double wx = x[10], wy = x[11], wz = x[12];
  xd[0] = x[7]; xd[1] = x[8]; xd[2] = x[9];
  xd[3] = 0.5 * (-qx * wx - qy * wy - qz * wz);
  xd[4] = 0.5 * (qw * wx + qy * wz - qz * wy);
  xd[5] = 0.5 * (qw * wy - qx * wz + qz * wx);
  xd[6] = 0.5 * (qw * wz + qx * wy - qy * wx);
  const double a = u[0] / QUAD_MASS;
  xd[7] = a * (2.0 * (qx * qz + qw * qy));
  xd[8] = a * (2.0 * (qy * qz - qw * qx));
  xd[9] = a * (1.0 - 2.0 * (qx * qx + qy * qy)) - QUAD_G;
  xd[10] = (u[1] - (QUAD_IZ - QUAD_IY) * wy * wz) / QUAD_IX;
  xd[11] = (u[2] - (QUAD_IX - QUAD_IZ) * wz * wx) / QUAD_IY;
  xd[12] = (u[3] - (QUAD_IY - QUAD_IX) * wx * wy) / QUAD_IZ;
}
/* jac (13 x 17) column-major = [df/dx df/du] */
void oracle_quadrotor13_jacobian(double* jac, const double* x, const double* u) {
  const int n = 13;
  memset(jac, 0, sizeof(double) * 13 * 17);
#define J(i, j) jac[(i) + (j) * n]
  const double qw = x[3], qx = x[4], qy = x[5], qz = x[6];
  const double wx = x[10], wy = x[11], wz = x[12];
  J(0, 7) = 1.0; J(1, 8) = 1.0; J(2, 9) = 1.0;
  /* quaternion kinematics */
  J(3, 4) = -0.5 * wx; J(3, 5) = -0.5 * wy; J(3, 6) = -0.5 * wz; J(3, 10) = -0.5 * qx; J(3, 11) = -0.5 * qy; J(3, 12) = -0.5 * qz;
  J(4, 3) = 0.5 * wx; J(4, 5) = 0.5 * wz; J(4, 6) = -0.5 * wy; J(4, 10) = 0.5 * qw; J(4, 11) = -0.5 * qz; J(4, 12) = 0.5 * qy;
  J(5, 3) = 0.5 * wy; J(5, 4) = -0.5 * wz; J(5, 6) = 0.5 * wx; J(5, 10) = 0.5 * qz; J(5, 11) = 0.5 * qw; J(5, 12) = -0.5 * qx;
  J(6, 3) = 0.5 * wz; J(6, 4) = 0.5 * wy; J(6, 5) = -0.5 * wx; J(6, 10) = -0.5 * qy; J(6, 11) = 0.5 * qx; J(6, 12) = 0.5 * qw;
  /* translational acceleration */
  const double a = u[0] / QUAD_MASS;
  J(7, 3) = 2.0 * a * qy; J(7, 4) = 2.0 * a * qz; J(7, 5) = 2.0 * a * qw; J(7, 6) = 2.0 * a * qx;
  J(7, 13) = 2.0 * (qx * qz + qw * qy) / QUAD_MASS;
  J(8, 3) = -2.0 * a * qx; J(8, 4) = -2.0 * a * qw; J(8, 5) = 2.0 * a * qz; J(8, 6) = 2.0 * a * qy;
  J(8, 13) = 2.0 * (qy * qz - qw * qx) / QUAD_MASS;
  J(9, 4) = -4.0 * a * qx; J(9, 5) = -4.0 * a * qy;
  J(9, 13) = (1.0 - 2.0 * (qx * qx + qy * qy)) / QUAD_MASS;
  /* body rates */
  J(10, 11) = -(QUAD_IZ - QUAD_IY) * wz / QUAD_IX; J(10, 12) = -(QUAD_IZ - QUAD_IY) * wy / QUAD_IX; J(10, 14) = 1.0 / QUAD_IX;
  J(11, 10) = -(QUAD_IX - QUAD_IZ) * wz / QUAD_IY; J(11, 12) = -(QUAD_IX - QUAD_IZ) * wx / QUAD_IY; J(11, 15) = 1.0 / QUAD_IY;
  J(12, 10) = -(QUAD_IY - QUAD_IX) * wy / QUAD_IZ; J(12, 11) = -(QUAD_IY - QUAD_IX) * wx / QUAD_IZ; J(12, 16) = 1.0 / QUAD_IZ;
#undef J
}

/* ---- generic continuous-model dispatch + midpoint rule (test_utils.cpp:84-132) --- */
typedef struct {
  int kind;            /* ORACLE_MODEL_* */
  int dim;             /* double integrator only */
  oracle_bicycle bike; /* bicycle only */
} oracle_model;

#define ORACLE_MAX_N 16
#define ORACLE_MAX_M 8

static void cont_dyn(const oracle_model* mdl, double* xdot, const double* x, const double* u) {
  if (mdl->kind == ORACLE_MODEL_PENDULUM) oracle_pendulum_dynamics(xdot, x, u);
  else if (mdl->kind == ORACLE_MODEL_QUADROTOR) oracle_quadrotor_dynamics(xdot, x, u);
  else if (mdl->kind == ORACLE_MODEL_QUADROTOR13) oracle_quadrotor13_dynamics(xdot, x, u);
  else oracle_bicycle_dynamics(&mdl->bike, xdot, x, u);
}
static void cont_jac(const oracle_model* mdl, double* jac, const double* x, const double* u) {
  if (mdl->kind == ORACLE_MODEL_PENDULUM) oracle_pendulum_jacobian(jac, x, u);
  else if (mdl->kind == ORACLE_MODEL_QUADROTOR) oracle_quadrotor_jacobian(jac, x, u);
  else if (mdl->kind == ORACLE_MODEL_QUADROTOR13) oracle_quadrotor13_jacobian(jac, x, u);
  else oracle_bicycle_jacobian(&mdl->bike, jac, x, u);
}

void oracle_model_dims(const oracle_model* mdl, int* n, int* m) {
  switch (mdl->kind) {
    case ORACLE_MODEL_DOUBLE_INTEGRATOR: *n = 2 * mdl->dim; *m = mdl->dim; break;
    case ORACLE_MODEL_PENDULUM: *n = 2; *m = 1; break;
    case ORACLE_MODEL_QUADROTOR: *n = 12; *m = 4; break;
    case ORACLE_MODEL_QUADROTOR13: *n = 13; *m = 4; break;
    default: *n = 4; *m = 2; break;
  }
}

/* x+ = x + h f(x + h/2 f(x,u), u)   (test_utils.cpp:85-95; `h / 2` is float arithmetic) */
void oracle_discrete_dynamics(const oracle_model* mdl, double* xn, const double* x,
                              const double* u, float h) {
  if (mdl->kind == ORACLE_MODEL_DOUBLE_INTEGRATOR) {
    oracle_di_dynamics(xn, x, u, h, mdl->dim);
    return;
  }
  int n, m;
  oracle_model_dims(mdl, &n, &m);
  double xm[ORACLE_MAX_N];
  cont_dyn(mdl, xm, x, u);
  for (int i = 0; i < n; ++i) xm[i] *= h / 2;
  for (int i = 0; i < n; ++i) xm[i] += x[i];
  cont_dyn(mdl, xn, xm, u);
  for (int i = 0; i < n; ++i) xn[i] = x[i] + h * xn[i];
}

/* jac (n x (n+m)) col-major:  A = I + h Am (I + h/2 A),  B = h (Am h/2 B + Bm)
 * (test_utils.cpp:113-129)                                                          */
void oracle_discrete_jacobian(const oracle_model* mdl, double* jac, const double* x,
                              const double* u, float h) {
  if (mdl->kind == ORACLE_MODEL_DOUBLE_INTEGRATOR) {
    oracle_di_jacobian(jac, x, u, h, mdl->dim);
    return;
  }
  int n, m;
  oracle_model_dims(mdl, &n, &m);
  double xm[ORACLE_MAX_N];
  double J0[ORACLE_MAX_N * (ORACLE_MAX_N + ORACLE_MAX_M)];
  double Jm[ORACLE_MAX_N * (ORACLE_MAX_N + ORACLE_MAX_M)];
  cont_dyn(mdl, xm, x, u);
  for (int i = 0; i < n; ++i) xm[i] = x[i] + h / 2 * xm[i];
  cont_jac(mdl, J0, x, u);
  cont_jac(mdl, Jm, xm, u);
  const double* A = J0;
  const double* B = J0 + n * n;
  const double* Am = Jm;
  const double* Bm = Jm + n * n;
  /* T = I + h/2 A */
  double T[ORACLE_MAX_N * ORACLE_MAX_N];
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) T[i + j * n] = (i == j ? 1.0 : 0.0) + h / 2 * A[i + j * n];
  /* A_d = I + (h Am) T */
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      double s = 0;
      for (int k = 0; k < n; ++k) s += (h * Am[i + k * n]) * T[k + j * n];
      jac[i + j * n] = (i == j ? 1.0 : 0.0) + s;
    }
  /* B_d = h ((Am h/2) B + Bm) */
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < n; ++i) {
      double s = 0;
      for (int k = 0; k < n; ++k) s += (Am[i + k * n] * (double)(h / 2)) * B[k + j * n];
      jac[i + (n + j) * n] = h * (s + Bm[i + j * n]);
    }
}
