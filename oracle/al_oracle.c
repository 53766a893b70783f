/*
 * oracle/al_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C restatement of the reference's augmented-Lagrangian / conic machinery:
 *     /root/reference/src/altro/solver/cones.cpp:13-38    second-order-cone projection
 *     /root/reference/src/altro/solver/cones.cpp:40-77    its Jacobian
 *     /root/reference/src/altro/solver/cones.cpp:79-123   its Hessian-vector term d/dx [J(x)^T b]
 *     /root/reference/src/altro/solver/cones.cpp:125-202  dispatch per cone (zero / identity / orthant / SOC)
 *     /root/reference/src/altro/solver/cones.hpp:13-56    dual cones, linearity
 *     /root/reference/src/altro/solver/knotpoint_data.cpp:489-613  violations, projected duals, conic
 *                                                        Jacobians / Hessians, AL cost, gradient, Hessian
 * Constraints are LINEAR functions c(x,u) = G [x;u] - g (every constraint in the reference's tests is
 * of this form); G is (p x (n+m)) column-major.  Pinned, through oracle_ilqr_solve, by the reference's
 * end-to-end iteration counts 3 / 5 / 9 (test/double_integrator_test.cpp:255,374,491).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { ORACLE_CONE_EQUALITY = 0, ORACLE_CONE_IDENTITY = 1, ORACLE_CONE_INEQUALITY = 2, ORACLE_CONE_SOC = 3 };
#define ORACLE_MAX_CON 8
#define ORACLE_MAX_P 32
#define ORACLE_MAX_W 64   /* n + m up to 32 + 32 (plan GENERIC's shapes) */

typedef struct {
  int type, p;
  double G[ORACLE_MAX_P * ORACLE_MAX_W];   /* p x w column-major */
  double g[ORACLE_MAX_P];
  double val[ORACLE_MAX_P], z[ORACLE_MAX_P], z_est[ORACLE_MAX_P], z_proj[ORACLE_MAX_P], proj_jvp[ORACLE_MAX_P];
  double proj_jac[ORACLE_MAX_P * ORACLE_MAX_P], proj_hess[ORACLE_MAX_P * ORACLE_MAX_P];
  double jac_tmp[ORACLE_MAX_P * ORACLE_MAX_W], hess[ORACLE_MAX_W * ORACLE_MAX_W];
  double rho;
} oracle_con;

typedef struct {
  int ncon;
  oracle_con con[ORACLE_MAX_CON];
} oracle_kp_cons;

static int dual_cone(int cone) { /* cones.hpp:13-30 */
  if (cone == ORACLE_CONE_EQUALITY) return ORACLE_CONE_IDENTITY;
  if (cone == ORACLE_CONE_IDENTITY) return ORACLE_CONE_EQUALITY;
  return cone;
}

void oracle_cone_projection(int cone, int dim, const double* x, double* px) { /* cones.cpp:13-38,125-151 */
  switch (cone) {
    case ORACLE_CONE_EQUALITY: for (int i = 0; i < dim; ++i) px[i] = 0; break;
    case ORACLE_CONE_IDENTITY: for (int i = 0; i < dim; ++i) px[i] = x[i]; break;
    case ORACLE_CONE_INEQUALITY: for (int i = 0; i < dim; ++i) px[i] = fmin(0.0, x[i]); break;
    default: {
      int n = dim - 1;
      double s = x[n], a = 0.0;
      for (int i = 0; i < n; ++i) a += x[i] * x[i];
      a = sqrt(a);
      if (a <= -s) { for (int i = 0; i < dim; ++i) px[i] = 0.0; }
      else if (a <= s) { for (int i = 0; i < dim; ++i) px[i] = x[i]; }
      else {
        double c = 0.5 * (1 + s / a);
        for (int i = 0; i < n; ++i) px[i] = c * x[i];
        px[n] = c * a;
      }
    }
  }
}

void oracle_cone_jacobian(int cone, int dim, const double* x, double* J) { /* cones.cpp:40-77,153-178 */
  memset(J, 0, sizeof(double) * dim * dim);
  switch (cone) {
    case ORACLE_CONE_EQUALITY: break;
    case ORACLE_CONE_IDENTITY: for (int i = 0; i < dim; ++i) J[i + i * dim] = 1.0; break;
    case ORACLE_CONE_INEQUALITY: for (int i = 0; i < dim; ++i) J[i + i * dim] = (x[i] <= 0) ? 1 : 0; break;
    default: {
      int n = dim - 1;
      double s = x[n], a = 0.0;
      for (int i = 0; i < n; ++i) a += x[i] * x[i];
      a = sqrt(a);
      if (a <= -s) break;
      if (a <= s) { for (int i = 0; i < dim; ++i) J[i + i * dim] = 1.0; break; }
      double c = 0.5 * (1 + s / a);
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
          J[i + j * dim] = -0.5 * s / (a * a * a) * x[i] * x[j];
          J[i + j * dim] += (i == j) ? c : 0;
        }
      for (int i = 0; i < n; ++i) J[i + n * dim] = 0.5 * x[i] / a;
      for (int j = 0; j < n; ++j) J[n + j * dim] = ((-0.5 * s / (a * a)) + c / a) * x[j];
      J[n + n * dim] = 0.5;
    }
  }
}

void oracle_cone_hessian(int cone, int dim, const double* x, const double* b, double* H) { /* :79-123,180-202 */
  memset(H, 0, sizeof(double) * dim * dim);
  if (cone != ORACLE_CONE_SOC) return;
  int n = dim - 1;
  double s = x[n], bs = b[n], vbv = 0, a = 0;
  for (int i = 0; i < n; ++i) { a += x[i] * x[i]; vbv += x[i] * b[i]; }
  a = sqrt(a);
  if (a <= -s || a <= s) return;
  for (int i = 0; i < n; ++i) {
    double hi = 0;
    for (int j = 0; j < n; ++j) {
      double Hij = -x[i] * x[j] / (a * a);
      Hij += (i == j) ? 1 : 0;
      hi += Hij * b[j];
    }
    H[i + n * dim] = hi / (2 * a);
    H[n + i * dim] = hi / (2 * a);
    for (int j = 0; j <= i; ++j) {
      double vij = x[i] * x[j];
      double H1 = hi * x[j] * (-s / (a * a * a));
      double H2 = vij * (2 * vbv) / (a * a * a * a) - x[i] * b[j] / (a * a);
      double H3 = -vij / (a * a);
      if (i == j) { H2 -= vbv / (a * a); H3 += 1; }
      H2 *= s / a;
      H3 *= bs / a;
      H[i + j * dim] = (H1 + H2 + H3) / 2.0;
      H[j + i * dim] = (H1 + H2 + H3) / 2.0;
    }
  }
}

/* ---- per-knot-point AL terms (knotpoint_data.cpp:473-613); w = n + m ------------------------------ */
void oracle_al_constraints(oracle_kp_cons* kc, int n, int m, const double* x, const double* u) {
  const int w = n + m;
  for (int j = 0; j < kc->ncon; ++j) {
    oracle_con* c = &kc->con[j];
    for (int i = 0; i < c->p; ++i) {
      double s = 0.0;
      for (int e = 0; e < n; ++e) s += c->G[i + e * c->p] * x[e];
      for (int e = 0; e < m; ++e) s += c->G[i + (n + e) * c->p] * u[e];
      c->val[i] = s - c->g[i];
    }
    (void)w;
  }
}
void oracle_al_projected_duals(oracle_kp_cons* kc) { /* :523-535 */
  for (int j = 0; j < kc->ncon; ++j) {
    oracle_con* c = &kc->con[j];
    for (int i = 0; i < c->p; ++i) c->z_est[i] = c->z[i] - c->rho * c->val[i];
    oracle_cone_projection(dual_cone(c->type), c->p, c->z_est, c->z_proj);
  }
}
double oracle_al_cost(oracle_kp_cons* kc) { /* :572-581 */
  double cost = 0;
  oracle_al_projected_duals(kc);
  for (int j = 0; j < kc->ncon; ++j) {
    oracle_con* c = &kc->con[j];
    double s = 0;
    for (int i = 0; i < c->p; ++i) s += c->z_proj[i] * c->z_proj[i];
    cost += s / (2 * c->rho);
  }
  return cost;
}
void oracle_al_gradient(oracle_kp_cons* kc, int n, int m, int terminal, double* lx, double* lu) { /* :537-547,583-595 */
  for (int j = 0; j < kc->ncon; ++j) {
    oracle_con* c = &kc->con[j];
    oracle_cone_jacobian(dual_cone(c->type), c->p, c->z_est, c->proj_jac);
    for (int i = 0; i < c->p; ++i) {
      double s = 0;
      for (int k = 0; k < c->p; ++k) s += c->proj_jac[k + i * c->p] * c->z_proj[k];
      c->proj_jvp[i] = s;
    }
    for (int e = 0; e < n; ++e) {
      double s = 0;
      for (int i = 0; i < c->p; ++i) s += c->G[i + e * c->p] * c->proj_jvp[i];
      lx[e] -= s;
    }
    if (!terminal)
      for (int e = 0; e < m; ++e) {
        double s = 0;
        for (int i = 0; i < c->p; ++i) s += c->G[i + (n + e) * c->p] * c->proj_jvp[i];
        lu[e] -= s;
      }
  }
}
void oracle_al_hessian(oracle_kp_cons* kc, int n, int m, int terminal, double* lxx, double* luu, double* lux) { /* :549-570,597-613 */
  const int w = n + m;
  for (int j = 0; j < kc->ncon; ++j) {
    oracle_con* c = &kc->con[j];
    const int p = c->p;
    for (int e = 0; e < w; ++e)
      for (int i = 0; i < p; ++i) {
        double s = 0;
        for (int k = 0; k < p; ++k) s += c->proj_jac[i + k * p] * c->G[k + e * p];
        c->jac_tmp[i + e * p] = s;
      }
    for (int b = 0; b < w; ++b)
      for (int a = 0; a < w; ++a) {
        double s = 0;
        for (int k = 0; k < p; ++k) s += c->jac_tmp[k + a * p] * c->jac_tmp[k + b * p];
        c->hess[a + b * w] = c->rho * s;
      }
    if (dual_cone(c->type) == ORACLE_CONE_SOC) {
      oracle_cone_hessian(ORACLE_CONE_SOC, p, c->z_est, c->z_proj, c->proj_hess);
      for (int e = 0; e < w; ++e)
        for (int i = 0; i < p; ++i) {
          double s = 0;
          for (int k = 0; k < p; ++k) s += c->proj_hess[i + k * p] * c->G[k + e * p];
          c->jac_tmp[i + e * p] = s;
        }
      for (int b = 0; b < w; ++b)
        for (int a = 0; a < w; ++a) {
          double s = 0;
          for (int k = 0; k < p; ++k) s += c->G[k + a * p] * c->jac_tmp[k + b * p];
          c->hess[a + b * w] += c->rho * s;
        }
    }
    for (int b = 0; b < n; ++b)
      for (int a = 0; a < n; ++a) lxx[a + b * n] += c->hess[a + b * w];
    if (!terminal) {
      for (int b = 0; b < m; ++b)
        for (int a = 0; a < m; ++a) luu[a + b * m] += c->hess[(n + a) + (n + b) * w];
      for (int b = 0; b < n; ++b)
        for (int a = 0; a < m; ++a) lux[a + b * m] += c->hess[(n + a) + b * w];
    }
  }
}
double oracle_al_violation(oracle_kp_cons* kc) { /* :489-501 */
  double viol = 0;
  for (int j = 0; j < kc->ncon; ++j) {
    oracle_con* c = &kc->con[j];
    double v[ORACLE_MAX_P];
    oracle_cone_projection(c->type, c->p, c->val, v);
    for (int i = 0; i < c->p; ++i) viol = fmax(viol, fabs(v[i] - c->val[i]));
  }
  return viol;
}
void oracle_al_dual_update(oracle_kp_cons* kc) { /* :503-510 */
  for (int j = 0; j < kc->ncon; ++j) memcpy(kc->con[j].z, kc->con[j].z_proj, sizeof(double) * kc->con[j].p);
}
void oracle_al_penalty_update(oracle_kp_cons* kc, double scaling, double pmax) { /* :512-517 */
  for (int j = 0; j < kc->ncon; ++j) kc->con[j].rho = fmin(kc->con[j].rho * scaling, pmax);
}

/* One knot point's AL terms from given (x, u, z, rho): the sequence CalcConstraints -> CalcConstraintCosts
 * -> CalcConstraintCostGradients -> CalcConstraintCostHessians of knotpoint_data_test.cpp:233-524, used to
 * pin this file against that test's constants.  lx/lu/lxx/luu/lux start from zero; hess is the (n+m)^2
 * constraint_hess_ block. */
double oracle_al_knot_eval(int type, int p, int n, int m, const double* G, const double* g, const double* x,
                           const double* u, const double* z, double rho, double* lx, double* lu, double* lxx,
                           double* luu, double* lux, double* hess, double* val) {
  oracle_kp_cons kc;
  memset(&kc, 0, sizeof(kc));
  kc.ncon = 1;
  oracle_con* c = &kc.con[0];
  c->type = type; c->p = p; c->rho = rho;
  memcpy(c->G, G, sizeof(double) * p * (n + m));
  memcpy(c->g, g, sizeof(double) * p);
  memcpy(c->z, z, sizeof(double) * p);
  oracle_al_constraints(&kc, n, m, x, u);
  double cost = oracle_al_cost(&kc);
  memset(lx, 0, sizeof(double) * n); memset(lu, 0, sizeof(double) * m);
  memset(lxx, 0, sizeof(double) * n * n); memset(luu, 0, sizeof(double) * m * m); memset(lux, 0, sizeof(double) * m * n);
  oracle_al_gradient(&kc, n, m, 0, lx, lu);
  oracle_al_hessian(&kc, n, m, 0, lxx, luu, lux);
  memcpy(hess, c->hess, sizeof(double) * (n + m) * (n + m));
  memcpy(val, c->val, sizeof(double) * p);
  return cost;
}
