/*
 * oracle/ilqr_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C fp64 restatement of the reference's iLQR inner loop around the TVLQR kernel,
 * with the augmented-Lagrangian / conic terms of al_oracle.c:
 *     /root/reference/src/altro/solver/solver.cpp:116-131   OpenLoopRollout
 *     /root/reference/src/altro/solver/solver.cpp:148-157   CopyTrajectory
 *     /root/reference/src/altro/solver/solver.cpp:189-201   CalcExpansions
 *     /root/reference/src/altro/solver/solver.cpp:207-222   Stationarity
 *     /root/reference/src/altro/solver/solver.cpp:237-271   ForwardPass
 *     /root/reference/src/altro/solver/solver.cpp:273-355   MeritFunction
 *     /root/reference/src/altro/solver/solver.cpp:360-378   BackwardPass
 *     /root/reference/src/altro/solver/solver.cpp:414-511   Solve
 *     /root/reference/src/altro/solver/knotpoint_data.cpp:406-419, 616-719
 *                                                  expansions, quadratic/diagonal cost, dynamics
 *     /root/reference/src/altro/altro_solver.cpp:138-172    SetLQRCost -> diagonal cost
 * Pinned by: solver_impl_test.cpp:236-237,261-262 (merit values), :309-315 (alpha == 1),
 * :151-154 (stationarity), pendulum_test.cpp:110-114 (xN, iterations <= 10),
 * double_integrator_test.cpp:129-132 -- see tests/test_oracle_kat.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* --- pieces from the sibling oracle files --------------------------------------- */
typedef struct { int frame; double length; double lr; } oracle_bicycle;
typedef struct { int kind; int dim; oracle_bicycle bike; } oracle_model;
void oracle_discrete_dynamics(const oracle_model*, double*, const double*, const double*, float);
void oracle_discrete_jacobian(const oracle_model*, double*, const double*, const double*, float);

int oracle_tvlqr_BackwardPass(const int*, const int*, int, const double* const*,
                              const double* const*, const double* const*, const double* const*,
                              const double* const*, const double* const*, const double* const*,
                              const double* const*, double, double**, double**, double**,
                              double**, double*, double**, double**, double**, double**, double**,
                              double**, double**, double**, double**, double**, _Bool, _Bool);

typedef void (*oracle_merit_fn)(double alpha, double* phi, double* dphi, void* ctx);
typedef struct {
  int max_iters;
  double alpha_max, beta_increase, beta_decrease, min_interval_size, c1, c2;
  int try_cubic_first, use_backtracking;
  int status, n_iters, sufficient_decrease, curvature;
  double phi, dphi;
  double phi0, dphi0, phi_lo, phi_hi, dphi_lo, dphi_hi;
} oracle_linesearch;
void oracle_ls_defaults(oracle_linesearch*);
double oracle_ls_run(oracle_linesearch*, oracle_merit_fn, void*, double, double, double);

/* --- AL / conic pieces (al_oracle.c) ------------------------------------------------------------ */
#define ORACLE_MAX_CON 8
#define ORACLE_MAX_P 32
#define ORACLE_MAX_W 64   /* n + m up to 32 + 32 (plan GENERIC's shapes) */
typedef struct {
  int type, p;
  double G[ORACLE_MAX_P * ORACLE_MAX_W];
  double g[ORACLE_MAX_P];
  double val[ORACLE_MAX_P], z[ORACLE_MAX_P], z_est[ORACLE_MAX_P], z_proj[ORACLE_MAX_P], proj_jvp[ORACLE_MAX_P];
  double proj_jac[ORACLE_MAX_P * ORACLE_MAX_P], proj_hess[ORACLE_MAX_P * ORACLE_MAX_P];
  double jac_tmp[ORACLE_MAX_P * ORACLE_MAX_W], hess[ORACLE_MAX_W * ORACLE_MAX_W];
  double rho;
} oracle_con;
typedef struct { int ncon; oracle_con con[ORACLE_MAX_CON]; } oracle_kp_cons;
void oracle_al_constraints(oracle_kp_cons*, int, int, const double*, const double*);
void oracle_al_projected_duals(oracle_kp_cons*);
double oracle_al_cost(oracle_kp_cons*);
void oracle_al_gradient(oracle_kp_cons*, int, int, int, double*, double*);
void oracle_al_hessian(oracle_kp_cons*, int, int, int, double*, double*, double*);
double oracle_al_violation(oracle_kp_cons*);
void oracle_al_dual_update(oracle_kp_cons*);
void oracle_al_penalty_update(oracle_kp_cons*, double, double);

/* --- problem + solver state ------------------------------------------------------ */
enum { ORACLE_DYN_LINEAR = 0, ORACLE_DYN_MODEL = 1 };
enum { ORACLE_COST_QUADRATIC = 0, ORACLE_COST_DIAGONAL = 1 };

typedef struct {
  int N, n, m;
  float h;
  int dyn_kind;
  oracle_model model;
  int cost_kind;
  /* problem data, [k][block]; Q/q/c have N+1 entries, the rest N */
  double *A0, *B0, *aff;           /* linear dynamics data (dyn_kind == LINEAR) */
  double *Qc, *Rc, *Hc, *qc, *rc, *cc; /* cost: Qc n*n per k (diag in head(n)), Rc m*m, Hc m*n */
  double* x0;
  /* per-knot-point state (KnotPointData members, knotpoint_data.hpp:160-233) */
  double *x, *u, *y, *x_, *u_, *y_;
  double *A, *B, *f, *lxx, *luu, *lux, *lx, *lu;
  double *K, *d, *P, *p, *Qblk, *dx_da, *du_da;
  double delta_V[2];
  /* pointer arrays for the tvlqr entry point (solver.cpp:63-106) */
  int *nx, *nu;
  double** ptr;
  /* options (solver_options.hpp:16-39) */
  int iterations_max;
  double tol_stationarity, tol_primal_feasibility, tol_meritfun_gradient;
  int use_backtracking;
  double ls_c1, ls_c2; /* CubicLineSearch::SetOptimalityTolerances (linesearch.cpp), defaults 1e-4 / 0.9 */
  /* results */
  double phi0, dphi0, phi, dphi;
  int ls_iters, iterations, status, backward_status;
  int n_merit_evals;
  double last_alpha, last_stationarity;
  /* augmented Lagrangian (solver_options.hpp:27-29, solver.cpp:383-409) */
  oracle_kp_cons* cons;   /* [N+1] */
  double penalty_initial, penalty_scaling, penalty_max, rho, last_feasibility;
} oracle_ilqr;

static double* dalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

void* oracle_ilqr_create(int N, int n, int m, float h, int dyn_kind, int model_kind,
                         int model_dim, int cost_kind) {
  oracle_ilqr* s = (oracle_ilqr*)calloc(1, sizeof(oracle_ilqr));
  s->N = N; s->n = n; s->m = m; s->h = h;
  s->dyn_kind = dyn_kind;
  s->model.kind = model_kind; s->model.dim = model_dim;
  s->model.bike.frame = 0; s->model.bike.length = 2.7; s->model.bike.lr = 1.5;
  s->cost_kind = cost_kind;
  size_t np = (size_t)N + 1, nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
  s->A0 = dalloc(nn * N); s->B0 = dalloc(nm * N); s->aff = dalloc((size_t)n * N);
  s->Qc = dalloc(nn * np); s->Rc = dalloc(mm * N); s->Hc = dalloc(nm * N);
  s->qc = dalloc((size_t)n * np); s->rc = dalloc((size_t)m * N); s->cc = dalloc(np);
  s->x0 = dalloc(n);
  s->x = dalloc(n * np); s->u = dalloc((size_t)m * np); s->y = dalloc(n * np);
  s->x_ = dalloc(n * np); s->u_ = dalloc((size_t)m * np); s->y_ = dalloc(n * np);
  s->A = dalloc(nn * N); s->B = dalloc(nm * N); s->f = dalloc((size_t)n * N);
  s->lxx = dalloc(nn * np); s->luu = dalloc(mm * N); s->lux = dalloc(nm * N);
  s->lx = dalloc(n * np); s->lu = dalloc((size_t)m * N);
  s->K = dalloc(nm * N); s->d = dalloc((size_t)m * N); s->P = dalloc(nn * np); s->p = dalloc(n * np);
  s->Qblk = dalloc(2 * (nn + mm + nm + n + m) * N);
  s->dx_da = dalloc(n * np); s->du_da = dalloc((size_t)m * np);
  s->nx = (int*)calloc(np, sizeof(int)); s->nu = (int*)calloc(np, sizeof(int));
  for (size_t k = 0; k < np; ++k) { s->nx[k] = n; s->nu[k] = m; }
  s->ptr = (double**)calloc(24 * np, sizeof(double*));
  s->cons = (oracle_kp_cons*)calloc(np, sizeof(oracle_kp_cons));
  s->penalty_initial = 1.0; s->penalty_scaling = 10.0; s->penalty_max = 1e8;
  s->iterations_max = 200;
  s->tol_stationarity = 1e-4;
  s->tol_primal_feasibility = 1e-4;
  s->tol_meritfun_gradient = 1e-8;
  s->use_backtracking = 0;
  s->ls_c1 = 1e-4; s->ls_c2 = 0.9;
  return s;
}

void oracle_ilqr_destroy(void* h) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  if (!s) return;
  double* all[] = {s->A0, s->B0, s->aff, s->Qc, s->Rc, s->Hc, s->qc, s->rc, s->cc, s->x0, s->x,
                   s->u, s->y, s->x_, s->u_, s->y_, s->A, s->B, s->f, s->lxx, s->luu, s->lux,
                   s->lx, s->lu, s->K, s->d, s->P, s->p, s->Qblk, s->dx_da, s->du_da};
  for (size_t i = 0; i < sizeof(all) / sizeof(all[0]); ++i) free(all[i]);
  free(s->nx); free(s->nu); free(s->ptr); free(s->cons); free(s);
}

void oracle_ilqr_set_options(void* h, int iterations_max, double tol_stat, double tol_feas,
                             double tol_merit_grad, int use_backtracking) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  s->iterations_max = iterations_max;
  s->tol_stationarity = tol_stat;
  s->tol_primal_feasibility = tol_feas;
  s->tol_meritfun_gradient = tol_merit_grad;
  s->use_backtracking = use_backtracking;
}

void oracle_ilqr_set_bicycle(void* h, int frame, double length, double lr) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  s->model.bike.frame = frame; s->model.bike.length = length; s->model.bike.lr = lr;
}

/* KnotPointData::SetLinearDynamics (knotpoint_data.cpp:123-142); A,B,f are [k][block] */
void oracle_ilqr_set_linear_dynamics(void* h, const double* A, const double* B, const double* f) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  size_t nn = (size_t)s->n * s->n, nm = (size_t)s->n * s->m;
  memcpy(s->A0, A, sizeof(double) * nn * s->N);
  memcpy(s->B0, B, sizeof(double) * nm * s->N);
  if (f) memcpy(s->aff, f, sizeof(double) * s->n * s->N);
  memcpy(s->A, A, sizeof(double) * nn * s->N); /* A_, B_ ARE the problem data in linear mode */
  memcpy(s->B, B, sizeof(double) * nm * s->N);
}

/* KnotPointData::SetQuadraticCost (:64-85) for one knot point k (k == N: terminal) */
void oracle_ilqr_set_quadratic_cost(void* h, int k, const double* Q, const double* R,
                                    const double* H, const double* q, const double* r, double c) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  const int n = s->n, m = s->m;
  memcpy(s->Qc + (size_t)n * n * k, Q, sizeof(double) * n * n);
  memcpy(s->qc + (size_t)n * k, q, sizeof(double) * n);
  s->cc[k] = c;
  if (k < s->N) {
    memcpy(s->Rc + (size_t)m * m * k, R, sizeof(double) * m * m);
    memcpy(s->Hc + (size_t)m * n * k, H, sizeof(double) * m * n);
    memcpy(s->rc + (size_t)m * k, r, sizeof(double) * m);
  }
}

/* KnotPointData::SetDiagonalCost (:87-110): diagonal kept in the head of Q_/R_ */
void oracle_ilqr_set_diagonal_cost(void* h, int k, const double* Qd, const double* Rd,
                                   const double* q, const double* r, double c) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  const int n = s->n, m = s->m;
  memset(s->Qc + (size_t)n * n * k, 0, sizeof(double) * n * n);
  memcpy(s->Qc + (size_t)n * n * k, Qd, sizeof(double) * n);
  memcpy(s->qc + (size_t)n * k, q, sizeof(double) * n);
  s->cc[k] = c;
  if (k < s->N) {
    memset(s->Rc + (size_t)m * m * k, 0, sizeof(double) * m * m);
    memcpy(s->Rc + (size_t)m * m * k, Rd, sizeof(double) * m);
    memcpy(s->rc + (size_t)m * k, r, sizeof(double) * m);
    memset(s->Hc + (size_t)m * n * k, 0, sizeof(double) * m * n);
  }
}

/* ALTROSolver::SetLQRCost (altro_solver.cpp:138-172): q = -Q xref, r = -R uref,
 * c = 1/2 xref^T Q xref (+ 1/2 uref^T R uref when k < N)                           */
void oracle_ilqr_set_lqr_cost(void* h, int k, const double* Qd, const double* Rd,
                              const double* xref, const double* uref) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  const int n = s->n, m = s->m;
  double q[64], r[64], c = 0.0;
  for (int i = 0; i < n; ++i) { q[i] = -(Qd[i] * xref[i]); c += 0.5 * xref[i] * Qd[i] * xref[i]; }
  if (k < s->N)
    for (int i = 0; i < m; ++i) { r[i] = -(Rd[i] * uref[i]); c += 0.5 * uref[i] * Rd[i] * uref[i]; }
  oracle_ilqr_set_diagonal_cost(h, k, Qd, Rd, q, r, c);
}

void oracle_ilqr_set_initial_state(void* h, const double* x0) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  memcpy(s->x0, x0, sizeof(double) * s->n);
}
/* ALTROSolver::SetState / SetInput write the CANDIDATE trajectory x_, u_ (altro_solver.cpp:231-251) */
void oracle_ilqr_set_state(void* h, int k, const double* x) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  memcpy(s->x_ + (size_t)s->n * k, x, sizeof(double) * s->n);
}
void oracle_ilqr_set_input(void* h, int k, const double* u) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  memcpy(s->u_ + (size_t)s->m * k, u, sizeof(double) * s->m);
}

/* MPC methods: ALTROSolver::UpdateLinearCosts (altro_solver.cpp:266-281 -> knotpoint_data.cpp:193-226) and
 * ALTROSolver::ShiftTrajectory (altro_solver.cpp:283-293, sequential in-place copy of the candidates) */
int oracle_ilqr_update_linear_costs(void* h, int k, const double* q, const double* r, double c) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  if (r && k == s->N) return -1;   /* InvalidOptAtTerminalKnotPoint */
  if (q) memcpy(s->qc + (size_t)s->n * k, q, sizeof(double) * s->n);
  if (r) memcpy(s->rc + (size_t)s->m * k, r, sizeof(double) * s->m);
  s->cc[k] = c;
  return 0;
}
void oracle_ilqr_shift_trajectory(void* h) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  for (int k = 0; k < s->N; ++k) {
    memcpy(s->x_ + (size_t)s->n * k, s->x_ + (size_t)s->n * (k + 1), sizeof(double) * s->n);
    if (k < s->N - 1) memcpy(s->u_ + (size_t)s->m * k, s->u_ + (size_t)s->m * (k + 1), sizeof(double) * s->m);
  }
}

/* --- per-knot-point methods (knotpoint_data.cpp) -------------------------------- */
#define XK(a, k) ((a) + (size_t)s->n * (k))
#define UK(a, k) ((a) + (size_t)s->m * (k))
#define NNK(a, k) ((a) + (size_t)s->n * s->n * (k))
#define NMK(a, k) ((a) + (size_t)s->n * s->m * (k))
#define MMK(a, k) ((a) + (size_t)s->m * s->m * (k))

static void matvec(int rows, int cols, const double* M, const double* v, double* out, int add) {
  for (int i = 0; i < rows; ++i) {
    double acc = 0.0;
    for (int j = 0; j < cols; ++j) acc += M[i + (size_t)j * rows] * v[j];
    out[i] = add ? out[i] + acc : acc;
  }
}
static void matTvec(int rows, int cols, const double* M, const double* v, double* out, int add) {
  for (int j = 0; j < cols; ++j) {
    double acc = 0.0;
    for (int i = 0; i < rows; ++i) acc += M[i + (size_t)j * rows] * v[i];
    out[j] = add ? out[j] + acc : acc;
  }
}

/* CalcDynamics (:710-719): evaluated at the CANDIDATE point (x_, u_) */
static void kp_dynamics(oracle_ilqr* s, int k, double* xnext) {
  const int n = s->n, m = s->m;
  if (s->dyn_kind == ORACLE_DYN_LINEAR) {
    matvec(n, n, NNK(s->A0, k), XK(s->x_, k), xnext, 0);
    matvec(n, m, NMK(s->B0, k), UK(s->u_, k), xnext, 1);
    for (int i = 0; i < n; ++i) xnext[i] += XK(s->aff, k)[i];
  } else {
    oracle_discrete_dynamics(&s->model, xnext, XK(s->x_, k), UK(s->u_, k), s->h);
  }
}

/* CalcDynamicsExpansion (:406-419) */
static void kp_dynamics_expansion(oracle_ilqr* s, int k) {
  const int n = s->n, m = s->m;
  if (s->dyn_kind == ORACLE_DYN_MODEL) {
    double jac[16 * 24];
    oracle_discrete_jacobian(&s->model, jac, XK(s->x_, k), UK(s->u_, k), s->h);
    memcpy(NNK(s->A, k), jac, sizeof(double) * n * n);
    memcpy(NMK(s->B, k), jac + (size_t)n * n, sizeof(double) * n * m);
  } else {
    memset(XK(s->f, k), 0, sizeof(double) * n); /* f_.setZero() (:416) */
  }
}

/* CalcOriginalCost (:616-648) at the candidate point */
static double kp_original_cost(oracle_ilqr* s, int k) {
  const int n = s->n, m = s->m;
  const double *x = XK(s->x_, k), *u = UK(s->u_, k);
  const double* Q = NNK(s->Qc, k);
  double J = 0.0, tmp[64];
  int terminal = (k == s->N);
  if (s->cost_kind == ORACLE_COST_QUADRATIC) {
    matvec(n, n, Q, x, tmp, 0);
    double a = 0; for (int i = 0; i < n; ++i) a += x[i] * tmp[i];
    J = 0.5 * a;
    double b = 0; for (int i = 0; i < n; ++i) b += XK(s->qc, k)[i] * x[i];
    J += b;
    if (!terminal) {
      matvec(m, m, MMK(s->Rc, k), u, tmp, 0);
      a = 0; for (int i = 0; i < m; ++i) a += u[i] * tmp[i];
      J += 0.5 * a;
      b = 0; for (int i = 0; i < m; ++i) b += UK(s->rc, k)[i] * u[i];
      J += b;
      matvec(m, n, NMK(s->Hc, k), x, tmp, 0);
      b = 0; for (int i = 0; i < m; ++i) b += u[i] * tmp[i];
      J += b;
    }
    J += s->cc[k];
  } else {
    double a = 0; for (int i = 0; i < n; ++i) a += x[i] * (Q[i] * x[i]);
    J = 0.5 * a;
    double b = 0; for (int i = 0; i < n; ++i) b += XK(s->qc, k)[i] * x[i];
    J += b;
    if (!terminal) {
      const double* R = MMK(s->Rc, k);
      a = 0; for (int i = 0; i < m; ++i) a += u[i] * (R[i] * u[i]);
      J += 0.5 * a;
      b = 0; for (int i = 0; i < m; ++i) b += UK(s->rc, k)[i] * u[i];
      J += b;
    }
    J += s->cc[k];
  }
  return J;
}

/* CalcOriginalCostGradient (:650-681) at the candidate point */
static void kp_original_cost_gradient(oracle_ilqr* s, int k) {
  const int n = s->n, m = s->m;
  const double *x = XK(s->x_, k), *u = UK(s->u_, k);
  double* lx = XK(s->lx, k);
  int terminal = (k == s->N);
  if (s->cost_kind == ORACLE_COST_QUADRATIC) {
    matvec(n, n, NNK(s->Qc, k), x, lx, 0);
    for (int i = 0; i < n; ++i) lx[i] += XK(s->qc, k)[i];
    if (!terminal) {
      double* lu = UK(s->lu, k);
      matvec(m, m, MMK(s->Rc, k), u, lu, 0);
      for (int i = 0; i < m; ++i) lu[i] += UK(s->rc, k)[i];
      matvec(m, n, NMK(s->Hc, k), x, lu, 1);
      matTvec(m, n, NMK(s->Hc, k), u, lx, 1);
    }
  } else {
    const double* Q = NNK(s->Qc, k);
    for (int i = 0; i < n; ++i) lx[i] = Q[i] * x[i];
    for (int i = 0; i < n; ++i) lx[i] += XK(s->qc, k)[i];
    if (!terminal) {
      double* lu = UK(s->lu, k);
      const double* R = MMK(s->Rc, k);
      for (int i = 0; i < m; ++i) lu[i] = R[i] * u[i];
      for (int i = 0; i < m; ++i) lu[i] += UK(s->rc, k)[i];
    }
  }
}

/* CalcOriginalCostHessian (:683-708) */
static void kp_original_cost_hessian(oracle_ilqr* s, int k) {
  const int n = s->n, m = s->m;
  int terminal = (k == s->N);
  if (s->cost_kind == ORACLE_COST_QUADRATIC) {
    memcpy(NNK(s->lxx, k), NNK(s->Qc, k), sizeof(double) * n * n);
    if (!terminal) {
      memcpy(MMK(s->luu, k), MMK(s->Rc, k), sizeof(double) * m * m);
      memcpy(NMK(s->lux, k), NMK(s->Hc, k), sizeof(double) * m * n);
    }
  } else {
    double* lxx = NNK(s->lxx, k);
    memset(lxx, 0, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i) lxx[i + (size_t)i * n] = NNK(s->Qc, k)[i];
    if (!terminal) {
      double* luu = MMK(s->luu, k);
      memset(luu, 0, sizeof(double) * m * m);
      for (int i = 0; i < m; ++i) luu[i + (size_t)i * m] = MMK(s->Rc, k)[i];
      memset(NMK(s->lux, k), 0, sizeof(double) * m * n);
    }
  }
}

/* CalcConstraints / CalcCost / CalcCostGradient / CalcCostHessian with the AL terms (:421-448) */
static void kp_constraints(oracle_ilqr* s, int k) {
  oracle_al_constraints(&s->cons[k], s->n, s->m, XK(s->x_, k), UK(s->u_, k));
}
static double kp_cost(oracle_ilqr* s, int k) {
  kp_constraints(s, k);
  return kp_original_cost(s, k) + oracle_al_cost(&s->cons[k]);
}
static void kp_cost_gradient(oracle_ilqr* s, int k) {
  kp_original_cost_gradient(s, k);
  oracle_al_gradient(&s->cons[k], s->n, s->m, k == s->N, XK(s->lx, k), k < s->N ? UK(s->lu, k) : NULL);
}
static void kp_cost_hessian(oracle_ilqr* s, int k) {
  kp_original_cost_hessian(s, k);
  if (s->cons[k].ncon)
    oracle_al_hessian(&s->cons[k], s->n, s->m, k == s->N, NNK(s->lxx, k), k < s->N ? MMK(s->luu, k) : NULL,
                      k < s->N ? NMK(s->lux, k) : NULL);
}

/* ALTROSolver::SetConstraint with a linear function c = G [x;u] - g (G is p x (n+m), column-major) */
int oracle_ilqr_add_linear_constraint(void* h, int k, int type, int p, const double* G, const double* g) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  oracle_kp_cons* kc = &s->cons[k];
  if (kc->ncon >= ORACLE_MAX_CON || p > ORACLE_MAX_P || s->n + s->m > ORACLE_MAX_W) return -1;
  oracle_con* c = &kc->con[kc->ncon++];
  memset(c, 0, sizeof(*c));
  c->type = type; c->p = p; c->rho = 1.0;
  memcpy(c->G, G, sizeof(double) * p * (s->n + s->m));
  memcpy(c->g, g, sizeof(double) * p);
  return kc->ncon - 1;
}
void oracle_ilqr_set_penalty_options(void* h, double initial, double scaling, double pmax) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  s->penalty_initial = initial; s->penalty_scaling = scaling; s->penalty_max = pmax;
}
double oracle_ilqr_feasibility(void* h) { /* solver.cpp:224-231 */
  oracle_ilqr* s = (oracle_ilqr*)h;
  double viol = 0;
  for (int k = 0; k <= s->N; ++k) viol = fmax(viol, oracle_al_violation(&s->cons[k]));
  return viol;
}

/* KnotPointData::Initialize tail (:381-396) */
void oracle_ilqr_initialize(void* h) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  for (int k = 0; k <= s->N; ++k) {
    kp_original_cost_hessian(s, k);
    if (k == s->N) memcpy(XK(s->lx, k), XK(s->qc, k), sizeof(double) * s->n);
    if (k < s->N && s->dyn_kind == ORACLE_DYN_LINEAR) {
      memcpy(XK(s->lx, k), XK(s->qc, k), sizeof(double) * s->n);
      memcpy(UK(s->lu, k), UK(s->rc, k), sizeof(double) * s->m);
      memcpy(XK(s->f, k), XK(s->aff, k), sizeof(double) * s->n);
    }
  }
}

/* --- SolverImpl methods ----------------------------------------------------------- */
void oracle_ilqr_open_loop_rollout(void* h) { /* solver.cpp:116-131 */
  oracle_ilqr* s = (oracle_ilqr*)h;
  memcpy(s->x_, s->x0, sizeof(double) * s->n);
  for (int k = 0; k < s->N; ++k) kp_dynamics(s, k, XK(s->x_, k + 1));
}

void oracle_ilqr_copy_trajectory(void* h) { /* solver.cpp:148-157 */
  oracle_ilqr* s = (oracle_ilqr*)h;
  memcpy(s->x, s->x_, sizeof(double) * s->n * (s->N + 1));
  memcpy(s->y, s->y_, sizeof(double) * s->n * (s->N + 1));
  memcpy(s->u, s->u_, sizeof(double) * s->m * s->N);
}

double oracle_ilqr_calc_cost(void* h) { /* solver.cpp:163-174 */
  oracle_ilqr* s = (oracle_ilqr*)h;
  double cost = 0.0;
  for (int k = 0; k <= s->N; ++k) cost += kp_cost(s, k);
  return cost;
}

void oracle_ilqr_calc_cost_gradient(void* h) { /* solver.cpp:176-187 */
  oracle_ilqr* s = (oracle_ilqr*)h;
  for (int k = 0; k <= s->N; ++k) kp_cost_gradient(s, k);
}

void oracle_ilqr_calc_dynamics_expansions(void* h) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  for (int k = 0; k < s->N; ++k) kp_dynamics_expansion(s, k);
}

void oracle_ilqr_calc_expansions(void* h) { /* solver.cpp:189-201 */
  oracle_ilqr* s = (oracle_ilqr*)h;
  for (int k = 0; k <= s->N; ++k) kp_cost_hessian(s, k);
}

int oracle_ilqr_backward_pass(void* h) { /* solver.cpp:360-378: reg = 0, dense */
  oracle_ilqr* s = (oracle_ilqr*)h;
  const int N = s->N, n = s->n, m = s->m;
  const size_t np = (size_t)N + 1;
  double** pp = s->ptr;
  double **A = pp, **B = pp + np, **f = pp + 2 * np, **Q = pp + 3 * np, **R = pp + 4 * np,
         **H = pp + 5 * np, **q = pp + 6 * np, **r = pp + 7 * np, **K = pp + 8 * np,
         **d = pp + 9 * np, **P = pp + 10 * np, **p = pp + 11 * np, **Qxx = pp + 12 * np,
         **Quu = pp + 13 * np, **Qux = pp + 14 * np, **Qx = pp + 15 * np, **Qu = pp + 16 * np,
         **Qxxt = pp + 17 * np, **Quut = pp + 18 * np, **Quxt = pp + 19 * np,
         **Qxt = pp + 20 * np, **Qut = pp + 21 * np;
  const size_t per = (size_t)n * n + m * m + m * n + n + m;
  for (int k = 0; k <= N; ++k) {
    Q[k] = NNK(s->lxx, k); q[k] = XK(s->lx, k); P[k] = NNK(s->P, k); p[k] = XK(s->p, k);
    if (k < N) {
      A[k] = NNK(s->A, k); B[k] = NMK(s->B, k); f[k] = XK(s->f, k);
      R[k] = MMK(s->luu, k); H[k] = NMK(s->lux, k); r[k] = UK(s->lu, k);
      K[k] = NMK(s->K, k); d[k] = UK(s->d, k);
      double* blk = s->Qblk + 2 * per * k;
      Qxx[k] = blk; Quu[k] = Qxx[k] + n * n; Qux[k] = Quu[k] + m * m; Qx[k] = Qux[k] + m * n;
      Qu[k] = Qx[k] + n;
      Qxxt[k] = blk + per; Quut[k] = Qxxt[k] + n * n; Quxt[k] = Quut[k] + m * m;
      Qxt[k] = Quxt[k] + m * n; Qut[k] = Qxt[k] + n;
    }
  }
  int res = oracle_tvlqr_BackwardPass(
      s->nx, s->nu, N, (const double* const*)A, (const double* const*)B,
      (const double* const*)f, (const double* const*)Q, (const double* const*)R,
      (const double* const*)H, (const double* const*)q, (const double* const*)r, 0.0, K, d, P, p,
      s->delta_V, Qxx, Quu, Qux, Qx, Qu, Qxxt, Quut, Quxt, Qxt, Qut, 0, 0);
  s->backward_status = res;
  return res;
}

/* tvlqr_ForwardPass through SolverImpl::LinearRollout (solver.cpp:133-146) */
void oracle_ilqr_linear_rollout(void* h) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  const int N = s->N, n = s->n, m = s->m;
  memcpy(s->x_, s->x0, sizeof(double) * n);
  for (int k = 0; k < N; ++k) {
    double* u = UK(s->u_, k);
    memcpy(u, UK(s->d, k), sizeof(double) * m);
    for (int i = 0; i < m; ++i) {
      double acc = 0;
      for (int j = 0; j < n; ++j) acc += NMK(s->K, k)[i + (size_t)j * m] * XK(s->x_, k)[j];
      u[i] -= acc;
    }
    double* xn = XK(s->x_, k + 1);
    memcpy(xn, XK(s->f, k), sizeof(double) * n);
    matvec(n, n, NNK(s->A, k), XK(s->x_, k), xn, 1);
    matvec(n, m, NMK(s->B, k), u, xn, 1);
    matvec(n, n, NNK(s->P, k), XK(s->x_, k), XK(s->y_, k), 0);
    for (int i = 0; i < n; ++i) XK(s->y_, k)[i] += XK(s->p, k)[i];
  }
  matvec(n, n, NNK(s->P, N), XK(s->x_, N), XK(s->y_, N), 0);
  for (int i = 0; i < n; ++i) XK(s->y_, N)[i] += XK(s->p, N)[i];
}

double oracle_ilqr_stationarity(void* h) { /* solver.cpp:207-222 */
  oracle_ilqr* s = (oracle_ilqr*)h;
  const int N = s->N, n = s->n, m = s->m;
  double res_x = 0, res_u = 0, tmp[64];
  for (int k = 0; k < N; ++k) {
    matTvec(n, n, NNK(s->A, k), XK(s->y_, k + 1), tmp, 0);
    for (int i = 0; i < n; ++i)
      res_x = fmax(res_x, fabs(XK(s->lx, k)[i] + tmp[i] - XK(s->y_, k)[i]));
    matTvec(n, m, NMK(s->B, k), XK(s->y_, k + 1), tmp, 0);
    for (int i = 0; i < m; ++i) res_u = fmax(res_u, fabs(UK(s->lu, k)[i] + tmp[i]));
  }
  for (int i = 0; i < n; ++i) res_x = fmax(res_x, fabs(XK(s->lx, N)[i] - XK(s->y_, N)[i]));
  return fmax(res_x, res_u);
}

/* MeritFunction (solver.cpp:273-355); dphi may be NULL */
void oracle_ilqr_merit(void* h, double alpha, double* phi, double* dphi) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  const int N = s->N, n = s->n, m = s->m;
  const int calc_derivative = dphi != NULL;
  double phi_ = 0, dphi_ = 0, dx[64], tmp[64];
  s->n_merit_evals += 1;
  memcpy(s->x_, s->x0, sizeof(double) * n);
  memset(s->dx_da, 0, sizeof(double) * n);
  for (int k = 0; k < N; ++k) {
    double *xk_ = XK(s->x_, k), *uk_ = UK(s->u_, k);
    for (int i = 0; i < n; ++i) dx[i] = xk_[i] - XK(s->x, k)[i];
    /* du = -K dx + alpha d ; u_ = u + du */
    matvec(m, n, NMK(s->K, k), dx, tmp, 0);
    for (int i = 0; i < m; ++i) uk_[i] = UK(s->u, k)[i] + (-tmp[i] + alpha * UK(s->d, k)[i]);
    /* y_ = P dx + p */
    matvec(n, n, NNK(s->P, k), dx, XK(s->y_, k), 0);
    for (int i = 0; i < n; ++i) XK(s->y_, k)[i] += XK(s->p, k)[i];
    kp_dynamics(s, k, XK(s->x_, k + 1));
    phi_ += kp_cost(s, k);
    if (calc_derivative) {
      kp_dynamics_expansion(s, k);
      double* du_da = UK(s->du_da, k);
      matvec(m, n, NMK(s->K, k), XK(s->dx_da, k), tmp, 0);
      for (int i = 0; i < m; ++i) du_da[i] = -tmp[i] + UK(s->d, k)[i];
      double* dxn = XK(s->dx_da, k + 1);
      matvec(n, n, NNK(s->A, k), XK(s->dx_da, k), dxn, 0);
      matvec(n, m, NMK(s->B, k), du_da, dxn, 1);
      kp_cost_gradient(s, k);
      double a = 0; for (int i = 0; i < n; ++i) a += XK(s->lx, k)[i] * XK(s->dx_da, k)[i];
      dphi_ += a;
      a = 0; for (int i = 0; i < m; ++i) a += UK(s->lu, k)[i] * du_da[i];
      dphi_ += a;
    }
  }
  phi_ += kp_cost(s, N);
  for (int i = 0; i < n; ++i) dx[i] = XK(s->x_, N)[i] - XK(s->x, N)[i];
  matvec(n, n, NNK(s->P, N), dx, XK(s->y_, N), 0);
  for (int i = 0; i < n; ++i) XK(s->y_, N)[i] += XK(s->p, N)[i];
  *phi = phi_;
  if (calc_derivative) {
    kp_cost_gradient(s, N);
    double a = 0; for (int i = 0; i < n; ++i) a += XK(s->lx, N)[i] * XK(s->dx_da, N)[i];
    dphi_ += a;
    *dphi = dphi_;
  }
  s->phi = phi_;
  s->dphi = dphi_;
}

static void merit_cb(double alpha, double* phi, double* dphi, void* ctx) {
  oracle_ilqr_merit(ctx, alpha, phi, dphi);
}

/* ForwardPass (solver.cpp:237-271).  Returns 0 ok, 1 merit gradient too small, 2 failed. */
int oracle_ilqr_forward_pass(void* h, double* alpha) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  oracle_ilqr_merit(h, 0.0, &s->phi0, &s->dphi0);
  if (fabs(s->dphi0) < s->tol_meritfun_gradient) {
    *alpha = 0.0;
    return 1;
  }
  oracle_linesearch ls;
  oracle_ls_defaults(&ls);
  ls.try_cubic_first = 1;
  ls.use_backtracking = s->use_backtracking;
  ls.c1 = s->ls_c1; ls.c2 = s->ls_c2;
  *alpha = oracle_ls_run(&ls, merit_cb, h, 1.0, s->phi0, s->dphi0);
  s->phi = ls.phi;
  s->dphi = ls.dphi;
  s->ls_iters = ls.n_iters;
  if (s->use_backtracking && fabs(*alpha - 1.0) > 0) {
    for (int k = 0; k <= s->N; ++k) {
      if (k < s->N) kp_dynamics_expansion(s, k);
      kp_cost_gradient(s, k);
    }
  }
  if (isnan(*alpha) || !(ls.status == 1 /*MINIMUM_FOUND*/ || ls.status == 7 /*HIT_MAX*/)) return 2;
  return 0;
}

/* Solve (solver.cpp:414-511), unconstrained.  status: 0 Success, 1 Unsolved, 2 MaxIterations.
 * log (optional): per iteration [alpha, phi0, phi, dphi0, stationarity, ls_iters].       */
int oracle_ilqr_solve(void* h, double* log, int log_cap) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  oracle_ilqr_open_loop_rollout(h);
  oracle_ilqr_copy_trajectory(h);
  (void)oracle_ilqr_calc_cost(h);
  s->rho = s->penalty_initial;
  for (int k = 0; k <= s->N; ++k) {
    if (k < s->N) kp_dynamics_expansion(s, k);
    kp_cost_gradient(s, k);
    for (int j = 0; j < s->cons[k].ncon; ++j) s->cons[k].con[j].rho = s->penalty_initial;   /* SetPenalty (:429) */
  }
  int is_converged = 0, stop = 0, iter;
  s->status = 1;
  s->n_merit_evals = 0;
  double alpha = 0;
  for (iter = 0; iter < s->iterations_max; ++iter) {
    oracle_ilqr_calc_expansions(h);
    oracle_ilqr_backward_pass(h); /* return value ignored, as solver.cpp:449 does */
    int err = oracle_ilqr_forward_pass(h, &alpha);
    if (err == 2) stop = 1;
    double stationarity = oracle_ilqr_stationarity(h);
    double feasibility = oracle_ilqr_feasibility(h);
    s->last_feasibility = feasibility;
    oracle_ilqr_copy_trajectory(h);
    if (log && iter < log_cap) {
      double* L = log + 8 * (size_t)iter;
      L[0] = alpha; L[1] = s->phi0; L[2] = s->phi; L[3] = s->dphi0; L[4] = stationarity;
      L[5] = s->ls_iters;
      L[6] = feasibility; L[7] = s->rho;
    }
    s->last_alpha = alpha;
    s->last_stationarity = stationarity;
    if (fabs(stationarity) < s->tol_stationarity && feasibility < s->tol_primal_feasibility) {
      is_converged = 1;
      stop = 1;
      s->status = 0;
    }
    if (stationarity < sqrt(s->tol_stationarity)) {
      /* DualUpdate, PenaltyUpdate (only while infeasible), then refresh projected duals + gradients
       * from the constraint values cached by the forward pass (solver.cpp:474-489) */
      for (int k = 0; k <= s->N; ++k) oracle_al_dual_update(&s->cons[k]);
      if (feasibility > s->tol_primal_feasibility) {
        for (int k = 0; k <= s->N; ++k) oracle_al_penalty_update(&s->cons[k], s->penalty_scaling, s->penalty_max);
        s->rho = fmin(s->rho * s->penalty_scaling, s->penalty_max);
      }
      for (int k = 0; k <= s->N; ++k) {
        oracle_al_projected_duals(&s->cons[k]);
        kp_cost_gradient(s, k);
      }
    }
    if (stop) break;
  }
  if (!is_converged && iter == s->iterations_max) s->status = 2;
  s->iterations = iter + 1;
  return s->status;
}

/* --- getters ------------------------------------------------------------------------ */
#define GETTER(name, field, count)                                  \
  void oracle_ilqr_get_##name(void* h, double* out) {              \
    oracle_ilqr* s = (oracle_ilqr*)h;                               \
    memcpy(out, s->field, sizeof(double) * (count));                \
  }
GETTER(x, x, (size_t)s->n * (s->N + 1))
GETTER(u, u, (size_t)s->m * s->N)
GETTER(y, y, (size_t)s->n * (s->N + 1))
GETTER(x_cand, x_, (size_t)s->n * (s->N + 1))
GETTER(u_cand, u_, (size_t)s->m * s->N)
GETTER(y_cand, y_, (size_t)s->n * (s->N + 1))
GETTER(K, K, (size_t)s->n * s->m * s->N)
GETTER(d, d, (size_t)s->m * s->N)
GETTER(P, P, (size_t)s->n * s->n * (s->N + 1))
GETTER(p, p, (size_t)s->n * (s->N + 1))
GETTER(A, A, (size_t)s->n * s->n * s->N)
GETTER(B, B, (size_t)s->n * s->m * s->N)
GETTER(lx, lx, (size_t)s->n * (s->N + 1))
GETTER(lu, lu, (size_t)s->m * s->N)
GETTER(lxx, lxx, (size_t)s->n * s->n * (s->N + 1))
GETTER(luu, luu, (size_t)s->m * s->m * s->N)
GETTER(lux, lux, (size_t)s->n * s->m * s->N)
/* --- the pieces solver/test/alilqr_test.cpp:112-215 sequences by hand ------------- */
/* solver.ls_.SetOptimalityTolerances(c1, c2) (the solver's own line search object persists between passes) */
void oracle_ilqr_set_linesearch_tolerances(void* h, double c1, double c2) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  s->ls_c1 = c1; s->ls_c2 = c2;
}
void oracle_ilqr_dual_update(void* h) { /* solver.cpp:383-388 */
  oracle_ilqr* s = (oracle_ilqr*)h;
  for (int k = 0; k <= s->N; ++k) oracle_al_dual_update(&s->cons[k]);
}
void oracle_ilqr_penalty_update(void* h) { /* solver.cpp:390-395 */
  oracle_ilqr* s = (oracle_ilqr*)h;
  for (int k = 0; k <= s->N; ++k) oracle_al_penalty_update(&s->cons[k], s->penalty_scaling, s->penalty_max);
  s->rho = fmin(s->rho * s->penalty_scaling, s->penalty_max);
}
/* the per-knot-point refresh the test runs before each block of iterations (alilqr_test.cpp:126-133):
 * CalcDynamicsExpansion, CalcConstraints, CalcConstraintJacobians, CalcProjectedDuals, CalcConicJacobians,
 * CalcCostGradient -- on the candidate trajectory (x_, u_), which equals the nominal one after CopyTrajectory */
void oracle_ilqr_refresh_expansions(void* h) {
  oracle_ilqr* s = (oracle_ilqr*)h;
  for (int k = 0; k <= s->N; ++k) {
    if (k < s->N) kp_dynamics_expansion(s, k);
    kp_constraints(s, k);
    oracle_al_projected_duals(&s->cons[k]);
    kp_cost_gradient(s, k);
  }
}

int oracle_ilqr_iterations(void* h) { return ((oracle_ilqr*)h)->iterations; }
int oracle_ilqr_merit_evals(void* h) { return ((oracle_ilqr*)h)->n_merit_evals; }
double oracle_ilqr_delta_V(void* h, int i) { return ((oracle_ilqr*)h)->delta_V[i]; }
