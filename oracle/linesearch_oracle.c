/*
 * oracle/linesearch_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C restatement of the reference's scalar strong-Wolfe cubic line search:
 *     /root/reference/src/linesearch/cubicspline.c:18-42,111-181,229-246  (spline + argmin)
 *     /root/reference/src/linesearch/linesearch.cpp:37-217   (CubicLineSearch::Run)
 *     /root/reference/src/linesearch/linesearch.cpp:233-351  (Zoom)
 *     /root/reference/src/linesearch/linesearch.cpp:385-412  (SimpleBacktracking)
 * Unlike the Eigen-based files, src/linesearch/ DOES compile on this image, so this
 * restatement is pinned against the REAL reference code: tests/test_oracle_linesearch.py
 * runs both on the same merit functions through oracle/_ref/liblinesearch_ref.so
 * (built by `make -C oracle ref` from the reference sources where they lie).
 */
#include <math.h>
#include <stddef.h>

#define LS_TOL 1e-6

enum {
  OLS_NOERROR = 0,
  OLS_MINIMUM_FOUND,
  OLS_INVALID_POINTER,
  OLS_NOT_DESCENT_DIRECTION,
  OLS_WINDOW_TOO_SMALL,
  OLS_GOT_NONFINITE_STEP_SIZE,
  OLS_MAX_ITERATIONS,
  OLS_HIT_MAX_STEPSIZE
};

typedef void (*oracle_merit_fn)(double alpha, double* phi, double* dphi, void* ctx);

typedef struct {
  /* options (linesearch.hpp:41-47, :55-56) */
  int max_iters;
  double alpha_max, beta_increase, beta_decrease, min_interval_size, c1, c2;
  int try_cubic_first, use_backtracking;
  /* results */
  int status, n_iters, sufficient_decrease, curvature;
  double phi, dphi;
  /* internal */
  double phi0, dphi0, phi_lo, phi_hi, dphi_lo, dphi_hi;
} oracle_linesearch;

void oracle_ls_defaults(oracle_linesearch* ls) {
  ls->max_iters = 25;
  ls->alpha_max = 2.0;
  ls->beta_increase = 1.5;
  ls->beta_decrease = 0.5;
  ls->min_interval_size = 1e-6;
  ls->c1 = 1e-4;
  ls->c2 = 0.9;
  ls->try_cubic_first = 0;
  ls->use_backtracking = 0;
  ls->status = OLS_NOERROR;
  ls->n_iters = 0;
  ls->sufficient_decrease = 0;
  ls->curvature = 0;
  ls->phi = ls->dphi = 0;
}

/* cubicspline.c:18-42.  Returns 0 if the two abscissae coincide. */
static int spline_from_2pts(double x1, double y1, double d1, double x2, double y2, double d2,
                            double* x0, double* b, double* c, double* d) {
  double delta = x2 - x1;
  if (fabs(delta) < LS_TOL) return 0;
  *x0 = x1;
  *b = d1;
  *c = 3 * (y2 - y1) / (delta * delta) - (d2 + 2 * d1) / delta;
  *d = (d2 + d1) / (delta * delta) - 2 * (y2 - y1) / (delta * delta * delta);
  return 1;
}

/* cubicspline.c:111-181 + :229-246.  Returns 1 and *xmin when a minimiser exists. */
static int spline_argmin(double x0, double b, double c, double d, double* xmin) {
  int is_quadratic = fabs(d) < LS_TOL;
  int is_linear = is_quadratic && (fabs(c) < LS_TOL);
  if (is_quadratic) {
    if (is_linear) return 0;
    if (c <= 0) return 0;
    *xmin = -b / (2 * c) + x0;
    return 1;
  }
  /* stationary points: roots of 3d t^2 + 2c t + b */
  double qa = 3 * d, qb = 2 * c, qc = b;
  if (fabs(qa) < LS_TOL) return 0;
  double s2 = qb * qb - 4 * qa * qc, s;
  if (fabs(s2) < LS_TOL) s = 0.0;
  else if (s2 < 0) return 0;
  else s = sqrt(s2);
  double d1 = (-qb + s) / (2 * qa);
  double d2 = (-qb - s) / (2 * qa);
  double curv1 = 2 * c + 6 * d * d1;
  double curv2 = 2 * c + 6 * d * d2;
  if (fabs(curv1) < LS_TOL && fabs(curv2) < LS_TOL) return 0;
  if (curv1 > 0 && curv2 < 0) { *xmin = d1 + x0; return 1; }
  if (curv1 < 0 && curv2 > 0) { *xmin = d2 + x0; return 1; }
  return 0;
}

/* Exposed for the spline unit tests (linesearch_tests.cpp:12-131). */
int oracle_cubic_argmin_2pts(double x1, double y1, double d1, double x2, double y2, double d2,
                             double* xmin) {
  double x0, b, c, d;
  if (!spline_from_2pts(x1, y1, d1, x2, y2, d2, &x0, &b, &c, &d)) return 0;
  return spline_argmin(x0, b, c, d, xmin);
}

/* linesearch.cpp:233-351 */
static double zoom(oracle_linesearch* ls, oracle_merit_fn f, void* ctx, double alo, double ahi) {
  double alpha = alo;
  if (!isfinite(alo) || !isfinite(ahi)) {
    ls->status = OLS_GOT_NONFINITE_STEP_SIZE;
    return 0;
  }
  const double c1 = ls->c1, c2 = ls->c2, phi0 = ls->phi0, dphi0 = ls->dphi0;
  double phi_lo = ls->phi_lo, phi_hi = ls->phi_hi, dphi_lo = ls->dphi_lo, dphi_hi = ls->dphi_hi;

  for (int zoom_iter = ls->n_iters + 1; zoom_iter < ls->max_iters; ++zoom_iter) {
    if (fabs(alo - ahi) < ls->min_interval_size) {
      alpha = (alo + ahi) / 2.0;
      ls->n_iters += 1;
      f(alpha, &ls->phi, &ls->dphi, ctx);
      ls->sufficient_decrease = ls->phi <= phi0 + c1 * alpha * dphi0;
      ls->curvature = fabs(ls->dphi) <= -c2 * dphi0;
      ls->status = (ls->sufficient_decrease && ls->curvature) ? OLS_MINIMUM_FOUND
                                                               : OLS_WINDOW_TOO_SMALL;
      return alpha;
    }
    double x0, b, c, d;
    int failed = 1;
    if (spline_from_2pts(alo, phi_lo, dphi_lo, ahi, phi_hi, dphi_hi, &x0, &b, &c, &d)) {
      double a;
      if (spline_argmin(x0, b, c, d, &a)) {
        alpha = a; /* the reference assigns alpha before the finiteness check */
        if (isfinite(a)) failed = 0;
      } else {
        alpha = NAN; /* CubicSpline_ArgMin returned NAN into alpha */
      }
    }
    if (failed) alpha = (alo + ahi) / 2;

    ls->n_iters += 1;
    f(alpha, &ls->phi, &ls->dphi, ctx);
    int sufficient_decrease = ls->phi <= phi0 + c1 * alpha * dphi0;
    int higher_than_lo = ls->phi > phi_lo;
    int curvature = fabs(ls->dphi) <= -c2 * dphi0;
    if (sufficient_decrease && curvature) {
      ls->sufficient_decrease = 1;
      ls->curvature = 1;
      ls->status = OLS_MINIMUM_FOUND;
      return alpha;
    }
    if (!sufficient_decrease || higher_than_lo) {
      ahi = alpha; phi_hi = ls->phi; dphi_hi = ls->dphi;
    } else {
      int reset_ahi = ls->dphi * (ahi - alo) <= 0;
      if (reset_ahi) { ahi = alo; phi_hi = phi_lo; dphi_hi = dphi_lo; }
      alo = alpha; phi_lo = ls->phi; dphi_lo = ls->dphi;
    }
  }
  ls->status = OLS_MAX_ITERATIONS;
  return alpha;
}

/* linesearch.cpp:385-412 */
static double backtracking(oracle_linesearch* ls, oracle_merit_fn f, void* ctx, double alpha0) {
  double alpha = alpha0;
  for (int iter = 1; iter < ls->max_iters; ++iter) {
    ls->n_iters += 1;
    f(alpha, &ls->phi, NULL, ctx);
    if (ls->phi <= ls->phi0 + ls->c1 * alpha * ls->dphi0) {
      ls->sufficient_decrease = 1;
      ls->curvature = 1;
      ls->status = OLS_MINIMUM_FOUND;
      return alpha;
    }
    alpha *= ls->beta_decrease;
  }
  return alpha;
}

/* linesearch.cpp:37-217 */
double oracle_ls_run(oracle_linesearch* ls, oracle_merit_fn f, void* ctx, double alpha0,
                     double phi0, double dphi0) {
  ls->phi0 = phi0;
  ls->dphi0 = dphi0;
  ls->n_iters = 0;
  ls->sufficient_decrease = 0;
  ls->curvature = 0;
  ls->status = OLS_NOERROR;
  if (dphi0 >= 0.0) {
    ls->status = OLS_NOT_DESCENT_DIRECTION;
    return 0.0;
  }
  double alpha_prev = 0.0, phi_prev = phi0, dphi_prev = dphi0;
  double alpha = alpha0;
  const double c1 = ls->c1, c2 = ls->c2;
  int hit_max_alpha = 0;

  for (int iter = 0; iter < ls->max_iters; ++iter) {
    ls->n_iters += 1;
    f(alpha, &ls->phi, &ls->dphi, ctx);
    const double phi = ls->phi, dphi = ls->dphi;
    int suff = phi <= phi0 + c1 * alpha * dphi0;
    int not_decreasing = phi >= phi_prev;
    int wolfe = fabs(dphi) <= -c2 * dphi0;
    if (suff && wolfe) {
      ls->sufficient_decrease = 1;
      ls->curvature = 1;
      ls->status = OLS_MINIMUM_FOUND;
      return alpha;
    } else if (iter == 0 && ls->try_cubic_first) {
      double x0, b, c, d, alpha_cubic = 0;
      int ok = 0;
      if (spline_from_2pts(0, phi0, dphi0, alpha, phi, dphi, &x0, &b, &c, &d)) {
        if (spline_argmin(x0, b, c, d, &alpha_cubic) && isfinite(alpha_cubic)) ok = 1;
      }
      if (ok) {
        ls->n_iters += 1;
        double phi_c, dphi_c;
        ++iter;
        f(alpha_cubic, &phi_c, &dphi_c, ctx);
        int suff_c = phi_c <= phi0 + c1 * alpha_cubic * dphi0;
        int wolfe_c = fabs(dphi_c) <= -c2 * dphi0;
        if (suff_c && wolfe_c) {
          ls->phi = phi_c;
          ls->dphi = dphi_c;
          ls->sufficient_decrease = 1;
          ls->curvature = 1;
          ls->status = OLS_MINIMUM_FOUND;
          return alpha_cubic;
        }
      }
    }
    if (ls->use_backtracking) return backtracking(ls, f, ctx, alpha0 * ls->beta_decrease);

    if (!suff || (iter > 0 && not_decreasing)) {
      ls->phi_lo = phi_prev; ls->dphi_lo = dphi_prev;
      ls->phi_hi = phi; ls->dphi_hi = dphi;
      return zoom(ls, f, ctx, alpha_prev, alpha);
    }
    if (dphi >= 0) {
      ls->phi_lo = phi; ls->dphi_lo = dphi;
      ls->phi_hi = phi_prev; ls->dphi_hi = dphi_prev;
      return zoom(ls, f, ctx, alpha, alpha_prev);
    }
    alpha_prev = alpha;
    alpha = alpha * ls->beta_increase;
    if (alpha > ls->alpha_max) {
      alpha = ls->alpha_max;
      if (hit_max_alpha) {
        ls->status = OLS_HIT_MAX_STEPSIZE;
        ls->sufficient_decrease = suff;
        ls->curvature = wolfe;
        return alpha;
      }
      hit_max_alpha = 1;
    }
    phi_prev = phi;
    dphi_prev = dphi;
  }
  return alpha;
}
