// linesearch_sm.h -- the strong-Wolfe cubic line search as a RESUMABLE state machine (host + device).
//
// The reference's CubicLineSearch (src/linesearch/linesearch.cpp:37-217 Run, :233-351 Zoom, :385-412
// SimpleBacktracking; spline helpers src/linesearch/cubicspline.c:18-42, :111-181, :229-246) calls the
// merit function through a std::function from inside its loops.  In the batched solver a merit
// evaluation is a kernel launch over ALL problems, so the search is turned inside out: every problem
// carries an LsState; `ls_begin` / `ls_feed` consume the (phi, dphi) of the step that was just
// evaluated and either finish or name the next trial step.  Fed the same numbers, the machine visits
// exactly the trial steps the reference does (tests/test_linesearch_sm.py checks this against the
// real reference code compiled into oracle/_ref).
#pragma once
#include "rtc_compat.h"

#if defined(__HIPCC__)
#define ALTRO_LS_HD __host__ __device__ inline
#else
#define ALTRO_LS_HD inline
#endif

#include "fp_contract.h"
ALTRO_FP_REGION_ON   // single-expression a * b + c only: the same rounding in every kernel these functions are inlined into (see models.h)
namespace altro_hip {

enum LsStatus {   // linesearch.hpp:16-25
  LS_NOERROR = 0, LS_MINIMUM_FOUND, LS_INVALID_POINTER, LS_NOT_DESCENT_DIRECTION, LS_WINDOW_TOO_SMALL,
  LS_GOT_NONFINITE_STEP_SIZE, LS_MAX_ITERATIONS, LS_HIT_MAX_STEPSIZE
};
enum LsStage { LS_STAGE_DONE = 0, LS_STAGE_RUN, LS_STAGE_CUBIC, LS_STAGE_ZOOM, LS_STAGE_ZOOM_MID, LS_STAGE_BACKTRACK };

struct LsOptions {   // linesearch.hpp:41-47, :55-56
  int max_iters;
  double alpha_max, beta_increase, beta_decrease, min_interval_size, c1, c2;
  int try_cubic_first, use_backtracking;
};
ALTRO_LS_HD LsOptions ls_default_options() {
  LsOptions o;
  o.max_iters = 25; o.alpha_max = 2.0; o.beta_increase = 1.5; o.beta_decrease = 0.5;
  o.min_interval_size = 1e-6; o.c1 = 1e-4; o.c2 = 0.9; o.try_cubic_first = 0; o.use_backtracking = 0;
  return o;
}

struct LsState {
  int stage, status, n_iters, iter, zoom_iter, bt_iter, hit_max_alpha;
  int sufficient_decrease, curvature;
  int want_derivative;        // does the pending evaluation need dphi?
  double alpha;               // the step to evaluate next (stage != DONE) / the accepted step (DONE)
  double alpha0, phi0, dphi0;
  double phi, dphi;           // "final merit values" (linesearch.cpp:32-35)
  double alpha_prev, phi_prev, dphi_prev;
  double alpha_first, phi_first, dphi_first;
  double alo, ahi, phi_lo, phi_hi, dphi_lo, dphi_hi;
};

namespace ls_detail {
constexpr double kTol = 1e-6;   // LINESEARCH_TOL, cubicspline.c:10
// Which way a comparison went, folded into a running code of the turns one feed takes (null: not recorded -- every existing caller;
// the recorder is how ls_feed_is_robust tells two feeds that took the same turns from two that did not).
ALTRO_LS_HD bool brc(int* br, bool c) {
  if (br) *br = *br * 31 + (c ? 2 : 1);
  return c;
}

// cubic through two points with slopes; returns false when the abscissae coincide (cubicspline.c:18-42)
ALTRO_LS_HD bool spline2(double x1, double y1, double d1, double x2, double y2, double d2, double* x0,
                         double* b, double* c, double* d, int* br = nullptr) {
  const double delta = x2 - x1;
  if (brc(br, fabs(delta) < kTol)) return false;
  *x0 = x1;
  *b = d1;
  *c = 3 * (y2 - y1) / (delta * delta) - (d2 + 2 * d1) / delta;
  *d = (d2 + d1) / (delta * delta) - 2 * (y2 - y1) / (delta * delta * delta);
  return true;
}
// minimiser of a + b t + c t^2 + d t^3 about x0 (cubicspline.c:111-181, :229-246)
ALTRO_LS_HD bool argmin(double x0, double b, double c, double d, double* xmin, int* br = nullptr) {
  const bool quad = brc(br, fabs(d) < kTol);
  if (quad) {
    if (brc(br, fabs(c) < kTol)) return false;
    if (brc(br, c <= 0)) return false;
    *xmin = -b / (2 * c) + x0;
    return true;
  }
  const double qa = 3 * d, qb = 2 * c, qc = b;
  if (brc(br, fabs(qa) < kTol)) return false;
  const double s2 = qb * qb - 4 * qa * qc;
  double s;
  if (brc(br, fabs(s2) < kTol)) s = 0.0;
  else if (brc(br, s2 < 0)) return false;
  else s = sqrt(s2);
  const double d1 = (-qb + s) / (2 * qa), d2 = (-qb - s) / (2 * qa);
  const double curv1 = 2 * c + 6 * d * d1, curv2 = 2 * c + 6 * d * d2;
  if (brc(br, fabs(curv1) < kTol && fabs(curv2) < kTol)) return false;
  if (brc(br, curv1 > 0 && curv2 < 0)) { *xmin = d1 + x0; return true; }
  if (brc(br, curv1 < 0 && curv2 > 0)) { *xmin = d2 + x0; return true; }
  return false;
}

ALTRO_LS_HD bool finish(LsState& s, double alpha) {
  s.alpha = alpha;
  s.stage = LS_STAGE_DONE;
  return false;
}
ALTRO_LS_HD bool request(LsState& s, int stage, double alpha, int want_derivative) {
  s.n_iters += 1;
  s.stage = stage;
  s.alpha = alpha;
  s.want_derivative = want_derivative;
  return true;
}

ALTRO_LS_HD bool run_top(LsState& s, const LsOptions& o, double alpha) {   // head of Run's for loop
  if (s.iter >= o.max_iters) return finish(s, alpha);
  return request(s, LS_STAGE_RUN, alpha, 1);
}

ALTRO_LS_HD bool zoom_top(LsState& s, const LsOptions& o, double last_alpha, int* br = nullptr) {   // head of Zoom's loop
  if (s.zoom_iter >= o.max_iters) {
    s.status = LS_MAX_ITERATIONS;
    return finish(s, last_alpha);
  }
  if (brc(br, fabs(s.alo - s.ahi) < o.min_interval_size))
    return request(s, LS_STAGE_ZOOM_MID, (s.alo + s.ahi) / 2.0, 1);
  double x0, b, c, d, a = 0.0;
  bool ok = false;
  if (spline2(s.alo, s.phi_lo, s.dphi_lo, s.ahi, s.phi_hi, s.dphi_hi, &x0, &b, &c, &d, br))
    ok = argmin(x0, b, c, d, &a, br) && isfinite(a);
  if (!brc(br, ok)) a = (s.alo + s.ahi) / 2;
  return request(s, LS_STAGE_ZOOM, a, 1);
}
ALTRO_LS_HD bool zoom_begin(LsState& s, const LsOptions& o, double alo, double ahi, double phi_lo,
                            double dphi_lo, double phi_hi, double dphi_hi, int* br = nullptr) {
  if (!isfinite(alo) || !isfinite(ahi)) {
    s.status = LS_GOT_NONFINITE_STEP_SIZE;
    return finish(s, 0.0);
  }
  s.alo = alo; s.ahi = ahi;
  s.phi_lo = phi_lo; s.dphi_lo = dphi_lo; s.phi_hi = phi_hi; s.dphi_hi = dphi_hi;
  s.zoom_iter = s.n_iters + 1;
  return zoom_top(s, o, alo, br);
}

ALTRO_LS_HD bool bt_top(LsState& s, const LsOptions& o, double alpha) {
  if (s.bt_iter >= o.max_iters) return finish(s, alpha);
  return request(s, LS_STAGE_BACKTRACK, alpha, 0);
}

// Run's loop body after the (optional) cubic first guess: alpha/phi/dphi are the FIRST evaluation's
ALTRO_LS_HD bool run_rest(LsState& s, const LsOptions& o, double alpha, double phi, double dphi, int* br = nullptr) {
  const bool suff = phi <= s.phi0 + o.c1 * alpha * s.dphi0;
  const bool not_decreasing = phi >= s.phi_prev;
  const bool wolfe = fabs(dphi) <= -o.c2 * s.dphi0;
  if (o.use_backtracking) {
    s.bt_iter = 1;
    return bt_top(s, o, s.alpha0 * o.beta_decrease);
  }
  if (brc(br, !suff || (s.iter > 0 && not_decreasing)))
    return zoom_begin(s, o, s.alpha_prev, alpha, s.phi_prev, s.dphi_prev, phi, dphi, br);
  if (brc(br, dphi >= 0)) return zoom_begin(s, o, alpha, s.alpha_prev, phi, dphi, s.phi_prev, s.dphi_prev, br);
  s.alpha_prev = alpha;
  double next = alpha * o.beta_increase;
  if (next > o.alpha_max) {
    next = o.alpha_max;
    if (s.hit_max_alpha) {
      s.status = LS_HIT_MAX_STEPSIZE;
      s.sufficient_decrease = suff;
      s.curvature = brc(br, wolfe);
      return finish(s, next);
    }
    s.hit_max_alpha = 1;
  }
  s.phi_prev = phi;
  s.dphi_prev = dphi;
  s.iter += 1;
  return run_top(s, o, next);
}
}  // namespace ls_detail

// Start a search from (alpha0, phi(0), dphi(0)).  Returns true when s.alpha must be evaluated.
ALTRO_LS_HD bool ls_begin(LsState& s, const LsOptions& o, double alpha0, double phi0, double dphi0) {
  s.stage = LS_STAGE_DONE; s.status = LS_NOERROR; s.n_iters = 0; s.iter = 0; s.zoom_iter = 0;
  s.bt_iter = 0; s.hit_max_alpha = 0; s.sufficient_decrease = 0; s.curvature = 0; s.want_derivative = 1;
  s.alpha0 = alpha0; s.phi0 = phi0; s.dphi0 = dphi0;
  s.alpha_prev = 0.0; s.phi_prev = phi0; s.dphi_prev = dphi0;
  if (dphi0 >= 0.0) {
    s.status = LS_NOT_DESCENT_DIRECTION;
    return ls_detail::finish(s, 0.0);
  }
  return ls_detail::run_top(s, o, alpha0);
}

// Feed the merit value (and derivative, when s.want_derivative) at s.alpha.  Returns true when
// another evaluation (of the new s.alpha) is needed; false when the search is over (s.alpha = result).
ALTRO_LS_HD bool ls_feed(LsState& s, const LsOptions& o, double phi, double dphi, int* br = nullptr) {
  using namespace ls_detail;
  const double alpha = s.alpha;
  switch (s.stage) {
    case LS_STAGE_RUN: {
      s.phi = phi; s.dphi = dphi;
      const bool suff = phi <= s.phi0 + o.c1 * alpha * s.dphi0;
      const bool wolfe = fabs(dphi) <= -o.c2 * s.dphi0;
      if (brc(br, suff && wolfe)) {
        s.sufficient_decrease = 1; s.curvature = 1; s.status = LS_MINIMUM_FOUND;
        return finish(s, alpha);
      }
      if (s.iter == 0 && o.try_cubic_first) {
        double x0, b, c, d, ac = 0.0;
        bool ok = false;
        if (spline2(0, s.phi0, s.dphi0, alpha, phi, dphi, &x0, &b, &c, &d, br))
          ok = argmin(x0, b, c, d, &ac, br) && isfinite(ac);
        if (brc(br, ok)) {
          s.alpha_first = alpha; s.phi_first = phi; s.dphi_first = dphi;
          s.iter += 1;
          return request(s, LS_STAGE_CUBIC, ac, 1);
        }
      }
      return run_rest(s, o, alpha, phi, dphi, br);
    }
    case LS_STAGE_CUBIC: {
      const bool suff = phi <= s.phi0 + o.c1 * alpha * s.dphi0;
      const bool wolfe = fabs(dphi) <= -o.c2 * s.dphi0;
      if (brc(br, suff && wolfe)) {
        s.phi = phi; s.dphi = dphi;
        s.sufficient_decrease = 1; s.curvature = 1; s.status = LS_MINIMUM_FOUND;
        return finish(s, alpha);
      }
      return run_rest(s, o, s.alpha_first, s.phi_first, s.dphi_first, br);
    }
    case LS_STAGE_ZOOM_MID: {
      s.phi = phi; s.dphi = dphi;
      s.sufficient_decrease = brc(br, phi <= s.phi0 + o.c1 * alpha * s.dphi0);
      s.curvature = brc(br, fabs(dphi) <= -o.c2 * s.dphi0);
      s.status = (s.sufficient_decrease && s.curvature) ? LS_MINIMUM_FOUND : LS_WINDOW_TOO_SMALL;
      return finish(s, alpha);
    }
    case LS_STAGE_ZOOM: {
      s.phi = phi; s.dphi = dphi;
      const bool suff = phi <= s.phi0 + o.c1 * alpha * s.dphi0;
      const bool higher = phi > s.phi_lo;
      const bool curv = fabs(dphi) <= -o.c2 * s.dphi0;
      if (brc(br, suff && curv)) {
        s.sufficient_decrease = 1; s.curvature = 1; s.status = LS_MINIMUM_FOUND;
        return finish(s, alpha);
      }
      if (brc(br, !suff || higher)) {
        s.ahi = alpha; s.phi_hi = phi; s.dphi_hi = dphi;
      } else {
        if (brc(br, dphi * (s.ahi - s.alo) <= 0)) { s.ahi = s.alo; s.phi_hi = s.phi_lo; s.dphi_hi = s.dphi_lo; }
        s.alo = alpha; s.phi_lo = phi; s.dphi_lo = dphi;
      }
      s.zoom_iter += 1;
      return zoom_top(s, o, alpha, br);
    }
    case LS_STAGE_BACKTRACK: {
      s.phi = phi;
      if (brc(br, phi <= s.phi0 + o.c1 * alpha * s.dphi0)) {
        s.sufficient_decrease = 1; s.curvature = 1; s.status = LS_MINIMUM_FOUND;
        return finish(s, alpha);
      }
      s.bt_iter += 1;
      return bt_top(s, o, alpha * o.beta_decrease);
    }
    default:
      return false;
  }
}

// The decision guard of evaluation forms that agree with the reference's only to rounding (the affine line-search rounds of plan
// MFMA16).  Would the search take the same turns if phi, phi' were off by a relative `margin`?  The state machine is run
// on copies of the state with the values pushed to the four corners of that box, each recording which way every comparison went
// (ls_detail::brc): Armijo, curvature, the bracket updates of linesearch.cpp:233-351, the degenerate-spline thresholds of
// cubicspline.c:18-42, :111-181 -- whatever the feed meets, without restating one of them here.  Same turns at every corner: the
// decision does not hang on the last bits.  False: evaluate this trial again in the reference's own order (a rollout) and feed that.
// `fed` (optional): the state after the feed, `need`: what that feed returned.
ALTRO_LS_HD bool ls_feed_is_robust(const LsState& s, const LsOptions& o, double phi, double dphi, double margin, LsState* fed = nullptr,
                                   bool* need_out = nullptr) {
  const double dp = margin * (fabs(phi) + fabs(s.phi0)), dd = margin * (fabs(dphi) + fabs(s.dphi0));
  LsState ref = s;
  int turns = 7;
  const bool need = ls_feed(ref, o, phi, dphi, &turns);
  if (fed) *fed = ref;
  if (need_out) *need_out = need;
  if (!(margin > 0.0)) return true;
  for (int c = 0; c < 4; ++c) {
    LsState t = s;
    int tc = 7;
    const bool nd = ls_feed(t, o, phi + ((c & 1) ? dp : -dp), dphi + ((c & 2) ? dd : -dd), &tc);
    if (nd != need || tc != turns || t.stage != ref.stage || t.status != ref.status) return false;
  }
  return true;
}

}  // namespace altro_hip
ALTRO_FP_REGION_END   // back to the including translation unit's own mode (fp_contract.h)
