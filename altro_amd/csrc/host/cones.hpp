// cones.hpp -- conic projections for the augmented-Lagrangian terms (host side).
// Mathematics of the reference's src/altro/solver/cones.cpp:13-202 and cones.hpp:13-56:
//   zero cone {0} (EQUALITY), whole space (IDENTITY), negative orthant (INEQUALITY), and the
//   second-order cone {[v; s] : ||v|| <= s}; their projections, projection Jacobians and, for the SOC,
//   the derivative of J(x)^T b with respect to x.  Dual cones: EQUALITY <-> IDENTITY, the other two
//   are self-dual.  Column-major (dim x dim) outputs.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "altro/solver/typedefs.hpp"

namespace altro {
namespace cones {

inline ConstraintType DualCone(ConstraintType cone) {
  switch (cone) {
    case ConstraintType::EQUALITY: return ConstraintType::IDENTITY;
    case ConstraintType::IDENTITY: return ConstraintType::EQUALITY;
    default: return cone;
  }
}
inline bool ProjectionIsLinear(ConstraintType cone) { return cone != ConstraintType::SECOND_ORDER_CONE; }

inline double soc_norm(int nv, const double* x) {
  double a = 0.0;
  for (int i = 0; i < nv; ++i) a += x[i] * x[i];
  return std::sqrt(a);
}

inline void Projection(ConstraintType cone, int dim, const double* x, double* px) {
  switch (cone) {
    case ConstraintType::EQUALITY:
      for (int i = 0; i < dim; ++i) px[i] = 0.0;
      break;
    case ConstraintType::IDENTITY:
      for (int i = 0; i < dim; ++i) px[i] = x[i];
      break;
    case ConstraintType::INEQUALITY:
      for (int i = 0; i < dim; ++i) px[i] = std::min(0.0, x[i]);
      break;
    case ConstraintType::SECOND_ORDER_CONE: {
      const int nv = dim - 1;
      const double s = x[nv], a = soc_norm(nv, x);
      if (a <= -s) {            // below the cone
        for (int i = 0; i < dim; ++i) px[i] = 0.0;
      } else if (a <= s) {      // inside
        for (int i = 0; i < dim; ++i) px[i] = x[i];
      } else {                  // outside: scale onto the boundary
        const double c = 0.5 * (1 + s / a);
        for (int i = 0; i < nv; ++i) px[i] = c * x[i];
        px[nv] = c * a;
      }
    } break;
  }
}

inline void ProjectionJacobian(ConstraintType cone, int dim, const double* x, double* J) {
  std::fill(J, J + (size_t)dim * dim, 0.0);
  switch (cone) {
    case ConstraintType::EQUALITY: break;
    case ConstraintType::IDENTITY:
      for (int i = 0; i < dim; ++i) J[i + (size_t)i * dim] = 1.0;
      break;
    case ConstraintType::INEQUALITY:
      for (int i = 0; i < dim; ++i) J[i + (size_t)i * dim] = (x[i] <= 0) ? 1.0 : 0.0;
      break;
    case ConstraintType::SECOND_ORDER_CONE: {
      const int nv = dim - 1;
      const double s = x[nv], a = soc_norm(nv, x);
      if (a <= -s) break;
      if (a <= s) {
        for (int i = 0; i < dim; ++i) J[i + (size_t)i * dim] = 1.0;
        break;
      }
      // outside: P(v, s) = ((a + s) / 2) (u, 1) with u = v / a, hence
      //   J = 1/2 [ I + (s/a) (I - u u^T)   u ]
      //           [ u^T                     1 ]
      std::vector<double> u((size_t)nv);
      const double inv_a = 1.0 / a, t = s * inv_a;
      for (int i = 0; i < nv; ++i) u[i] = x[i] * inv_a;
      for (int j = 0; j < nv; ++j) {
        for (int i = 0; i < nv; ++i) J[i + (size_t)j * dim] = 0.5 * (((i == j) ? 1.0 + t : 0.0) - t * u[i] * u[j]);
        J[nv + (size_t)j * dim] = 0.5 * u[j];
        J[j + (size_t)nv * dim] = 0.5 * u[j];
      }
      J[nv + (size_t)nv * dim] = 0.5;
    } break;
  }
}

// d/dx [ J(x)^T b ] ; zero for the polyhedral cones
inline void ProjectionHessian(ConstraintType cone, int dim, const double* x, const double* b, double* H) {
  std::fill(H, H + (size_t)dim * dim, 0.0);
  if (cone != ConstraintType::SECOND_ORDER_CONE) return;
  // Second derivative of g(x) = b^T P(x) outside the cone, P(v, s) = ((a + s) / 2) (u, 1), a = |v|, u = v / a.
  // With Pi = I - u u^T, gamma = u^T b_v and w = Pi b_v (the part of b_v orthogonal to v):
  //   d2g/dv dv^T = 1/(2a) [ (b_s - (s/a) gamma) Pi - (s/a) (w u^T + u w^T) ]
  //   d2g/dv ds   = w / (2a),        d2g/ds ds = 0
  // (the u u^T terms of the individual pieces cancel); inside or below the cone P is linear and the Hessian vanishes.
  const int nv = dim - 1;
  const double s = x[nv], a = soc_norm(nv, x);
  if (a <= -s || a <= s) return;
  std::vector<double> u((size_t)nv), w((size_t)nv);
  const double inv_a = 1.0 / a, half = 0.5 * inv_a, t = s * inv_a;
  double gamma = 0.0;
  for (int i = 0; i < nv; ++i) { u[i] = x[i] * inv_a; gamma += u[i] * b[i]; }
  for (int i = 0; i < nv; ++i) w[i] = b[i] - gamma * u[i];
  const double kappa = b[nv] - t * gamma;
  for (int j = 0; j < nv; ++j) {
    for (int i = 0; i < nv; ++i)
      H[i + (size_t)j * dim] = half * (kappa * (((i == j) ? 1.0 : 0.0) - u[i] * u[j]) - t * (w[i] * u[j] + u[i] * w[j]));
    H[nv + (size_t)j * dim] = half * w[j];
    H[j + (size_t)nv * dim] = half * w[j];
  }
}

}  // namespace cones
}  // namespace altro
