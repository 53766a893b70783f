// cones.hpp -- conic projections for the augmented-Lagrangian terms (host side).
// Mathematics of the reference's src/altro/solver/cones.cpp:13-202 and cones.hpp:13-56:
//   zero cone {0} (EQUALITY), whole space (IDENTITY), negative orthant (INEQUALITY), and the
//   second-order cone {[v; s] : ||v|| <= s}; their projections, projection Jacobians and, for the SOC,
//   the derivative of J(x)^T b with respect to x.  Dual cones: EQUALITY <-> IDENTITY, the other two
//   are self-dual.  Column-major (dim x dim) outputs.
#pragma once
#include <algorithm>
#include <cmath>

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
      const double c = 0.5 * (1 + s / a);
      for (int j = 0; j < nv; ++j)
        for (int i = 0; i < nv; ++i) {
          double v = -0.5 * s / (a * a * a) * x[i] * x[j];
          v += (i == j) ? c : 0;
          J[i + (size_t)j * dim] = v;
        }
      for (int i = 0; i < nv; ++i) J[i + (size_t)nv * dim] = 0.5 * x[i] / a;
      for (int j = 0; j < nv; ++j) J[nv + (size_t)j * dim] = ((-0.5 * s / (a * a)) + c / a) * x[j];
      J[nv + (size_t)nv * dim] = 0.5;
    } break;
  }
}

// d/dx [ J(x)^T b ] ; zero for the polyhedral cones
inline void ProjectionHessian(ConstraintType cone, int dim, const double* x, const double* b, double* H) {
  std::fill(H, H + (size_t)dim * dim, 0.0);
  if (cone != ConstraintType::SECOND_ORDER_CONE) return;
  const int nv = dim - 1;
  const double s = x[nv], bs = b[nv];
  double vbv = 0.0, a = 0.0;
  for (int i = 0; i < nv; ++i) { a += x[i] * x[i]; vbv += x[i] * b[i]; }
  a = std::sqrt(a);
  if (a <= -s || a <= s) return;
  for (int i = 0; i < nv; ++i) {
    double hi = 0.0;
    for (int j = 0; j < nv; ++j) {
      double Hij = -x[i] * x[j] / (a * a);
      Hij += (i == j) ? 1 : 0;
      hi += Hij * b[j];
    }
    H[i + (size_t)nv * dim] = hi / (2 * a);
    H[nv + (size_t)i * dim] = hi / (2 * a);
    for (int j = 0; j <= i; ++j) {
      const double vij = x[i] * x[j];
      const double H1 = hi * x[j] * (-s / (a * a * a));
      double H2 = vij * (2 * vbv) / (a * a * a * a) - x[i] * b[j] / (a * a);
      double H3 = -vij / (a * a);
      if (i == j) { H2 -= vbv / (a * a); H3 += 1; }
      H2 *= s / a;
      H3 *= bs / a;
      const double v = (H1 + H2 + H3) / 2.0;
      H[i + (size_t)j * dim] = v;
      H[j + (size_t)i * dim] = v;
    }
  }
}

}  // namespace cones
}  // namespace altro
