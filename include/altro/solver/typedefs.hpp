// typedefs.hpp -- public scalar / callback / enum types of the altro::ALTROSolver API.
// Same names, meanings and enumerator ORDER as the reference's src/altro/solver/typedefs.hpp:12-68
// (they are part of the API contract: callers pass these by value).
#pragma once

#include <functional>

namespace altro {

using a_float = double;

class ALTROSolver;

constexpr int LastIndex = -1;
constexpr int AllIndices = -2;

enum class SolveStatus {
  Success,
  Unsolved,
  MaxIterations,
  MaxObjectiveExceeded,
  StateOutOfBounds,
  InputOutOfBounds,
  MeritFunGradientTooSmall,
};

using CallbackFunction = std::function<void(const ALTROSolver*)>;

// x+ = f(x, u, h) and its Jacobian [df/dx df/du], (n2 x (n+m)) column-major; h is a float
using ExplicitDynamicsFunction = std::function<void(double* xnext, const double* x, const double* u, float h)>;
using ExplicitDynamicsJacobian = std::function<void(double* jac, const double* x, const double* u, float h)>;
using ImplicitDynamicsFunction =
    std::function<void(double* err, const double* x1, const double* u1, const double* x2, const double* u2, float h)>;
using ImplicitDynamicsJacobian = std::function<void(double* jac1, double* jac2, const double x1, const double* u1,
                                                    const double* x2, const double* u2, float h)>;

using CostFunction = std::function<a_float(const a_float* x, const a_float* u)>;
using CostGradient = std::function<void(a_float* dx, a_float* du, const a_float* x, const a_float* u)>;
using CostHessian = std::function<void(a_float* ddx, a_float* ddu, a_float* dxdu, const a_float* x, const a_float* u)>;

using ConstraintFunction = std::function<void(a_float* val, const a_float* x, const a_float* u)>;
using ConstraintJacobian = std::function<void(a_float* jac, const a_float* x, const a_float* u)>;

enum class ConstraintType { EQUALITY, IDENTITY, INEQUALITY, SECOND_ORDER_CONE };

class ConstraintIndex {
 public:
  int KnotPointIndex() const { return k; }
  friend ALTROSolver;

 private:
  ConstraintIndex(int k, int i) : k(k), i(i) {}
  int k;
  int i;
};

}  // namespace altro
