// typedefs.hpp -- scalar, callback and enum types of the altro::ALTROSolver API.
//
// Every name, signature and enumerator VALUE below is API: callers hand these to the solver by value and
// compare results against them, so they mirror the reference's src/altro/solver/typedefs.hpp:12-68 -- including
// two things that look like slips but are part of the contract: the time step `h` of the dynamics callbacks is a
// `float`, and `ImplicitDynamicsJacobian` takes `x1` by value (`const double`, not a pointer).
#pragma once

#include <functional>

namespace altro {

class ALTROSolver;

using a_float = double;  // the solver's floating-point type

// Knot-point index sentinels accepted wherever an index (or the end of a range) is expected.
constexpr int LastIndex = -1;   // the terminal knot point N
constexpr int AllIndices = -2;  // every knot point the call applies to

// Why Solve() stopped (SolverStats::status, ALTROSolver::GetStatus()).
enum class SolveStatus : int {
  Success = 0,        // stationarity and primal feasibility under their tolerances
  Unsolved = 1,       // Solve() has not finished (also what a failed backward pass / line search leaves behind)
  MaxIterations = 2,  // iterations_max reached first
  // declared by the reference but never assigned by its solver, nor by this one:
  MaxObjectiveExceeded = 3,
  StateOutOfBounds = 4,
  InputOutOfBounds = 5,
  MeritFunGradientTooSmall = 6,
};

// The four cones a constraint value can be projected onto (AL term 1/2rho (|Pi_K(lambda - rho c)|^2 - |lambda|^2)).
enum class ConstraintType : int {
  EQUALITY = 0,           // c = 0        (zero cone)
  IDENTITY = 1,           // no constraint (the whole space)
  INEQUALITY = 2,         // c <= 0       (negative orthant)
  SECOND_ORDER_CONE = 3,  // |c[0..p-2]| <= c[p-1]
};

template <class Signature>
using Callback = std::function<Signature>;

// Registered with ALTROSolver::SetCallback.
using CallbackFunction = Callback<void(const ALTROSolver*)>;

// Discrete dynamics x+ = f(x, u, h) and its Jacobian [df/dx df/du], n2 x (n + m) column-major.
using ExplicitDynamicsFunction = Callback<void(double* xnext, const double* x, const double* u, float h)>;
using ExplicitDynamicsJacobian = Callback<void(double* jac, const double* x, const double* u, float h)>;

// Implicit form err = g(x1, u1, x2, u2, h) = 0 with one Jacobian per knot point (accepted by the API; the
// reference never evaluates them either).
using ImplicitDynamicsFunction =
    Callback<void(double* err, const double* x1, const double* u1, const double* x2, const double* u2, float h)>;
using ImplicitDynamicsJacobian = Callback<void(double* jac1, double* jac2, const double x1, const double* u1,
                                               const double* x2, const double* u2, float h)>;

// Stage cost l(x, u), its gradient (dx: n, du: m) and Hessian blocks (ddx: n x n, ddu: m x m, dxdu: n x m).
using CostFunction = Callback<a_float(const a_float* x, const a_float* u)>;
using CostGradient = Callback<void(a_float* dx, a_float* du, const a_float* x, const a_float* u)>;
using CostHessian = Callback<void(a_float* ddx, a_float* ddu, a_float* dxdu, const a_float* x, const a_float* u)>;

// Constraint value c(x, u) (p entries) and Jacobian (p x (n + m), column-major).
using ConstraintFunction = Callback<void(a_float* val, const a_float* x, const a_float* u)>;
using ConstraintJacobian = Callback<void(a_float* jac, const a_float* x, const a_float* u)>;

// Handle returned by ALTROSolver::SetConstraint: (knot point, position in that knot point's constraint list).
// Only the solver can make one.
class ConstraintIndex {
  friend ALTROSolver;
  ConstraintIndex(int knot_point, int position) : k(knot_point), i(position) {}
  int k;
  int i;

 public:
  int KnotPointIndex() const { return k; }
};

}  // namespace altro
