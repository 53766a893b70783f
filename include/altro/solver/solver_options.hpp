// solver_options.hpp -- run-time options of altro::ALTROSolver.
//
// Field names, types and defaults are API and follow the reference's src/altro/solver/solver_options.hpp:14-39
// (yes, `use_backtracking_linesearch` is declared `double` there and callers assign `true` to it).  One deliberate
// difference: the reference leaves max_state_value / max_input_value uninitialised in the struct and sets them in
// SolverImpl; here they carry their effective default (+inf) from the start.
#pragma once

#include <limits>

#include "typedefs.hpp"

namespace altro {

enum class Verbosity { Silent, Outer, Inner, LineSearch };

struct AltroOptions {
  AltroOptions() = default;

  // -- termination ------------------------------------------------------------------------------------------
  int iterations_max = 200;              // total iLQR iterations over all AL outer loops
  double tol_stationarity = 1e-4;        // inf-norm of the Lagrangian gradient; sqrt(tol) also gates dual updates
  double tol_primal_feasibility = 1e-4;  // max constraint violation
  double tol_meritfun_gradient = 1e-8;   // |phi'(0)| below this skips the line search (step length 0)

  // -- accepted for source compatibility; consulted neither here nor by the reference's solver -----------------
  double tol_cost = 1e-4;               // (cost-decrease tolerances)
  double tol_cost_intermediate = 1e-4;  //
  double max_state_value = std::numeric_limits<double>::infinity();
  double max_input_value = std::numeric_limits<double>::infinity();
  double max_solve_time = std::numeric_limits<a_float>::infinity();

  // -- augmented Lagrangian ---------------------------------------------------------------------------------
  double penalty_initial = 1.0;   // rho of every constraint at the start of Solve()
  double penalty_scaling = 10.0;  // rho <- min(rho * scaling, penalty_max) after each dual update
  double penalty_max = 1e8;       // cap on rho

  // -- line search / reporting ------------------------------------------------------------------------------
  double use_backtracking_linesearch = false;  // nonzero: simple backtracking instead of the cubic strong-Wolfe search
  Verbosity verbose = Verbosity::Silent;       // above Silent: one line per iteration; LineSearch: also each trial
  bool throw_errors = true;                    // accepted; error reporting is chosen at compile time (exceptions.hpp)
};

}  // namespace altro
