// solver_stats.hpp -- solve statistics (reference: src/altro/solver/solver_stats.hpp:14-25).
// The reference only ever fills `status` and `iterations`; this implementation also fills
// solve_time, objective_value, stationarity and primal_feasibility.
#pragma once

#include <chrono>

#include "typedefs.hpp"

namespace altro {

struct AltroStats {
  using millisd = std::chrono::duration<double, std::milli>;

  SolveStatus status = SolveStatus::Unsolved;  // why Solve() stopped
  int iterations = 0;                          // iLQR iterations, over all outer loops
  int outer_iterations = 0;                    // declared by the API; left at 0 (as upstream)
  millisd solve_time{0.0};                     // wall clock of Solve(), host side
  double objective_value = 0.0;                // merit value phi at the accepted step of the last iteration
  double stationarity = 0.0;                   // last value compared with tol_stationarity
  double primal_feasibility = 0.0;             // last value compared with tol_primal_feasibility
  double complimentarity = 0.0;                // declared by the API (spelling included); never computed
};

}  // namespace altro
