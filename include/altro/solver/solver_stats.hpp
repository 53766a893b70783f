// solver_stats.hpp -- solve statistics (reference: src/altro/solver/solver_stats.hpp:14-25).
// The reference only ever fills `status` and `iterations`; this implementation also fills
// solve_time, objective_value, stationarity and primal_feasibility.
#pragma once

#include <chrono>

#include "typedefs.hpp"

namespace altro {

struct AltroStats {
  using millisd = std::chrono::duration<double, std::milli>;
  SolveStatus status = SolveStatus::Unsolved;
  millisd solve_time{0.0};
  int iterations = 0;
  int outer_iterations = 0;
  double objective_value = 0.0;
  double stationarity = 0.0;
  double primal_feasibility = 0.0;
  double complimentarity = 0.0;
};

}  // namespace altro
