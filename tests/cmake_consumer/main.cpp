// What examples/cmake/basic_cmake_project/main.cpp does with the public API (printf instead of fmt).
#include <cstdio>

#include "altro/altro.hpp"

int main() {
  altro::ALTROSolver solver(10);
  const altro::ErrorCodes err = solver.SetDimension(4, 2, 0, altro::LastIndex);
  std::printf("Solver Initialized! (%d)\n", (int)err);
  return err == altro::ErrorCodes::NoError ? 0 : 1;
}
