// altro.hpp -- umbrella header, as consumers of the reference write `#include "altro/altro.hpp"`
// (examples/cmake/basic_cmake_project/main.cpp:5).
#pragma once
#include "altro/altro_solver.hpp"
