// al_types.h -- plain types of the augmented-Lagrangian constraint tables (shared by the kernels in
// al_lane.hip and the host code of the C ABI, which fills them).
#pragma once
#include <stdint.h>

namespace altro_hip {

constexpr int AL_MAXC = 2;     // constraint blocks per knot point
constexpr int AL_MAXP = 8;     // rows per zero / identity / orthant block
constexpr int AL_MAXSOC = 4;   // rows per second-order-cone block
constexpr int AL_MAXDEF = 16;  // distinct blocks per handle

enum { CONE_EQUALITY = 0, CONE_IDENTITY = 1, CONE_INEQUALITY = 2, CONE_SOC = 3 };   // typedefs.hpp:29-34

struct AlDef {
  int cone, p, g_per_problem;
  int G_off;       // into the G pool (elements)
  int64_t g_off;   // into the g pool (elements): [p] shared, or [p][batch]
};
struct AlKnot {          // everything a kernel needs about knot point k in ONE wave-uniform record
  int ncon;
  int def[AL_MAXC];
  int z_off[AL_MAXC];   // first row of this block's dual in z[rows][batch]
  int cone[AL_MAXC], p[AL_MAXC], g_per_problem[AL_MAXC], G_off[AL_MAXC];
  int64_t g_off[AL_MAXC];
};
template <typename T>
struct AlTable {
  const AlKnot* knots;   // [N + 1]
  const T* G;
  const T* g;
  T* z;
  int enabled;
};
// Tables and the shared G / g blocks are read through the constant address space: with a wave-uniform
// address that is an s_load into SGPRs (scalar cache) instead of a vector load + v_readlane waterfall.
#define ALTRO_CONST_AS __attribute__((address_space(4)))

}  // namespace altro_hip
