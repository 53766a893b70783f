// al_types.h -- plain types of the augmented-Lagrangian constraint tables (shared by the kernels in
// al_lane.hip and the host code of the C ABI, which fills them).
#pragma once
#include "../rtc_compat.h"

namespace altro_hip {

constexpr int AL_MAXC = 2;     // constraint blocks per knot point on plan LANE (its kernels unroll two)
constexpr int AL_MAXP = 8;     // rows per zero / identity / orthant block (plan LANE); rows per SLOT on plan MFMA16
// Plan MFMA16 (round 6): a knot point's record holds up to AL_TILE_MAXC SLOTS of at most AL_MAXP rows.  A caller's block in the zero /
// identity / orthant cones with more rows than a slot is laid out over consecutive slots by the host (those cones project row by row:
// cones.cpp:13-38, so rows 8s .. 8s + 7 of a block ARE a block of their own with the same duals in the same place); a second-order-cone
// block takes one slot.  6 x 8 = 48 rows per knot point: e.g. an input box (8 rows) and a state box (24) on a (12, 4) problem with two
// slots to spare.  The row-layout kernels loop over the slots a knot point has; the merit kernel, which carries per-slot values in
// registers across the ring of knot points, is instantiated for 2 and for AL_TILE_MAXC slots (AlTable::max_ncon picks).
constexpr int AL_TILE_MAXC = 6;
constexpr int AL_TILE_MAXSLOTDEF = 32;   // distinct slots per handle on plan MFMA16 (their padded Jacobians sit in the merit kernel's LDS: 32 x 1296 B)
constexpr int AL_MAXSOC = 4;   // rows per second-order-cone block (plans LANE / MFMA16; plan GENERIC: up to GEN_MAXSOC)
constexpr int AL_MAXDEF = 16;  // distinct blocks per handle
// AlTable::Gpad: every block as 9 rows x 16 tile columns, rows padded to 18 (16-byte aligned, eight lanes reading one column hit
// eight bank groups): rows >= p and row 8 are zero, so a lane reads `its` row (min(lane, 8)) or any column without a select
constexpr int AL_GP_LD = 18, AL_GP_DEF = 9 * AL_GP_LD;

// Plan GENERIC has its own, larger table (kernels/ilqr_generic.hip loops over the blocks instead of unrolling two of them, one lane per
// row): the reference takes any number of constraints of any dimension per knot point (knotpoint_data.hpp:16, knotpoint_data.cpp:155-161)
constexpr int GEN_MAXC = 8;     // constraint blocks per knot point on plan GENERIC
constexpr int GEN_MAXSOC = 32;  // rows per second-order-cone block on plan GENERIC (one lane per row; cones.cpp:13-123 takes any dimension)
constexpr int GEN_MAXP = 64;    // rows per zero / identity / orthant block on plan GENERIC: one lane each, so up to n + m = 64
constexpr int GEN_MAXDEF = 64;  // distinct blocks per handle on plan GENERIC

enum { CONE_EQUALITY = 0, CONE_IDENTITY = 1, CONE_INEQUALITY = 2, CONE_SOC = 3 };   // typedefs.hpp:29-34

struct AlDef {
  int cone, p, g_per_problem;
  int G_off;       // into the G pool (elements)
  int64_t g_off;   // into the g pool (elements): [p] shared, or [p][batch]
  int user;        // 0: c = G [x;u] - g;  id + 1: rows and Jacobian come from the caller's run-time compiled source
  int w = 0;       // columns of G as given: n + m of the handle, or nx[k] + nu[k] of the block's knot points (per-knot-point dimensions)
};
struct AlKnot {          // everything a kernel needs about knot point k in ONE wave-uniform record
  int ncon;               // plan LANE: blocks (<= AL_MAXC); plan MFMA16: slots (<= AL_TILE_MAXC)
  int def[AL_TILE_MAXC];
  int z_off[AL_TILE_MAXC];   // first row of this block's dual in z[rows][batch]
  int cone[AL_TILE_MAXC], p[AL_TILE_MAXC], g_per_problem[AL_TILE_MAXC], G_off[AL_TILE_MAXC];
  int64_t g_off[AL_TILE_MAXC];
  // Bound-type blocks: every row of G is +-e_idx (control / state bounds, goal pins -- most constraints of an MPC
  // problem).  sel != 0 marks such a block; sidx[row] = +(idx + 1) or -(idx + 1).  The kernels then skip G.
  int sel[AL_TILE_MAXC];
  int sidx[AL_TILE_MAXC][AL_MAXP];
  // id + 1 of a block whose value c(x, u) and Jacobian dc/d[x;u] the caller's source computes (altro_hip_add_user_constraint:
  // ALTROSolver::SetConstraint with a general callback pair, altro_solver.cpp:192-223); 0 for c = G [x;u] - g
  int user[AL_TILE_MAXC];
  // the slot's Jacobian in the zero-padded pool AlTable::Gpad (plan MFMA16's row-layout kernels): slot definition index * AL_GP_DEF
  int Gp_off[AL_TILE_MAXC];
};
struct AlKnotBig {       // the host's record of a knot point on every plan, and plan GENERIC's device table entry
  int ncon;
  int def[GEN_MAXC];
  int z_off[GEN_MAXC];
  int cone[GEN_MAXC], p[GEN_MAXC], g_per_problem[GEN_MAXC], G_off[GEN_MAXC];
  int64_t g_off[GEN_MAXC];
};
template <typename T>
struct AlTable {
  const AlKnot* knots;   // [N + 1]   (plans LANE, MFMA16)
  const AlKnotBig* big = nullptr;   // [N + 1]   (plan GENERIC)
  const T* G;
  const T* g;
  T* z;
  int enabled;
  // uniform != 0: knot points 0 .. N-1 carry the same blocks (the usual MPC problem), so the kernels read the table
  // entry of knot point 0 at every step -- the same constant-memory address, a scalar-cache hit instead of a fresh
  // miss per step -- and offset the duals by k * rows_per_knot.
  int uniform, rows_per_knot, N;
  int G_count;           // elements of the G pool (kernels that keep it in LDS: kernels/ilqr_merit2_dpp.hip)
  const T* Gpad;         // [def][9][AL_GP_LD], see AL_GP_DEF (plan MFMA16 only; else null)
  int Gpad_count;        // its elements
  int all_sel;           // every block is bound-type (AlKnot::sel): the Gauss-Newton blocks are diagonal
  int has_soc;           // some block is a second-order cone (kernels instantiated without the cone's code serve handles that have none)
  // plan GENERIC: per block definition [1 + GEN_MAXP] ints -- [0] != 0: every row of G is +-e_idx (a bound-type block: the Gauss-Newton
  // term of such a block is diagonal), [1 + i] = +(idx + 1) / -(idx + 1) for row i = +e_idx / -e_idx (the caller's column order) -- or null
  const int* gsel = nullptr;
  int max_ncon = 0;      // most blocks / slots any knot point has (plan MFMA16: above AL_MAXC the merit kernel's wide instantiation runs)
};
#define ALTRO_CONST_AS __attribute__((address_space(4)))
// table entry and dual-row shift for knot point k
template <typename T>
__device__ __forceinline__ const AlKnot ALTRO_CONST_AS& al_knot(const AlTable<T>& t, int k, int& zshift) {
  const bool u = t.uniform != 0 && k < t.N;
  zshift = u ? k * t.rows_per_knot : 0;
  return *(const AlKnot ALTRO_CONST_AS*)(t.knots + (u ? 0 : k));
}
// Tables and the shared G / g blocks are read through the constant address space: with a wave-uniform
// address that is an s_load into SGPRs (scalar cache) instead of a vector load + v_readlane waterfall.

}  // namespace altro_hip
