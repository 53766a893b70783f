// oracle/ref_linesearch_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// A C-callable door into the REAL reference line search.  This file contains no reference
// code: it includes the reference's own header where it lies (-I/root/reference/src/linesearch)
// and is linked against objects compiled from the reference's own cubicspline.c and
// linesearch.cpp by `make -C oracle ref`.  Output: oracle/_ref/liblinesearch_ref.so.
#include "linesearch.hpp"

extern "C" {
typedef void (*ref_merit_fn)(double alpha, double* phi, double* dphi, void* ctx);

// Runs linesearch::CubicLineSearch::Run (linesearch.cpp:37) with default options except the
// two flags SolverImpl sets (solver.cpp:248, :417).  Returns the step; fills the outputs.
double ref_ls_run(ref_merit_fn f, void* ctx, double alpha0, double phi0, double dphi0,
                  int try_cubic_first, int use_backtracking, int* status, int* iters,
                  double* phi, double* dphi) {
  linesearch::CubicLineSearch ls;
  ls.try_cubic_first = try_cubic_first != 0;
  ls.use_backtracking_linesearch = use_backtracking != 0;
  auto merit = [f, ctx](double a, double* p, double* dp) { f(a, p, dp, ctx); };
  double alpha = ls.Run(merit, alpha0, phi0, dphi0);
  *status = static_cast<int>(ls.GetStatus());
  *iters = ls.Iterations();
  ls.GetFinalMeritValues(phi, dphi);
  return alpha;
}
}
