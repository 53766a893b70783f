// Test infrastructure (never part of the product): the three tvlqr_* entry points (include/tvlqr/tvlqr.h) bound to the CPU oracle
// (oracle/tvlqr_oracle.c, the restatement of src/tvlqr/tvlqr.cpp).  Linked into a test program it interposes the definitions
// libaltro_hip.so exports, so the SAME ALTROSolver (host callbacks, line search, AL updates) runs with its backward sweeps on the host
// CPU: the rate a reference user has today, printed beside the rate through the GPU seam (tests/test_gpu_scotty.py).
#include "tvlqr/tvlqr.h"

extern "C" {
int oracle_tvlqr_TotalMemSize(const int* nx, const int* nu, int num_horizon, bool is_diag);
int oracle_tvlqr_BackwardPass(const int* nx, const int* nu, int num_horizon, const double* const* A, const double* const* B,
                              const double* const* f, const double* const* Q, const double* const* R, const double* const* H,
                              const double* const* q, const double* const* r, double reg, double** K, double** d, double** P, double** p,
                              double* delta_V, double** Qxx, double** Quu, double** Qux, double** Qx, double** Qu, double** Qxx_tmp,
                              double** Quu_tmp, double** Qux_tmp, double** Qx_tmp, double** Qu_tmp, bool linear_only_update, bool is_diag);
int oracle_tvlqr_ForwardPass(const int* nx, const int* nu, int num_horizon, const double* const* A, const double* const* B,
                             const double* const* f, const double* const* K, const double* const* d, const double* const* P,
                             const double* const* p, const double* x0, double** x, double** u, double** y);
}

int tvlqr_TotalMemSize(const int* nx, const int* nu, int num_horizon, bool is_diag) {
  return oracle_tvlqr_TotalMemSize(nx, nu, num_horizon, is_diag);
}
int tvlqr_BackwardPass(const int* nx, const int* nu, int num_horizon, const lqr_float* const* A, const lqr_float* const* B,
                       const lqr_float* const* f, const lqr_float* const* Q, const lqr_float* const* R, const lqr_float* const* H,
                       const lqr_float* const* q, const lqr_float* const* r, lqr_float reg, lqr_float** K, lqr_float** d, lqr_float** P,
                       lqr_float** p, lqr_float* delta_V, lqr_float** Qxx, lqr_float** Quu, lqr_float** Qux, lqr_float** Qx,
                       lqr_float** Qu, lqr_float** Qxx_tmp, lqr_float** Quu_tmp, lqr_float** Qux_tmp, lqr_float** Qx_tmp,
                       lqr_float** Qu_tmp, bool linear_only_update, bool is_diag) {
  return oracle_tvlqr_BackwardPass(nx, nu, num_horizon, A, B, f, Q, R, H, q, r, reg, K, d, P, p, delta_V, Qxx, Quu, Qux, Qx, Qu, Qxx_tmp,
                                   Quu_tmp, Qux_tmp, Qx_tmp, Qu_tmp, linear_only_update, is_diag);
}
int tvlqr_ForwardPass(const int* nx, const int* nu, int num_horizon, const lqr_float* const* A, const lqr_float* const* B,
                      const lqr_float* const* f, const lqr_float* const* K, const lqr_float* const* d, const lqr_float* const* P,
                      const lqr_float* const* p, const lqr_float* x0, lqr_float** x, lqr_float** u, lqr_float** y) {
  return oracle_tvlqr_ForwardPass(nx, nu, num_horizon, A, B, f, K, d, P, p, x0, x, u, y);
}
