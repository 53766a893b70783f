/*
 * tvlqr.h -- the reference's kernel boundary, kept signature for signature.
 *
 * Same three functions, same argument order and meaning, same return convention as the reference's
 * src/tvlqr/tvlqr.h:11-33 (including C++ linkage: that header has no extern "C", so callers such as
 * SolverImpl::BackwardPass, solver.cpp:366-371, and tvlqr_test.cpp:176-202 mangle these names).
 * Link against libaltro_hip.so instead of the reference's tvlqr target and the calls below run on
 * the MI355X: host blocks are gathered into one device arena, the GENERIC HIP kernel
 * (altro_amd/csrc/kernels/tvlqr_generic.hip, one wavefront) performs the recursion with the
 * reference's operation order, and the results are scattered back to the caller's pointers.
 *   - per-knot-point dimensions nx[k], nu[k] are honoured (max 32 each);
 *   - return value: TVLQR_SUCCESS (-1) or the knot point whose Quu + reg I is not positive definite,
 *     with K_k = Qux, d_k = -Qu left unsolved exactly like tvlqr.cpp:162-164;
 *   - all of Qxx..Qu and the five *_tmp blocks are written, as the reference leaves them;
 *   - no host allocation after the first call of a given problem size (cached workspace);
 *   - there is no CPU fallback: without a HIP device the functions return TVLQR_NO_DEVICE (-2).
 */
#pragma once

#include <stdbool.h>

#define TVLQR_SUCCESS -1
#define TVLQR_NO_DEVICE -2

typedef double lqr_float;

int tvlqr_TotalMemSize(const int *nx, const int *nu, int num_horizon, bool is_diag);

int tvlqr_BackwardPass(const int *nx, const int *nu, int num_horizon,
                       const lqr_float *const *A, const lqr_float *const *B, const lqr_float *const *f,
                       const lqr_float *const *Q, const lqr_float *const *R, const lqr_float *const* H,
                       const lqr_float *const *q, const lqr_float *const *r, lqr_float reg,
                       lqr_float **K, lqr_float **d,
                       lqr_float **P, lqr_float **p, lqr_float *delta_V,
                       lqr_float **Qxx, lqr_float **Quu, lqr_float **Qux,
                       lqr_float **Qx, lqr_float **Qu,
                       lqr_float **Qxx_tmp, lqr_float **Quu_tmp, lqr_float **Qux_tmp,
                       lqr_float **Qx_tmp, lqr_float **Qu_tmp,
                       bool linear_only_update, bool is_diag);

int tvlqr_ForwardPass(const int *nx, const int *nu, int num_horizon,
                      const lqr_float *const *A, const lqr_float *const *B, const lqr_float *const *f,
                      const lqr_float *const *K, const lqr_float *const *d,
                      const lqr_float *const *P, const lqr_float *const *p,
                      const lqr_float *x0, lqr_float **x, lqr_float **u, lqr_float **y);
