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

#define TVLQR_SUCCESS -1   /* both passes: no knot point failed */
#define TVLQR_NO_DEVICE -2 /* this library only: no usable HIP device (the reference has no such case) */
#define TVLQR_UNSUPPORTED_SIZE -3 /* this library only: some nx[k] or nu[k] exceeds 256 (the reference has no limit; up to round 4: 32) */

typedef double lqr_float;

/* One pointer per knot point, each to a column-major block owned by the caller.  The aliases spell the same
 * types the reference spells out, so the C++-mangled names are unchanged. */
typedef const lqr_float *const *tvlqr_blocks_in; /* read by the pass  */
typedef lqr_float **tvlqr_blocks_out;            /* written by the pass */

/* Bytes a caller would need to hold every block of one problem (inputs, gains, cost-to-go, Q blocks and
 * scratch); `is_diag` sizes Q, R as diagonals. */
int tvlqr_TotalMemSize(const int *nx, const int *nu, int num_horizon, bool is_diag);

/* Riccati recursion k = num_horizon .. 0 over x+ = A x + B u + f with stage cost
 * 1/2 x'Qx + 1/2 u'Ru + u'Hx + q'x + r'u.  Returns TVLQR_SUCCESS or the failing knot point. */
int tvlqr_BackwardPass(
    const int *nx,           /* [num_horizon + 1] state dimensions   */
    const int *nu,           /* [num_horizon + 1] input dimensions   */
    int num_horizon,         /* N: knot points 0..N                  */
    tvlqr_blocks_in A,       /* [N]   nx[k+1] x nx[k]                */
    tvlqr_blocks_in B,       /* [N]   nx[k+1] x nu[k]                */
    tvlqr_blocks_in f,       /* [N]   nx[k+1]                        */
    tvlqr_blocks_in Q,       /* [N+1] nx x nx, or nx if is_diag      */
    tvlqr_blocks_in R,       /* [N]   nu x nu, or nu if is_diag      */
    tvlqr_blocks_in H,       /* [N]   nu x nx                        */
    tvlqr_blocks_in q,       /* [N+1] nx                             */
    tvlqr_blocks_in r,       /* [N]   nu                             */
    lqr_float reg,           /* added to the diagonal of Quu         */
    tvlqr_blocks_out K,      /* [N]   nu x nx feedback gain          */
    tvlqr_blocks_out d,      /* [N]   nu feedforward                 */
    tvlqr_blocks_out P,      /* [N+1] nx x nx cost-to-go Hessian     */
    tvlqr_blocks_out p,      /* [N+1] nx cost-to-go gradient         */
    lqr_float *delta_V,      /* [2]   sum d'Qu, sum 1/2 d'Quu d       */
    tvlqr_blocks_out Qxx,    /* [N+1] action-value expansion ...     */
    tvlqr_blocks_out Quu,    /* [N]                                  */
    tvlqr_blocks_out Qux,    /* [N]                                  */
    tvlqr_blocks_out Qx,     /* [N+1]                                */
    tvlqr_blocks_out Qu,     /* [N]                                  */
    tvlqr_blocks_out Qxx_tmp, /* [N+1] ... and the scratch blocks    */
    tvlqr_blocks_out Quu_tmp, /* [N]   (Quu_tmp holds the factor)    */
    tvlqr_blocks_out Qux_tmp, /* [N]                                 */
    tvlqr_blocks_out Qx_tmp,  /* [N+1]                               */
    tvlqr_blocks_out Qu_tmp,  /* [N]                                 */
    bool linear_only_update, /* accepted, unused (as upstream)       */
    bool is_diag);

/* Closed-loop rollout u = d - K x, x+ = A x + B u + f from x0, with co-states y = P x + p (skipped when y is NULL). */
int tvlqr_ForwardPass(
    const int *nx, const int *nu, int num_horizon,
    tvlqr_blocks_in A, tvlqr_blocks_in B, tvlqr_blocks_in f, /* dynamics, as above     */
    tvlqr_blocks_in K, tvlqr_blocks_in d,                    /* gains from the backward pass */
    tvlqr_blocks_in P, tvlqr_blocks_in p,                    /* cost-to-go from the backward pass */
    const lqr_float *x0,                                     /* nx[0] initial state    */
    tvlqr_blocks_out x,                                      /* [N+1] states           */
    tvlqr_blocks_out u,                                      /* [N]   inputs           */
    tvlqr_blocks_out y);                                     /* [N+1] co-states        */

/* This library only (the reference has nothing to warm up): one throw-away backward pass on blocks of this shape, so that device
 * initialisation, the kernels' load and the workspace allocation do not land in the caller's first sweep.  ALTROSolver::Initialize
 * calls it.  Returns TVLQR_SUCCESS, TVLQR_NO_DEVICE or TVLQR_UNSUPPORTED_SIZE. */
int tvlqr_hip_warmup(const int *nx, const int *nu, int num_horizon);
