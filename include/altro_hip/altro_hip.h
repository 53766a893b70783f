/*
 * altro_hip.h -- the drop-in boundary: a C ABI (plain pointers and sizes, no C++/torch types)
 * in front of the hand-written HIP (gfx950) implementation of ALTRO's iLQR inner loop.
 *
 * Every entry point names the reference interface it stands in for (paths relative to the
 * reference tree bjack205/altro):
 *
 *   altro_hip_backward      <-> tvlqr_BackwardPass   src/tvlqr/tvlqr.h:17-27, tvlqr.cpp:65-195,
 *                               as called by SolverImpl::BackwardPass  src/altro/solver/solver.cpp:360-378
 *   altro_hip_forward_ltv   <-> tvlqr_ForwardPass    src/tvlqr/tvlqr.h:29-33, tvlqr.cpp:197-248,
 *                               as called by SolverImpl::LinearRollout src/altro/solver/solver.cpp:133-146
 *   altro_hip_merit         <-> SolverImpl::MeritFunction              src/altro/solver/solver.cpp:273-355
 *   altro_hip_expand        <-> SolverImpl::CalcExpansions + KnotPointData::CalcDynamicsExpansion /
 *                               CalcCostGradient     solver.cpp:189-201, knotpoint_data.cpp:406-471
 *   altro_hip_set_*         <-> the pointer arrays SolverImpl::Initialize builds over KnotPointData
 *                               members (solver.cpp:63-106) and the setters that fill them
 *                               (altro_solver.cpp:68-81,118-172,177-190, knotpoint_data.cpp:64-153)
 *   altro_hip_get_*         <-> KnotPointData public members K_,d_,P_,p_,x_,u_,y_,Qxx_...
 *                               (knotpoint_data.hpp:160-233) and ALTROSolver::GetState/GetInput/
 *                               GetFeedbackGain/GetFeedforwardGain (altro_solver.hpp:413-422)
 *   altro_hip_ilqr_solve    <-> SolverImpl::Solve incl. ForwardPass + CubicLineSearch and the outer dual / penalty
 *                               updates (solver.cpp:237-271, 383-409, 414-511; linesearch.cpp:37-412), per problem
 *   altro_hip_add_linear_constraint <-> ALTROSolver::SetConstraint (altro_solver.cpp:192-223) with the AL / conic
 *                               terms of knotpoint_data.cpp:489-613 and cones.cpp:13-202
 *   altro_hip_shift_trajectory / update_linear_costs / get_knot <-> the MPC methods (altro_solver.cpp:266-293, 323-347)
 *   altro_hip_stats_*       <-> AltroStats (solver_stats.hpp:14-25) as SolverImpl::Solve fills it (solver.cpp:464-469,
 *                               :492-509), reduced per batch on the device and across GPUs by RCCL
 *
 * The single-problem kernel boundary itself (the three tvlqr_* functions with the reference's exact
 * C++ signatures) is declared in include/tvlqr/tvlqr.h and exported by the same library.
 *
 * Conventions
 *   - One handle = `batch` independent problems with the same horizon N and dims (n, m), bound to one
 *     HIP device and one stream.  One host thread per handle (the reference is non-re-entrant per
 *     solver instance too: solver.cpp:277-278).
 *   - "Reference layout" for host buffers: [batch][k][block], every block column-major exactly as the
 *     reference stores it (tvlqr.cpp:13-16): A n*n, B n*m, f n, Q n*n (or n when is_diag), R m*m (or m),
 *     H m*n, q n, r m, K m*n, d m, P n*n, p n, x n, u m, y n.  Q, q, P, p, x, y have N+1 knot points,
 *     everything else N.  Host buffers are borrowed for the duration of the call only.
 *   - Broadcast: `k_stride_zero` / `batch_stride_zero` flags say the host buffer holds ONE knot point
 *     and/or ONE problem that every k / every problem shares (the device copy is still expanded: every
 *     (problem, knot point) owns its blocks in HBM, the general time-varying case).
 *   - No allocation happens inside backward / forward / merit / sweep (tvlqr_test.cpp:174-182 asserts
 *     the same of the reference).
 *   - Return value: 0 on success, a negative altro_hip_error otherwise; altro_hip_last_error() gives
 *     the message.  Per-problem results of the backward pass follow the reference's convention:
 *     status[b] == -1 (TVLQR_SUCCESS, tvlqr.h:11) or the knot-point index whose Quu was not positive
 *     definite (tvlqr.cpp:162-164).
 *   - There is NO CPU fallback: without a HIP device every compute entry point fails with
 *     ALTRO_HIP_ERR_NO_DEVICE.
 */
#ifndef ALTRO_HIP_H_
#define ALTRO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALTRO_HIP_VERSION 300
#define ALTRO_HIP_TVLQR_SUCCESS (-1) /* tvlqr.h:11 */

typedef struct altro_hip_batch altro_hip_batch; /* opaque handle */

typedef enum altro_hip_dtype {
  ALTRO_HIP_F64 = 0, /* the reference's lqr_float / a_float (tvlqr.h:13, typedefs.hpp:12) */
  ALTRO_HIP_F32 = 1  /* extension: BASELINE.json configs[4] */
} altro_hip_dtype;

typedef enum altro_hip_error {
  ALTRO_HIP_OK = 0,
  ALTRO_HIP_ERR_NO_DEVICE = -1,
  ALTRO_HIP_ERR_BAD_ARGUMENT = -2,
  ALTRO_HIP_ERR_UNSUPPORTED = -3,
  ALTRO_HIP_ERR_HIP = -4,
  ALTRO_HIP_ERR_NOT_SET = -5,
  ALTRO_HIP_ERR_OUT_OF_MEMORY = -6
} altro_hip_error;

/* Which kernel family a handle runs.  AUTO picks by measured sweep cost (tools/shape_cliff.py, profiles/r06b_shape_cliff.txt):
 * (12, 4) -> MFMA16; n <= 6, m <= 3 -> LANE, except n >= 5 with m >= 2 below 6144 problems ((6, 3): 8192), where a lane carrying
 * whole 5 x 5 / 6 x 6 blocks is a long single-wave chain and the zero-padded tile is up to 2.7 x faster -> MFMA16 (such a handle moves
 * to LANE by itself when altro_hip_set_model names a compiled-in model only LANE carries, provided nothing else was set on it yet);
 * other n <= 12, m <= 4 -> MFMA16 (padded); fp64 problems with uniform dimensions, n <= 31, m <= 8, n + m <= 32 past that tile ->
 * MFMA32 (the same arrays and iLQR loop as GENERIC, the sweeps on 2 x 2 matrix-core tiles: 3-6 x faster than GENERIC); anything else
 * (<= 256) -> GENERIC (the TVLQR sweep for any size; the iLQR loop for n, m <= 64 with dynamics given as data, a quadratic cost and
 * linear constraint blocks in every cone, kernels/ilqr_generic.hip: correctness first; no device models, no regularisation retry).
 * What AUTO does NOT promise is the CPU path's bits: plans LANE and GENERIC repeat it operation for operation (bit-identical K, d, P,
 * p), the matrix-core plans MFMA16 / MFMA32 agree with it to rounding (K, d within 1e-8 absolute, measured 1e-13).  A caller that
 * needs the bits names ALTRO_HIP_PLAN_LANE / ALTRO_HIP_PLAN_GENERIC; one that passes a flag only plan LANE honours
 * (ALTRO_HIP_LANE_FUSED) stays on LANE under AUTO.  altro_hip_ilqr_solve_async needs plan LANE with a compiled-in device model
 * (altro_hip_set_model on an AUTO handle moves it there while nothing but the initial state has been set). */
typedef enum altro_hip_plan {
  ALTRO_HIP_PLAN_AUTO = 0,
  ALTRO_HIP_PLAN_GENERIC = 1, /* wave-per-problem, any (n, m) <= 256: blocks staged in LDS, or (past ~32) worked on in global memory;
                                 the iLQR loop of this plan: n, m <= 64                            */
  ALTRO_HIP_PLAN_MFMA16 = 2,  /* wave-per-problem, 16x16x4 MFMA tiles: (n, m) = (12, 4), and any n <= 12,
                                 m <= 4 on zero-padded records (same results, the (12, 4) cost)        */
  ALTRO_HIP_PLAN_LANE = 3,    /* lane-per-problem, batch structure-of-arrays, n <= 6 and m <= 3         */
  ALTRO_HIP_PLAN_MFMA32 = 4   /* wave-per-problem, 2 x 2 tiles of v_mfma_f64_16x16x4 (kernels/tvlqr_tile32.hip): fp64, uniform dimensions,
                                 12 < n <= 31, m <= 8, n + m <= 32 (and n <= 12 with 4 < m <= 8).  Plan GENERIC's arrays and iLQR loop with
                                 matrix-core sweeps: K, d within 1e-8 of the CPU path (measured 1e-13), not bit for bit.  MeritFunction
                                 runs a row-layout kernel per shape (kernels/ilqr_row32.hip: plan GENERIC's values)      */
} altro_hip_plan;

/* create flags */
#define ALTRO_HIP_STORE_QBLOCKS 0x1u /* also write Qxx,Quu,Qux,Qx,Qu (knotpoint_data.hpp:211-215) */
#define ALTRO_HIP_F32_PURE 0x2u      /* ALTRO_HIP_F32 on plan MFMA16: backward sweep in pure fp32 on
                                        v_mfma_f32_16x16x4_f32 (cost-to-go carried in fp32: ~5e-4 relative).
                                        Default for F32 is fp32 storage with fp64 tile arithmetic (2e-5): on
                                        MI355X both are bound by the same fp32 record traffic */
#define ALTRO_HIP_GENERIC_MATRIX_CORES 0x8u /* plan GENERIC, fp64: the backward sweep's products as v_mfma_f64_16x16x4 tiles (any
                                      * n, m, per-knot-point dimensions included) instead of one multiply-add at a time in the CPU
                                      * path's order: 1.0-1.9 x faster from n = 14 up, results equal to rounding
                                      * (1e-12) instead of bit for bit -- a whole AL-iLQR solve may then take a line-search decision
                                      * differently from the CPU path.  Off unless given; the tvlqr_* drop-in is always exact.      */

#define ALTRO_HIP_LANE_FUSED 0x4u    /* plan LANE: let the TVLQR kernels fuse a * b + c into one FMA.  Default off: the
                                        unfused kernels repeat the CPU path operation for operation and give
                                        bit-identical results; fused ones differ in the last bits (1e-12 relative)
                                        and are ~20 % faster where the sweep is instruction-bound (small batches)  */

/* Device dynamics/cost models for the nonlinear forward pass (user std::function callbacks of
 * typedefs.hpp:31-53 cannot run on the device; these are the compiled-in equivalents of the
 * reference's own test models, test/test_utils.cpp:18-238).                                    */
typedef enum altro_hip_model {
  ALTRO_HIP_MODEL_LINEAR = 0,            /* x+ = A_k x + B_k u + f_k from the uploaded data      */
  ALTRO_HIP_MODEL_DOUBLE_INTEGRATOR = 1, /* test_utils.cpp:18-41                                  */
  ALTRO_HIP_MODEL_PENDULUM = 2,          /* test_utils.cpp:43-82 + midpoint :84-132               */
  ALTRO_HIP_MODEL_BICYCLE = 3,           /* test_utils.cpp:134-238 + midpoint                     */
  ALTRO_HIP_MODEL_USER = 4,              /* set by altro_hip_set_model_source (not a value to pass) */
  ALTRO_HIP_MODEL_QUADROTOR = 5,         /* 12 states, 4 inputs, plan MFMA16 (csrc/models.h) + midpoint; not a reference model */
  ALTRO_HIP_MODEL_QUADROTOR13 = 6        /* 13 states (quaternion attitude), 4 inputs, plans GENERIC / MFMA32 + midpoint; not a reference model */
} altro_hip_model;

/* ---- library ------------------------------------------------------------------------------- */
int altro_hip_version(void);
const char* altro_hip_last_error(void);
int altro_hip_device_count(void);
/* name[cap], CU count and wave size of `device`; ALTRO_HIP_ERR_NO_DEVICE when there is none */
int altro_hip_device_info(int device, char* name, int cap, int* compute_units, int* wave_size);
/* "domain:bus:device.function" of a HIP device (hipDeviceGetPCIBusId): which physical GPU a rank of a multi-GPU run is
 * bound to -- bench.py prints it per rank.  cap >= 16.                                                                */
int altro_hip_device_pci_bus_id(int device, char* buf, int cap);

/* ---- lifecycle ------------------------------------------------------------------------------- */
/* `stream` is a hipStream_t (NULL = a stream owned by the handle).  `plan` is an altro_hip_plan. */
int altro_hip_batch_create(altro_hip_batch** out, int horizon_N, int n, int m, int batch,
                           int dtype, int plan, unsigned flags, int device, void* stream);
/* The same with PER-KNOT-POINT dimensions nx[0..N], nu[0..N-1] (ALTROSolver::SetDimension per range, altro_solver.cpp:26-47; the
 * kernel boundary takes nx[k], nu[k] throughout, tvlqr.cpp:65-248: A_k is nx[k+1] x nx[k], B_k nx[k+1] x nu[k], K_k nu[k] x nx[k]).
 * Plan GENERIC.  Such a handle serves
 *   - the TVLQR sweeps (every dimension in [1, 256]): altro_hip_set_dynamics / _set_cost / _set_initial_state, altro_hip_backward /
 *     _forward_ltv / _sweep, the getters, the statistics;
 *   - the iLQR loop (every dimension in [1, 64]): altro_hip_set_quadratic_cost, _set_input_guess, _set_state_guess,
 *     _add_linear_constraint (G is p x (nx[k] + nu[k]), p x nx[N] for a block of the terminal knot point alone; the knot points of
 *     one block must share their dimensions -- register it per range, like ALTROSolver::SetConstraint per index), altro_hip_open_loop_rollout / _merit / _expand / _accept / _stationarity,
 *     altro_hip_ilqr_solve, altro_hip_get_nominal / _get_knot (x: nx[k], u: nu[k] entries);
 * with every bulk array packed [b][k][block_k] (block_k column-major with knot point k's own dimensions: the reference's per-knot-
 * point blocks end to end; c is [b][N + 1]; altro_hip_set_tracking_cost takes Qd, xref [b][sum nx], Rd, uref [b][sum nu];
 * altro_hip_update_linear_costs takes the blocks of its range end to end).  k_stride_zero is refused, batch_stride_zero works.  The
 * calls whose meaning needs ONE dimension say so: altro_hip_shift_trajectory, the device models.                                  */
int altro_hip_batch_create_dims(altro_hip_batch** out, int horizon_N, const int* nx, const int* nu, int batch, int dtype,
                                unsigned flags, int device, void* stream);
void altro_hip_batch_destroy(altro_hip_batch* h);
int altro_hip_batch_plan(const altro_hip_batch* h);         /* the plan actually chosen          */
size_t altro_hip_batch_device_bytes(const altro_hip_batch* h);

/* ---- problem data (reference layout in, converted once to the device layout) ----------------- */
int altro_hip_set_dynamics(altro_hip_batch* h, const double* A, const double* B, const double* f,
                           int k_stride_zero, int batch_stride_zero);
/* Q, R symmetric, as the reference requires (altro_solver.hpp:183).  Plan MFMA16 stores the symmetric blocks once:
 * of Q_0..Q_{N-1} only the upper triangle is read, and altro_hip_get_P returns P_0..P_{N-1} mirrored from the upper
 * triangle the sweep keeps (exactly symmetric, within rounding of the reference's P). */
int altro_hip_set_cost(altro_hip_batch* h, const double* Q, const double* R, const double* H,
                       const double* q, const double* r, int is_diag, int k_stride_zero,
                       int batch_stride_zero);
int altro_hip_set_initial_state(altro_hip_batch* h, const double* x0, int batch_stride_zero);
/* Pointer mode of the BULK arrays: with device_pointers != 0 the A, B, f, Q, R, H, q, r, x0, u arguments of
 * altro_hip_set_dynamics / set_cost / set_initial_state / set_input_guess / update_linear_costs and the outputs of
 * altro_hip_get_K .. get_y / get_nominal / get_knot are DEVICE pointers (fp64, the same reference layout) on the
 * handle's device: kernels on the handle's stream read / write them in place, no staging copy -- for callers whose
 * linearisation already lives in HBM.  Everything else (status, delta_V, alpha / phi, results, duals, tracking-cost
 * arguments) stays in host memory.  Default 0.  Stream ordering is the caller's: the handle works on its own
 * stream, so work other streams still have pending on those arrays must be complete before the call; on return
 * the call's own reads / writes are complete (it synchronises its stream).                                */
int altro_hip_set_pointer_mode(altro_hip_batch* h, int device_pointers);
/* Host arrays of the following set_dynamics / set_cost / set_tracking_cost / set_input_guess calls hold
 * only `host_batch` distinct problems, tiled (b mod host_batch) over the batch on the device; 0 = off.
 * Lets a large synthetic batch be staged without a batch-sized host copy (plans MFMA16 and LANE).   */
int altro_hip_set_host_batch(altro_hip_batch* h, int host_batch);

/* ---- the hot path ----------------------------------------------------------------------------- */
/* tvlqr_BackwardPass over the batch: K, d, P, p, delta_V, status for every problem.             */
int altro_hip_backward(altro_hip_batch* h, double reg);
/* tvlqr_ForwardPass over the batch: x, u, y from x0 through u = d - K x, x+ = f + A x + B u.     */
int altro_hip_forward_ltv(altro_hip_batch* h);
/* One "sweep" of BASELINE.json's metric: backward followed by forward, no host sync in between. */
int altro_hip_sweep(altro_hip_batch* h, double reg);
/* Wait for the handle's stream. */
int altro_hip_synchronize(altro_hip_batch* h);

/* ---- results (reference layout out) ------------------------------------------------------------ */
int altro_hip_get_K(altro_hip_batch* h, double* K);             /* [batch][N][m*n]   */
int altro_hip_get_d(altro_hip_batch* h, double* d);             /* [batch][N][m]     */
int altro_hip_get_P(altro_hip_batch* h, double* P);             /* [batch][N+1][n*n] */
int altro_hip_get_p(altro_hip_batch* h, double* p);             /* [batch][N+1][n]   */
int altro_hip_get_x(altro_hip_batch* h, double* x);             /* [batch][N+1][n]   */
int altro_hip_get_u(altro_hip_batch* h, double* u);             /* [batch][N][m]     */
int altro_hip_get_y(altro_hip_batch* h, double* y);             /* [batch][N+1][n]   */
int altro_hip_get_delta_V(altro_hip_batch* h, double* delta_V); /* [batch][2]        */
int altro_hip_get_status(altro_hip_batch* h, int* status);      /* [batch]           */
/* Qxx|Quu|Qux|Qx|Qu per knot point, [batch][N][n*n+m*m+m*n+n+m]; needs ALTRO_HIP_STORE_QBLOCKS */
int altro_hip_get_qblocks(altro_hip_batch* h, double* qblocks);


/* ---- the iLQR loop around the sweep ------------------------------------------------------------------
 * Plan GENERIC (n or m beyond the tile, up to 64): dynamics are DATA, tracking or dense quadratic cost, MPC operations; one wave per
 * problem (kernels/ilqr_generic.hip), linear constraint blocks in every cone (round 4).  Device models and the regularisation retry:
 * plans LANE / MFMA16.
 * Plan LANE (n <= 6: BASELINE.json configs[2], [3]): nonlinear dynamics from a compiled-in device model
 * (altro_hip_set_model), constraint blocks, MPC operations.
 * Plan MFMA16 ((n, m) = (12, 4): configs[1], [4]): dynamics are DATA -- the A, B, f given to
 * altro_hip_set_dynamics, the reference's SetLinearDynamics path (knotpoint_data.cpp:123-142, :406-419,
 * :710-719) -- with the tracking cost below; no altro_hip_set_model.  From altro_hip_set_tracking_cost on, the
 * backward sweep of such a handle ignores f, as the reference's expansion does (f_.setZero(), :416), while the
 * rollout keeps it.  Constraint blocks (all four cones) and the regularisation schedule work on both plans.  */
/* Device model standing in for SetExplicitDynamics' host callbacks (altro_solver.cpp:68-81).  Plan LANE: ALTRO_HIP_MODEL_DOUBLE_INTEGRATOR,
 * _PENDULUM, _BICYCLE (the reference's test models).  Plan MFMA16 with (n, m) = (12, 4) and fp64 records: ALTRO_HIP_MODEL_QUADROTOR
 * (12-state rigid-body quadrotor, csrc/models.h; not a reference model) -- NONLINEAR dynamics on the tile plan: rollout and merit
 * evaluation step the model (explicit midpoint, test/test_utils.cpp:84-132), the expansion writes A, B into the records the backward
 * sweep reads (knotpoint_data.cpp:406-419); no altro_hip_set_dynamics then.                                                      */
int altro_hip_set_model(altro_hip_batch* h, int model, float timestep, int bicycle_frame,
                        double bicycle_length, double bicycle_lr);
/* ALTROSolver::SetExplicitDynamics (altro_solver.cpp:68-81) for the batched path, for dynamics the library does not ship:
 * the caller's host callbacks (typedefs.hpp:31-53) cannot run on the device, so the caller hands their device-side SOURCE
 * instead -- hand-written HIP, compiled with hiprtc into the library's own lane-per-problem kernels, once per (source, n, m,
 * dtype) and process (plan LANE: n <= 6, m <= 3).  `source` must define, for T = double and T = float,
 *     template <typename T> __device__ void altro_user_dynamics(const T* x, const T* u, T* xdot);   // xdot = f(x, u)
 *     template <typename T> __device__ void altro_user_jacobian(const T* x, const T* u, T* J);      // J = [df/dx df/du],
 * J column-major n x (n + m); the discretisation is the explicit midpoint rule with `timestep` (and its chain rule for
 * A, B), as in the reference's own test models (test/test_utils.cpp:84-132).  No system headers: the HIP device API and
 * the math functions (sin, cos, sincos, sqrt, ...) are built in.  A source that does not compile gives
 * ALTRO_HIP_ERR_BAD_ARGUMENT with the compiler's log in altro_hip_last_error().  Whole solves of such a handle run on the
 * launch-sequenced loop.                                                                                              */
int altro_hip_set_model_source(altro_hip_batch* h, const char* source, float timestep);
/* Plans GENERIC / MFMA32: 1 when the handle's device model (compiled in, or the caller's source) runs the loop's ROW-LAYOUT model kernels
 * (kernels/ilqr_row32.hip: two problems per wave, every lane evaluates the model -- plan MFMA32's shapes), 0 when it runs the
 * wave-per-problem ones.  A model from source gets the row layout when hiprtc compiled those kernels without scratch memory (a sparse
 * Jacobian written entry by entry does; a dense one of many states, or one filled by loops the compiler cannot unroll, does not) --
 * same results either way (the sums are taken in the same order), about 3 x the speed.  0 on other plans.                         */
int altro_hip_model_row_layout(const altro_hip_batch* h);
/* ALTROSolver::SetConstraint with a general (nonlinear) callback pair (altro_solver.cpp:192-223, typedefs.hpp:41-53): when
 * the source above also defines
 *     template <typename T> __device__ void altro_user_constraint(int id, const T* x, const T* u, T* c);            // c[p]
 *     template <typename T> __device__ void altro_user_constraint_jacobian(int id, const T* x, const T* u, T* J);   // p x (n+m)
 * (J column-major; u is zero at the terminal knot point), this registers block `id` of them -- p rows in `cone` at knot
 * points k_first..k_last -- next to the linear blocks; the AL terms treat it as the reference treats any constraint
 * (knotpoint_data.cpp:489-613: projected duals, Gauss-Newton in the Jacobian).  Returns the block id or a negative error. */
int altro_hip_add_user_constraint(altro_hip_batch* h, int k_first, int k_last, int cone, int p, int id);
/* ALTROSolver::SetLQRCost (altro_solver.cpp:138-172): Qd, xref [batch][N+1][n]; Rd, uref [batch][N][m];
 * with k_stride_zero Qd/xref hold {running, terminal} and Rd/uref one knot point.                   */
int altro_hip_set_tracking_cost(altro_hip_batch* h, const double* Qd, const double* Rd, const double* xref,
                                const double* uref, int k_stride_zero, int batch_stride_zero);
/* ALTROSolver::SetQuadraticCost (altro_solver.cpp:118-136 -> KnotPointData::SetQuadraticCost, knotpoint_data.cpp:64-85) for the
 * device iLQR loop: the dense cost  1/2 x'Q x + 1/2 u'R u + u'H x + q'x + r'u + c  per knot point, evaluated as
 * CalcOriginalCost / CalcOriginalCostGradient / CalcOriginalCostHessian do (knotpoint_data.cpp:624-634, :659-668, :691-698:
 * lx = Q x + H'u + q, lu = R u + H x + r, lxx = Q, luu = R, lux = H).  Column-major blocks as the reference takes them:
 * Q [batch][N+1][n*n] (symmetric, altro_solver.hpp:183), R [batch][N][m*m], H [batch][N][m*n], q [batch][N+1][n],
 * r [batch][N][m], c [batch][N+1] or NULL (zero); with k_stride_zero Q / q / c hold {running, terminal} and R / H / r one
 * knot point.  Replaces a tracking cost set before (and vice versa); altro_hip_update_linear_costs then updates q, r, c of
 * this cost.  Every plan: LANE (device models and run-time compiled ones; whole solves run on the launch-sequenced loop), MFMA16,
 * GENERIC.                                                                                                                          */
int altro_hip_set_quadratic_cost(altro_hip_batch* h, const double* Q, const double* R, const double* H, const double* q,
                                 const double* r, const double* c, int k_stride_zero, int batch_stride_zero);
/* ALTROSolver::SetInput over all knot points (altro_solver.cpp:242-251): u [batch][N][m]             */
int altro_hip_set_input_guess(altro_hip_batch* h, const double* u, int k_stride_zero, int batch_stride_zero);
/* ALTROSolver::SetState over all knot points (altro_solver.cpp:229-240): x [batch][N+1][n] into the CANDIDATE states x_.  A whole
 * solve overwrites them with its initial rollout (solver.cpp:421), so this matters to callers who sequence the phases themselves
 * (accept / expand / backward / merit on a state trajectory that is not a rollout: solver_impl_test.cpp:186-271 does exactly that). */
int altro_hip_set_state_guess(altro_hip_batch* h, const double* x, int k_stride_zero, int batch_stride_zero);
int altro_hip_open_loop_rollout(altro_hip_batch* h); /* SolverImpl::OpenLoopRollout solver.cpp:116-131 */
int altro_hip_accept(altro_hip_batch* h);            /* SolverImpl::CopyTrajectory  solver.cpp:148-157 */
int altro_hip_expand(altro_hip_batch* h);            /* dynamics + cost expansion at the candidate     */
/* SolverImpl::MeritFunction (solver.cpp:273-355): alpha [batch] (or one value when alpha_is_uniform),
 * phi / dphi [batch] out; get_x/u/y then return the candidate trajectory x_, u_, y_.               */
int altro_hip_merit(altro_hip_batch* h, const double* alpha, int alpha_is_uniform, int want_derivative,
                    double* phi, double* dphi);
int altro_hip_stationarity(altro_hip_batch* h, double* stationarity); /* solver.cpp:207-222, [batch]  */
int altro_hip_get_nominal(altro_hip_batch* h, double* x, double* u);
int altro_hip_get_expansion(altro_hip_batch* h, double* A, double* B, double* lx, double* lu);

/* ---- augmented-Lagrangian constraint blocks ---------------------------
 * ALTROSolver::SetConstraint (altro_solver.cpp:175-215) for constraints of the form
 *     c(x_k, u_k) = G [x_k; u_k] - g   in cone K,    k = k_first .. k_last (inclusive),
 * which covers every constraint of the reference's tests (goal, control bounds, second-order-cone bound).
 * G: host, p x (n+m) column-major, shared by the batch.  g: host, [p] or (g_per_problem) [batch][p].
 * cone: ConstraintType order of typedefs.hpp:29-34: 0 EQUALITY, 1 IDENTITY, 2 INEQUALITY (c <= 0),
 * 3 SECOND_ORDER_CONE (||c[0:p-1]|| <= c[p-1]).
 * Capacity (the reference appends constraints without limit, knotpoint_data.cpp:155-161; going past a limit here is an error that
 * names it, never a truncation):
 *   plan GENERIC       (and plan MFMA32, which shares its loop) 8 blocks per knot point, p <= 64 rows per block, a second-order cone
 *                      p <= 32 (one lane per row; the projection's Jacobian and curvature applied from their closed forms), 64 blocks
 *                      per handle;
 *   plan MFMA16        (the (12, 4) tile, fp64) 6 SLOTS of 8 rows per knot point -- a block in the zero / identity / orthant cones
 *                      takes ceil(p / 8) consecutive slots (p <= 48: those cones project row by row, so the host lays the rows out;
 *                      duals stay [p] per block), a second-order cone (p <= 4) one -- and 32 slots per handle: e.g. an input box
 *                      (8 rows) and a state box (24 rows) at every knot point with two slots to spare.  Knot points with up to 2 / 4 /
 *                      6 slots run the merit kernel's 2- / 4- / 6-slot instantiation (the last at one wave per SIMD: C1 with both
 *                      boxes solves in 1.27 x the time of the input box alone).  fp32 records: 2 slots;
 *   plan LANE          2 blocks per knot point, p <= 8 (SOC: p <= 4), 16 blocks per handle (the blocks ride registers of the
 *                      lane-per-problem kernels); rows of one cone can be stacked into one block.
 * Returns the block id (>= 0) or a negative error.  Duals and penalties live on the device per problem and,
 * like the reference's, persist from one solve to the next (warm-started MPC) until reset.  Every plan (GENERIC: n, m <= 64,
 * one wave per knot point).                                                                              */
int altro_hip_add_linear_constraint(altro_hip_batch* h, int k_first, int k_last, int cone, int p,
                                    const double* G, const double* g, int g_per_problem);
int altro_hip_clear_constraints(altro_hip_batch* h);
int altro_hip_reset_duals(altro_hip_batch* h, double penalty);
int altro_hip_get_duals(altro_hip_batch* h, int k, int slot, double* z /* [batch][p] */);
/* SolverImpl::Feasibility (solver.cpp:224-231) of the candidate trajectory, [batch].                  */
int altro_hip_feasibility(altro_hip_batch* h, double* out);

/* ---- MPC receding-horizon operations on the resident batch -------------
 * The caller pattern of test/bicycle_test.cpp:302-337: solve, read u_0, move the reference, set the new
 * initial state, shift the trajectory -- without repacking the problem between solves.
 *   altro_hip_shift_trajectory      ALTROSolver::ShiftTrajectory   (altro_solver.cpp:283-293)
 *   altro_hip_update_linear_costs   ALTROSolver::UpdateLinearCosts (altro_solver.cpp:266-281): q [nb][nk][n],
 *                                   r [nb][nk][m] or NULL, c [nb][nk] or NULL for knot points
 *                                   k_first..k_last (inclusive); nk = 1 if kz, nb = 1 if bz
 *   altro_hip_get_knot              ALTROSolver::GetState / GetInput (altro_solver.cpp:323-347) for the batch
 * (altro_hip_set_initial_state is ALTROSolver::SetInitialState, altro_solver.cpp:177-190.)               */
int altro_hip_shift_trajectory(altro_hip_batch* h);
int altro_hip_update_linear_costs(altro_hip_batch* h, const double* q, const double* r, const double* c,
                                  int k_first, int k_last, int kz, int bz);
int altro_hip_get_knot(altro_hip_batch* h, int k, double* x /* [batch][n] */, double* u /* [batch][m] */);

/* ALWAYS start from altro_hip_default_solve_options(&o) and change fields afterwards: the struct grows at its END between library
 * versions (altro_hip_version(); stop_when_running_at_most came with version 200), and a caller that fills it field by field leaves
 * the new fields uninitialised.                                                                                                  */
typedef struct altro_hip_solve_options { /* AltroOptions, solver_options.hpp:16-39 */
  int iterations_max;
  double tol_stationarity;
  double tol_primal_feasibility;
  double tol_meritfun_gradient;
  int use_backtracking_linesearch;
  double penalty_initial; /* solver_options.hpp:27-29; used when constraint blocks exist              */
  double penalty_scaling;
  double penalty_max;
  /* EXTENSION beyond the reference: the reference calls the backward pass with
   * reg = 0 and ignores a failed Cholesky (tvlqr.cpp:159-164, solver.cpp:363, :449).  With reg_retry_max > 0
   * a problem whose backward pass fails repeats it with reg <- max(reg * reg_scale, reg_min) (at most
   * reg_retry_max times per sweep, never above reg_max) and relaxes reg by reg_scale after a success.
   * Defaults (0, 0) reproduce the reference.                                                              */
  double reg_initial;
  int reg_retry_max;
  double reg_scale, reg_min, reg_max;
  /* EXTENSION beyond the reference, for batches: a batched solve lasts as long as its slowest problem (the reference solves one
   * problem per call and has no such notion).  With stop_when_running_at_most = k > 0 the call returns after the first sweep
   * that leaves at most k problems running; those keep their latest accepted iterate and report status 1 (Unsolved) with the
   * iterations they took.  Every problem that stopped on its own is untouched by this: its result is bit for bit the one of
   * the full solve.  Honoured by the launch-sequenced loop (plans MFMA16 and GENERIC; plan LANE runs that loop instead of its
   * one-launch kernel when k > 0).  Default 0: solve every problem to its own end, like the reference.                        */
  int stop_when_running_at_most;
  /* ---- version 300 ----
   * HOW the solve is executed: ALTRO_HIP_FORM_* bits, OR-ed with the handle's (altro_hip_set_forms).  0 = the defaults.  These were
   * process-environment switches (ALTRO_HIP_AFFINE, _NO_SPECULATION, _MERIT2, ...) until version 200: what changes iterates -- even
   * in the last bits -- belongs to the call, not to the process.                                                                   */
  unsigned forms;
  /* plan LANE: hand the problems still running after this many sweeps of the one-launch solve kernel over to the launch-sequenced
   * loop (the hand-over is exact at any sweep: a test hook).  0: the one-launch kernel runs every sweep.  -n: the one-launch solve
   * re-lists its still-running problems after every n sweeps whatever the batch size (the straggler compaction of batches beyond
   * one workgroup per compute unit, forced: a test hook too -- results are those of the single launch bit for bit).             */
  int fused_sweeps;
  /* The decision guard of the affine line-search rounds (plan MFMA16, dynamics as data), opt-in: with a margin > 0 a trial
   * whose phi, phi' would turn the search another way if they were off by that RELATIVE margin -- any comparison of
   * linesearch.cpp:37-351 and cubicspline.c, found by running the state machine at the corners of the box (linesearch_sm.h:
   * ls_feed_is_robust) -- is evaluated again as a rollout (SolverImpl::MeritFunction's own order, solver.cpp:285-316) and decided on
   * that.  With ALTRO_HIP_FORM_AFFINE_EXACT on top, values the search would KEEP and steps it ACCEPTS are evaluated as rollouts too,
   * and the solve is the rollout form's bit for bit (tools/fuzz_affine.py: 0 of 5617 problems differ, |dx| = 0).  Default 0 = off,
   * because a batch pays for it: some problem of 4096 is at a boundary in nearly every round, every such round then carries a rollout
   * launch, and the constrained C1 solve goes from 43 ms to 84 ms -- slower than ALTRO_HIP_FORM_ROLLOUT_ROUNDS (70 ms), which is the
   * form to ask for when the reference's own evaluation order matters more than the time.                                           */
  double decision_margin;
} altro_hip_solve_options;
/* altro_hip_solve_options::forms / altro_hip_set_forms.  Bit-for-bit equal to the default unless said otherwise. */
#define ALTRO_HIP_FORM_NO_SPECULATION 0x0001u    /* one line-search step per launch (no two-trial pass, no speculative backtracking steps) */
#define ALTRO_HIP_FORM_NO_RUNAHEAD 0x0002u       /* the host waits for every verdict before it enqueues on                    */
#define ALTRO_HIP_FORM_NO_MERIT2 0x0004u         /* plan MFMA16: phi(0) and the first step in separate launches                             */
#define ALTRO_HIP_FORM_MERIT_LDS 0x0008u         /* plan MFMA16: line-search rounds / altro_hip_merit in the LDS form (equal to rounding)   */
#define ALTRO_HIP_FORM_MERIT_DPP_ALWAYS 0x0010u  /* ... in the row-layout form whatever the launcher's rule                                 */
#define ALTRO_HIP_FORM_EXPAND_LDS 0x0020u        /* plan MFMA16: the LDS form of the expansion                                              */
#define ALTRO_HIP_FORM_ALROWS_LDS 0x0040u        /* plan MFMA16: dual update / feasibility in the LDS form                                  */
#define ALTRO_HIP_FORM_ROLLOUT_ROUNDS 0x0080u    /* plan MFMA16: every line-search trial a rollout, no affine rounds (equal to rounding, 1e-13;
                                                    the reference's own evaluation order)                                                    */
#define ALTRO_HIP_FORM_SEQUENCED 0x0100u         /* plan LANE: the launch-sequenced loop instead of the one-launch solve kernel             */
#define ALTRO_HIP_FORM_MERIT_ONE_LAUNCH 0x0200u  /* plan LANE: MeritFunction in one launch instead of three                                 */
#define ALTRO_HIP_FORM_LANE_QUAD_OFF 0x0400u     /* plan LANE, (4, 2) / (2, 1): one lane per problem whatever the batch                     */
#define ALTRO_HIP_FORM_LANE_QUAD_ON 0x0800u      /* ... four lanes per problem whatever the batch                                           */
#define ALTRO_HIP_FORM_GENERIC_LATE_Q_OFF 0x1000u /* plan GENERIC: the backward kernel with a block for Qxx whatever the occupancy rule     */
#define ALTRO_HIP_FORM_GENERIC_LATE_Q_ON 0x2000u
#define ALTRO_HIP_FORM_AFFINE_EXACT 0x8000u      /* plan MFMA16: affine rounds used for robust decisions only -- kept values and accepted steps are
                                                    evaluated as rollouts: the rollout form's results bit for bit, slower than ROLLOUT_ROUNDS (a checking form) */
#define ALTRO_HIP_FORM_FUSED_CLOCK 0x4000u       /* plan LANE: per-phase clock of the one-launch kernel on stderr (a tuning aid)            */
#define ALTRO_HIP_FORM_NO_COMPACTION 0x10000u    /* plan LANE: the one-launch solve of a batch beyond one workgroup per compute unit stays ONE
                                                    launch (default: chunks of sweeps over the list of still-running problems)              */
#define ALTRO_HIP_FORM_GENERIC_MERIT_LDS 0x20000u /* plans GENERIC / MFMA32: MeritFunction by the wave-per-problem LDS kernel also where the
                                                    row-layout one runs (uniform n <= 31, m <= 8, n + m <= 32: kernels/ilqr_row32.hip)        */
/* forms of a HANDLE: what altro_hip_merit / _expand / _sweep and every solve on it run with (a solve ORs its options' bits in) */
int altro_hip_set_forms(altro_hip_batch* h, unsigned forms);
unsigned altro_hip_get_forms(const altro_hip_batch* h);
typedef struct altro_hip_solve_result { /* AltroStats per problem, solver_stats.hpp:14-25 */
  int status;     /* SolveStatus: 0 Success, 1 Unsolved, 2 MaxIterations (typedefs.hpp:19-27)        */
  int iterations; /* solver.cpp:506                                                                  */
  double stationarity;
  double final_alpha;
  double final_phi;
  double primal_feasibility; /* AltroStats::primal_feasibility (solver.cpp:508)                      */
  double penalty;            /* the constraints' penalty when the problem stopped                     */
  int dual_updates;          /* outer (dual) updates taken                                            */
  int reg_retries;           /* extension: backward passes repeated with a larger regularisation      */
} altro_hip_solve_result;
void altro_hip_default_solve_options(altro_hip_solve_options* opts);
/* SolverImpl::Solve (solver.cpp:414-511) for the whole batch; results [batch] (may be NULL).
 * While the problems still searching leave part of the GPU idle, line-search steps that are known in advance are
 * evaluated speculatively -- the first step (alpha = 1) in the launch that evaluates phi(0), the backtracking steps
 * alpha beta^j several per launch -- and consumed by the search in its own order: every result is bit-identical to
 * the one-step-per-launch sequence, only altro_hip_last_solve_counts' merit_launches drops.  ALTRO_HIP_FORM_NO_SPECULATION
 * restores one step per launch.
 * Plan MFMA16, fp64, dynamics given as data: the closed-loop rollout of SolverImpl::MeritFunction (solver.cpp:273-355) is affine in
 * the step, x_k(alpha) = x_k(0) + alpha dx_k/dalpha, so the line-search rounds after a sweep's first step evaluate the knot points
 * independently from that pair instead of rolling out again (16 per wavefront; a round then costs the same whatever the horizon).
 * Their trial points equal the rollout's to rounding (1e-13), as everything on this plan does: a search whose decision sits on a
 * boundary at that level may turn another way (tools/fuzz_affine.py: 25 of 5617 random constrained problems -- all of them problems
 * whose search FAILS in one of the two forms after 13-37 sweeps -- end with another status or iteration count; the converged rest
 * within 5e-10).  ALTRO_HIP_FORM_ROLLOUT_ROUNDS keeps every trial a rollout; altro_hip_solve_options::decision_margin and
 * ALTRO_HIP_FORM_AFFINE_EXACT are the guarded forms (slower than either).                                                        */
int altro_hip_ilqr_solve(altro_hip_batch* h, const altro_hip_solve_options* opts,
                         altro_hip_solve_result* results);
int altro_hip_last_solve_counts(const altro_hip_batch* h, int* sweeps, int* merit_launches);

/* Results WHILE the batch still solves.  A batched solve lasts as long as its slowest problem (8192 steering-bounded bicycles:
 * 3.3 sweeps on average, a handful run all 80), but an MPC caller can use every problem the moment it stops.  On the
 * one-launch path (plan LANE with a compiled-in device model) the solve kernel publishes each problem into a record in
 * pinned host memory when it stops: its AltroStats and the first input of its solution.
 *   altro_hip_ilqr_solve_async  starts the solve and returns at once (ALTRO_HIP_ERR_UNSUPPORTED where only the
 *                               launch-sequenced loop can run: use altro_hip_ilqr_solve there)
 *   altro_hip_ilqr_poll         never blocks: *n_done = records published so far, *records = the [batch] array (valid until
 *                               the handle's next solve or its destruction); record b is complete once records[b].done != 0
 *   altro_hip_ilqr_wait         blocks until the launch has ended; results [batch] as altro_hip_ilqr_solve fills them (may be
 *                               NULL).  Any other call on the handle waits for the launch as well.                          */
typedef struct altro_hip_poll_record {
  altro_hip_solve_result result;
  double u0[4];      /* u_0 of the solution (entries >= m are 0): what a receding-horizon loop applies (bicycle_test.cpp:313) */
  int done;          /* set last (release order) */
  int reserved;
} altro_hip_poll_record;
int altro_hip_ilqr_solve_async(altro_hip_batch* h, const altro_hip_solve_options* opts);
int altro_hip_ilqr_poll(altro_hip_batch* h, int* n_done, const altro_hip_poll_record** records);
int altro_hip_ilqr_wait(altro_hip_batch* h, altro_hip_solve_result* results);

/* The batched solver's line search is CubicLineSearch (src/linesearch/linesearch.cpp:37-412) recast as
 * a resumable state machine (altro_amd/csrc/linesearch_sm.h).  This host entry drives that same code
 * with a callback so that it can be pinned against the reference line search without a GPU.         */
typedef void (*altro_hip_merit_fn)(double alpha, double* phi, double* dphi, void* ctx);
double altro_hip_linesearch_host(altro_hip_merit_fn f, void* ctx, double alpha0, double phi0, double dphi0,
                                 int try_cubic_first, int use_backtracking, double c1, double c2,
                                 int* status, int* iters, double* phi, double* dphi);

/* ---- statistics: the only thing that ever crosses GPUs ---------------------
 * What SolverImpl::Solve reports per solver (solver.cpp:464-469, :492-509; AltroStats, solver_stats.hpp:14-25),
 * reduced over the problems of a batch ON THE DEVICE (one deterministic two-stage reduction kernel pair; nothing
 * but this struct travels to the host), and -- across GPUs -- by two in-place ncclAllReduce calls (RCCL over
 * xGMI) on the handle's stream: the sums as one ncclSum over doubles (counts are exact below 2^53), the maxima
 * as one ncclMax.  Problem instances are sharded; there is no other collective on the path.                  */
typedef struct altro_hip_stats {
  int64_t problems;          /* problems reduced over (batch, or the global batch after an all-reduce)       */
  int64_t cholesky_failures; /* problems whose last backward pass failed (status != -1, tvlqr.cpp:162-164)   */
  int64_t converged;         /* SolveStatus::Success after the last altro_hip_ilqr_solve (0 before any)      */
  int64_t iterations;        /* sum of AltroStats::iterations (solver.cpp:506)                               */
  double sum_cost;           /* sum of the final merit value phi (AltroStats::objective_value)               */
  double sum_delta_V0;       /* sum of delta_V[0] over the problems whose backward pass succeeded            */
  double sum_delta_V1;       /* sum of delta_V[1]                                                            */
  double max_stationarity;   /* max over problems (solver.cpp:207-222)                                       */
  double max_feasibility;    /* max over problems (solver.cpp:224-231)                                       */
  double max_abs_xN;         /* max |x_N| over problems (after a forward pass / solve)                       */
  int64_t non_finite;        /* problems whose cost, stationarity, feasibility or x_N is NaN / Inf (summed over
                                GPUs like the other counts: ncclMax need not carry a NaN, this count does)       */
} altro_hip_stats;
/* Within a batch a NaN in any problem's stationarity / feasibility / x_N makes the corresponding maximum NaN (it is
 * not dropped, matching the per-problem results); across GPUs rely on non_finite.                                  */
/* this handle's problems only */
int altro_hip_stats_reduce(altro_hip_batch* h, altro_hip_stats* out);

/* One communicator rank per GPU.  RCCL is resolved at run time (dlopen of the librccl.so.1 the process already
 * uses, else the system one): the library has no link-time dependency on it.
 *   one process per GPU (the launch model of bench.py / torchrun / mpirun): rank 0 calls altro_hip_comm_unique_id,
 *     the host program hands the 128 bytes to every rank by whatever channel it has, every rank calls
 *     altro_hip_comm_create (ncclCommInitRank);
 *   one process driving several GPUs: altro_hip_comm_create_all (ncclCommInitAll), then
 *     altro_hip_stats_allreduce_multi with one handle per device (the calls are grouped, ncclGroupStart/End). */
typedef struct altro_hip_comm altro_hip_comm;
#define ALTRO_HIP_COMM_ID_BYTES 128
int altro_hip_comm_unique_id(void* id /* [ALTRO_HIP_COMM_ID_BYTES] out */);
int altro_hip_comm_create(altro_hip_comm** out, int device, int rank, int world, const void* id);
int altro_hip_comm_create_all(altro_hip_comm** out /* [ndev] */, int ndev, const int* devices);
void altro_hip_comm_destroy(altro_hip_comm* c);
/* what the communicator itself was built with: rank, number of ranks (= GPUs taking part), HIP device */
int altro_hip_comm_rank(const altro_hip_comm* c);
int altro_hip_comm_world(const altro_hip_comm* c);
int altro_hip_comm_device(const altro_hip_comm* c);
/* local device-side reduction, then the two all-reduces on h's stream; `out` holds the global statistics on every rank */
int altro_hip_stats_allreduce(altro_hip_batch* h, altro_hip_comm* comm, altro_hip_stats* out);
int altro_hip_stats_allreduce_multi(altro_hip_batch* const* handles, altro_hip_comm* const* comms, int n,
                                    altro_hip_stats* out);

/* ---- measurement -------------------------------------------------------------------------------- */
/* Every launch of the sweep kernels is bracketed by hipEvents ON THE HANDLE'S STREAM (torch.cuda.Event would only see
 * torch's current stream) and the elapsed times are accumulated per kernel: slot 0 = backward kernel, 1 = forward
 * kernel.  enable = 1: the call waits for each launch (use outside timed regions); enable = 2: events only -- nothing
 * waits between launches, altro_hip_profile_get / _get_range synchronise the stream and read up to 4096 recorded
 * launches -- so the launches of a timed region are measured where they run; 0 = off.                           */
int altro_hip_profile_enable(altro_hip_batch* h, int enable);
int altro_hip_profile_reset(altro_hip_batch* h);
int altro_hip_profile_get(altro_hip_batch* h, int slot, int* launches, double* total_ms,
                          const char** kernel_name);
int altro_hip_profile_get_range(altro_hip_batch* h, int slot, double* min_ms, double* max_ms);
/* enable = 2: launches of the slot's kernel issued after the 4096-launch event window was full since the last reset
 * (they ran, but are not part of the totals above); 0 means the averages cover every launch.                        */
int altro_hip_profile_dropped(altro_hip_batch* h, int slot);
/* Algorithmic bytes one launch of the slot's kernel must move.             */
double altro_hip_algorithmic_bytes(const altro_hip_batch* h, int slot);

#ifdef __cplusplus
}
#endif
#endif /* ALTRO_HIP_H_ */
