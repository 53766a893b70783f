// kernels/ilqr_tile_model.hip -- NONLINEAR dynamics on the (12, 4) tile plan (plan MFMA16's iLQR loop in the row layout of
// ilqr_merit2_dpp.hip): what KnotPointData::CalcDynamics / CalcDynamicsExpansion do with the caller's callback pair
// (knotpoint_data.cpp:406-419, :710-719; ALTROSolver::SetExplicitDynamics, altro_solver.cpp:68-81) for a device model
// (models.h: the continuous f and its Jacobian; the explicit midpoint rule and its chain rule, test/test_utils.cpp:84-132).
//
// The point [x; u] of a (problem, trial) lives in the registers of one row of 16 lanes (lane j < 12: x_j, lane 12 + i: u_i).
// One step:
//   * every lane gathers the whole [x; u] (16 DPP moves) and evaluates the continuous model itself -- the same instructions
//     in every lane, so one evaluation's worth of issue per wave -- at (x, u) and at the midpoint; lane j keeps x+_j.  What a
//     model can share over the row it does: MODEL_QUADROTOR takes ONE sincos per lane and evaluation (tile_quad_trig);
//   * for the expansion lane j keeps ROW j of the two continuous Jacobians J0 = [df/dx df/du](x, u), Jm = (...)(xm, u) and forms
//     row j of Z = [A B],  A = I + h Jm_x (I + h/2 J0_x),  B = h (Jm_x h/2 J0_u + Jm_u):  the products  sum_i Jm[j][i] J0[i][c]
//     are sums over the NONZERO entries of J0, which every lane holds whole (it evaluated the model itself) -- nothing crosses LDS,
//     nothing is broadcast; the lane picks its row of Jm only.
// wave_merit_dpp_kernel<.., MK> (ilqr_merit2_dpp.hip) calls tile_model_step at every knot point and leaves the rows in the DYN
// records for the next backward sweep; here are the two kernels around it: the open-loop rollout and the expansion of a stored
// trajectory (the head of Solve, and the re-expansion after a speculative line-search step).
#pragma once

namespace altro_hip {

// lanes 0..15 of the row, to every lane of the row
__device__ __forceinline__ void md_gather16(double v, double (&o)[16]) {
  asm volatile("s_nop 4\n"
               "v_mov_b64_dpp %0, %16" MD_BC(0) "v_mov_b64_dpp %1, %16" MD_BC(1) "v_mov_b64_dpp %2, %16" MD_BC(2) "v_mov_b64_dpp %3, %16" MD_BC(3)
               "v_mov_b64_dpp %4, %16" MD_BC(4) "v_mov_b64_dpp %5, %16" MD_BC(5) "v_mov_b64_dpp %6, %16" MD_BC(6) "v_mov_b64_dpp %7, %16" MD_BC(7)
               "v_mov_b64_dpp %8, %16" MD_BC(8) "v_mov_b64_dpp %9, %16" MD_BC(9) "v_mov_b64_dpp %10, %16" MD_BC(10) "v_mov_b64_dpp %11, %16" MD_BC(11)
               "v_mov_b64_dpp %12, %16" MD_BC(12) "v_mov_b64_dpp %13, %16" MD_BC(13) "v_mov_b64_dpp %14, %16" MD_BC(14) "v_mov_b64_dpp %15, %16" MD_BC(15)
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]),
        "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15])
      : "v"(v));
}

// Row jr of a 12 x 16 column-major Jacobian every lane holds in full: a chain of selects per column.  Entries that are the same
// constant in every row (the structural zeros of a model's Jacobian) fold away, so this costs a select per NONZERO, not per entry.
__device__ __forceinline__ void tile_pick_row(const double (&J)[192], int jr, double (&row)[16]) {
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < 12; ++r) v = (jr == r) ? J[r + 12 * c] : v;
    row[c] = v;
  }
}

// MODEL_QUADROTOR's trigonometry once per ROW: lane 3, 4, 5 hold phi, theta, psi, every lane takes sin / cos of ONE number (the
// other lanes of 0) and the six results reach the row by DPP -- a third of the library calls of quadrotor_trig, same values.
__device__ __forceinline__ QuadTrig<double> tile_quad_trig(double own) {
  double s, c;
  sincos_hd<double>(own, &s, &c);
  QuadTrig<double> t;
  asm volatile("s_nop 4\n"
               "v_mov_b64_dpp %0, %6" MD_BC(3) "v_mov_b64_dpp %1, %7" MD_BC(3) "v_mov_b64_dpp %2, %6" MD_BC(4) "v_mov_b64_dpp %3, %7" MD_BC(4)
               "v_mov_b64_dpp %4, %6" MD_BC(5) "v_mov_b64_dpp %5, %7" MD_BC(5)
      : "=&v"(t.sp), "=&v"(t.cp), "=&v"(t.st), "=&v"(t.ct), "=&v"(t.ss), "=&v"(t.cs)
      : "v"(s), "v"(c));
  t.ict = 1.0 / t.ct;
  return t;
}

// The continuous model at (x, u), every lane the whole of it; `jr` is the lane's row (for the models that share work over the row).
template <int MK, bool JAC>
__device__ __forceinline__ void tile_cont(const ModelParams& mp, const double* x, const double* u, int jr, double* xdot, double* J) {
  using M = DiscreteModel<MK, 12, 4, double>;
  if constexpr (MK == MODEL_QUADROTOR) {
    const double own = jr == 3 ? x[3] : (jr == 4 ? x[4] : (jr == 5 ? x[5] : 0.0));
    const QuadTrig<double> t = tile_quad_trig(own);
    quadrotor_f_from<double>(t, x, u, xdot);
    if constexpr (JAC) quadrotor_J_from<double>(t, x, u, J);
  } else {
    if constexpr (JAC) M::cont_fJ(mp, x, u, xdot, J);
    else M::cont_f(mp, x, u, xdot);
  }
}

// One explicit-midpoint step of model MK from the row's registers: lane j < 12 gets x+_j and, with JAC, row j of Z = [A B] of the
// step (lanes 12..15 compute along as row 11 and must not use either).  Must be called by every lane of the row.
template <int MK, bool JAC>
__device__ __forceinline__ void tile_model_step(const ModelParams& mp, double w, int jr, double& xnext, double (&zrow)[16]) {
  double z[16];
  md_gather16(w, z);
  const double h = (double)mp.h, h2 = (double)(mp.h / 2);     // (h / 2 in float arithmetic, like the reference's harness)
  double k1[12], xm[12], k2[12];
  if constexpr (JAC) {
    double J0[192], Jm[192], jm[16];
    tile_cont<MK, true>(mp, z, z + 12, jr, k1, J0);
#pragma unroll
    for (int i = 0; i < 12; ++i) xm[i] = z[i] + h2 * k1[i];
    tile_cont<MK, true>(mp, xm, z + 12, jr, k2, Jm);
    tile_pick_row(Jm, jr, jm);                          // the lane's row of Jm; J0 it holds whole (every lane evaluated it)
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      double s = 0.0;                                   // sum_i Jm[j][i] J0[i][c], without the terms whose J0[i][c] is a structural zero of the
#pragma unroll                                          // model (the quadrotor keeps 38 of 192); fused like the DPP chain this replaces
      for (int i = 0; i < 12; ++i) {
        if (__builtin_constant_p(J0[i + 12 * c]) && J0[i + 12 * c] == 0.0) continue;
        s = __builtin_fma(J0[i + 12 * c], jm[i], s);
      }
      if (c < 12) zrow[c] = ((jr == c) ? 1.0 : 0.0) + h * (jm[c] + h2 * s);
      else zrow[c] = h * (h2 * s + jm[c]);
    }
  } else {
    tile_cont<MK, false>(mp, z, z + 12, jr, k1, nullptr);
#pragma unroll
    for (int i = 0; i < 12; ++i) xm[i] = z[i] + h2 * k1[i];
    tile_cont<MK, false>(mp, xm, z + 12, jr, k2, nullptr);
  }
  double kj = 0.0;
#pragma unroll
  for (int i = 0; i < 12; ++i) kj = (jr == i) ? k2[i] : kj;
  xnext = w + h * kj;                                   // (w is x_j in the lanes that keep the result)
}

// SolverImpl::OpenLoopRollout (solver.cpp:116-131) with a device model: x_0 = x0, x_{k+1} = F(x_k, u_k) on the candidate
// trajectory, whose inputs are the guess already stored there.  Four problems per wave, one per row of 16 lanes; no LDS.
template <typename S, int MK>
__global__ __launch_bounds__(64) void wave_rollout_model_kernel(IlqrWaveArgs<S> a) {
  const ModelParams mp = a.mp;
  const int lane = threadIdx.x, j = lane & 15;
  const int b0 = (int)blockIdx.x * 4, b_own = b0 + (lane >> 4);
  bool on = b_own < a.batch;
  if (on && a.active && !a.active[b_own]) on = false;
  const unsigned long long onm = __ballot(on);
  if (onm == 0ull) return;
  const int b = on ? b_own : b0 + (__builtin_ctzll(onm) >> 4);   // a row without a problem shadows a live one and stores nothing
  const bool isx = j < 12;
  const int jr = isx ? j : 11;
  S* __restrict__ candb = a.cand + (size_t)b * a.xuy_bs;
  double x = (double)a.x0[(size_t)b * 12 + jr];
  double unext = (double)candb[12 + j];            // u_(j-12) of knot point 0 for the lanes that hold an input (x lanes: unused)
  for (int k = 0; k < a.N; ++k) {
    const double w = isx ? x : unext;
    unext = (double)candb[(size_t)(k + 1 < a.N ? k + 1 : k) * a.xuy_ks + 12 + j];
    if (on && isx) candb[(size_t)k * a.xuy_ks + j] = (S)x;
    double xn, unused[16];
    tile_model_step<MK, false>(mp, w, jr, xn, unused);
    x = xn;
  }
  if (on && isx) candb[(size_t)a.N * a.xuy_ks + j] = (S)x;
}

// CalcDynamicsExpansion (knotpoint_data.cpp:406-419) at every knot point of the stored candidate trajectory: Z_k = [A_k B_k]
// (and f_k = 0: the expansion carries no affine term, :416) into the DYN records the backward sweep reads.  Four (problem, knot
// point) pairs per wave like wave_expand_dpp_kernel.
template <typename S, int MK>
__global__ __launch_bounds__(64) void wave_expand_dyn_kernel(IlqrWaveArgs<S> a) {
  const ModelParams mp = a.mp;
  const int lane = threadIdx.x, j = lane & 15;
  const int wpk = (a.batch + 3) >> 2;
  const int k = (int)(blockIdx.x / wpk), b0 = (int)(blockIdx.x % wpk) * 4, b_own = b0 + (lane >> 4);
  if (k >= a.N) return;
  bool on = b_own < a.batch;
  if (on && a.active && !a.active[b_own]) on = false;
  const unsigned long long onm = __ballot(on);
  if (onm == 0ull) return;
  const int b = on ? b_own : b0 + (__builtin_ctzll(onm) >> 4);
  const bool isx = j < 12;
  const int jr = isx ? j : 11;
  const S* c = a.cand + (size_t)b * a.xuy_bs + (size_t)k * a.xuy_ks;
  const double w = isx ? (double)c[j] : (double)c[12 + j];
  double xn, zrow[16];
  tile_model_step<MK, true>(mp, w, jr, xn, zrow);
  if (on && isx) {
    S* z = const_cast<S*>(a.dyn) + (size_t)b * a.dyn_bs + (size_t)k * a.dyn_ks;
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) z[MF_OFF_Z + j * 16 + cc] = (S)zrow[cc];
    z[MF_OFF_F + j] = S(0);
  }
}

}  // namespace altro_hip
