// kernels/ilqr_merit2_dpp.hip -- plan MFMA16's iLQR kernels in the ROW LAYOUT: the point [x; u] of a (problem, trial) lives in the
// registers of one row of 16 lanes and every product against it is a chain of `v_fmac_f64_dpp ... row_newbcast` instructions.
//   wave_merit_dpp_kernel<S, AL, DUAL>   MeritFunction (solver.cpp:273-355): the two-trial pass and the line-search rounds
//   wave_expand_dpp_kernel               CalcExpansions with constraint blocks (solver.cpp:189-201, knotpoint_data.cpp:537-613)
//   wave_dual_update_dpp_kernel          DualUpdate (solver.cpp:383-388, knotpoint_data.cpp:503-510)
//   wave_feasibility_dpp_kernel          Feasibility (solver.cpp:224-231)
// Each replaces an LDS-form kernel of ilqr_mfma16.hip and is bit-identical to it (tests/test_gpu_merit2.py, tools/fuzz_dpp.py).
//
// The first of them: wave_merit2_kernel's two-trial merit evaluation (phi(0) and the line search's first step alpha0 = 1 in one
// pass over the records) with the operand broadcast moved from the LDS pipe to the VALU's data-parallel primitives, and TWO
// PROBLEMS PER WAVE.
//
// wave_merit2_kernel is bound by the LDS pipe: ~100 LDS instructions per knot point, most of them every
// lane re-reading the vector its row multiplies.  tools/ldsbench.hip prices an LDS wave instruction at 1.2-2.1 ns of the
// CU's one pipe (whatever the width, broadcast or not, 16 lanes active or 64) and `v_fmac_f64_dpp ... row_newbcast:c` --
// acc += (lane c of my row of 16 lanes) * coefficient, gfx90a+ -- at the price of a plain v_fma_f64 on the SIMD's own VALU.
// So a trial lives in ONE ROW OF 16 LANES:
//     lane j < 12 : x_j, dx_j/dalpha           -- row j of Z = [A B] (next state), row j of [P | p] (y)
//     lane 12 + i : u_i, du_i/dalpha           -- row i of Kt = [K | -d]
// the vector [x; u] IS the row's registers, every product is sixteen (twelve) DPP multiply-adds, and nothing of the
// recursion goes through LDS.  A wave has four such rows: two problems x two trials (lane = 32 * slot + 16 * trial + j).
// What still crosses LDS is the transposition of the records -- loaded coalesced, 16 bytes per lane, each half wave its own
// problem's; read back as rows of coefficients -- and one column read of Z per knot point for the stationarity.
//
// Per trial every sum is taken in wave_merit2_kernel's order with its expressions (same FMA contraction), and the final
// wave sums are taken over the same 32-entry arrangement, so phi, dphi, the candidate trajectory, the gradient and the
// stationarity are BIT-IDENTICAL to that kernel's (tests/test_gpu_merit2.py).  Constraint blocks ride in the same rows (dpp_al_rows below).
#pragma once

namespace altro_hip {

constexpr int MD_ZLD = 18;                   // Z rows padded to 18: rows stay 16-byte aligned, 16 lanes reading a column hit 16 bank pairs
constexpr int MD_OUT0 = 12 * MD_ZLD;         // 216
constexpr int MD_F0 = MD_OUT0 + MF_OUT;      // 360
constexpr int MD_NOM0 = MD_F0 + 12;          // 372
constexpr int MD_CP0 = MD_NOM0 + MF_NOM;     // 388
constexpr int MD_IMG = MD_CP0 + MF_COSTP;    // 424
constexpr int MD_IMG_DENSE = MD_CP0 + MF_COST;   // 548: a dense quadratic cost's record (IlqrWaveArgs::costd) instead of the 36 parameters
static_assert(MF_DYN % 2 == 0 && MF_OUT % 2 == 0 && MF_NOM % 2 == 0 && MF_COSTP % 2 == 0 && MD_IMG % 2 == 0, "records move as pairs");

typedef double md_d2 __attribute__((ext_vector_type(2)));
typedef float md_f2 __attribute__((ext_vector_type(2)));
template <typename S> struct md_pair_of;
template <> struct md_pair_of<double> { typedef md_d2 type; };
template <> struct md_pair_of<float> { typedef md_f2 type; };
template <typename S>
__device__ __forceinline__ md_d2 md_ld(const S* __restrict__ p, int pair) {
  const typename md_pair_of<S>::type v = *reinterpret_cast<const typename md_pair_of<S>::type*>(p + 2 * pair);
  return md_d2{(double)v[0], (double)v[1]};
}
// one knot point's records of one problem, spread over 32 lanes  (DENSE: the 160-element dense cost record, 80 pairs)
template <bool DENSE = false>
struct MeritPairRegs { md_d2 z[3], f, o[3], nm, cp[DENSE ? 3 : 1]; };
// (MODEL: the dynamics come from a device model, kernels/ilqr_tile_model.hip -- Z and f are neither loaded nor staged)
template <typename S, bool DENSE = false, bool MODEL = false>
__device__ __forceinline__ void merit_pair_load(MeritPairRegs<DENSE>& r, const S* __restrict__ z, const S* __restrict__ o,
                                                const S* __restrict__ nm, const S* __restrict__ cp, int hl) {
  if constexpr (!MODEL) {
#pragma unroll
    for (int c = 0; c < 3; ++c) r.z[c] = md_ld<S>(z, c * 32 + hl);         // Z: 96 pairs
    r.f = md_ld<S>(z, 96 + (hl < 6 ? hl : 5));                             // f: 6 pairs
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) r.o[c] = md_ld<S>(o, c * 32 + hl);           // OUT: 72 pairs
  r.o[2] = md_ld<S>(o, 64 + (hl & 7));
  r.nm = md_ld<S>(nm, hl & 7);                                             // nominal: 8 pairs
  if constexpr (DENSE) {                                                   // dense cost record: 80 pairs
    r.cp[0] = md_ld<S>(cp, hl); r.cp[1] = md_ld<S>(cp, 32 + hl); r.cp[2] = md_ld<S>(cp, 64 + (hl & 15));
  } else {
    r.cp[0] = md_ld<S>(cp, hl < 18 ? hl : 17);                             // cost parameters: 18 pairs
  }
}
template <bool DENSE, bool MODEL = false>
__device__ __forceinline__ void merit_pair_stage(const MeritPairRegs<DENSE>& r, double* __restrict__ L, int hl) {
  if constexpr (!MODEL) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int pi = c * 32 + hl;
      *reinterpret_cast<md_d2*>(L + (pi >> 3) * MD_ZLD + (pi & 7) * 2) = r.z[c];
    }
    *reinterpret_cast<md_d2*>(L + MD_F0 + 2 * (hl < 6 ? hl : 5)) = r.f;
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) *reinterpret_cast<md_d2*>(L + MD_OUT0 + 2 * (c * 32 + hl)) = r.o[c];
  *reinterpret_cast<md_d2*>(L + MD_OUT0 + 2 * (64 + (hl & 7))) = r.o[2];
  *reinterpret_cast<md_d2*>(L + MD_NOM0 + 2 * (hl & 7)) = r.nm;
  if constexpr (DENSE) {
    *reinterpret_cast<md_d2*>(L + MD_CP0 + 2 * hl) = r.cp[0];
    *reinterpret_cast<md_d2*>(L + MD_CP0 + 2 * (32 + hl)) = r.cp[1];
    *reinterpret_cast<md_d2*>(L + MD_CP0 + 2 * (64 + (hl & 15))) = r.cp[2];
  } else {
    *reinterpret_cast<md_d2*>(L + MD_CP0 + 2 * (hl < 18 ? hl : 17)) = r.cp[0];
  }
}

// acc += (lane N of this lane's row of 16 lanes of v) * coef.  The hazard recogniser does not look inside inline assembly:
// a DPP read needs 2 wait states after a VALU write of its source (5 after a VALU write of EXEC) -- every block of these
// instructions therefore starts with `s_nop 4`, and no block reads a register that the block itself writes through DPP.
// (Accumulators are early-clobber operands: a block writes them before it has read all its inputs, so an input that happens to
//  hold the same value -- a literal zero next to a sum that starts at zero -- must not be given the accumulator's register.)
// (volatile: a block must stay where the whole wave executes it -- sunk into a branch only some lanes take, its broadcasts
//  would read lanes that are switched off.)
#define MD_BC(N) " row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n"
// block A: %0 += bc(%2) * coef, %1 += bc(%3) * coef      (two sums against the same twelve coefficients %4 .. %15)
#define MD_A(N, OP) "v_fmac_f64_dpp %0, %2, %" #OP MD_BC(N) "v_fmac_f64_dpp %1, %3, %" #OP MD_BC(N)
__device__ __forceinline__ void md_rows12(double& acc, double& acc2, double v, double v2, const double (&c)[13]) {
  asm volatile("s_nop 4\n" MD_A(0, 4) MD_A(1, 5) MD_A(2, 6) MD_A(3, 7) MD_A(4, 8) MD_A(5, 9) MD_A(6, 10) MD_A(7, 11) MD_A(8, 12) MD_A(9, 13)
      MD_A(10, 14) MD_A(11, 15)
      : "+&v"(acc), "+&v"(acc2)
      : "v"(v), "v"(v2), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]), "v"(c[8]), "v"(c[9]),
        "v"(c[10]), "v"(c[11]));
}
// block B: rows of Z = [A B] against [x; u] (%4) and its sensitivity (%5): the A part into %0 / %1, the B part into %2 / %3
#define MD_B(N, OP) "v_fmac_f64_dpp %0, %4, %" #OP MD_BC(N) "v_fmac_f64_dpp %1, %5, %" #OP MD_BC(N)
#define MD_BU(N, OP) "v_fmac_f64_dpp %2, %4, %" #OP MD_BC(N) "v_fmac_f64_dpp %3, %5, %" #OP MD_BC(N)
__device__ __forceinline__ void md_rows16(double& acc, double& acc2, double& s2, double& t2, double w, double dw, const double (&c)[16]) {
  asm volatile("s_nop 4\n" MD_B(0, 6) MD_B(1, 7) MD_B(2, 8) MD_B(3, 9) MD_B(4, 10) MD_B(5, 11) MD_B(6, 12) MD_B(7, 13) MD_B(8, 14) MD_B(9, 15)
      MD_B(10, 16) MD_B(11, 17) MD_BU(12, 18) MD_BU(13, 19) MD_BU(14, 20) MD_BU(15, 21)
      : "+&v"(acc), "+&v"(acc2), "+&v"(s2), "+&v"(t2)
      : "v"(w), "v"(dw), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]), "v"(c[8]), "v"(c[9]),
        "v"(c[10]), "v"(c[11]), "v"(c[12]), "v"(c[13]), "v"(c[14]), "v"(c[15]));
}
// block C: one sum against twelve coefficients
#define MD_C(N, OP) "v_fmac_f64_dpp %0, %1, %" #OP MD_BC(N)
__device__ __forceinline__ void md_col12(double& acc, double v, const double (&c)[12]) {
  asm volatile("s_nop 4\n" MD_C(0, 2) MD_C(1, 3) MD_C(2, 4) MD_C(3, 5) MD_C(4, 6) MD_C(5, 7) MD_C(6, 8) MD_C(7, 9) MD_C(8, 10) MD_C(9, 11)
      MD_C(10, 12) MD_C(11, 13)
      : "+&v"(acc)
      : "v"(v), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]), "v"(c[8]), "v"(c[9]), "v"(c[10]),
        "v"(c[11]));
}

// ---- constraint blocks in the row layout ------------------------------------------------------------------------------------------
// wave_al_rows / wave_al_col (ilqr_mfma16.hip) with the point [x; u] in the row's registers: lane i < 8 of the row owns
// row i of each of the (at most two) blocks -- c_i = G_i [x; u] - g_i is one more 16-term DPP chain, its coefficients the
// lane's row of G --, forms the estimated / projected dual, the AL cost share, the violation and (J^T z_proj)_i, which STAYS in
// the lane; the gradient's column sums  sum_c sum_i G_c[i][e] (J^T z_proj)_i  are a 16-term chain in lane e against those.  A
// second-order-cone block (p <= 4) is evaluated by every lane of the row from the four row values (broadcast), as the one
// lane of wave_al_rows does.  Same expressions, same order of the sums (terms that do not exist enter as exact zeros).
#define MD_K(N, OP) "v_fmac_f64_dpp %0, %1, %" #OP MD_BC(N)
__device__ __forceinline__ void md_chain16(double& acc, double v, const double (&c)[16]) {
  asm volatile("s_nop 4\n" MD_K(0, 2) MD_K(1, 3) MD_K(2, 4) MD_K(3, 5) MD_K(4, 6) MD_K(5, 7) MD_K(6, 8) MD_K(7, 9) MD_K(8, 10) MD_K(9, 11)
      MD_K(10, 12) MD_K(11, 13) MD_K(12, 14) MD_K(13, 15) MD_K(14, 16) MD_K(15, 17)
      : "+&v"(acc)
      : "v"(v), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]), "v"(c[8]), "v"(c[9]), "v"(c[10]),
        "v"(c[11]), "v"(c[12]), "v"(c[13]), "v"(c[14]), "v"(c[15]));
}
__device__ __forceinline__ void md_chain8(double& acc, double v, const double (&c)[8]) {
  asm volatile("s_nop 4\n" MD_K(0, 2) MD_K(1, 3) MD_K(2, 4) MD_K(3, 5) MD_K(4, 6) MD_K(5, 7) MD_K(6, 8) MD_K(7, 9)
      : "+&v"(acc)
      : "v"(v), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]));
}
// lanes 0..3 of the row, to every lane of the row
__device__ __forceinline__ void md_gather4(double v, double (&o)[4]) {
  asm volatile("s_nop 4\n"
               "v_mov_b64_dpp %0, %4" MD_BC(0) "v_mov_b64_dpp %1, %4" MD_BC(1) "v_mov_b64_dpp %2, %4" MD_BC(2) "v_mov_b64_dpp %3, %4" MD_BC(3)
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(v));
}
// (DU: DualUpdate, knotpoint_data.cpp:503-510 -- the projected dual becomes the dual; `store`: this row has a problem of its own)
template <typename S, bool DU = false, bool SOC = true>
__device__ __forceinline__ void dpp_al_rows(const AlTable<S>& t, int k, int b, int64_t B, double w, bool terminal, double rho_est, int j,
                                            double (&jvr)[AL_TILE_MAXC], double& cost, double& viol, bool store = false) {
  int zshift;
  const AlKnot ALTRO_CONST_AS& kn = al_knot<S>(t, k, zshift);
#pragma unroll
  for (int c = 0; c < AL_TILE_MAXC; ++c) {
    jvr[c] = 0.0;
    if (c >= kn.ncon) continue;                     // (wave-uniform: the table is the handle's)
    const int p = kn.p[c], cone = kn.cone[c];
    const S* G = t.G + kn.G_off[c];
    const bool rl = j < p;                          // this lane owns a row of the block
    const int jr = rl ? j : 0;
    double cG[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const double ge = (double)G[jr + e * p];
      cG[e] = (rl && !(terminal && e >= 12)) ? ge : 0.0;
    }
    double sacc = 0.0;
    md_chain16(sacc, w, cG);
    S* const zp_ = t.z + (int64_t)(kn.z_off[c] + zshift + jr) * B + b;
    const double gi = rl ? (kn.g_per_problem[c] ? (double)t.g[kn.g_off[c] + (int64_t)jr * B + b] : (double)t.g[kn.g_off[c] + jr]) : 0.0;
    const double zi = (double)*zp_;
    const double val = sacc - gi;
    const double ze = rl ? zi - rho_est * val : 0.0;
    if (!SOC || cone != CONE_SOC) {   // (!SOC: the handle has no second-order cone -- al.has_soc -- and the cone's code is not in the kernel)
      if (rl) {
        double zp = 0.0, mkv = 0.0;
        if (cone == CONE_EQUALITY) { zp = ze; mkv = 1.0; viol = fmax(viol, fabs(val)); }
        else if (cone == CONE_INEQUALITY) { zp = fmin(0.0, ze); mkv = (ze <= 0.0) ? 1.0 : 0.0; viol = fmax(viol, fabs(fmin(0.0, val) - val)); }
        cost += zp * zp / (2.0 * rho_est);
        jvr[c] = mkv * zp;
        if (DU && store) *zp_ = (S)zp;
      }
    } else {
      double valv[AL_MAXSOC], zev[AL_MAXSOC], zpv[AL_MAXSOC], pv[AL_MAXSOC];
      md_gather4(val, valv);
      md_gather4(ze, zev);
      soc_projection<double>(p, zev, zpv);
      soc_projection<double>(p, valv, pv);
      double sq = 0.0;
#pragma unroll
      for (int r = 0; r < AL_MAXSOC; ++r)
        if (r < p) { sq += zpv[r] * zpv[r]; viol = fmax(viol, fabs(pv[r] - valv[r])); }
      if (j == 0) cost += sq / (2.0 * rho_est);
      if (DU && store && rl) {
        double zown = 0.0;
#pragma unroll
        for (int r = 0; r < AL_MAXSOC; ++r)
          if (j == r) zown = zpv[r];
        *zp_ = (S)zown;
      }
      double Jc[AL_MAXSOC * AL_MAXSOC];
      soc_jacobian<double>(p, zev, Jc);
#pragma unroll
      for (int r = 0; r < AL_MAXSOC; ++r) {
        double sj = 0.0;
#pragma unroll
        for (int q = 0; q < AL_MAXSOC; ++q) sj += Jc[q + r * AL_MAXSOC] * zpv[q];     // (J^T z_proj)_r
        if (r < p && j == r) jvr[c] = sj;
      }
    }
  }
}
// ---- the same rows for the sweeps of wave_merit_dpp_kernel: what a knot point's chain does not have to wait for, it does not ----
// * the table entry of a knot point (AlKnot, constant address space) is read ONCE per knot point into scalar registers (AlpKnot)
//   and handed to the three users (duals a step ahead, rows, gradient columns) -- for the usual uniform table once per sweep;
// * the Jacobians come from the zero-padded pool AlTable::Gpad (al_types.h: AL_GP_DEF) in LDS: a lane reads row min(lane, 8) --
//   rows >= p and row 8 are zero -- as eight 16-byte reads, and any column as eight 8-byte ones, with no select per element;
// * a lane that owns no row carries exact zeros through the same arithmetic instead of branching around it (val = ze = 0:
//   cost share + 0, violation max(., 0), J^T z_proj = 0).
// Same values, same expressions, same order of the sums as dpp_al_rows (and as wave_al_rows / wave_al_col of the LDS form):
// bit-identical (tests/test_gpu_merit2.py).
// (NC: the slots the instantiation carries -- AL_MAXC = 2, today's kernels register for register, or AL_TILE_MAXC; al_types.h)
template <int NC>
struct AlpKnot {
  int ncon, p[NC], cone[NC], gp_off[NC], z_off[NC], gpp[NC];
  int64_t g_off[NC];
};
template <typename S, int NC>
__device__ __forceinline__ void alp_knot(const AlTable<S>& t, int k, AlpKnot<NC>& s) {
  const AlKnot ALTRO_CONST_AS& kn = *(const AlKnot ALTRO_CONST_AS*)(t.knots + k);
  s.ncon = kn.ncon;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    s.p[c] = kn.p[c]; s.cone[c] = kn.cone[c]; s.gp_off[c] = kn.Gp_off[c]; s.z_off[c] = kn.z_off[c]; s.gpp[c] = kn.g_per_problem[c];
    s.g_off[c] = kn.g_off[c];
  }
}
// (z_i, g_i) of the knot point whose entry is s (zshift: al_knot's, for uniform tables)
template <typename S, int NC>
__device__ __forceinline__ void alp_fetch(const AlTable<S>& t, const AlpKnot<NC>& s, int zshift, int b, int64_t B, int j, double (&zg)[NC][2]) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    zg[c][0] = 0.0; zg[c][1] = 0.0;
    if (c >= s.ncon) continue;
    const bool rl = j < s.p[c];
    const int jr = rl ? j : 0;
    zg[c][0] = (double)t.z[(int64_t)(s.z_off[c] + zshift + jr) * B + b];
    zg[c][1] = rl ? (s.gpp[c] ? (double)t.g[s.g_off[c] + (int64_t)jr * B + b] : (double)t.g[s.g_off[c] + jr]) : 0.0;
  }
}
// rowb: min(lane of the row, 8) * AL_GP_LD;  pre: this knot point's (z_i, g_i)
// FUSE (the wide tables): the gradient's column sum  sum_c sum_i G_c[i][jcol] (J^T z_proj)_i  is taken here, slot by slot, right after a
// slot's rows -- alp_col's sum in alp_col's order, so the same bits -- and (J^T z_proj)_i is not kept per slot: six slots cost 12
// registers less, which is what lets the six-slot instantiation run two waves per SIMD.  jvr is then a single scratch value.
template <bool SOC, int NC, bool FUSE = false>
__device__ __forceinline__ void alp_rows(const AlpKnot<NC>& s, double w, double rho_est, int j, int rowb, double (&jvr)[FUSE ? 1 : NC], double& cost,
                                         double& viol, const double* Gp, const double (&pre)[NC][2], int jcol = 0, double* colsum = nullptr) {
  if constexpr (FUSE) *colsum = 0.0;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    double& jv = jvr[FUSE ? 0 : c];
    jv = 0.0;
    if (c >= s.ncon) continue;                      // (wave-uniform)
    const int p = s.p[c], cone = s.cone[c];
    const md_d2* Gr = reinterpret_cast<const md_d2*>(Gp + s.gp_off[c] + rowb);
    double cG[16];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const md_d2 v = Gr[e]; cG[2 * e] = v[0]; cG[2 * e + 1] = v[1]; }
    double sacc = 0.0;
    md_chain16(sacc, w, cG);
    const bool rl = j < p;
    const double val = sacc - pre[c][1];
    const double ze = rl ? pre[c][0] - rho_est * val : 0.0;
    if (!SOC || cone != CONE_SOC) {
      double zp = 0.0, mkv = 0.0;
      if (cone == CONE_EQUALITY) { zp = ze; mkv = 1.0; viol = fmax(viol, fabs(val)); }
      else if (cone == CONE_INEQUALITY) { zp = fmin(0.0, ze); mkv = (ze <= 0.0) ? 1.0 : 0.0; viol = fmax(viol, fabs(fmin(0.0, val) - val)); }
      cost += zp * zp / (2.0 * rho_est);
      jv = mkv * zp;
    } else {
      double valv[AL_MAXSOC], zev[AL_MAXSOC], zpv[AL_MAXSOC], pv[AL_MAXSOC];
      md_gather4(val, valv);
      md_gather4(ze, zev);
      soc_projection<double>(p, zev, zpv);
      soc_projection<double>(p, valv, pv);
      double sq = 0.0;
#pragma unroll
      for (int r = 0; r < AL_MAXSOC; ++r)
        if (r < p) { sq += zpv[r] * zpv[r]; viol = fmax(viol, fabs(pv[r] - valv[r])); }
      if (j == 0) cost += sq / (2.0 * rho_est);
      double Jc[AL_MAXSOC * AL_MAXSOC];
      soc_jacobian<double>(p, zev, Jc);
#pragma unroll
      for (int r = 0; r < AL_MAXSOC; ++r) {
        double sj = 0.0;
#pragma unroll
        for (int q = 0; q < AL_MAXSOC; ++q) sj += Jc[q + r * AL_MAXSOC] * zpv[q];     // (J^T z_proj)_r
        if (r < p && j == r) jv = sj;
      }
    }
    if constexpr (FUSE) {                             // alp_col's chain for this slot
      const double* Gc = Gp + s.gp_off[c] + jcol;
      double cC[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) cC[i] = Gc[i * AL_GP_LD];
      md_chain8(*colsum, jv, cC);
    }
  }
}
template <int NC>
__device__ __forceinline__ double alp_col(const AlpKnot<NC>& s, int j, const double (&jvr)[NC], const double* Gp) {
  double sum = 0.0;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (c >= s.ncon) continue;
    const double* Gc = Gp + s.gp_off[c] + j;
    double cC[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cC[i] = Gc[i * AL_GP_LD];
    md_chain8(sum, jvr[c], cC);
  }
  return sum;
}

}  // namespace altro_hip
#include "ilqr_tile_model.hip"   // nonlinear device models in this layout: tile_model_step (uses the DPP blocks above)
namespace altro_hip {

// DUAL: the two rows of a problem are trial 0 (alpha = a.alpha[b]) and trial 1 (alpha = 1) of wave_merit2_kernel (IK_MERIT2).
// !DUAL: wave_merit_kernel (IK_MERIT, a line-search round): the rows are the speculative trials 2 blockIdx.y and 2 blockIdx.y + 1
// of IlqrArgs::spec_trials (trial 0 = the step the search asked for), every row with its own step, candidate buffer and
// outputs; a row without a trial computes along and stores nothing.  The wave sums are then taken over that kernel's 64-entry
// arrangement (state terms 0..11, input terms 16..19, constraint rows 48..55) by the butterfly it uses.
// DENSE: the cost is the dense quadratic one of ALTROSolver::SetQuadraticCost (knotpoint_data.cpp:616-708) -- lane j of a row also
// owns row j of W = [Q H^T; H R] (staged with the other records: IlqrWaveArgs::costd), (W [x; u])_j is one more 16-term DPP chain,
// the cost share 1/2 w_j (W [x; u])_j + [q r]_j w_j and the gradient (W [x; u])_j + [q r]_j.
// MK != 0: the dynamics are the device model MK (models.h) instead of the DYN records' data -- every knot point steps the model
// from the row's registers and, where phi' is wanted, forms row j of Z = [A B] at the point (tile_model_step,
// ilqr_tile_model.hip); a pass that stores the expansion leaves those rows in the DYN records for the next backward sweep
// (what KnotPointData::CalcDynamicsExpansion leaves in A_, B_: knotpoint_data.cpp:406-419).  Z and f are then not loaded.
// AFF ("affine trials", dynamics as data only): with x+ = A x + B u + f the closed-loop rollout is affine in the step --
//     x_k(alpha) = x_k(alpha_b) + (alpha - alpha_b) s_k ,   s_k = dx_k / dalpha  (the same for every alpha)
// -- so once the sweep's phi(0) evaluation has left x_k(alpha_b) and s_k behind (IlqrWaveArgs::sens; alpha_b = 0), every
// further trial is a sum over knot points that do not depend on each other: the wave takes MD_AFF_CHUNK knot points (blockIdx.z),
// forms x_k(alpha) from the two stored vectors instead of from the step before, does everything else this kernel does at a knot
// point -- u_, y_, cost share, constraint rows, gradient, phi' share, the candidate record -- except the product with Z = [A B],
// and leaves its share of phi / phi' in IlqrWaveArgs::aff_part; wave_aff_reduce_kernel adds the shares in chunk order.  A
// line-search round then costs a chunk's walk (16 knot points) instead of the horizon's (256 for C1: 0.36 ms however few problems
// search).  The trial points equal the rollout's to rounding (1e-13), not bit for bit -- like everything on this plan.
constexpr int MD_AFF_CHUNK = 16;                    // even (the image ping-pong is the parity of k)
template <typename S, bool AL, bool DUAL, bool DENSE = false, int MK = 0, bool SOC = true, bool AFF = false, int NC = AL_MAXC>
__global__ __launch_bounds__(64, ((MK != 0 || (AL && DENSE && sizeof(S) == 8)) ? 1 : 2)) void wave_merit_dpp_kernel(IlqrWaveArgs<S> a) {
  static_assert(NC >= AL_MAXC && NC <= AL_TILE_MAXC, "two to AL_TILE_MAXC slots per knot point");
  static_assert(!AFF || (!DUAL && MK == 0), "affine trials: the single-trial rounds of dynamics given as data");
  constexpr bool NOZ = MK != 0 || AFF;              // Z and f are neither loaded nor staged
  constexpr int DEPTH = 2;                          // also the image ping-pong: parity of k == dd
  constexpr bool kStat = sizeof(S) == 8;            // stored values == computed values only without a rounding store
  constexpr int IMG = DENSE ? MD_IMG_DENSE : MD_IMG;
  constexpr int CPE = DENSE ? MF_COST : MF_COSTP;   // elements of a knot point's cost record
  __shared__ double img[2][2][IMG];                 // [parity][slot]; after the sweep: the final sums (red, below)
  extern __shared__ double Gdyn[];                  // AlTable::Gpad as doubles: the constraint Jacobians, read at every knot point (the launchers pass Gpad_count * 8 bytes)
  const int lane = threadIdx.x;
  const int npairs = (a.batch + 1) >> 1;
  const int pr = mf_problem(blockIdx.x, npairs);
  if (pr >= npairs) return;
  const int slot = lane >> 5, h = (lane >> 4) & 1, j = lane & 15, hl = lane & 31;
  const int b_own = 2 * pr + slot, b_oth = 2 * pr + (1 - slot);
  const bool ok_own = b_own < a.batch && !(a.active && !a.active[b_own]);
  const bool ok_oth = b_oth < a.batch && !(a.active && !a.active[b_oth]);
  if (!ok_own && !ok_oth) return;                   // (the same two answers in every lane: wave-uniform)
  const int b = ok_own ? b_own : b_oth;             // a slot without a problem shadows the other one and stores nothing
  const bool wr = ok_own;
  const int N = a.N;
  // this row's trial: its step, where its candidate goes, what it stores
  int trial = h;
  double alpha = a.alpha ? a.alpha[b] : a.alpha_const;
  bool store = true, deriv = true, row_on = wr;
  S* __restrict__ candb = a.cand + (size_t)b * a.xuy_bs;
  if constexpr (DUAL) {
    if (h) alpha = 1.0;
  } else {
    trial = 2 * (int)blockIdx.y + h;
    row_on = wr && trial < (a.spec_trials > 1 ? a.spec_trials : 1);
    if (trial > 0 && a.spec_pre) {                  // (see wave_merit_kernel)
      if (trial > 1) row_on = false;
      alpha = 1.0;
      candb = a.cand_spec + (size_t)b * a.xuy_bs;
      store = false;
    } else if (trial > 0) {
      const LsState& ls = a.prob[b].ls;
      if (ls.stage == LS_STAGE_BACKTRACK) {
        if (ls.bt_iter + trial >= a.ls_max_iters) row_on = false;
      } else if (ls.stage == LS_STAGE_CUBIC) {
        if (trial >= a.ls_max_iters) row_on = false;
        alpha = ls.alpha0;
      } else {
        row_on = false;
      }
      for (int t = 0; t < trial; ++t) alpha = alpha * a.ls_beta;
      candb = a.cand_spec + (size_t)(trial - 1) * a.spec_stride + (size_t)b * a.xuy_bs;
    }
    deriv = a.want_derivative != 0 && (trial == 0 || a.spec_pre);
    if (!__any(row_on)) return;
  }
  constexpr bool al = AL;
  const double rho = al ? a.prob[b].rho : 1.0;
  if (al)                                           // (the first barrier of the sweep below orders these stores before their readers)
    for (int e = lane; e < a.al.Gpad_count; e += 64) Gdyn[e] = (double)a.al.Gpad[e];
  const bool al_uni = al && a.al.uniform != 0;
  const int rowb = (j < 8 ? j : 8) * AL_GP_LD;
  AlpKnot<NC> kc_s, kn_s;                           // the table entries of the knot point in hand and of the next (scalar registers)
  const bool isx = j < 12;
  const bool cand = DUAL ? (h == 1 && wr) : row_on; // DUAL: trial 1 writes the candidate trajectory and the expansion
  const bool wqr = DUAL ? cand : (row_on && deriv && store);
  const int jr = isx ? j : 11;                      // a valid row for the lanes that own none
  const S* __restrict__ dynb = a.dyn + (size_t)b * a.dyn_bs;
  const S* __restrict__ outb = a.out + (size_t)b * a.out_bs;
  const S* __restrict__ nomb = a.nom + (size_t)b * MF_NOM;
  const S* __restrict__ cpb = (DENSE ? a.costd : a.costp) + (size_t)b * CPE;
  const size_t nom_ks = (size_t)a.batch * MF_NOM, cp_ks = (size_t)a.batch * CPE;
  // where this lane's coefficients sit in the image: row j of [P | p] (lanes < 12) or row j - 12 of Kt (lanes 12..15)
  int ra[13];
#pragma unroll
  for (int c = 0; c < 12; ++c) ra[c] = isx ? MD_OUT0 + MF_OFF_P + mf_sym(j, c) : MD_OUT0 + (j - 12) * 13 + c;
  ra[12] = isx ? MD_OUT0 + MF_OFF_p + j : MD_OUT0 + (j - 12) * 13 + 12;
  // DENSE: row j of W = [Q H^T; H R] inside the staged cost record (triu(Q) | c | pad | [H R] | [q r])
  int rw[DENSE ? 16 : 1];
  if constexpr (DENSE) {
#pragma unroll
    for (int c = 0; c < 16; ++c)
      rw[c] = MD_CP0 + (isx ? (c < 12 ? MF_OFF_Q + mf_sym(j, c) : MF_OFF_HR + (c - 12) * 16 + j) : MF_OFF_HR + (j - 12) * 16 + c);
  }
  // AFF: this wave's knot points kb .. ke - 1 (the last chunk also takes the terminal knot point)
  const int kb = AFF ? (int)blockIdx.z * MD_AFF_CHUNK : 0;
  const int ke = AFF ? (kb + MD_AFF_CHUNK < N ? kb + MD_AFF_CHUNK : N) : N;
  const double dalpha = AFF ? alpha - a.sens_alpha[b] : 0.0;
  const double* const sensb = AFF || a.sens ? a.sens + (size_t)b * 24 : nullptr;   // [k][b][24]: x_k(alpha_b) 12 | s_k 12
  const size_t sens_ks = (size_t)a.batch * 24;
  // the row that leaves the base trajectory and its sensitivity behind: the sweep's phi(0) evaluation (trial 0 of the pass the host marks)
  const bool sens_out = !AFF && a.sens != nullptr && wr && (DUAL ? h == 0 : (row_on && trial == 0));
  if (sens_out && j == 0) a.sens_alpha[b] = alpha;
  double x = isx ? (double)a.x0[(size_t)b * 12 + j] : 0.0;
  double dxda = 0.0;
  double J = 0.0, Jal = 0.0, dJ = 0.0, res = 0.0, viol = 0.0;   // (Jal: the constraint rows' cost shares, lanes 0..7)
  constexpr bool FUSE = NC > AL_MAXC;               // the wide tables: the gradient's column sums are taken with the rows (alp_rows)
  double jvr[FUSE ? 1 : NC] = {0.0};
  double colsum = 0.0;
  double zg[NC][2] = {{0.0, 0.0}, {0.0, 0.0}};   // (z_i, g_i) of the knot point in hand
  if (al) {
    alp_knot<S, NC>(a.al, al_uni ? 0 : kb, kc_s); kn_s = kc_s;
    alp_fetch<S, NC>(a.al, kc_s, al_uni ? kb * a.al.rows_per_knot : 0, b, a.batch, j, zg);
  }
  double lprev = 0.0, yprev = 0.0;                  // gradient and y of knot point k - 1 (the stationarity's lag)
  MeritPairRegs<DENSE> ring[DEPTH];
  double rxb[DEPTH] = {0.0, 0.0}, rsv[DEPTH] = {0.0, 0.0};   // AFF: x_k(alpha_b)_j and s_k_j of the knot points in the ring
#pragma unroll
  for (int dd = 0; dd < DEPTH; ++dd) {
    const size_t kk = kb + dd < ke ? kb + dd : ke - 1;
    merit_pair_load<S, DENSE, NOZ>(ring[dd], dynb + kk * a.dyn_ks, outb + kk * a.out_ks, nomb + kk * nom_ks, cpb + kk * cp_ks, hl);
    if constexpr (AFF) { rxb[dd] = sensb[kk * sens_ks + jr]; rsv[dd] = sensb[kk * sens_ks + 12 + jr]; }
  }
  const int Npad = ((ke - kb + DEPTH - 1) / DEPTH) * DEPTH;
  for (int k0 = kb; k0 < kb + Npad; k0 += DEPTH) {
#pragma unroll
   for (int dd = 0; dd < DEPTH; ++dd) {
    const int k = k0 + dd;
    const bool live = k < ke;
    const int kc = live ? k : ke - 1;
    double* const L = img[dd][slot];
    merit_pair_stage<DENSE, NOZ>(ring[dd], L, hl);
    if constexpr (AFF) { x = isx ? rxb[dd] + dalpha * rsv[dd] : 0.0; dxda = isx ? rsv[dd] : 0.0; }   // x_k(alpha), dx_k / dalpha
    {
      const size_t kn = (k + DEPTH < ke) ? k + DEPTH : ke - 1;
      merit_pair_load<S, DENSE, NOZ>(ring[dd], dynb + kn * a.dyn_ks, outb + kn * a.out_ks, nomb + kn * nom_ks, cpb + kn * cp_ks, hl);
      if constexpr (AFF) { rxb[dd] = sensb[kn * sens_ks + jr]; rsv[dd] = sensb[kn * sens_ks + 12 + jr]; }
    }
    if (sens_out && live && isx) {                    // (x, dxda: this knot point's, before the step below moves them on)
      double* sp_ = const_cast<double*>(sensb) + (size_t)k * sens_ks;
      sp_[j] = x; sp_[12 + j] = dxda;
    }
    __syncthreads();
    // (1) rows of [P | p] and of Kt against dx and dx/dalpha
    double cA[13];
#pragma unroll
    for (int c = 0; c < 13; ++c) cA[c] = L[ra[c]];
    const double wnom = L[MD_NOM0 + j];                      // nominal x_j | u_(j-12)
    const double cq = DENSE ? 0.0 : L[MD_CP0 + j];           // Qd | Rd and q | r line up with [x; u]
    const double cl = L[MD_CP0 + (DENSE ? MF_OFF_QR : 16) + j];
    const double dx = x - wnom;
    double acc = 0.0, acc2 = 0.0;
    md_rows12(acc, acc2, dx, dxda, cA);
    const double aff = cA[12];                               // p_j | -d_i
    const double d = -aff;
    const double uval = wnom + (-acc + alpha * d);           // u_ = u + (-K dx + alpha d)        (lanes 12..15)
    const double duval = -acc2 + d;                          // du/dalpha = -K dx/dalpha + d
    const double y = acc + aff;                              // y_ = P dx + p                     (lanes 0..11)
    const double w = isx ? x : uval, dw = isx ? dxda : duval;
    if (al) {   // both trials' constraint rows at the candidate point [x; u]; the feasibility that counts is trial 1's
      double Ja = 0.0, vv = 0.0;
      if (live) {
        if (k + 1 >= N) alp_knot<S, NC>(a.al, N, kn_s);           // (k + 1 <= N: the terminal knot point's too)
        else if (!al_uni) alp_knot<S, NC>(a.al, k + 1, kn_s);
        if constexpr (NC > AL_MAXC) {
          // the wide table: the next knot point's (z_i, g_i) are asked for AFTER this one's have been used, into the same registers --
          // the loads then have the rest of this step and the head of the next to land in, and six slots cost 24 registers less
          // than with a second set (with alp_rows' FUSE: the six-slot instantiation at two waves per SIMD, no spills)
          alp_rows<SOC, NC, FUSE>(kc_s, w, rho, j, rowb, jvr, Ja, vv, Gdyn, zg, j, &colsum);
          alp_fetch<S, NC>(a.al, kn_s, (al_uni && k + 1 < N) ? (k + 1) * a.al.rows_per_knot : 0, b, a.batch, j, zg);
        } else {
          double zgn[NC][2];
          alp_fetch<S, NC>(a.al, kn_s, (al_uni && k + 1 < N) ? (k + 1) * a.al.rows_per_knot : 0, b, a.batch, j, zgn);
          alp_rows<SOC, NC>(kc_s, w, rho, j, rowb, jvr, Ja, vv, Gdyn, zg);
#pragma unroll
          for (int c = 0; c < NC; ++c) { zg[c][0] = zgn[c][0]; zg[c][1] = zgn[c][1]; }
        }
        Jal += Ja;
      } else {
        alp_rows<SOC, NC, FUSE>(kc_s, w, rho, j, rowb, jvr, Ja, vv, Gdyn, zg, j, &colsum);   // a padding step: discarded
      }
      if (cand && live) viol = fmax(viol, vv);               // (live: a padding step's point is not on the trajectory)
    }
    // (2) the stationarity at knot point k - 1 now that y_k is known: column j of Z_(k-1) against y_k
    if (DUAL && kStat && live && k >= 1) {
      const double* const Lp = img[dd ^ 1][slot];
      double cZ[12];
#pragma unroll
      for (int rr = 0; rr < 12; ++rr) cZ[rr] = Lp[rr * MD_ZLD + j];
      double sy = 0.0;
      md_col12(sy, y, cZ);
      const double g = lprev + sy;
      if (cand) res = fmax(res, fabs(isx ? g - yprev : g));
    }
    // (3) rows of Z against [x; u] and its sensitivity; costs, gradient, dphi
    double cR[16];
    double zacc = 0.0, zacc2 = 0.0, s2 = 0.0, t2 = 0.0;
    double xn;
    if constexpr (AFF) {       // the next knot point's state comes from the stored pair, not from this step
      xn = 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c) cR[c] = 0.0;
    } else if constexpr (MK != 0) {   // a device model: x+ = F(x, u), and row j of Z = [A B] at (x, u) where phi' is wanted
      if (DUAL || deriv) {
        tile_model_step<MK, true>(a.mp, w, jr, xn, cR);
      } else {
        tile_model_step<MK, false>(a.mp, w, jr, xn, cR);
#pragma unroll
        for (int c = 0; c < 16; ++c) cR[c] = 0.0;
      }
      md_rows16(zacc, zacc2, s2, t2, w, dw, cR);
      if (DUAL && kStat && cand && isx) {                    // the candidate's Z for the stationarity of the next step (column reads)
#pragma unroll
        for (int c = 0; c < 16; ++c) L[jr * MD_ZLD + c] = cR[c];
      }
      if (live && wqr && isx) {                              // CalcDynamicsExpansion's A_, B_ for the next backward sweep (f = 0)
        S* zd = const_cast<S*>(a.dyn) + (size_t)b * a.dyn_bs + (size_t)kc * a.dyn_ks;
#pragma unroll
        for (int c = 0; c < 16; ++c) zd[MF_OFF_Z + j * 16 + c] = (S)cR[c];
        zd[MF_OFF_F + j] = S(0);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 16; ++c) cR[c] = L[jr * MD_ZLD + c];
      md_rows16(zacc, zacc2, s2, t2, w, dw, cR);
      xn = (zacc + s2) + L[MD_F0 + jr];                      // x+ = A x + B u + f
    }
    const double dxn = zacc2 + t2;                           // dx+/dalpha = A dx/dalpha + B du/dalpha
    double l;                                                // lx_j | lu_(j-12)
    if constexpr (DENSE) {   // (W [x; u])_j: this lane's row of W against the row's registers (knotpoint_data.cpp:624-634, :659-668)
      double cW[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) cW[c] = L[rw[c]];
      double g = 0.0;
      md_chain16(g, w, cW);
      if (live) {
        J += 0.5 * (w * g) + cl * w;
        if (j == 0) J += L[MD_CP0 + MF_COSTD_C];
      }
      l = g + cl;
    } else {
      if (live) {
        J += 0.5 * (w * (cq * w)) + cl * w;
        if (j == 0) J += L[MD_CP0 + 32];
      }
      // (an explicit fma: cq * w also feeds the cost above, and a product with two uses is not contracted -- wave_merit2_kernel
      //  forms this gradient from its own loads, where it is)
      l = __builtin_fma(cq, w, cl);
    }
    if (al) {
      if constexpr (FUSE) l -= colsum;
      else l -= alp_col(kc_s, j, jvr, Gdyn);
      if (live) kc_s = kn_s;
    }
    if (live) dJ += l * dw;
    if (cand) {   // trial 1's candidate record x | y | u and its [lx lu]
      S* c = candb + (size_t)(live ? k : N) * a.xuy_ks;
      if (isx) c[j] = (S)x;
      c[12 + j] = (S)(isx ? y : uval);
      if (live && wqr) a.cin[(size_t)b * a.cin_bs + (size_t)kc * a.cin_ks + MF_OFF_QR + j] = (S)l;
    }
    if (live) { lprev = l; yprev = y; if (isx) { x = xn; dxda = dxn; } }
   }
  }
  if (!AFF || ke == N) {   // terminal knot point (solver.cpp:319-332), both trials (AFF: the last chunk's)
    if constexpr (AFF) {
      const double xb = sensb[(size_t)N * sens_ks + jr], sv = sensb[(size_t)N * sens_ks + 12 + jr];
      x = isx ? xb + dalpha * sv : 0.0; dxda = isx ? sv : 0.0;
    }
    if (sens_out && isx) {
      double* sp_ = const_cast<double*>(sensb) + (size_t)N * sens_ks;
      sp_[j] = x; sp_[12 + j] = dxda;
    }
    const S* nm = a.nom + ((size_t)N * a.batch + b) * MF_NOM;
    const S* on = a.outn + (size_t)b * MF_TERM;
    S* c = candb + (size_t)N * a.xuy_ks;
    const double dxN = x - (double)nm[jr];
    double Qd = 0.0, q, gN = 0.0;                            // (gN: DENSE, (Q_N x)_j)
    if constexpr (DENSE) {
      const S* cT = a.costd_term + (size_t)b * MF_TERM;      // Q_N rows | q_N
      double cQ[12];
#pragma unroll
      for (int cc = 0; cc < 12; ++cc) cQ[cc] = (double)cT[jr * 12 + cc];
      q = (double)cT[144 + jr];
      md_col12(gN, isx ? x : 0.0, cQ);
      if (isx) {
        J += 0.5 * (x * gN) + q * x;
        if (j == 0) J += (double)a.costd[((size_t)N * a.batch + b) * MF_COST + MF_COSTD_C];
      }
    } else {
      const S* cp = a.costp + ((size_t)N * a.batch + b) * MF_COSTP;
      Qd = (double)cp[jr]; q = (double)cp[16 + jr];
      if (isx) {
        J += 0.5 * (x * (Qd * x)) + q * x;
        if (j == 0) J += (double)cp[32];
      }
    }
    if (al) {
      double Ja = 0.0, vv = 0.0;
      if constexpr (DUAL) {
        alp_rows<SOC, NC, FUSE>(kc_s, isx ? x : 0.0, rho, j, rowb, jvr, Ja, vv, Gdyn, zg, jr, &colsum);
        Jal += Ja;
      } else {   // (wave_merit_kernel adds the terminal blocks' shares to its running sum one by one)
        alp_rows<SOC, NC, FUSE>(kc_s, isx ? x : 0.0, rho, j, rowb, jvr, Jal, vv, Gdyn, zg, jr, &colsum);
      }
      if (cand) viol = fmax(viol, vv);
    }
    double cP[13];
#pragma unroll
    for (int cc = 0; cc < 13; ++cc) cP[cc] = (double)on[jr * 13 + cc];
    double sacc = 0.0, unused = 0.0;
    md_rows12(sacc, unused, dxN, dxda, cP);
    const double yN = sacc + cP[12];
    double lx = DENSE ? gN + q : __builtin_fma(Qd, x, q);
    if (al) {
      if constexpr (FUSE) lx -= colsum;
      else lx -= alp_col(kc_s, jr, jvr, Gdyn);
    }
    if (isx) dJ += lx * dxda;
    if (cand) {
      if (isx) {
        c[j] = (S)x;
        c[12 + j] = (S)yN;
        if (wqr) a.term[(size_t)b * MF_TERM + 144 + j] = (S)lx;
      } else {
        c[12 + j] = S(0);
      }
    }
    if (DUAL && kStat) {
      const double* const Lp = img[(N - 1) & 1][slot];     // the last LIVE step's image (a padding step writes the other parity)
      double cZ[12];
#pragma unroll
      for (int rr = 0; rr < 12; ++rr) cZ[rr] = Lp[rr * MD_ZLD + j];
      double sy = 0.0;
      md_col12(sy, yN, cZ);
      const double g = lprev + sy;
      if (cand) {
        res = fmax(res, fabs(isx ? g - yprev : g));
        if (isx) res = fmax(res, fabs(lx - yN));
      }
    }
  }
  // the sums, over the LDS-form kernels' arrangement of the addends.  DUAL: a trial's 32 entries hold the state terms at 0..11,
  // the input terms at 16..19 (dphi: 12..15), the constraint rows' at 16..23, and are added by the butterfly of offsets 16 .. 1
  __syncthreads();
  static_assert(2 * 2 * IMG >= 2 * 4 * 64, "the final sums reuse the record images");
  double (*red)[4][64] = reinterpret_cast<double (*)[4][64]>(&img[0][0][0]);   // the final sums, in the LDS-form kernels' lane arrangement
  for (int e = lane; e < 2 * 4 * 64; e += 64) (&red[0][0][0])[e] = 0.0;
  __syncthreads();
  red[0][lane >> 4][isx ? j : j + 4] = J;
  red[1][lane >> 4][j] = dJ;
  __syncthreads();
  if (al) {   // the row lanes of wave_merit2_kernel are entries 16..23 ((J + Jal) there), of wave_merit_kernel 48..55
    if (j < AL_MAXP) red[0][lane >> 4][(DUAL ? 16 : 48) + j] += Jal;
    __syncthreads();
  }
  if constexpr (DUAL) {
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int tr = lane >> 5, e = lane & 31;
      const double phi = half_sum(red[0][2 * sl + tr][e]), dphi = half_sum(red[1][2 * sl + tr][e]);
      const int bs = 2 * pr + sl;
      const bool oks = sl == slot ? ok_own : ok_oth;
      if (e == 0 && oks) {
        a.phi[(size_t)tr * a.batch + bs] = phi;
        a.dphi[(size_t)tr * a.batch + bs] = dphi;
      }
    }
    if (kStat) {
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) res = fmax(res, __shfl_xor(res, o, 64));   // over this slot's 32 lanes
      if (al) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) viol = fmax(viol, __shfl_xor(viol, o, 64));
      }
      if (hl == 0 && wr) { a.prob[b].stationarity = res; a.prob[b].feasibility = viol; }
    }
    if (al && hl == 0 && wr) a.prob[b].rho_est = rho;
  } else {
    const unsigned long long on_mask = __ballot(row_on);
#pragma unroll
    for (int q = 0; q < 4; ++q) {                     // row q of the wave: slot q / 2, trial 2 blockIdx.y + q % 2
      const double phi = wave_sum(red[0][q][lane]), dphi = wave_sum(red[1][q][lane]);
      const int tq = 2 * (int)blockIdx.y + (q & 1), bs = 2 * pr + (q >> 1);
      if (lane == 0 && ((on_mask >> (16 * q)) & 1ull)) {
        if constexpr (AFF) {   // this chunk's share; chunk 0 says that (trial, problem) is being evaluated
          double* pt = a.aff_part + (((size_t)blockIdx.z * ILQR_SPEC_TRIALS + tq) * a.batch + bs) * 2;
          pt[0] = phi; pt[1] = dphi;
          if (blockIdx.z == 0) a.aff_on[(size_t)tq * a.batch + bs] = 1;
        } else {
          a.phi[(size_t)tq * a.batch + bs] = phi;
          if (a.want_derivative != 0 && (tq == 0 || a.spec_pre)) a.dphi[(size_t)tq * a.batch + bs] = dphi;
        }
      }
    }
    if (al && j == 0 && row_on && (!AFF || blockIdx.z == 0)) a.prob[b].rho_est = rho;
  }
}

// phi / phi' of the affine trials: the chunks' shares added in chunk order (one thread per (trial, problem))
template <typename S>
__global__ void wave_aff_reduce_kernel(IlqrWaveArgs<S> a, int chunks) {
  const int64_t total = (int64_t)ILQR_SPEC_TRIALS * a.batch;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    if (!a.aff_on[t]) continue;
    a.aff_on[t] = 0;                                  // (consumed: the next round finds the flags clear)
    const int tq = (int)(t / a.batch);
    double phi = 0.0, dphi = 0.0;
    for (int c = 0; c < chunks; ++c) {
      const double* pt = a.aff_part + ((size_t)c * ILQR_SPEC_TRIALS * a.batch + t) * 2;
      phi += pt[0]; dphi += pt[1];
    }
    a.phi[t] = phi;
    if (a.want_derivative != 0 && tq == 0) a.dphi[t] = dphi;
  }
}

// ---- the expansion with constraint blocks (wave_expand_kernel) in the row layout ------------------------------------------------
// One (problem, knot point) per row of 16 lanes, four per wave (wave_expand_kernel spends a whole wave on one).  Lane j holds
// [x; u]_j and column j of every constraint Jacobian: the rows' values are 16-term DPP chains (dpp_al_rows' arithmetic), the
// gradient's column sums 8-term ones, and the Gauss-Newton block  rho G^T J^T J G  is built row by row as outer products
//     tile[r][j] += (J G)_(i r) (J G)_(i j) = bcast_r(a_i) * a_i ,   a_i = lane j's entry of row i of J G ,
// 8 DPP multiply-adds per tile row and block (second-order cones: J G with the cone's full 4 x 4 Jacobian, plus the curvature
// term  G^T (d/dz J^T z_proj) G  the same way).  Same expressions and order of the sums as wave_expand_kernel: bit-identical.
template <int R>
__device__ __forceinline__ void md_outer8(double& acc, const double (&v)[8]) {
  asm volatile("s_nop 4\n"
               "v_fmac_f64_dpp %0, %1, %1 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %2, %2 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %0, %3, %3 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %4, %4 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %0, %5, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %6, %6 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %0, %7, %7 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %8, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
      : "+&v"(acc)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "n"(R));
}
// acc += bcast_R(g_i) * h_i, i = 0..3   (the second-order cone's curvature term)
template <int R>
__device__ __forceinline__ void md_cross4(double& acc, const double (&g)[8], const double (&hc)[4]) {
  asm volatile("s_nop 4\n"
               "v_fmac_f64_dpp %0, %1, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %2, %6 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %0, %3, %7 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %4, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
      : "+&v"(acc)
      : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(hc[0]), "v"(hc[1]), "v"(hc[2]), "v"(hc[3]), "n"(R));
}
__device__ __forceinline__ void md_gather8(double v, double (&o)[8]) {
  asm volatile("s_nop 4\n"
               "v_mov_b64_dpp %0, %8" MD_BC(0) "v_mov_b64_dpp %1, %8" MD_BC(1) "v_mov_b64_dpp %2, %8" MD_BC(2) "v_mov_b64_dpp %3, %8" MD_BC(3)
               "v_mov_b64_dpp %4, %8" MD_BC(4) "v_mov_b64_dpp %5, %8" MD_BC(5) "v_mov_b64_dpp %6, %8" MD_BC(6) "v_mov_b64_dpp %7, %8" MD_BC(7)
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(v));
}
template <int R>
__device__ __forceinline__ void md_outer_rows(double (&tile)[16], const double (&ai)[8]) {
  md_outer8<R>(tile[R], ai);
  if constexpr (R + 1 < 16) md_outer_rows<R + 1>(tile, ai);
}
template <int R>
__device__ __forceinline__ void md_cross_rows(double (&tile)[16], const double (&g)[8], const double (&hc)[4]) {
  md_cross4<R>(tile[R], g, hc);
  if constexpr (R + 1 < 16) md_cross_rows<R + 1>(tile, g, hc);
}

// DENSE: the cost's own Hessian is W = [Q H^T; H R] of ALTROSolver::SetQuadraticCost instead of diag(Qd, Rd) and its gradient
// W [x; u] + [q r] (knotpoint_data.cpp:659-668, :691-698); lane j reads row j of W (= column j: W is symmetric) from the dense
// cost record, the gradient is one more 16-term chain.
// BOUNDS: every block of the handle is bound-type (rows +-e_idx: AlTable::all_sel) and the cost is diagonal -- the Hessian blocks
// are then diagonal: diag(Qd, Rd) + rho * (active rows per variable).  No 16 x 16 tile is formed; and once a solve has stored the
// full blocks, later expansions store the diagonal only (EXPAND_DIAG: what is off it cannot have changed).
// NC: the constraint slots of a knot point the instantiation loops over (AL_MAXC, or AL_TILE_MAXC: ilqr_launch_mfma16_wide.hip)
template <typename S, bool DENSE = false, bool BOUNDS = false, int NC = AL_MAXC>
__global__ __launch_bounds__(64) void wave_expand_dpp_kernel(IlqrWaveArgs<S> a) {
  static_assert(!(DENSE && BOUNDS), "the diagonal form is the diagonal cost's");
  const int lane = threadIdx.x, j = lane & 15;
  // four problems of ONE knot point per wave (the constraint table entry is the knot point's: wave-uniform control flow)
  const int wpk = (a.batch + 3) >> 2;
  const int k = (int)(blockIdx.x / wpk), b0 = (int)(blockIdx.x % wpk) * 4, b_own = b0 + (lane >> 4);
  if (k > a.N) return;
  // a row without a problem of its own (past the end of the batch, or one that is not taking part) shadows the wave's first
  // live row and stores nothing: the DPP blocks below need every lane switched on
  const bool in_batch = b_own < a.batch;
  // EXPAND_DUAL: this launch is also the sweep's DualUpdate (wave_dual_update_dpp_kernel) and looks ahead of its PenaltyUpdate
  // (ilqr_penalty_update_logic, which runs AFTER it): a problem whose sweep asked for the update (IlqrProb::dual) gets its
  // projected duals stored, and its gradient / Hessians are formed from those and from the penalty the update will leave
  const bool du = (a.mode & EXPAND_DUAL) != 0;
  const int dual_own = (du && in_batch) ? a.prob[b_own].dual : 0;
  const bool on_g = du ? dual_own != 0 : (in_batch && !(a.active && !a.active[b_own]));   // stores the gradient
  const bool on_h = (a.mode & EXPAND_NEXT) ? (in_batch && a.prob[b_own].running != 0) : on_g;   // stores the Hessian blocks
  const bool on = on_g || on_h;
  const unsigned long long onm = __ballot(on);
  if (onm == 0ull) return;
  const int b = on ? b_own : b0 + (__builtin_ctzll(onm) >> 4);
  const bool terminal = k == a.N;
  const bool grad = (a.mode & EXPAND_GRADIENT) != 0, hess = (a.mode & EXPAND_HESSIAN) != 0;
  const S* c = a.cand + (size_t)b * a.xuy_bs + (size_t)k * a.xuy_ks;
  const double w = j < 12 ? (double)c[j] : (terminal ? 0.0 : (double)c[12 + j]);      // x_j | u_(j-12)   (c[24 + j - 12])
  double cq = 0.0, cl, cW[DENSE ? 16 : 1];
  if constexpr (DENSE) {
    if (!terminal) {
      const S* cd = a.costd + ((size_t)k * a.batch + b) * MF_COST;
#pragma unroll
      for (int cc = 0; cc < 16; ++cc)
        cW[cc] = (double)cd[j < 12 ? (cc < 12 ? MF_OFF_Q + mf_sym(j, cc) : MF_OFF_HR + (cc - 12) * 16 + j) : MF_OFF_HR + (j - 12) * 16 + cc];
      cl = (double)cd[MF_OFF_QR + j];
    } else {
      const S* cT = a.costd_term + (size_t)b * MF_TERM;      // Q_N rows | q_N
      const int jr = j < 12 ? j : 11;
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) {
        const double v = (double)cT[jr * 12 + (cc < 12 ? cc : 11)];
        cW[cc] = (j < 12 && cc < 12) ? v : 0.0;
      }
      const double qv = (double)cT[144 + jr];
      cl = j < 12 ? qv : 0.0;
    }
  } else {
    const S* cp = a.costp + ((size_t)k * a.batch + b) * MF_COSTP;
    cq = (double)cp[j]; cl = (double)cp[16 + j];
  }
  double gW = 0.0;                                           // DENSE: (W [x; u])_j
  if constexpr (DENSE) md_chain16(gW, w, cW);
  const int dual = du ? a.prob[b].dual : 0;
  const double rho_est0 = a.prob[b].rho_est, rho0 = a.prob[b].rho;
  const double rho = (dual == 2) ? fmin(rho0 * a.penalty_scaling, a.penalty_max) : rho0;   // (what PenaltyUpdate will set)
  const double rho_est = dual ? rho : rho_est0;
  int zshift;
  const AlKnot ALTRO_CONST_AS& kn = al_knot<S>(a.al, k, zshift);
  double tile[BOUNDS ? 1 : 16];
#pragma unroll
  for (int r = 0; r < (BOUNDS ? 1 : 16); ++r) tile[r] = 0.0;
  double dacc = 0.0;                                // BOUNDS: this lane's diagonal entry of (J G)^T (J G), summed over the blocks
  double scol = 0.0;
  bool tile_counts = true;
#pragma unroll
  for (int cidx = 0; cidx < NC; ++cidx) {
    if (cidx >= kn.ncon) continue;
    const int p = kn.p[cidx], cone = kn.cone[cidx];
    // row min(j, 8) and column j of the block from the zero-padded pool (al_types.h: AL_GP_DEF; rows >= p are zero, and at the
    // terminal knot point the input lanes of w are): no select per element
    const S* Gp = a.al.Gpad + kn.Gp_off[cidx];
    const bool rl = j < p;
    const int jr = rl ? j : 0;
    double cC[8];
    double sacc = 0.0;
    if constexpr (BOUNDS) {
      // every row of the slot is +-e_idx (AlKnot::sidx, scalar registers): nothing of G is loaded.  Row i's value is +-w_idx -- the
      // 16-term chain would add that one product to fifteen exact zeros -- fetched from lane idx of this lane's row of sixteen; column
      // j's entries are the signs of the rows that select j.  (The pool's loads were what bound this kernel: 16 vector loads per slot
      // and wave through the CU's one texture path -- C1 with an input box and a state box, four slots: 1.0 ms; like this: see DESIGN.)
      int sx = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int si = kn.sidx[cidx][i];                       // (rows >= p: never read -- i < p below)
        if (i < p) {
          if (j == i) sx = si;
          cC[i] = (j == (si < 0 ? -si : si) - 1) ? (si < 0 ? -1.0 : 1.0) : 0.0;
        } else {
          cC[i] = 0.0;
        }
      }
      const int idx = (sx < 0 ? -sx : sx) - 1;                 // (lanes without a row: sx = 0, idx = -1 -> their own lane, value unused)
      const double wsel = __shfl(w, idx < 0 ? j : idx, 16);
      sacc = (sx == 0) ? 0.0 : (sx < 0 ? -wsel : wsel);
    } else {
      double cG[16];
#pragma unroll
      for (int e = 0; e < 8; ++e) { const md_d2 v = md_ld<S>(Gp + (j < 8 ? j : 8) * AL_GP_LD, e); cG[2 * e] = v[0]; cG[2 * e + 1] = v[1]; }
#pragma unroll
      for (int i = 0; i < 8; ++i) cC[i] = (double)Gp[i * AL_GP_LD + j];
      md_chain16(sacc, w, cG);
    }
    const double gi = rl ? (kn.g_per_problem[cidx] ? (double)a.al.g[kn.g_off[cidx] + (int64_t)jr * a.batch + b] : (double)a.al.g[kn.g_off[cidx] + jr]) : 0.0;
    const double val = sacc - gi;
    S* const zp_ = a.al.z + (int64_t)(kn.z_off[cidx] + zshift + jr) * a.batch + b;
    double zi = (double)*zp_;
    if (du) {   // DualUpdate (knotpoint_data.cpp:503-510): the projected dual, formed with the penalty in force, becomes the dual.
                // (`du` and `cone` are the wave's; the problems' own `dual` flags only select at the end -- the cone's gathers are DPP)
      const double ze0 = rl ? zi - rho_est0 * val : 0.0;
      double znew = 0.0;
      if (cone == CONE_EQUALITY) znew = ze0;
      else if (cone == CONE_INEQUALITY) znew = fmin(0.0, ze0);
      else if constexpr (!BOUNDS) {
        double zev0[AL_MAXSOC], zpv0[AL_MAXSOC];
        md_gather4(ze0, zev0);
        soc_projection<double>(p, zev0, zpv0);
#pragma unroll
        for (int r = 0; r < AL_MAXSOC; ++r)
          if (j == r) znew = zpv0[r];
      }
      if (dual != 0 && rl) {
        if (on && on_g) *zp_ = (S)znew;
        zi = (double)(S)znew;                      // (what a later pass would read back)
      }
    }
    const double ze = rl ? zi - rho_est * val : 0.0;
    double jv = 0.0;
    if (BOUNDS || cone != CONE_SOC) {
      double mkv = 0.0;
      if (rl) {
        double zp = 0.0;
        if (cone == CONE_EQUALITY) { zp = ze; mkv = 1.0; }
        else if (cone == CONE_INEQUALITY) { zp = fmin(0.0, ze); mkv = (ze <= 0.0) ? 1.0 : 0.0; }
        jv = mkv * zp;
      }
      if (grad) md_chain8(scol, jv, cC);
      if (hess) {
        double mk[8];
        md_gather8(mkv, mk);
        if constexpr (BOUNDS) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (i < p) {
              const int sx = kn.sidx[cidx][i];
              dacc += (j == (sx < 0 ? -sx : sx) - 1) ? mk[i] * mk[i] : 0.0;
            }
        } else if (kn.sel[cidx] && tile_counts) {
          // a bound-type block (every row of G is +-e_idx: AlKnot::sel): (J G)^T (J G) is diagonal, entry idx = the number of
          // its rows that are active there -- the 128 multiply-adds below would add exactly these ones and zeros, one by one
          // (tile_counts: the tile holds such counts only so far, so adding them in one go rounds nowhere)
          double dsum = 0.0;
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (i < p) {
              const int sx = kn.sidx[cidx][i];
              dsum += (j == (sx < 0 ? -sx : sx) - 1) ? mk[i] * mk[i] : 0.0;
            }
#pragma unroll
          for (int r = 0; r < 16; ++r) tile[r] += (r == j) ? dsum : 0.0;
        } else {
          double ai[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) ai[i] = mk[i] * cC[i];      // (J G)_(i j): the projection's Jacobian is diagonal
          md_outer_rows<0>(tile, ai);
          tile_counts = false;
        }
      }
    } else if constexpr (!BOUNDS) {
      double valv[AL_MAXSOC], zev[AL_MAXSOC], zpv[AL_MAXSOC];
      md_gather4(val, valv);
      md_gather4(ze, zev);
      soc_projection<double>(p, zev, zpv);
      double Jc[AL_MAXSOC * AL_MAXSOC];
      soc_jacobian<double>(p, zev, Jc);
#pragma unroll
      for (int r = 0; r < AL_MAXSOC; ++r) {
        double sj = 0.0;
#pragma unroll
        for (int q = 0; q < AL_MAXSOC; ++q) sj += Jc[q + r * AL_MAXSOC] * zpv[q];
        if (r < p && j == r) jv = sj;
      }
      if (grad) md_chain8(scol, jv, cC);
      if (hess) {
        double ai[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          double jr2 = 0.0;
          if (i < AL_MAXSOC) {
#pragma unroll
            for (int q = 0; q < AL_MAXSOC; ++q)
              if (q < p) jr2 += Jc[i + q * AL_MAXSOC] * cC[q];
          }
          ai[i] = (i < p) ? jr2 : 0.0;
        }
        md_outer_rows<0>(tile, ai);
        tile_counts = false;
        double Hp[AL_MAXSOC * AL_MAXSOC], hc[4];
        soc_hessian<double>(p, zev, zpv, Hp);
#pragma unroll
        for (int i = 0; i < AL_MAXSOC; ++i) {
          double hh = 0.0;
#pragma unroll
          for (int q = 0; q < AL_MAXSOC; ++q)
            if (q < p) hh += Hp[i + q * AL_MAXSOC] * cC[q];
          hc[i] = (i < p) ? hh : 0.0;
        }
        md_cross_rows<0>(tile, cC, hc);
      }
    }
  }
  if (!on) return;
  if (grad && on_g) {
    double l = DENSE ? gW + cl : cq * w + cl;
    l -= scol;
    if (!terminal) a.cin[(size_t)b * a.cin_bs + (size_t)k * a.cin_ks + MF_OFF_QR + j] = (S)l;
    else if (j < 12) a.term[(size_t)b * MF_TERM + 144 + j] = (S)l;
  }
  if (hess && on_h) {
    const bool diag_only = BOUNDS && (a.mode & EXPAND_DIAG) != 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (terminal && (r >= 12 || j >= 12)) continue;
      if (r < 12 && j >= 12) continue;             // the H^T block is not stored
      if (!terminal && r < 12 && j < r) continue;  // nor is the lower triangle of Q
      if (diag_only && r != j) continue;           // (stored by this solve's full expansion; cannot have changed)
      double v = DENSE ? cW[r] : ((r == j) ? cq : 0.0);
      if constexpr (BOUNDS) v += rho * ((r == j) ? dacc : 0.0);   // (the tile's entry: the count on the diagonal, an exact zero off it)
      else v += rho * tile[r];
      if (terminal) a.term[(size_t)b * MF_TERM + r * 12 + j] = (S)v;
      else if (r < 12) a.cin[(size_t)b * a.cin_bs + (size_t)k * a.cin_ks + MF_OFF_Q + mf_sym(r, j)] = (S)v;
      else a.cin[(size_t)b * a.cin_bs + (size_t)k * a.cin_ks + MF_OFF_HR + (r - 12) * 16 + j] = (S)v;
    }
  }
}

// DualUpdate (wave_dual_update_kernel) and the candidate's feasibility (the second half of wave_stationarity_kernel) in the row
// layout: four problems per wave, the rows' values by DPP, no LDS and no barriers.  wave_dual_update_kernel spends a wave per
// (problem, knot point); the feasibility loop of wave_stationarity_kernel a wave per problem with eight lanes at work and two
// barriers per knot point.  Maxima and stores only: bit-identical by construction.
template <typename S>
__global__ __launch_bounds__(64) void wave_dual_update_dpp_kernel(IlqrWaveArgs<S> a) {
  const int lane = threadIdx.x, j = lane & 15;
  const int wpk = (a.batch + 3) >> 2;
  const int k = (int)(blockIdx.x / wpk), b0 = (int)(blockIdx.x % wpk) * 4, b_own = b0 + (lane >> 4);
  if (k > a.N || !a.al.enabled) return;
  const bool on = b_own < a.batch && a.prob[b_own < a.batch ? b_own : 0].dual != 0;
  const unsigned long long onm = __ballot(on);
  if (onm == 0ull) return;
  const int b = on ? b_own : b0 + (__builtin_ctzll(onm) >> 4);
  const bool terminal = k == a.N;
  const S* c = a.cand + (size_t)b * a.xuy_bs + (size_t)k * a.xuy_ks;
  const double w = j < 12 ? (double)c[j] : (terminal ? 0.0 : (double)c[12 + j]);
  double jvr[AL_TILE_MAXC], cost = 0.0, viol = 0.0;
  dpp_al_rows<S, true>(a.al, k, b, a.batch, w, terminal, a.prob[b].rho_est, j, jvr, cost, viol, on);
}
// One wave per (four problems, knot point), like the dual update; the maximum over the knot points is an atomic maximum on the
// bit pattern (violations are non-negative doubles, whose order is their bit patterns' order as unsigned integers) into the
// control block, which wave_stationarity_kernel -- launched just before, same stream -- has set to zero (STAT_NO_FEAS).
template <typename S>
__global__ __launch_bounds__(64) void wave_feasibility_dpp_kernel(IlqrWaveArgs<S> a) {
  const int lane = threadIdx.x, j = lane & 15;
  const int wpk = (a.batch + 3) >> 2;
  const int k = (int)(blockIdx.x / wpk), b0 = (int)(blockIdx.x % wpk) * 4, b_own = b0 + (lane >> 4);
  if (k > a.N) return;
  bool on = b_own < a.batch;
  if (on && a.active && !a.active[b_own]) on = false;
  if (on && a.skip && a.skip[b_own]) on = false;
  const unsigned long long onm = __ballot(on);
  if (onm == 0ull) return;
  const int b = on ? b_own : b0 + (__builtin_ctzll(onm) >> 4);
  const bool terminal = k == a.N;
  const S* c = a.cand + (size_t)b * a.xuy_bs + (size_t)k * a.xuy_ks;
  const double w = j < 12 ? (double)c[j] : (terminal ? 0.0 : (double)c[12 + j]);
  double jvr[AL_TILE_MAXC], cost = 0.0, viol = 0.0;
  dpp_al_rows<S>(a.al, k, b, a.batch, w, terminal, a.prob[b].rho, j, jvr, cost, viol);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) viol = fmax(viol, __shfl_xor(viol, o, 64));
  if (j == 0 && on && viol > 0.0)
    atomicMax(reinterpret_cast<unsigned long long*>(&a.prob[b].feasibility), (unsigned long long)__double_as_longlong(viol));
}

// (The open-loop rollout was built the same way -- one problem per row of 16 lanes, two per wave -- and measured against
//  wave_rollout_kernel in alternation on one box: 0.579 vs 0.564 ms on C1.  That kernel's LDS traffic is a third of the merit
//  evaluation's and it already runs at the record stream's rate, so the LDS form stays; profiles/r03r_rollout_ab.txt.)

}  // namespace altro_hip
