// kernels/ilqr_row32.hip -- MeritFunction (solver.cpp:273-355) for plan MFMA32's shapes (uniform n <= 31, m <= 8, n + m <= 32 past the
// (12, 4) tile) in a ROW LAYOUT of 32 lanes per problem, on plan GENERIC's arrays (reference layout on the device, generic_arrays.h):
// a drop-in for generic_merit_kernel, which stays the kernel of every other case (per-knot-point dimensions, other shapes, device
// models, fp32, second-order cones, blocks of more than 32 rows) and the form this one is compared with (tests/test_gpu_row32.py).
//
// Why: after round 6 gave these shapes a matrix-core TVLQR pair, a whole solve at (13, 4) spent half to three quarters of its time in
// generic_merit_kernel -- one wave per problem, every vector through LDS, four to five barriers and a dozen dependent global round
// trips per knot point: 1.87 ms per evaluation at 4096 problems x 128 knot points where the backward sweep takes 0.72
// (profiles/r06h_solve_13_4_before_row32.txt).  The (12, 4) tile's answer (kernels/ilqr_merit2_dpp.hip) carries over:
//   * a problem lives in HALF a wave: lane position q = lane % 32 holds x_q (q < n) or u_e at q = 31 - e (inputs from the top down:
//     their positions do not depend on n); two problems per wave;
//   * a vector of 32 positions is two registers per lane -- positions 0..15 and 16..31 as seen from BOTH rows of 16 lanes of the half
//     (one cross-row exchange per vector and knot point) -- so every product against it is a chain of `v_fmac_f64_dpp ... row_newbcast`
//     instructions with the lane's own row of coefficients; input lanes ride the same instructions with rows of K, [H R] where state
//     lanes have rows of P, [A B], [Q H^T];
//   * a knot point's seven matrices go from HBM to an image in LDS verbatim (column-major blocks as they lie: 256-byte runs per half
//     wave and load, fetched one knot point ahead into a ring of registers), and "coefficient j of every lane's row" is then a
//     conflict-free ds_read_b64 -- no barriers: one wave is the whole workgroup, its LDS operations execute in order;
//   * n and m are COMPILE-TIME (one instantiation per shape, like plan MFMA32's backward kernel: row32_unit.inc): the chains have
//     exactly n and m terms, the ring exactly the registers the shape needs, no select or branch asks how large the problem is.  (The
//     first form of this kernel took n, m at run time: 1500 instructions per knot point, a third of them moving spilled scalars, and
//     ran at generic_merit_kernel's speed: profiles/r06i_row32_runtime_shape.txt.)
// Every sum is taken in generic_merit_kernel's order with its expressions, the final sums over its 64-entry arrangement (state rows
// at 0..31, input rows at 32..63) with its butterfly: the tests hold the results to 1e-13 relative against that kernel and report
// whether they are bit-identical; whole solves are held to the oracle like plan GENERIC's.
#pragma once

namespace altro_hip {

typedef __attribute__((address_space(3))) double r32_lds_t;   // (an explicit LDS pointer: a generic one becomes flat loads on some paths)

#define R32_DPP " row_mask:0xf bank_mask:0xf\n"
// CNT (1..4) terms of two sums against the same coefficients: acc += bc_L(v) c, acc2 += bc_L(v2) c for lanes L0, L0 + DL, ... of the
// row of 16.  (Hazards: see ilqr_merit2_dpp.hip -- every block starts with `s_nop 4`, accumulators are early-clobber, `volatile`
// keeps a block where the whole wave executes it.)
template <int L0, int DL, int CNT>
__device__ __forceinline__ void r32_two(double& acc, double& acc2, double v, double v2, const double (&c)[4]) {
  static_assert(CNT >= 1 && CNT <= 4 && L0 >= 0 && L0 <= 15 && L0 + DL * (CNT - 1) >= 0 && L0 + DL * (CNT - 1) <= 15, "a row of 16 lanes");
#define R32_T2(C, L) "v_fmac_f64_dpp %0, %2, %" #C " row_newbcast:%" #L R32_DPP "v_fmac_f64_dpp %1, %3, %" #C " row_newbcast:%" #L R32_DPP
  if constexpr (CNT == 4)
    asm volatile("s_nop 4\n" R32_T2(4, 8) R32_T2(5, 9) R32_T2(6, 10) R32_T2(7, 11)
                 : "+&v"(acc), "+&v"(acc2)
                 : "v"(v), "v"(v2), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "n"(L0), "n"(L0 + DL), "n"(L0 + 2 * DL), "n"(L0 + 3 * DL));
  else if constexpr (CNT == 3)
    asm volatile("s_nop 4\n" R32_T2(4, 7) R32_T2(5, 8) R32_T2(6, 9)
                 : "+&v"(acc), "+&v"(acc2)
                 : "v"(v), "v"(v2), "v"(c[0]), "v"(c[1]), "v"(c[2]), "n"(L0), "n"(L0 + DL), "n"(L0 + 2 * DL));
  else if constexpr (CNT == 2)
    asm volatile("s_nop 4\n" R32_T2(4, 6) R32_T2(5, 7) : "+&v"(acc), "+&v"(acc2) : "v"(v), "v"(v2), "v"(c[0]), "v"(c[1]), "n"(L0), "n"(L0 + DL));
  else
    asm volatile("s_nop 4\n" R32_T2(4, 5) : "+&v"(acc), "+&v"(acc2) : "v"(v), "v"(v2), "v"(c[0]), "n"(L0));
#undef R32_T2
}
template <int L0, int DL, int CNT>
__device__ __forceinline__ void r32_one(double& acc, double v, const double (&c)[4]) {
  static_assert(CNT >= 1 && CNT <= 4 && L0 >= 0 && L0 <= 15 && L0 + DL * (CNT - 1) >= 0 && L0 + DL * (CNT - 1) <= 15, "a row of 16 lanes");
#define R32_T1(C, L) "v_fmac_f64_dpp %0, %1, %" #C " row_newbcast:%" #L R32_DPP
  if constexpr (CNT == 4)
    asm volatile("s_nop 4\n" R32_T1(2, 6) R32_T1(3, 7) R32_T1(4, 8) R32_T1(5, 9)
                 : "+&v"(acc)
                 : "v"(v), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "n"(L0), "n"(L0 + DL), "n"(L0 + 2 * DL), "n"(L0 + 3 * DL));
  else if constexpr (CNT == 3)
    asm volatile("s_nop 4\n" R32_T1(2, 5) R32_T1(3, 6) R32_T1(4, 7)
                 : "+&v"(acc) : "v"(v), "v"(c[0]), "v"(c[1]), "v"(c[2]), "n"(L0), "n"(L0 + DL), "n"(L0 + 2 * DL));
  else if constexpr (CNT == 2)
    asm volatile("s_nop 4\n" R32_T1(2, 4) R32_T1(3, 5) : "+&v"(acc) : "v"(v), "v"(c[0]), "v"(c[1]), "n"(L0), "n"(L0 + DL));
  else
    asm volatile("s_nop 4\n" R32_T1(2, 3) : "+&v"(acc) : "v"(v), "v"(c[0]), "n"(L0));
#undef R32_T1
}

// a vector of the half's 32 positions as the two registers the chains read: positions 0..15 (lo) and 16..31 (hi) in both rows
struct R32Vec { double lo, hi; };
__device__ __forceinline__ R32Vec r32_spread(double v, bool upper_row) {
  const double o = __shfl_xor(v, 16, 64);
  return R32Vec{upper_row ? o : v, upper_row ? v : o};
}

// ---- chains with the coefficients in LDS: coefficient j of this lane's row at L[at + j * stride] (at, stride: the lane's) ----------
// Four coefficients are read a block ahead of the block being summed.  A lane without a row reads from its `at` (somewhere inside the
// image) and sums what it finds: nobody looks at its sums, and no chain broadcasts from its position -- the same instructions for
// every lane, not one select.
template <int CNT>
__device__ __forceinline__ void r32_read4(double (&c)[4], const r32_lds_t* L, int& at, int stride) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t < CNT) { c[t] = L[at]; at += stride; }
}
// positions 0 .. CNT-1 from the bottom (states), ascending, two vectors
template <int CNT, int J0 = 0>
__device__ __forceinline__ void r32_lo2(double& acc, double& acc2, const R32Vec& v, const R32Vec& v2, const r32_lds_t* L, int& at, int stride,
                                        double (&cur)[4]) {
  if constexpr (J0 < CNT) {
    constexpr int here = CNT - J0 < 4 ? CNT - J0 : 4;
    constexpr int next = CNT - J0 - 4 < 4 ? CNT - J0 - 4 : 4;
    double nxt[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (next > 0) r32_read4<next>(nxt, L, at, stride);
    if constexpr (J0 < 16) r32_two<J0, 1, here>(acc, acc2, v.lo, v2.lo, cur);
    else r32_two<J0 - 16, 1, here>(acc, acc2, v.hi, v2.hi, cur);
    if constexpr (next > 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) cur[t] = nxt[t];
      r32_lo2<CNT, J0 + 4>(acc, acc2, v, v2, L, at, stride, cur);
    }
  }
}
template <int CNT, int J0 = 0>
__device__ __forceinline__ void r32_lo1(double& acc, const R32Vec& v, const r32_lds_t* L, int& at, int stride, double (&cur)[4]) {
  if constexpr (J0 < CNT) {
    constexpr int here = CNT - J0 < 4 ? CNT - J0 : 4;
    constexpr int next = CNT - J0 - 4 < 4 ? CNT - J0 - 4 : 4;
    double nxt[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (next > 0) r32_read4<next>(nxt, L, at, stride);
    if constexpr (J0 < 16) r32_one<J0, 1, here>(acc, v.lo, cur);
    else r32_one<J0 - 16, 1, here>(acc, v.hi, cur);
    if constexpr (next > 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) cur[t] = nxt[t];
      r32_lo1<CNT, J0 + 4>(acc, v, L, at, stride, cur);
    }
  }
}
template <int CNT>
__device__ __forceinline__ void r32_states2(double& acc, double& acc2, const R32Vec& v, const R32Vec& v2, const r32_lds_t* L, int at, int stride) {
  double cur[4] = {0.0, 0.0, 0.0, 0.0};
  r32_read4<(CNT < 4 ? CNT : 4)>(cur, L, at, stride);
  r32_lo2<CNT>(acc, acc2, v, v2, L, at, stride, cur);
}
template <int CNT>
__device__ __forceinline__ void r32_states1(double& acc, const R32Vec& v, const r32_lds_t* L, int at, int stride) {
  double cur[4] = {0.0, 0.0, 0.0, 0.0};
  r32_read4<(CNT < 4 ? CNT : 4)>(cur, L, at, stride);
  r32_lo1<CNT>(acc, v, L, at, stride, cur);
}
// the input positions e = 0 .. CNT-1 (position 31 - e: lane 15 - e of the upper row), ascending e
template <int CNT>
__device__ __forceinline__ void r32_in2(double& acc, double& acc2, const R32Vec& v, const R32Vec& v2, const r32_lds_t* L, int at, int stride) {
  double c[4] = {0.0, 0.0, 0.0, 0.0};
  r32_read4<(CNT < 4 ? CNT : 4)>(c, L, at, stride);
  r32_two<15, -1, (CNT < 4 ? CNT : 4)>(acc, acc2, v.hi, v2.hi, c);
  if constexpr (CNT > 4) {
    r32_read4<CNT - 4>(c, L, at, stride);
    r32_two<11, -1, CNT - 4>(acc, acc2, v.hi, v2.hi, c);
  }
}
template <int CNT>
__device__ __forceinline__ void r32_in1(double& acc, const R32Vec& v, const r32_lds_t* L, int at, int stride) {
  double c[4] = {0.0, 0.0, 0.0, 0.0};
  r32_read4<(CNT < 4 ? CNT : 4)>(c, L, at, stride);
  r32_one<15, -1, (CNT < 4 ? CNT : 4)>(acc, v.hi, c);
  if constexpr (CNT > 4) {
    r32_read4<CNT - 4>(c, L, at, stride);
    r32_one<11, -1, CNT - 4>(acc, v.hi, c);
  }
}

// ---- a block of a knot point on its way from HBM to the image: the PAIR of elements 2 (q + 32 t), + 1 in lane position q of the
// problem's half -- each half a 512-byte run per load (16 bytes per lane; with 8 the texture path was the busiest unit of the kernel:
// 74 %, 19 cache accesses per load instruction).  The address is a wave-uniform base (the wave's first problem, moved on by scalar
// adds) plus the lane's 32-bit offset (the half's problem and the pair): one offset register per array instead of a pointer.
typedef double r32_d2 __attribute__((ext_vector_type(2), aligned(8)));   // (blocks start on 8-byte boundaries: 169 doubles per knot point at n = 13)
// LANES: the lanes that share the block's runs -- 32 (a half wave per problem) or 64 (one problem per wave: the two-trial pass)
template <int COUNT, int LANES = 32>
struct R32Block {
  static constexpr int PAIRS = (COUNT + 1) / 2, T = (PAIRS + LANES - 1) / LANES;
  r32_d2 r[T];
  // g: the block of the wave's first problem; lane_off: elements from there to this lane's first pair (the half's problem included)
  template <typename S>
  __device__ __forceinline__ void fetch(const S* g, int lane_off, int q) {
    static_assert(sizeof(S) == 8, "fp64 records");
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int e0 = 2 * (q + LANES * t);
      if constexpr (COUNT < 2) {
        r[t] = r32_d2{(double)g[lane_off - 2 * q], 0.0};
      } else if (2 * LANES * (t + 1) <= COUNT) {
        r[t] = *reinterpret_cast<const r32_d2*>(g + lane_off + 2 * LANES * t);
      } else {   // the last run: past the block's end its last pair again (put() takes what belongs to it)
        const int back = e0 + 1 < COUNT ? 0 : e0 - (COUNT - 2);
        r[t] = *reinterpret_cast<const r32_d2*>(g + lane_off + 2 * LANES * t - back);
      }
    }
  }
  // the same pairs back to global memory (a copy of the block: the fetch's address pattern with stores)
  template <typename S>
  __device__ __forceinline__ void store(S* g, int lane_off, int q) const {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int e0 = 2 * (q + LANES * t);
      if constexpr (COUNT < 2) {
        if (q == 0) g[lane_off] = (S)r[t][0];
      } else if (2 * LANES * (t + 1) <= COUNT) {
        *reinterpret_cast<r32_d2*>(g + lane_off + 2 * LANES * t) = r[t];
      } else if (e0 + 1 < COUNT) {
        *reinterpret_cast<r32_d2*>(g + lane_off + 2 * LANES * t) = r[t];
      } else if (e0 < COUNT) {
        g[lane_off + 2 * LANES * t] = (S)r[t][1];
      }
    }
  }
  __device__ __forceinline__ void put(r32_lds_t* L, int q) const {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int e0 = 2 * (q + LANES * t);
      if constexpr (COUNT < 2) {
        if (q == 0) L[0] = r[t][0];
      } else if (2 * LANES * (t + 1) <= COUNT) {
        L[e0] = r[t][0]; L[e0 + 1] = r[t][1];
      } else if (e0 + 1 < COUNT) {
        L[e0] = r[t][0]; L[e0 + 1] = r[t][1];
      } else if (e0 < COUNT) {                          // (an odd count's last element: the fetched pair is (COUNT - 2, COUNT - 1))
        L[e0] = r[t][1];
      }
    }
  }
};

// ---- constraint blocks in this layout (gen_al_rows / gen_al_col of ilqr_generic.hip: zero / identity / orthant blocks of up to 32 rows) ----
// Lane position r owns row r of EVERY block of the knot point: c_r = G_r [x; u] - g_r is one more chain over the positions (its
// coefficients the lane's row of G in global memory -- the blocks are the handle's, a few KB that stay in L2 --, column e of the
// caller's [x; u] order); the projected dual, the cost share and (J^T z_proj)_r stay in the lane.  The gradient's column sums
// sum_c sum_i G_c[i][col] (J^T z_proj)_i  are chains over the ROWS (positions 0..p-1) against the lane's own column of G, taken block
// by block right after the block's rows -- gen_al_col's sum in gen_al_col's order.
template <typename T, int CNT, bool TOP, int J0 = 0>
__device__ __forceinline__ void r32_g1(double& acc, const R32Vec& w, const T* g, int64_t stride) {
  if constexpr (J0 < CNT) {
    constexpr int here = CNT - J0 < 4 ? CNT - J0 : 4;
    double c[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < here; ++t) c[t] = (double)g[(int64_t)(J0 + t) * stride];
    if constexpr (TOP) r32_one<15 - J0, -1, here>(acc, w.hi, c);          // inputs e = J0.. at positions 31 - e
    else if constexpr (J0 < 16) r32_one<J0, 1, here>(acc, w.lo, c);       // states j = J0.. at positions j
    else r32_one<J0 - 16, 1, here>(acc, w.hi, c);
    r32_g1<T, CNT, TOP, J0 + 4>(acc, w, g, stride);
  }
}
// the column sums run over p <= 32 rows: blocks of four skipped (wave-uniformly) past p
template <typename T, int J0 = 0>
__device__ __forceinline__ void r32_gcol(double& acc, const R32Vec& jv, const T* g, int p) {
  if constexpr (J0 < 32) {
    if (J0 < p) {
      double c[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) c[t] = (double)g[J0 + t < p ? J0 + t : p - 1];   // (past p: a real entry against (J^T z_proj) = 0: an exact zero)
      if constexpr (J0 < 16) r32_one<J0, 1, 4>(acc, jv.lo, c);
      else r32_one<J0 - 16, 1, 4>(acc, jv.hi, c);
      r32_gcol<T, J0 + 4>(acc, jv, g, p);
    }
  }
}
// (bound-type blocks -- AlTable::gsel: every row +-e_idx -- load nothing of G: a row's value is +-[x; u]_idx, fetched from the lane
//  that holds it; a column's coefficients are the signs of the rows that select it, compared out of the table's scalars)
template <int J0 = 0>
__device__ __forceinline__ void r32_selcol(double& acc, const R32Vec& jv, const int* sd, int p, int col1) {
  if constexpr (J0 < 32) {
    if (J0 < p) {
      double c[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int se = sd[1 + (J0 + t < p ? J0 + t : p - 1)];   // (wave-uniform: a scalar load)
        c[t] = (J0 + t < p && se == col1) ? 1.0 : ((J0 + t < p && se == -col1) ? -1.0 : 0.0);
      }
      if constexpr (J0 < 16) r32_one<J0, 1, 4>(acc, jv.lo, c);
      else r32_one<J0 - 16, 1, 4>(acc, jv.hi, c);
      r32_selcol<J0 + 4>(acc, jv, sd, p, col1);
    }
  }
}
// the same walk with coefficient 1 wherever a row selects the column, whatever its sign: sum_i [row i selects col] v_i
template <int J0 = 0>
__device__ __forceinline__ void r32_selcount(double& acc, const R32Vec& v, const int* sd, int p, int col1) {
  if constexpr (J0 < 32) {
    if (J0 < p) {
      double c[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int se = sd[1 + (J0 + t < p ? J0 + t : p - 1)];
        c[t] = (J0 + t < p && (se == col1 || se == -col1)) ? 1.0 : 0.0;
      }
      if constexpr (J0 < 16) r32_one<J0, 1, 4>(acc, v.lo, c);
      else r32_one<J0 - 16, 1, 4>(acc, v.hi, c);
      r32_selcount<J0 + 4>(acc, v, sd, p, col1);
    }
  }
}
// diag (or null): the diagonal Gauss-Newton term of this lane's variable, sum over the bound-type blocks' rows that select it of J_ii^2
// (the host asks for it only when every block is bound-type: EXPAND_DIAG)
template <typename T, int NX, int NU>
__device__ __forceinline__ void r32_al(const AlTable<T>& t, int k, int b, int64_t B, const R32Vec& w, double wv, bool terminal, double rho_est, int q,
                                       bool upper_row, bool want_col, int col, bool has_col, double& cost, double& viol, double& colsum,
                                       double& diag, bool want_diag) {
  int zshift;
  const AlKnotBig ALTRO_CONST_AS& kn = gen_knot<T>(t, k, zshift);
  const int ncon = kn.ncon;
  colsum = 0.0;
  diag = 0.0;
#pragma unroll 1
  for (int c = 0; c < ncon; ++c) {                      // (wave-uniform trip count; nothing is carried per block but the two sums)
    const int p = kn.p[c], cone = kn.cone[c];
    const T* G = t.G + kn.G_off[c];
    const bool rl = q < p;
    const int* sd = t.gsel ? t.gsel + (int64_t)kn.def[c] * (1 + GEN_MAXP) : nullptr;
    const bool bsel = sd && sd[0];                      // (wave-uniform)
    double s = 0.0;
    if (bsel) {
      const int se = rl ? sd[1 + q] : 1, e = (se < 0 ? -se : se) - 1;
      const double wsel = __shfl(wv, e < NX ? e : 31 - (e - NX), 32);   // (this half's lane that holds [x; u]_e)
      if (e < NX || !terminal) s = se < 0 ? -wsel : wsel;
    } else {
      r32_g1<T, NX, false>(s, w, G + (rl ? q : 0), p);
      if (!terminal) r32_g1<T, NU, true>(s, w, G + (rl ? q : 0) + (int64_t)NX * p, p);
    }
    double jv = 0.0, jd2 = 0.0;
    if (rl) {
      const double gi = kn.g_per_problem[c] ? (double)t.g[kn.g_off[c] + (int64_t)q * B + b] : (double)t.g[kn.g_off[c] + q];
      const double val = s - gi;
      const double ze = (double)t.z[(int64_t)(kn.z_off[c] + zshift + q) * B + b] - rho_est * val;
      double zp = 0.0, mkv = 0.0;
      if (cone == CONE_EQUALITY) { zp = ze; mkv = 1.0; viol = fmax(viol, fabs(val)); }
      else if (cone == CONE_INEQUALITY) { zp = fmin(0.0, ze); mkv = (ze <= 0.0) ? 1.0 : 0.0; viol = fmax(viol, fabs(fmin(0.0, val) - val)); }
      cost += zp * zp / (2.0 * rho_est);
      jv = mkv * zp;
      jd2 = (mkv * 1.0) * (mkv * 1.0);
    }
    if (want_diag && bsel) {                            // (wave-uniform)
      const R32Vec jdv = r32_spread(jd2, upper_row);
      r32_selcount(diag, jdv, sd, p, has_col ? col + 1 : 0);
    }
    if (want_col) {                                     // (wave-uniform)
      const R32Vec jvv = r32_spread(jv, upper_row);
      if (bsel) r32_selcol(colsum, jvv, sd, p, has_col ? col + 1 : 0);
      else r32_gcol<T>(colsum, jvv, G + (int64_t)(has_col ? col : 0) * p, p);   // (rows 0..p-1 sit at positions 0..p-1, like states)
    }
  }
}

// ---- a compiled-in device model (models.h) in this layout: generic_merit_kernel<.., MK>'s values (there lane 0 evaluates the model and
// its two Jacobians for the wave, through LDS; the rows of A_k = I + h Am (I + h/2 A0), B_k = h (Am h/2 B0 + Bm) are 13 x 17 inner
// loops of 13 terms per lane: 94 % of a quadrotor NMPC step, profiles/r06o_quad13_nmpc_4096.txt) the way the (12, 4) tile steps its
// models (kernels/ilqr_tile_model.hip): every lane gathers the half's [x; u] (32 DPP moves) and evaluates the continuous model itself
// -- the same instructions in every lane: one evaluation's issue per wave, for two problems --, keeps row `ix` of the midpoint Jacobian
// and sums over the terms whose factor in the first Jacobian is not a structural zero of the model (a skipped term is (h Am) * (0 + h/2
// * 0) = +-0 added to a sum that starts at +0: the sums, in generic_merit_kernel's order with its expressions, keep their bits).
// lanes L0, L0 + DL, ... (CNT of them, 1..4) of the row of 16, to every lane of the row
template <int L0, int DL, int CNT>
__device__ __forceinline__ void r32_gather4(double v, double* o) {
  static_assert(CNT >= 1 && CNT <= 4 && L0 >= 0 && L0 <= 15 && L0 + DL * (CNT - 1) >= 0 && L0 + DL * (CNT - 1) <= 15, "a row of 16 lanes");
#define R32_MV(O, L) "v_mov_b64_dpp %" #O ", %" #L R32_DPP
  if constexpr (CNT == 4)
    asm volatile("s_nop 4\n" "v_mov_b64_dpp %0, %4 row_newbcast:%5" R32_DPP "v_mov_b64_dpp %1, %4 row_newbcast:%6" R32_DPP
                 "v_mov_b64_dpp %2, %4 row_newbcast:%7" R32_DPP "v_mov_b64_dpp %3, %4 row_newbcast:%8" R32_DPP
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(v), "n"(L0), "n"(L0 + DL), "n"(L0 + 2 * DL), "n"(L0 + 3 * DL));
  else if constexpr (CNT == 3)
    asm volatile("s_nop 4\n" "v_mov_b64_dpp %0, %3 row_newbcast:%4" R32_DPP "v_mov_b64_dpp %1, %3 row_newbcast:%5" R32_DPP
                 "v_mov_b64_dpp %2, %3 row_newbcast:%6" R32_DPP
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]) : "v"(v), "n"(L0), "n"(L0 + DL), "n"(L0 + 2 * DL));
  else if constexpr (CNT == 2)
    asm volatile("s_nop 4\n" "v_mov_b64_dpp %0, %2 row_newbcast:%3" R32_DPP "v_mov_b64_dpp %1, %2 row_newbcast:%4" R32_DPP
                 : "=&v"(o[0]), "=&v"(o[1]) : "v"(v), "n"(L0), "n"(L0 + DL));
  else
    asm volatile("s_nop 4\n" "v_mov_b64_dpp %0, %1 row_newbcast:%2" R32_DPP : "=&v"(o[0]) : "v"(v), "n"(L0));
#undef R32_MV
}
// the states (positions 0 .. NX-1) and the inputs (position 31 - e) of a vector, whole, in every lane of the half
template <int NX, int NU, int J0 = 0>
__device__ __forceinline__ void r32_gather_states(const R32Vec& v, double* xl) {
  if constexpr (J0 < NX) {
    constexpr int here = NX - J0 < 4 ? NX - J0 : 4;
    constexpr int stop = (J0 < 16 && J0 + here > 16) ? 16 - J0 : here;   // (a block does not straddle the rows of 16)
    if constexpr (J0 < 16) r32_gather4<J0, 1, stop>(v.lo, xl + J0);
    else r32_gather4<J0 - 16, 1, stop>(v.hi, xl + J0);
    r32_gather_states<NX, NU, J0 + stop>(v, xl);
  }
}
template <int NU>
__device__ __forceinline__ void r32_gather_inputs(const R32Vec& v, double* ul) {
  r32_gather4<15, -1, (NU < 4 ? NU : 4)>(v.hi, ul);
  if constexpr (NU > 4) r32_gather4<11, -1, NU - 4>(v.hi, ul + 4);
}
// One explicit-midpoint step (test_utils.cpp:84-132) from the half's [x; u] (vw: states at positions 0.., input e at 31 - e): state
// lane ix gets x+_ix and, with JAC, row ix of [A_k B_k] (lanes without a state compute row 0 along and must not use it).
template <int MK, int NX, int NU, bool JAC>
__device__ __forceinline__ void r32_model_step(const ModelParams& mp, const R32Vec& vw, double own, int ix, double& xnext, double (&zrow)[NX + NU]) {
  using M = DiscreteModel<MK, NX, NU, double>;
  double xl[NX], ul[NU], k1[NX], xm[NX], k2[NX];
  r32_gather_states<NX, NU>(vw, xl);
  r32_gather_inputs<NU>(vw, ul);
  const float h = mp.h;
  if constexpr (JAC) {
    double J0[NX * (NX + NU)], Jm[NX * (NX + NU)], jm[NX + NU];
    M::cont_fJ(mp, xl, ul, k1, J0);
#pragma unroll
    for (int e = 0; e < NX; ++e) xm[e] = xl[e] + (double)(h / 2) * k1[e];
    M::cont_fJ(mp, xm, ul, k2, Jm);
#pragma unroll
    for (int c = 0; c < NX + NU; ++c) {                 // the lane's row of Jm: a select per entry that is not the same constant in every row
      double v = 0.0;
#pragma unroll
      for (int r = 0; r < NX; ++r) v = (ix == r) ? Jm[r + NX * c] : v;
      jm[c] = v;
    }
#pragma unroll
    for (int jc = 0; jc < NX; ++jc) {
      double sm = 0.0;
#pragma unroll
      for (int kk = 0; kk < NX; ++kk) {
        const double a0 = J0[kk + jc * NX];
        if (kk != jc && __builtin_constant_p(a0) && a0 == 0.0) continue;
        sm += ((double)h * jm[kk]) * ((kk == jc ? 1.0 : 0.0) + (double)(h / 2) * a0);
      }
      zrow[jc] = (ix == jc ? 1.0 : 0.0) + sm;
    }
#pragma unroll
    for (int jc = 0; jc < NU; ++jc) {
      double sm = 0.0;
#pragma unroll
      for (int kk = 0; kk < NX; ++kk) {
        const double b0 = J0[kk + (NX + jc) * NX];
        if (__builtin_constant_p(b0) && b0 == 0.0) continue;
        sm += (jm[kk] * (double)(h / 2)) * b0;
      }
      zrow[NX + jc] = (double)h * (sm + jm[NX + jc]);
    }
  } else {
    M::cont_f(mp, xl, ul, k1);
#pragma unroll
    for (int e = 0; e < NX; ++e) xm[e] = xl[e] + (double)(h / 2) * k1[e];
    M::cont_f(mp, xm, ul, k2);
  }
  double kj = 0.0;
#pragma unroll
  for (int r = 0; r < NX; ++r) kj = (ix == r) ? k2[r] : kj;
  xnext = own + (double)h * kj;
}

constexpr int r32_image_doubles(int n, int m) { return ((3 * n * n + 3 * n * m + m * m + 1) & ~1); }

// two problems per wave: lanes 0..31 problem 2 pr, lanes 32..63 problem 2 pr + 1
// DUAL (the sweep's first evaluation, IK_MERIT2): ONE problem per wave, its two halves the two evaluations ForwardPass starts with --
// phi(0) at alpha = a.alpha[b] in half 0 and the line search's first step alpha = 1 in half 1 (solver.cpp:241-252) -- sharing one
// image: the knot point's matrices are fetched once, by all 64 lanes.  Half 1 writes the candidate trajectory and the expansion (what
// the second of two launches would have left behind), both halves their phi / phi' (rows 0 and 1 of IlqrGenArgs::phi / dphi); per
// half the single-trial kernel's instructions on the same data, so the same values.
// MK != 0: a compiled-in device model of NX states and NU inputs in the place of x+ = A x + B u + f (generic_merit_kernel<.., MK>): the
// blocks A_k, B_k are not fetched but formed (r32_model_step), written into their slots of the image -- the sensitivity's chain runs
// over them as it does over fetched ones -- and, with the derivative, stored into the sweep's arrays for the backward pass
// (MeritFunction with derivative refreshes the dynamics expansion: solver.cpp:300-305).
template <typename T, int NX, int NU, int WPS, bool DUAL = false, int MK = 0>
__global__ __launch_bounds__(64, WPS) void row32_merit_kernel(IlqrGenArgs<T> a) {
  static_assert(NX >= 1 && NU >= 1 && NU <= 8 && NX + NU <= 32, "a problem's [x; u] fits half a wave, the inputs its top eight positions");
  constexpr int NN = NX * NX, NM = NX * NU, MM = NU * NU;
  constexpr int oP = 0, oK = NN, oA = oK + NM, oB = oA + NN, oQ = oB + NM, oH = oQ + NN, oR = oH + NM, IMG = r32_image_doubles(NX, NU);
  constexpr int LN = DUAL ? 64 : 32;                  // lanes that share a block's runs
  __shared__ double red[2][2][64];                    // [phi | phi'][half][generic_merit_kernel's lane arrangement]
  // [half][P | K | A | B | Q | H | R], every block column-major as in HBM.  (DUAL with a model: the two trials share the fetched blocks
  // but each forms its own A_k, B_k -- two more A | B slots behind the image, one per half.)
  constexpr int IMGX = IMG + ((DUAL && MK != 0) ? 2 * (NN + NM) : 0);
  __shared__ double img[DUAL ? 1 : 2][IMGX];
  const int lane = threadIdx.x, half = lane >> 5, q = lane & 31;
  const int ql = DUAL ? lane : q;                     // this lane's place among the lanes that fetch a block
  const bool upper_row = (lane & 16) != 0;
  const int b_own = DUAL ? (int)blockIdx.x : 2 * (int)blockIdx.x + half, b_oth = DUAL ? (int)blockIdx.x : 2 * (int)blockIdx.x + (1 - half);
  const bool ok_own = b_own < a.batch && !(a.active && !a.active[b_own]);
  const bool ok_oth = b_oth < a.batch && !(a.active && !a.active[b_oth]);
  if (!ok_own && !ok_oth) return;                     // (the same two answers in every lane)
  const int b = ok_own ? b_own : b_oth;               // a half without a problem shadows the other one and stores nothing
  const bool wr = DUAL ? half == 1 : ok_own;          // (DUAL: the first step's half writes the candidate and the expansion)
  const int N = a.N;
  const bool isx = q < NX, isu = q >= 32 - NU, has = isx || isu;
  const int iu = isu ? 31 - q : 0, ix = isx ? q : 0;
  const bool al = a.al.enabled != 0;
  const double rho = al ? a.prob[b].rho : 1.0;
  double viol = 0.0, r32_nodiag = 0.0;
  const double alpha = (DUAL && half == 1) ? 1.0 : (a.alpha ? a.alpha[b] : a.alpha_const);
  const bool deriv = a.want_derivative != 0;
  const int oAm = (DUAL && MK != 0) ? IMG + half * (NN + NM) : oA, oBm = oAm + NN;   // where this half's rows of A_k, B_k live
  const r32_lds_t* const L = (const r32_lds_t*)&img[DUAL ? 0 : half][0];
  r32_lds_t* const Lw = (r32_lds_t*)&img[DUAL ? 0 : half][0];
  // uniform dimensions: knot point k's block of an array starts k strides after knot point 0's (the offset table's rows 0 and 1 say
  // both).  Per lane: one pointer per array, moved on by the stride after every knot point.
  const int64_t* off0 = a.off;
  const int64_t* off1 = a.off + (N > 1 ? G_NUM : 0);
#define R32_STRIDE(arr) (off1[arr] - off0[arr])
  const int b0 = DUAL ? b : 2 * (int)blockIdx.x, hb = b - b0;   // the wave's first problem; this half's problem relative to it (0 or 1)
  const T* gP = a.P + (int64_t)b0 * a.P_bs + off0[G_P];   const int64_t sP = R32_STRIDE(G_P);   const int vP = hb * (int)a.P_bs + 2 * ql;
  const T* gK = a.K + (int64_t)b0 * a.K_bs + off0[G_K];   const int64_t sK = R32_STRIDE(G_K);   const int vK = hb * (int)a.K_bs + 2 * ql;
  const T* gA = a.A + (int64_t)b0 * a.A_bs + off0[G_A];   const int64_t sA = R32_STRIDE(G_A);   const int vA = hb * (int)a.A_bs + 2 * ql;
  const T* gB = a.B + (int64_t)b0 * a.B_bs + off0[G_B];   const int64_t sB = R32_STRIDE(G_B);   const int vB = hb * (int)a.B_bs + 2 * ql;
  const T* gQ = a.cQ + (int64_t)b0 * a.sQ + off0[G_Q];    const int64_t sQ = R32_STRIDE(G_Q);   const int vQ = hb * (int)a.sQ + 2 * ql;
  const T* gH = a.cH + (int64_t)b0 * a.sH + off0[G_H];    const int64_t sH = R32_STRIDE(G_H);   const int vH = hb * (int)a.sH + 2 * ql;
  const T* gR = a.cR + (int64_t)b0 * a.sR + off0[G_R];    const int64_t sR = R32_STRIDE(G_R);   const int vR = hb * (int)a.sR + 2 * ql;
  // the vectors: this lane's entry (state lanes: row ix, input lanes: row iu)
  const T* gxn = a.xn + (int64_t)b * a.sx + off0[G_x] + ix;  const int64_t sx_ = R32_STRIDE(G_x);
  T* gx = a.x + (int64_t)b * a.x_bs + off0[G_x] + ix;
  T* gy = a.y + (int64_t)b * a.y_bs + off0[G_y] + ix;        const int64_t sy_ = R32_STRIDE(G_y);
  const T* gp = a.p + (int64_t)b * a.p_bs + off0[G_p] + ix;  const int64_t sp_ = R32_STRIDE(G_p);
  const T* gf = a.f + (int64_t)b * a.f_bs + off0[G_f] + ix;  const int64_t sf_ = R32_STRIDE(G_f);
  const T* gcq = a.cq + (int64_t)b * a.sx + off0[G_q] + ix;  const int64_t sq_ = R32_STRIDE(G_q);
  T* glx = a.q + (int64_t)b * a.q_bs + off0[G_q] + ix;
  const T* gd = a.d + (int64_t)b * a.d_bs + off0[G_d] + iu;  const int64_t sd_ = R32_STRIDE(G_d);
  const T* gun = a.un + (int64_t)b * a.su + off0[G_u] + iu;  const int64_t su_ = R32_STRIDE(G_u);
  T* gu = a.u + (int64_t)b * a.u_bs + off0[G_u] + iu;
  const T* gcr = a.cr + (int64_t)b * a.su + off0[G_r] + iu;  const int64_t sr_ = R32_STRIDE(G_r);
  T* glu = a.r + (int64_t)b * a.r_bs + off0[G_r] + iu;
  const T* gcc = a.cc + (int64_t)b * (N + 1);
  T* gAw = const_cast<T*>(a.A) + (int64_t)b * a.A_bs + off0[G_A] + ix;   // (MK: this lane's row of A_k, B_k in the sweep's arrays)
  T* gBw = const_cast<T*>(a.B) + (int64_t)b * a.B_bs + off0[G_B] + ix;
#undef R32_STRIDE
  double x = isx ? (double)a.x0[(int64_t)b * a.x0_stride + q] : 0.0, dxda = 0.0;
  double J0 = 0.0, J1 = 0.0, dJ = 0.0;                // J0: this position's constraint rows and state row; J1: its input row
  R32Block<NN, LN> rP, rA, rQ;
  R32Block<NM, LN> rK, rB, rH;
  R32Block<MM, LN> rR;
  double vxn, vp, vf, vcq, vd = 0.0, vun = 0.0, vcr = 0.0, vcc;   // the knot point's entries of the vectors, fetched with the matrices
  auto fetch = [&]() {
    rP.fetch(gP, vP, ql); rK.fetch(gK, vK, ql); rQ.fetch(gQ, vQ, ql); rH.fetch(gH, vH, ql); rR.fetch(gR, vR, ql);
    if constexpr (MK == 0) { rA.fetch(gA, vA, ql); rB.fetch(gB, vB, ql); vf = (double)*gf; }
    else vf = 0.0;
    vxn = (double)*gxn; vp = (double)*gp; vcq = (double)*gcq;
    vd = (double)*gd; vun = (double)*gun; vcr = (double)*gcr;   // (every lane: lanes without an input row read row 0's and do not use it)
  };
  fetch();
  vcc = (double)gcc[0];
  for (int k = 0; k < N; ++k) {
    // this knot point's matrices into the image (every read of the last one's has been issued), the next one's on their way.
    // One wave: the LDS pipe executes its writes before the reads that follow, for all lanes -- nothing to wait for (a __syncthreads
    // would wait for the loads just issued: the whole round trip, every knot point); the compiler only has to keep the order.
    rP.put(Lw + oP, ql); rK.put(Lw + oK, ql); rQ.put(Lw + oQ, ql); rH.put(Lw + oH, ql); rR.put(Lw + oR, ql);
    if constexpr (MK == 0) { rA.put(Lw + oA, ql); rB.put(Lw + oB, ql); }
    const double xnom = vxn, pk = vp, fk = vf, ql = vcq, dk = vd, unom = vun, rl = vcr, ck = vcc;
    if (k + 1 < N) {
      gP += sP; gK += sK; gA += sA; gB += sB; gQ += sQ; gH += sH; gR += sR;
      gxn += sx_; gp += sp_; gf += sf_; gcq += sq_; gd += sd_; gun += su_; gcr += sr_;
      fetch();
      vcc = (double)gcc[k + 1];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double dx = isx ? x - xnom : 0.0;
    if (isx && wr) *gx = (T)x;
    const R32Vec vdx = r32_spread(dx, upper_row), vda = r32_spread(isx ? dxda : 0.0, upper_row);
    // rows of P (state lanes) / K (input lanes) against dx and dx/dalpha
    double s = 0.0, s2 = 0.0;
    r32_states2<NX>(s, s2, vdx, vda, L, isx ? oP + ix : oK + iu, isx ? NX : NU);
    double uv = 0.0, du = 0.0;
    if (isu) {   // u_ = u + (-K dx + alpha d) ; du_da = -K dx_da + d
      uv = unom + (-s + alpha * dk);
      du = -s2 + dk;
      if (wr) *gu = (T)uv;
    }
    if (isx && wr) *gy = (T)(s + pk);                   // y_ = P dx + p
    const R32Vec vw = r32_spread(isx ? x : uv, upper_row), vdw = r32_spread(isx ? dxda : du, upper_row);
    double xmodel = 0.0;
    if constexpr (MK != 0) {   // the model at this knot point's [x; u]; with the derivative the state lanes' rows of [A_k B_k] into the image
      if (deriv) {
        double zrow[NX + NU];
        r32_model_step<MK, NX, NU, true>(a.mp, vw, x, ix, xmodel, zrow);
        if (isx) {
#pragma unroll
          for (int c = 0; c < NX; ++c) Lw[oAm + ix + c * NX] = zrow[c];
#pragma unroll
          for (int c = 0; c < NU; ++c) Lw[oBm + ix + c * NX] = zrow[NX + c];
          if (wr) {   // (a column per store instruction: the half's state lanes write NX consecutive entries)
#pragma unroll
            for (int c = 0; c < NX; ++c) gAw[c * NX] = (T)zrow[c];
#pragma unroll
            for (int c = 0; c < NU; ++c) gBw[c * NX] = (T)zrow[NX + c];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      } else {
        double unused[NX + NU];
        r32_model_step<MK, NX, NU, false>(a.mp, vw, x, ix, xmodel, unused);
      }
      gAw += sA; gBw += sB;
    }
    // rows of [A B] (state lanes) / [H R] (input lanes) against [x; u] and its sensitivity (MK: one sum over A's then B's terms, like
    // generic_merit_kernel's; without the derivative the state lanes' coefficients are whatever the image holds and nobody reads the sums)
    double sA_ = 0.0, tA = 0.0, sBv = 0.0, tB = 0.0;
    r32_states2<NX>(sA_, tA, vw, vdw, L, isx ? oAm + ix : oH + iu, isx ? NX : NU);
    if constexpr (MK == 0) r32_in2<NU>(sBv, tB, vw, vdw, L, isx ? oB + ix : oR + iu, isx ? NX : NU);
    else r32_in2<NU>(sBv, tA, vw, vdw, L, isx ? oBm + ix : oR + iu, isx ? NX : NU);
    double alcol = 0.0;
    if (al) {   // the constraint rows' cost shares at the candidate point and the gradient's column sums
      double Jal = 0.0;
      r32_al<T, NX, NU>(a.al, k, b, a.batch, vw, isx ? x : uv, false, rho, q, upper_row, deriv, isx ? ix : NX + iu, has, Jal, viol, alcol, r32_nodiag, false);
      J0 += Jal;
    }
    // rows of [Q H^T] (state lanes) against [x; u]
    double qx = 0.0, htu = 0.0;
    r32_states1<NX>(qx, vw, L, oQ + ix, NX);
    r32_in1<NU>(htu, vw, L, oH + ix * NU, 1);
    double lxu = 0.0;
    if (isx) {   // state row: cost share, lx
      J0 += x * (0.5 * qx + ql);
      if (q == 0) J0 += ck;
      lxu = (qx + htu) + ql;
    }
    if (isu) {   // input row: cost share (with the cross term u'Hx), lu
      const double ru = sBv, hx = sA_;
      J1 += uv * ((0.5 * ru + rl) + hx);
      lxu = (ru + hx) + rl;
    }
    if (al && deriv && has) lxu -= alcol;
    if (deriv) {
      if (isx) { dJ += lxu * dxda; if (wr) *glx = (T)lxu; }
      if (isu) { dJ += lxu * du; if (wr) *glu = (T)lxu; }
    }
    if (isx) {   // next state (uniform dimensions: n2 = n)
      const double xn = MK != 0 ? xmodel : (sA_ + sBv) + fk;
      dxda = MK != 0 ? tA : tA + tB;
      x = xn;
    }
    gx += sx_; gy += sy_; glx += sq_; gu += su_; glu += sr_;
  }
#define GOFFN(arr) (a.off[(int64_t)N * G_NUM + (arr)])
  {   // terminal knot point (solver.cpp:319-332): Q_N and P_N through the image's slots of Q and P
    R32Block<NN, LN> rQn, rPn;
    rQn.fetch(a.cQ + (int64_t)b0 * a.sQ + GOFFN(G_Q), vQ, ql);
    rPn.fetch(a.P + (int64_t)b0 * a.P_bs + GOFFN(G_P), vP, ql);
    const double xnom = isx ? (double)a.xn[(int64_t)b * a.sx + GOFFN(G_x) + ix] : 0.0;
    rQn.put(Lw + oQ, ql); rPn.put(Lw + oP, ql);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double dx = isx ? x - xnom : 0.0;
    if (isx && wr) a.x[(int64_t)b * a.x_bs + GOFFN(G_x) + ix] = (T)x;
    const R32Vec vx = r32_spread(isx ? x : 0.0, upper_row), vdx = r32_spread(dx, upper_row);
    double alcol = 0.0;
    if (al) r32_al<T, NX, NU>(a.al, N, b, a.batch, vx, isx ? x : 0.0, true, rho, q, upper_row, deriv, ix, isx, J0, viol, alcol, r32_nodiag, false);
    double qx = 0.0, s = 0.0;
    r32_states1<NX>(qx, vx, L, oQ + ix, NX);
    r32_states1<NX>(s, vdx, L, oP + ix, NX);
    if (isx) {
      const double ql = (double)a.cq[(int64_t)b * a.sx + GOFFN(G_q) + ix];
      J0 += x * (0.5 * qx + ql);
      if (q == 0) J0 += (double)a.cc[(int64_t)b * (N + 1) + N];
      if (wr) a.y[(int64_t)b * a.y_bs + GOFFN(G_y) + ix] = (T)(s + (double)a.p[(int64_t)b * a.p_bs + GOFFN(G_p) + ix]);
      double lx = qx + ql;
      if (al && deriv) lx -= alcol;
      if (deriv) { dJ += lx * dxda; if (wr) a.q[(int64_t)b * a.q_bs + GOFFN(G_q) + ix] = (T)lx; }
    }
  }
#undef GOFFN
  // the sums over generic_merit_kernel's arrangement: entry r < 32 what its lane r held (constraint row r and state row r), entry
  // 32 + e its input lane e
  for (int e = lane; e < 2 * 2 * 64; e += 64) (&red[0][0][0])[e] = 0.0;
  __syncthreads();
  red[0][half][q] = J0;
  red[1][half][q] = isx ? dJ : 0.0;
  __syncthreads();
  if (isu) { red[0][half][32 + iu] = J1; red[1][half][32 + iu] = dJ; }
  __syncthreads();
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const double phi = gen_wave_sum(red[0][hh][lane]), dphi = gen_wave_sum(red[1][hh][lane]);
    if constexpr (DUAL) {                               // row hh of phi / dphi: evaluation hh of this wave's problem
      if (lane == 0) {
        a.phi[(size_t)hh * a.batch + b] = phi;
        if (deriv) a.dphi[(size_t)hh * a.batch + b] = dphi;
        if (al && hh == 0) a.prob[b].rho_est = a.prob[b].rho;
      }
    } else {
      const int bs = 2 * (int)blockIdx.x + hh;
      if (lane == 0 && (hh == 0 ? ok_own : ok_oth)) {  // (lane 0 sits in half 0: its own problem is hh = 0)
        a.phi[bs] = phi;
        if (deriv) a.dphi[bs] = dphi;
        if (al) a.prob[bs].rho_est = a.prob[bs].rho;
      }
    }
  }
}

// ---- CalcDynamicsExpansion (knotpoint_data.cpp:406-419) of a stored candidate trajectory with a device model: A_k, B_k at every
// (problem, knot point), half a wave each (generic_model_expand_dyn_kernel: a thread each, its 13 x 17 Jacobians in scratch memory --
// 1.0 ms for 4096 vehicles x 30 knot points, once per solve).  r32_model_step's rows: jacobian()'s expressions (models.h).
template <typename T, int NX, int NU, int MK>
__global__ __launch_bounds__(64, 1) void row32_expand_dyn_kernel(IlqrGenArgs<T> a) {
  const int lane = threadIdx.x, half = lane >> 5, q = lane & 31;
  const bool upper_row = (lane & 16) != 0;
  const int64_t tot = (int64_t)a.batch * a.N;
  const int64_t t_own = 2 * (int64_t)blockIdx.x + half, t_oth = 2 * (int64_t)blockIdx.x + (1 - half);
  const bool ok_own = t_own < tot && !(a.active && !a.active[(int)(t_own / a.N)]);
  const bool ok_oth = t_oth < tot && !(a.active && !a.active[(int)(t_oth / a.N)]);
  if (!ok_own && !ok_oth) return;
  const int64_t t = ok_own ? t_own : t_oth;           // a half without a pair shadows the other one and stores nothing
  const int b = (int)(t / a.N), k = (int)(t % a.N);
  const bool isx = q < NX, isu = q >= 32 - NU;
  const int ix = isx ? q : 0, iu = isu ? 31 - q : 0;
  const double w = isx ? (double)a.x[(int64_t)b * a.x_bs + (int64_t)k * NX + ix] : (isu ? (double)a.u[(int64_t)b * a.u_bs + (int64_t)k * NU + iu] : 0.0);
  const R32Vec vw = r32_spread(w, upper_row);
  double xn, zrow[NX + NU];
  r32_model_step<MK, NX, NU, true>(a.mp, vw, w, ix, xn, zrow);
  if (ok_own && isx) {
    T* Ao = const_cast<T*>(a.A) + (int64_t)b * a.A_bs + (int64_t)k * NX * NX + ix;
    T* Bo = const_cast<T*>(a.B) + (int64_t)b * a.B_bs + (int64_t)k * NX * NU + ix;
#pragma unroll
    for (int c = 0; c < NX; ++c) Ao[c * NX] = (T)zrow[c];
#pragma unroll
    for (int c = 0; c < NU; ++c) Bo[c * NX] = (T)zrow[NX + c];
  }
}

// ---- Stationarity and Feasibility (solver.cpp:207-231) of the candidate trajectory in the same layout: generic_stationarity_kernel's
// values (maxima: no order to keep).  Lane j's product A_k^T y_{k+1} (B_k^T y_{k+1} in the input lanes) runs down COLUMN j of the
// block -- in global memory a dozen cache lines per load instruction, in the image a conflict-free read at stride one.
constexpr int R32_STAT_CHUNK = 32;
template <typename T, int NX, int NU, int WPS>
__global__ __launch_bounds__(64, WPS) void row32_stationarity_kernel(IlqrGenArgs<T> a) {
  constexpr int NN = NX * NX, NM = NX * NU, oA = 0, oB = NN, IMG = (NN + NM + 1) & ~1;
  __shared__ double img[2][IMG];
  const int lane = threadIdx.x, half = lane >> 5, q = lane & 31;
  const bool upper_row = (lane & 16) != 0;
  const int b0 = 2 * (int)blockIdx.x;
  const int b_own = b0 + half, b_oth = b0 + (1 - half);
  const bool ok_own = b_own < a.batch && !(a.active && !a.active[b_own]);
  const bool ok_oth = b_oth < a.batch && !(a.active && !a.active[b_oth]);
  if (!ok_own && !ok_oth) return;
  const int b = ok_own ? b_own : b_oth, hb = b - b0;
  const int N = a.N;
  const bool isx = q < NX, isu = q >= 32 - NU, has = isx || isu;
  const int iu = isu ? 31 - q : 0, ix = isx ? q : 0;
  const bool al = a.al.enabled != 0;
  const r32_lds_t* const L = (const r32_lds_t*)&img[half][0];
  r32_lds_t* const Lw = (r32_lds_t*)&img[half][0];
  const int64_t* off0 = a.off;
  const int64_t* off1 = a.off + (N > 1 ? G_NUM : 0);
#define R32_STRIDE(arr) (off1[arr] - off0[arr])
  const T* gA = a.A + (int64_t)b0 * a.A_bs + off0[G_A];   const int64_t sA = R32_STRIDE(G_A);   const int vA = hb * (int)a.A_bs + 2 * q;
  const T* gB = a.B + (int64_t)b0 * a.B_bs + off0[G_B];   const int64_t sB = R32_STRIDE(G_B);   const int vB = hb * (int)a.B_bs + 2 * q;
  const T* gy = a.y + (int64_t)b * a.y_bs + off0[G_y] + ix;    const int64_t sy_ = R32_STRIDE(G_y);
  const T* glx = a.q + (int64_t)b * a.q_bs + off0[G_q] + ix;   const int64_t sq_ = R32_STRIDE(G_q);
  const T* glu = a.r + (int64_t)b * a.r_bs + off0[G_r] + iu;   const int64_t sr_ = R32_STRIDE(G_r);
  const T* gx = a.x + (int64_t)b * a.x_bs + off0[G_x] + ix;    const int64_t sx_ = R32_STRIDE(G_x);
  const T* gu = a.u + (int64_t)b * a.u_bs + off0[G_u] + iu;    const int64_t su_ = R32_STRIDE(G_u);
#undef R32_STRIDE
  const double rho = al ? a.prob[b].rho : 1.0;
  double res = 0.0, viol = 0.0, r32_nodiag = 0.0;
  // the knot points are independent (maxima): blockIdx.y takes R32_STAT_CHUNK of them, the last chunk the terminal one too; with more
  // than one chunk the halves' maxima go to IlqrGenArgs::stat_part and row32_stat_reduce_kernel takes the maximum over the chunks
  const int kb = (int)blockIdx.y * R32_STAT_CHUNK, ke = (gridDim.y == 1 || kb + R32_STAT_CHUNK >= N) ? N : kb + R32_STAT_CHUNK;
  gA += kb * sA; gB += kb * sB; gy += kb * sy_; glx += kb * sq_; glu += kb * sr_; gx += kb * sx_; gu += kb * su_;
  R32Block<NN> rA;
  R32Block<NM> rB;
  rA.fetch(gA, vA, q); rB.fetch(gB, vB, q);
  double yk = (double)*gy;                               // y_k (state lanes), carried from knot point to knot point
  for (int k = kb; k < ke; ++k) {
    rA.put(Lw + oA, q); rB.put(Lw + oB, q);
    gy += sy_;
    const double yn = (double)*gy, lx = (double)*glx, lu = (double)*glu;
    double xv = 0.0, uv = 0.0;
    if (al) { xv = (double)*gx; uv = (double)*gu; }
    if (k + 1 < ke) { gA += sA; gB += sB; rA.fetch(gA, vA, q); rB.fetch(gB, vB, q); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const R32Vec vy = r32_spread(isx ? yn : 0.0, upper_row);
    double s = 0.0;
    r32_states1<NX>(s, vy, L, isx ? oA + ix * NX : oB + iu * NX, 1);
    if (isx) res = fmax(res, fabs((lx + s) - yk));
    if (isu) res = fmax(res, fabs(lu + s));
    if (al) {
      const double wv = isx ? xv : (isu ? uv : 0.0);
      const R32Vec vw = r32_spread(wv, upper_row);
      double cost = 0.0, colsum = 0.0;
      r32_al<T, NX, NU>(a.al, k, b, a.batch, vw, wv, false, rho, q, upper_row, false, 0, false, cost, viol, colsum, r32_nodiag, false);
    }
    yk = yn;
    glx += sq_; glu += sr_; gx += sx_; gu += su_;
  }
  if (ke == N && isx) res = fmax(res, fabs((double)*glx - yk));     // terminal: |lx_N - y_N|   (glx, yk stand at knot point N)
  if (ke == N && al) {
    const double wv = isx ? (double)*gx : 0.0;
    const R32Vec vw = r32_spread(wv, upper_row);
    double cost = 0.0, colsum = 0.0;
    r32_al<T, NX, NU>(a.al, N, b, a.batch, vw, wv, true, rho, q, upper_row, false, 0, false, cost, viol, colsum, r32_nodiag, false);
  }
  (void)has;
  // the maxima over the half's 32 lanes
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) { res = fmax(res, __shfl_xor(res, o, 64)); viol = fmax(viol, __shfl_xor(viol, o, 64)); }
  if (q == 0 && ok_own) {
    if (gridDim.y == 1) { a.prob[b].stationarity = res; a.prob[b].feasibility = viol; }
    else { double* pt = a.stat_part + ((size_t)blockIdx.y * a.batch + b) * 2; pt[0] = res; pt[1] = viol; }
  }
}
// the maximum over the chunks (one thread per problem; the same problems the chunks' waves ran for)
template <typename T>
__global__ void row32_stat_reduce_kernel(IlqrGenArgs<T> a, int chunks) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch || (a.active && !a.active[b])) return;
  double res = 0.0, viol = 0.0;
  for (int c = 0; c < chunks; ++c) {
    const double* pt = a.stat_part + ((size_t)c * a.batch + b) * 2;
    res = fmax(res, pt[0]); viol = fmax(viol, pt[1]);
  }
  a.prob[b].stationarity = res; a.prob[b].feasibility = viol;
}

// ---- the head of Solve for an unconstrained problem (solver.cpp:420-434) in one pass: OpenLoopRollout (generic_rollout_kernel),
// CopyTrajectory (generic_accept_kernel) and the first expansion (generic_expand_kernel: gradient and the cost's Hessian blocks) --
// their values, in their expressions' order.  The knot point's A, B, Q, H, R pass through the image like the merit kernel's; the
// Hessian blocks are written back out of the very registers they were fetched into.
template <typename T, int NX, int NU, int WPS>
__global__ __launch_bounds__(64, WPS) void row32_rollout_init_kernel(IlqrGenArgs<T> a) {
  constexpr int NN = NX * NX, NM = NX * NU, MM = NU * NU;
  constexpr int oA = 0, oB = NN, oQ = oB + NM, oH = oQ + NN, oR = oH + NM, IMG = (oR + MM + 1) & ~1;
  __shared__ double img[2][IMG];
  const int lane = threadIdx.x, half = lane >> 5, q = lane & 31;
  const bool upper_row = (lane & 16) != 0;
  const int b0 = 2 * (int)blockIdx.x;
  const int b_own = b0 + half, b_oth = b0 + (1 - half);
  const bool ok_own = b_own < a.batch && !(a.active && !a.active[b_own]);
  const bool ok_oth = b_oth < a.batch && !(a.active && !a.active[b_oth]);
  if (!ok_own && !ok_oth) return;
  const int b = ok_own ? b_own : b_oth, hb = b - b0;
  const bool wr = ok_own;
  const int N = a.N;
  const bool isx = q < NX, isu = q >= 32 - NU;
  const int iu = isu ? 31 - q : 0, ix = isx ? q : 0;
  const r32_lds_t* const L = (const r32_lds_t*)&img[half][0];
  r32_lds_t* const Lw = (r32_lds_t*)&img[half][0];
  const int64_t* off0 = a.off;
  const int64_t* off1 = a.off + (N > 1 ? G_NUM : 0);
#define R32_STRIDE(arr) (off1[arr] - off0[arr])
  const T* gA = a.A + (int64_t)b0 * a.A_bs + off0[G_A];   const int64_t sA = R32_STRIDE(G_A);   const int vA = hb * (int)a.A_bs + 2 * q;
  const T* gB = a.B + (int64_t)b0 * a.B_bs + off0[G_B];   const int64_t sB = R32_STRIDE(G_B);   const int vB = hb * (int)a.B_bs + 2 * q;
  const T* gQ = a.cQ + (int64_t)b0 * a.sQ + off0[G_Q];    const int64_t sQ = R32_STRIDE(G_Q);   const int vQ = hb * (int)a.sQ + 2 * q;
  const T* gH = a.cH + (int64_t)b0 * a.sH + off0[G_H];    const int64_t sH = R32_STRIDE(G_H);   const int vH = hb * (int)a.sH + 2 * q;
  const T* gR = a.cR + (int64_t)b0 * a.sR + off0[G_R];    const int64_t sR = R32_STRIDE(G_R);   const int vR = hb * (int)a.sR + 2 * q;
  // where the Hessian blocks go (the sweep's Q / R / H: same block sizes, their own problem strides)
  T* dQ = a.Q + (int64_t)b0 * a.Q_bs + off0[G_Q];   const int wQ = hb * (int)a.Q_bs + 2 * q;
  T* dH = a.H + (int64_t)b0 * a.H_bs + off0[G_H];   const int wH = hb * (int)a.H_bs + 2 * q;
  T* dR = a.R + (int64_t)b0 * a.R_bs + off0[G_R];   const int wR = hb * (int)a.R_bs + 2 * q;
  const T* gf = a.f + (int64_t)b * a.f_bs + off0[G_f] + ix;   const int64_t sf_ = R32_STRIDE(G_f);
  const T* gcq = a.cq + (int64_t)b * a.sx + off0[G_q] + ix;   const int64_t sq_ = R32_STRIDE(G_q);
  const T* gcr = a.cr + (int64_t)b * a.su + off0[G_r] + iu;   const int64_t sr_ = R32_STRIDE(G_r);
  T* gx = a.x + (int64_t)b * a.x_bs + off0[G_x] + ix;         const int64_t sx_ = R32_STRIDE(G_x);
  T* gxn = a.xn + (int64_t)b * a.sx + off0[G_x] + ix;
  const T* gu = a.u + (int64_t)b * a.u_bs + off0[G_u] + iu;   const int64_t su_ = R32_STRIDE(G_u);
  T* gun = a.un + (int64_t)b * a.su + off0[G_u] + iu;
  T* glx = a.q + (int64_t)b * a.q_bs + off0[G_q] + ix;
  T* glu = a.r + (int64_t)b * a.r_bs + off0[G_r] + iu;
#undef R32_STRIDE
  double x = isx ? (double)a.x0[(int64_t)b * a.x0_stride + q] : 0.0;
  R32Block<NN> rA, rQ;
  R32Block<NM> rB, rH;
  R32Block<MM> rR;
  double vf, vcq, vcr, vu;
  auto fetch = [&]() {
    rA.fetch(gA, vA, q); rB.fetch(gB, vB, q); rQ.fetch(gQ, vQ, q); rH.fetch(gH, vH, q); rR.fetch(gR, vR, q);
    vf = (double)*gf; vcq = (double)*gcq; vcr = (double)*gcr; vu = (double)*gu;
  };
  fetch();
  for (int k = 0; k < N; ++k) {
    rA.put(Lw + oA, q); rB.put(Lw + oB, q); rQ.put(Lw + oQ, q); rH.put(Lw + oH, q); rR.put(Lw + oR, q);
    if (wr) { rQ.store(dQ, wQ, q); rH.store(dH, wH, q); rR.store(dR, wR, q); }     // CalcCostHessian: the cost's own blocks
    const double fk = vf, ql = vcq, rl = vcr, uv = isu ? vu : 0.0;
    if (k + 1 < N) {
      gA += sA; gB += sB; gQ += sQ; gH += sH; gR += sR; gf += sf_; gcq += sq_; gcr += sr_; gu += su_;
      fetch();
    }
    dQ += sQ; dH += sH; dR += sR;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (wr) {
      if (isx) { *gx = (T)x; *gxn = (T)x; }
      if (isu) *gun = (T)uv;
    }
    const R32Vec vw = r32_spread(isx ? x : uv, upper_row);
    // rows of [A B] (state lanes) / [H R] (input lanes) against [x; u]
    double sA_ = 0.0, sBv = 0.0;
    r32_states1<NX>(sA_, vw, L, isx ? oA + ix : oH + iu, isx ? NX : NU);
    r32_in1<NU>(sBv, vw, L, isx ? oB + ix : oR + iu, isx ? NX : NU);
    // rows of [Q H^T] (state lanes)
    double qx = 0.0, htu = 0.0;
    r32_states1<NX>(qx, vw, L, oQ + ix, NX);
    r32_in1<NU>(htu, vw, L, oH + ix * NU, 1);
    if (wr) {   // generic_expand_kernel's sums: (Q x + q) + H^T u  |  (R u + r) + H x
      if (isx) *glx = (T)((qx + ql) + htu);
      if (isu) *glu = (T)((sBv + rl) + sA_);
    }
    if (isx) x = (sA_ + sBv) + fk;
    gx += sx_; gxn += sx_; gun += su_; glx += sq_; glu += sr_;
  }
  {   // terminal knot point
    R32Block<NN> rQn;
    rQn.fetch(a.cQ + (int64_t)b0 * a.sQ + a.off[(int64_t)N * G_NUM + G_Q], vQ, q);
    const double ql = (double)a.cq[(int64_t)b * a.sx + a.off[(int64_t)N * G_NUM + G_q] + ix];
    rQn.put(Lw + oQ, q);
    if (wr) rQn.store(a.Q + (int64_t)b0 * a.Q_bs + a.off[(int64_t)N * G_NUM + G_Q], wQ, q);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (wr && isx) { *gx = (T)x; *gxn = (T)x; }
    const R32Vec vx = r32_spread(isx ? x : 0.0, upper_row);
    double qx = 0.0;
    r32_states1<NX>(qx, vx, L, oQ + ix, NX);
    if (wr && isx) *glx = (T)(qx + ql);
  }
}

constexpr int R32_EXPAND_CHUNK = 16;
// ---- the expansion with constraint blocks (generic_expand_al_kernel: one wave per (problem, knot point), every gradient entry a walk
// over a row in global memory) as a walk of the row layout over a problem's knot points: EXPAND_GRADIENT -- lx, lu with the AL terms
// (knotpoint_data.cpp:583-595), that kernel's sums ((Q x + q) + H^T u - G^T J^T z_proj) -- and EXPAND_HESSIAN in its diagonal form
// (EXPAND_DIAG: every block bound-type; cost entry + rho * the selecting rows' J_ii^2).  The full Hessian stays that kernel's.
template <typename T, int NX, int NU, int WPS>
__global__ __launch_bounds__(64, WPS) void row32_expand_kernel(IlqrGenArgs<T> a) {
  constexpr int NN = NX * NX, NM = NX * NU, MM = NU * NU;
  constexpr int oQ = 0, oH = NN, oR = oH + NM, IMG = (oR + MM + 1) & ~1;
  __shared__ double img[2][IMG];
  const int lane = threadIdx.x, half = lane >> 5, q = lane & 31;
  const bool upper_row = (lane & 16) != 0;
  const int b0 = 2 * (int)blockIdx.x;
  const int b_own = b0 + half, b_oth = b0 + (1 - half);
  const bool ok_own = b_own < a.batch && !(a.active && !a.active[b_own]);
  const bool ok_oth = b_oth < a.batch && !(a.active && !a.active[b_oth]);
  if (!ok_own && !ok_oth) return;
  const int b = ok_own ? b_own : b_oth, hb = b - b0;
  const bool wr = ok_own;
  const int N = a.N;
  const bool grad = (a.mode & EXPAND_GRADIENT) != 0, hess = (a.mode & EXPAND_HESSIAN) != 0;
  const bool isx = q < NX, isu = q >= 32 - NU, has = isx || isu;
  const int iu = isu ? 31 - q : 0, ix = isx ? q : 0;
  const r32_lds_t* const L = (const r32_lds_t*)&img[half][0];
  r32_lds_t* const Lw = (r32_lds_t*)&img[half][0];
  const int64_t* off0 = a.off;
  const int64_t* off1 = a.off + (N > 1 ? G_NUM : 0);
#define R32_STRIDE(arr) (off1[arr] - off0[arr])
  const T* gQ = a.cQ + (int64_t)b0 * a.sQ + off0[G_Q];    const int64_t sQ = R32_STRIDE(G_Q);   const int vQ = hb * (int)a.sQ + 2 * q;
  const T* gH = a.cH + (int64_t)b0 * a.sH + off0[G_H];    const int64_t sH = R32_STRIDE(G_H);   const int vH = hb * (int)a.sH + 2 * q;
  const T* gR = a.cR + (int64_t)b0 * a.sR + off0[G_R];    const int64_t sR = R32_STRIDE(G_R);   const int vR = hb * (int)a.sR + 2 * q;
  // the diagonal entries this lane owns: Q(ix, ix) / R(iu, iu) of the cost, and where they go in the sweep's blocks
  const T* cQd = a.cQ + (int64_t)b * a.sQ + off0[G_Q] + ix + (int64_t)ix * NX;
  const T* cRd = a.cR + (int64_t)b * a.sR + off0[G_R] + iu + (int64_t)iu * NU;
  T* dQd = a.Q + (int64_t)b * a.Q_bs + off0[G_Q] + ix + (int64_t)ix * NX;
  T* dRd = a.R + (int64_t)b * a.R_bs + off0[G_R] + iu + (int64_t)iu * NU;
  const T* gcq = a.cq + (int64_t)b * a.sx + off0[G_q] + ix;   const int64_t sq_ = R32_STRIDE(G_q);
  const T* gcr = a.cr + (int64_t)b * a.su + off0[G_r] + iu;   const int64_t sr_ = R32_STRIDE(G_r);
  const T* gx = a.x + (int64_t)b * a.x_bs + off0[G_x] + ix;   const int64_t sx_ = R32_STRIDE(G_x);
  const T* gu = a.u + (int64_t)b * a.u_bs + off0[G_u] + iu;   const int64_t su_ = R32_STRIDE(G_u);
  T* glx = a.q + (int64_t)b * a.q_bs + off0[G_q] + ix;
  T* glu = a.r + (int64_t)b * a.r_bs + off0[G_r] + iu;
#undef R32_STRIDE
  const double rho = a.prob[b].rho, rho_est = a.prob[b].rho_est;
  // the knot points are independent here: blockIdx.y takes R32_EXPAND_CHUNK of them (the last chunk the terminal one too), so that a
  // launch is 8 x as many waves walking an eighth of the horizon each (one walk of 128 knot points: 0.41 ms whatever the batch)
  const int kb = (int)blockIdx.y * R32_EXPAND_CHUNK, ke = kb + R32_EXPAND_CHUNK < N ? kb + R32_EXPAND_CHUNK : N;
  gQ += kb * sQ; gH += kb * sH; gR += kb * sR;
  cQd += kb * sQ; cRd += kb * sR; dQd += kb * sQ; dRd += kb * sR;
  gcq += kb * sq_; gcr += kb * sr_; gx += kb * sx_; gu += kb * su_; glx += kb * sq_; glu += kb * sr_;
  R32Block<NN> rQ;
  R32Block<NM> rH;
  R32Block<MM> rR;
  if (grad) { rQ.fetch(gQ, vQ, q); rH.fetch(gH, vH, q); rR.fetch(gR, vR, q); }
  for (int k = kb; k < ke; ++k) {
    const double xv = (double)*gx, uv0 = (double)*gu;
    double ql = 0.0, rl = 0.0, cqd = 0.0, crd = 0.0;
    if (grad) {
      rQ.put(Lw + oQ, q); rH.put(Lw + oH, q); rR.put(Lw + oR, q);
      ql = (double)*gcq; rl = (double)*gcr;
      if (k + 1 < ke) { gQ += sQ; gH += sH; gR += sR; rQ.fetch(gQ, vQ, q); rH.fetch(gH, vH, q); rR.fetch(gR, vR, q); }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (hess) { cqd = (double)*cQd; crd = (double)*cRd; }
    const double wv = isx ? xv : (isu ? uv0 : 0.0);
    const R32Vec vw = r32_spread(wv, upper_row);
    double cost = 0.0, viol = 0.0, alcol = 0.0, dg = 0.0;
    r32_al<T, NX, NU>(a.al, k, b, a.batch, vw, wv, false, rho_est, q, upper_row, grad, isx ? ix : NX + iu, has, cost, viol, alcol, dg, hess);
    if (grad) {
      double s1 = 0.0, s2 = 0.0;     // state lanes: Q x, H^T u;  input lanes: H x, R u
      r32_states1<NX>(s1, vw, L, isx ? oQ + ix : oH + iu, isx ? NX : NU);
      r32_in1<NU>(s2, vw, L, isx ? oH + ix * NU : oR + iu, isx ? 1 : NU);
      if (wr) {
        if (isx) *glx = (T)(((s1 + ql) + s2) - alcol);
        if (isu) *glu = (T)(((s2 + rl) + s1) - alcol);
      }
    }
    if (hess && wr) {
      if (isx) { double v = cqd; v += rho * dg; *dQd = (T)v; }
      if (isu) { double v = crd; v += rho * dg; *dRd = (T)v; }
    }
    gx += sx_; gu += su_; gcq += sq_; gcr += sr_; glx += sq_; glu += sr_;
    cQd += sQ; cRd += sR; dQd += sQ; dRd += sR;
  }
  if (ke == N) {   // terminal knot point (the last chunk's): states only
    const int64_t oNQ = a.off[(int64_t)N * G_NUM + G_Q];
    R32Block<NN> rQn;
    if (grad) { rQn.fetch(a.cQ + (int64_t)b0 * a.sQ + oNQ, vQ, q); rQn.put(Lw + oQ, q); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double xv = isx ? (double)*gx : 0.0;
    const R32Vec vx = r32_spread(xv, upper_row);
    double cost = 0.0, viol = 0.0, alcol = 0.0, dg = 0.0;
    r32_al<T, NX, NU>(a.al, N, b, a.batch, vx, xv, true, rho_est, q, upper_row, grad, ix, isx, cost, viol, alcol, dg, hess);
    if (grad) {
      double s1 = 0.0;
      r32_states1<NX>(s1, vx, L, oQ + ix, NX);
      if (wr && isx) *glx = (T)((s1 + (double)*gcq) - alcol);
    }
    if (hess && wr && isx) {
      double v = (double)a.cQ[(int64_t)b * a.sQ + oNQ + ix + (int64_t)ix * NX];
      v += rho * dg;
      a.Q[(int64_t)b * a.Q_bs + oNQ + ix + (int64_t)ix * NX] = (T)v;
    }
  }
}

}  // namespace altro_hip
