// tvlqr_mfma16.hip -- plan MFMA16: the (n, m) = (12, 4) fp64 TVLQR pair on CDNA4 matrix cores.
//
// Replaces tvlqr_BackwardPass / tvlqr_ForwardPass (src/tvlqr/tvlqr.cpp:65-195, :197-248) for the
// shape BASELINE.json quotes its metric on.  One wavefront (64 lanes) owns one problem and walks the
// horizon serially; nothing but the knot point's own blocks ever moves through HBM.
//
// Why MFMA here: n + m = 16, so the whole action-value expansion is two chained 16x16 tiles
//
//        D1 = [P' | t]^T-as-A-operand x Z         (12+1 rows used, K = 12)          3 x v_mfma_f64_16x16x4
//        G  = [Q H^T; H R] + Z^T x D1             (16x16, K = 12, 100 % tile use)   3 x v_mfma_f64_16x16x4
//
// with Z = [A B] (12x16, the reference's dynamics_jac_ layout, knotpoint_data.cpp:412-414) and
// t = p' (+ P' f).  G holds Qxx, Qux, Quu at once (tvlqr.cpp:135-143) and row 12 of D1 is Z^T t, the
// gradient part (tvlqr.cpp:147-152).  The key hardware fact: the f64 16x16x4 accumulator layout
// (lane l, reg r  <->  row (l>>4)+4r, col l&15) IS the B-operand layout of K-chunk r, so D1 feeds the
// second product straight from registers -- no LDS, no shuffles.  The cost-to-go update
// (tvlqr.cpp:173-186) is two more K=4 MFMAs in homogeneous form with Kt = [K | -d], Qt = [Qux | Qu]:
//
//        [P | p] = [Qxx | Qx] + Kt^T (Quu Kt - Qt) - Qt^T Kt                          2 x v_mfma_f64_16x16x4
//
// Only the 4x4 Cholesky + 13 triangular solves (tvlqr.cpp:159-166) run on the VALU, every lane
// solving its own column redundantly in its four row-groups, fed by one 640-byte LDS exchange.
// 8 MFMAs (512 matrix-pipe cycles) per knot point against 5088 algorithmic bytes (4064 after storing the symmetric
// blocks once, see below): HBM-bound.
//
// Symmetry note: the tile product uses P'^T where the reference uses P' (and returns G rather than
// G^T); for the symmetric Q, R the API requires (altro_solver.hpp:183) P_k is symmetric up to rounding,
// so this changes results only at the 1e-16 relative level.  Parity is asserted at 1e-8 on K, d.
// The cost-to-go the recursion CARRIES is the one it STORES: after every step the lower triangle of the
// new P is replaced by the mirrored upper one (one LDS transposition per knot point).  Without it the
// antisymmetric part of P -- pure rounding noise, 1e-16 -- is propagated by the recursion, and that
// propagation is not the contracting closed-loop map the symmetric part sees: on a 12-state quadrotor
// linearisation (N = 40) it doubles every step and reached 2e-6 in K_0 (round 4, tests/test_gpu_tile_model.py);
// the double integrator and the random LTV problems of the benchmark configs never showed it.
//
// Device layout (private to this plan; pack_mfma16.hip converts from/to the reference layout).  Q and P are symmetric
// (altro_solver.hpp:183 requires it of Q; P inherits it up to rounding): only their upper triangles go through HBM.
//   DYN [k][b][204]  = Z = [A B] 3x64 fragment (== row-major 12x16) | f 12                       (1632 B)
//   COST[k][b][160]  = triu(Q) 78 row-major packed | pad 2 | [H R] 4x16 | [q r] 16               (1280 B; 2912 B in all)
//   TERM[b][156]     = Q_N rows 12x12 | q_N 12
//   OUT [k][b][144]  = Kt 4x13 row-major | triu(P) 78 | p 12 | pad 2                             (1152 B)
//   OUTN[b][156]     = [P_N p_N] 12x13 row-major
// 508 elements per knot point move instead of the 636 the full blocks would take (mfma16_layout.h).
// Every load/store below is `lane -> consecutive 8-byte element` over a contiguous run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfma16_layout.h"

namespace altro_hip {

typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f64x4 mfma_f64_16x16x4(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---- DPP helpers: butterfly all-reduce over the 16 lanes of a DPP row -------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
  union { double d; int i[2]; } in, out;
  in.d = v;
  // old == src: every lane is written by these permutations, so no zero-initialising v_mov is needed
  out.i[0] = __builtin_amdgcn_update_dpp(in.i[0], in.i[0], CTRL, 0xf, 0xf, false);
  out.i[1] = __builtin_amdgcn_update_dpp(in.i[1], in.i[1], CTRL, 0xf, 0xf, false);
  return out.d;
}
__device__ __forceinline__ double row16_allreduce(double v) {
  v += dpp_mov_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov_f64<0x141>(v);  // row_half_mirror
  v += dpp_mov_f64<0x140>(v);  // row_mirror
  return v;
}
// Lanes 13..15 of every DPP row take lane 12's value (quad_perm [0,0,0,0] applied to bank 3 only);
// all other lanes keep theirs.  Lets the three padding columns of a 16-wide tile store a duplicate of
// column 12 to column 12's address, so that tile stores need no exec mask (and no branch).
__device__ __forceinline__ double dpp_col12_dup(double v) {
  union { double d; int i[2]; } in, out;
  in.d = v;
  out.i[0] = __builtin_amdgcn_update_dpp(in.i[0], in.i[0], 0x00, 0xf, 0x8, false);
  out.i[1] = __builtin_amdgcn_update_dpp(in.i[1], in.i[1], 0x00, 0xf, 0x8, false);
  return out.d;
}
// sum over the four 16-lane row groups (lanes l, l^16, l^32, l^48): result in every lane
__device__ __forceinline__ double group4_allreduce(double v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// S = storage type of the HBM arrays (double, or float for ALTRO_HIP_F32 handles: fp32 storage with
// fp64 tile arithmetic -- half the bytes, same kernels; element order and record sizes are unchanged).
template <typename S>
struct Mfma16Args {
  const S* in;           // IN
  // Element strides between problems (bs) / knot points (ks) of IN, OUT and the forward output.
  // Knot-point-major ([k][b][record]: bs = record, ks = batch*record) keeps all the records touched at
  // one time step in one contiguous slab of HBM (the batch structure-of-arrays picture); 0 = shared.
  int64_t in_bs, in_ks, out_bs, out_ks, xuy_bs, xuy_ks, cin_bs, cin_ks;
  const S* cin;          // COST records (the forward sweep never touches them: two fully-used streams
                         // instead of one half-used one -- partial-record reads waste DRAM pages)
  const S* term;         // TERM
  S* out;                // OUT
  S* outn;               // OUTN
  S* qblk;               // optional [b][k][MF_QB]
  S* trash;              // [b][MF_OUT] dump record for the masked stores of failed problems
  const S* x0;           // [b][12]
  S* xuy;                // forward output [b][k][28] = x 12 | y 12 | u 4 ; terminal [b][N][..]
  S* delta_V;            // [b][2]
  int* status;           // [b]
  int N;
  int batch;
  double reg;
  int has_f;
  const int* active;     // optional per-problem mask: the batched solver skips problems that have stopped (backward only)
  const double* reg_pp;  // optional per-problem regularisation (overrides reg): the solver's retry schedule
};

// ------------------------------------------------------------------------------------------------
// Backward sweep.
// ------------------------------------------------------------------------------------------------
// Record loads.  Measured on MI355X (round 1): nontemporal loads (`__builtin_nontemporal_load`) make the
// backward sweep 9 % SLOWER (1.155 vs 1.06 ms) and leave the forward sweep unchanged, so plain loads.  Nontemporal STORES
// of the OUT records (round 2, same-box A/B, three pairs): 0.8424 vs 0.8428 ms -- no difference, plain stores.
template <typename S>
__device__ __forceinline__ double ld_stream(const S* p) {
  return (double)(*p);
}

constexpr int MF_PT_LD = 17;   // row pitch of the [P | p] exchange tile: 16 columns + 1 (a column read hits 16 different bank pairs)

struct Mfma16Knot {  // one knot point's inputs, in registers (11 doubles / lane)
  double z[3], q[3], hr, qr, f[3];
};
// Element offsets, inside a cost record, of the three entries Q[g + 4r][j] of this lane's G-tile registers: a gather
// from the packed upper triangle (the 64 lanes of one load touch the same ten cache lines the triangle occupies).
struct Mfma16QOff { int o[3]; };
__device__ __forceinline__ Mfma16QOff mfma16_q_offsets(int j, int g) {
  const int jq = (j < 12) ? j : 11;
  Mfma16QOff q;
#pragma unroll
  for (int r = 0; r < 3; ++r) q.o[r] = MF_OFF_Q + mf_sym(g + 4 * r, jq);
  return q;
}

// All loads are unconditional and branch-free (lanes that have no element read a clamped, valid
// address and discard it): exec-masked load branches make hipcc's s_waitcnt insertion collapse to
// vmcnt(0) at the loop head, which serialises the prefetch against its own issue.
template <bool HAS_F, typename S>
__device__ __forceinline__ void mfma16_load_knot(Mfma16Knot& kn, const S* __restrict__ rec,
                                                 const S* __restrict__ crec, int lane, int j, int g,
                                                 const Mfma16QOff& qo) {
#pragma unroll
  for (int c = 0; c < 3; ++c) kn.z[c] = ld_stream(&rec[MF_OFF_Z + c * 64 + lane]);
#pragma unroll
  for (int r = 0; r < 3; ++r) kn.q[r] = ld_stream(&crec[qo.o[r]]);
  kn.hr = ld_stream(&crec[MF_OFF_HR + lane]);
  kn.qr = ld_stream(&crec[MF_OFF_QR + j]);
  if (HAS_F) {
#pragma unroll
    for (int r = 0; r < 3; ++r) kn.f[r] = ld_stream(&rec[MF_OFF_F + g + 4 * r]);
  } else {
    kn.f[0] = kn.f[1] = kn.f[2] = 0.0;
  }
}

// 1/sqrt(x) to ~1 ulp: v_rsq_f64 seed (~2^-24 relative) + ONE third-order step
//   e = 1 - x y^2 ;  y <- y (1 + e/2 + 3 e^2 / 8)        (error ~ e^3 ~ 2^-70)
// The Cholesky needs only the reciprocal pivots (L_kk itself is never used), so there is no sqrt
// and no division anywhere in the sweep.
__device__ __forceinline__ double rsqrt_nr(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-x, y * y, 1.0);
  const double t = __builtin_fma(0.375, e, 0.5);
  return __builtin_fma(y * e, t, y);
}

template <bool STORE_Q, bool HAS_F, typename S>
__global__ __launch_bounds__(64, 4) void mfma16_backward_kernel(Mfma16Args<S> a) {
  // the one and only LDS object: 4x16 tile [Qux | Quu] then [Qx | Qu]
  // lds[0..63] = [Qux | Quu] (row g, col j), lds[64..79] = [Qx | Qu], lds[80] = 0.0 (the "zero slot" padding
  // lanes read instead of selecting: their loop-invariant LDS addresses point here)
  __shared__ __attribute__((aligned(16))) double lds[64 + 16 + 2];
  __shared__ double ptile[12 * MF_PT_LD];   // the new [P | p] tile, written row-major and read back mirrored (see the symmetry note)
  const int lane = threadIdx.x;
  const int j = lane & 15, g = lane >> 4;
  const int b = mf_problem(blockIdx.x, a.batch);
  if (b >= a.batch) return;
  if (a.active && !a.active[b]) return;   // wave-uniform: the whole problem is skipped, its outputs stay as they are
  const int N = a.N;
  if (lane < 2) lds[80 + lane] = 0.0;
  // loop-invariant LDS read addresses (element indices into S)
  int rhs_idx[4];   // column j of Qt = [Qux | Qu]; zero for the padding columns 13..15
#pragma unroll
  for (int r = 0; r < 4; ++r) rhs_idx[r] = (j < 12) ? (r * 16 + j) : ((j == 12) ? (64 + 12 + r) : 80);
  const int qmine_idx = (j < 12) ? lane : ((j == 12) ? (64 + 12 + g) : 80);   // Qt[g][j]
  int qx_idx[3];    // column 12 of [Qxx | Qx]: Qx[g + 4r]; zero for columns 13..15 (unused for j < 12)
#pragma unroll
  for (int r = 0; r < 3; ++r) qx_idx[r] = (j == 12) ? (64 + g + 4 * r) : 80;
  const S* __restrict__ in = a.in + (size_t)b * a.in_bs;
  const S* __restrict__ cin = a.cin + (size_t)b * a.cin_bs;
  S* __restrict__ out = a.out + (size_t)b * a.out_bs;
  const bool col_ok = (j <= 12);
  const int jc = col_ok ? j : 12;
  S* __restrict__ trash = a.trash + (size_t)b * MF_OUT;
  const Mfma16QOff qo = mfma16_q_offsets(j, g);
  // where this lane's three [P | p] registers (rows g + 4r, column j) go inside an OUT record: the packed upper
  // triangle, the p block, or -- for the entries below the diagonal and the padding columns -- the record's two
  // spare slots (garbage nobody reads; same cache lines, so no extra traffic and no exec-masked store)
  int p_off[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int row = g + 4 * r;
    p_off[r] = (j == 12) ? MF_OFF_p + row : ((j < 12 && j >= row) ? MF_OFF_P + mf_sym(row, j) : MF_OFF_PAD + (j > 12 ? 1 : 0));
  }

  // the tile exchange that keeps the carried P symmetric: every lane writes its three entries (row g + 4r, column j) and reads
  // them back -- the entries BELOW the diagonal from the mirrored position above it, every other one from its own slot (no
  // selects, no masked accesses)
  int pt_wr[3], pt_rd[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int row = g + 4 * r;
    pt_wr[r] = row * MF_PT_LD + j;
    pt_rd[r] = (j < 12 && j < row) ? j * MF_PT_LD + row : pt_wr[r];
  }

  // terminal cost-to-go: P_N = Q_N, p_N = q_N (tvlqr.cpp:81-90) -> tile [P | p]
  double Pt[3];
  {
    const S* term = a.term + (size_t)b * MF_TERM;
    S* on = a.outn + (size_t)b * MF_TERM;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      double v = 0.0;
      if (j < 12) v = (double)term[r * 48 + g * 12 + j];
      else if (j == 12) v = (double)term[144 + g + 4 * r];
      Pt[r] = v;
      if (col_ok) on[(g + 4 * r) * 13 + j] = (S)v;
    }
  }
  // per-lane partial sums of the expected decrease; only column 12 is meaningful, reduced at the end
  double dv0 = 0.0, dv1 = 0.0;
  int fail_k = -1;
  const double reg = a.reg_pp ? a.reg_pp[b] : a.reg;

  Mfma16Knot cur, nxt;
  mfma16_load_knot<HAS_F, S>(cur, in + (size_t)(N - 1) * a.in_ks, cin + (size_t)(N - 1) * a.cin_ks, lane, j, g, qo);
  // Drain the VMEM queue before entering the loop: hipcc merges the pre-header's scoreboard into the
  // loop header's, and a pending first load there turns into `s_waitcnt vmcnt(0)` at the top of EVERY
  // iteration -- which would drain each step's stores before the next step may start.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched

  for (int k = N - 1; k >= 0; --k) {
    // prefetch the next knot point (k-1) while this one computes (k == 0 re-reads record 0: harmless)
    {
      const size_t kp = (k > 0) ? k - 1 : 0;
      mfma16_load_knot<HAS_F, S>(nxt, in + kp * a.in_ks, cin + kp * a.cin_ks, lane, j, g, qo);
    }

    // ---- D1 = [P'|t]^T Z : rows 0..11 = P'^T Z, row 12 = t^T Z --------------------------------
    f64x4 D1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 3; ++c) D1 = mfma_f64_16x16x4(Pt[c], cur.z[c], D1);

    // ---- G = [Q H^T; H R] + Z^T D1 : Qxx (rows<12, cols<12), [Qux | Quu] (rows 12..15) ---------
    // (columns 12..15 of rows 0..11 -- the Qxu block -- are never used, so the clamped Q loads of the
    //  padding lanes need no zeroing)
    f64x4 G = {cur.q[0], cur.q[1], cur.q[2], cur.hr};
#pragma unroll
    for (int c = 0; c < 3; ++c) G = mfma_f64_16x16x4(cur.z[c], D1[c], G);

    // ---- gradient [Qx; Qu] = [q; r] + Z^T t (+ Z^T P' f) : valid in lanes g == 0 -----------------
    double gv = D1[3] + cur.qr;
    if (HAS_F) {  // Z^T P' f = (P'^T Z)^T f : sum_i f[i] D1[i][j]
      double s = cur.f[0] * D1[0] + cur.f[1] * D1[1] + cur.f[2] * D1[2];
      gv += group4_allreduce(s);
    }

    // ---- LDS exchange ------------------------------------------------------------------------------
    __syncthreads();  // previous iteration's readers are done (single-wave block: free)
    lds[lane] = G[3];                 // [Qux | Quu], row g, col j
    if (g == 0) lds[64 + j] = gv;     // [Qx | Qu]
    __syncthreads();
    // lower triangle of Quu (Eigen's LLT<Lower> reads only that), same in every lane
    const double a00 = lds[0 * 16 + 12];
    const double a10 = lds[1 * 16 + 12], a11 = lds[1 * 16 + 13];
    const double a20 = lds[2 * 16 + 12], a21 = lds[2 * 16 + 13], a22 = lds[2 * 16 + 14];
    const double a30 = lds[3 * 16 + 12], a31 = lds[3 * 16 + 13], a32 = lds[3 * 16 + 14], a33 = lds[3 * 16 + 15];
    double rhs[4];  // column j of Qt = [Qux | Qu]
#pragma unroll
    for (int r = 0; r < 4; ++r) rhs[r] = lds[rhs_idx[r]];
    double quu_row[4];  // row g of the UNregularised Quu
#pragma unroll
    for (int c = 0; c < 4; ++c) quu_row[c] = lds[g * 16 + 12 + c];
    f64x4 Pn;           // accumulator init: column j of [Qxx | Qx], rows g + 4r
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double qx = lds[qx_idx[r]];
      Pn[r] = (j < 12) ? G[r] : qx;
    }
    Pn[3] = 0.0;
    const double q_mine = lds[qmine_idx];  // Qt[g][j] (0 in the padding columns)

    // ---- Cholesky of Quu + reg I (lower; fail when a pivot is <= 0: tvlqr.cpp:159-164) -----------
    // Only the reciprocal pivots i_k = 1/L_kk are needed by the substitutions below.
    const double x0 = a00 + reg;
    const double i0 = rsqrt_nr(x0);
    const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
    const double x1 = (a11 + reg) - l10 * l10;
    const double i1 = rsqrt_nr(x1);
    const double l21 = (a21 - l20 * l10) * i1, l31 = (a31 - l30 * l10) * i1;
    const double x2 = (a22 + reg) - l20 * l20 - l21 * l21;
    const double i2 = rsqrt_nr(x2);
    const double l32 = (a32 - l30 * l20 - l31 * l21) * i2;
    const double x3 = (a33 + reg) - l30 * l30 - l31 * l31 - l32 * l32;
    const double i3 = rsqrt_nr(x3);
    // Failure is wave-uniform (Quu is the same in every lane).  Reference: `return k` with
    // K_k = Qux, d_k = -Qu left unsolved and P_k, p_k untouched.  To keep the loop a plain counted
    // loop (no mid-body exits: they wreck hipcc's s_waitcnt placement) a failed problem keeps
    // iterating on garbage with its stores masked off; failures are rare and cost nothing extra.
    const bool fail = !(x0 > 0.0) || !(x1 > 0.0) || !(x2 > 0.0) || !(x3 > 0.0);
    const bool was_alive = (fail_k < 0);
    if (was_alive && fail) fail_k = k;
    const bool alive = (fail_k < 0);
    // ---- Kt[:, j] = (L L^T)^-1 Qt[:, j]  (tvlqr.cpp:165-166; column 12 gives -d) ------------------
    const double y0 = rhs[0] * i0;
    const double y1 = (rhs[1] - l10 * y0) * i1;
    const double y2 = (rhs[2] - l20 * y0 - l21 * y1) * i2;
    const double y3 = (rhs[3] - l30 * y0 - l31 * y1 - l32 * y2) * i3;
    const double k3 = y3 * i3;
    const double k2 = (y2 - l32 * k3) * i2;
    const double k1 = (y1 - l21 * k2 - l31 * k3) * i1;
    const double k0 = (y0 - l10 * k1 - l20 * k2 - l30 * k3) * i0;
    const double k_mine = (g == 0) ? k0 : (g == 1) ? k1 : (g == 2) ? k2 : k3;  // Kt[g][j]
    // (Quu Kt)[g][j] with the UNregularised Quu (tvlqr.cpp:174), then W = Quu Kt - Qt
    const double qk = quu_row[0] * k0 + quu_row[1] * k1 + quu_row[2] * k2 + quu_row[3] * k3;
    const double w_mine = qk - q_mine;   // padding columns: rhs = 0 => Kt = 0, W = 0 without masking
    // ---- expected decrease (tvlqr.cpp:189-191): in column 12 Kt = -d, Qt = Qu, so
    //      d.Qu = -sum_g Kt[g] Qt[g]  and  1/2 d.Quu d = 1/2 sum_g Kt[g] (Quu Kt)[g]
    dv0 = alive ? __builtin_fma(-k_mine, q_mine, dv0) : dv0;
    dv1 = alive ? __builtin_fma(0.5 * k_mine, qk, dv1) : dv1;

    // ---- [P | p] = [Qxx | Qx] + Kt^T W - Qt^T Kt  (tvlqr.cpp:173-186) ----------------------------
    Pn = mfma_f64_16x16x4(k_mine, w_mine, Pn);
    Pn = mfma_f64_16x16x4(q_mine, -k_mine, Pn);
    // ---- the carried P is the stored P: lower triangle <- mirrored upper triangle ------------------
#pragma unroll
    for (int r = 0; r < 3; ++r) ptile[pt_wr[r]] = Pn[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 3; ++r) Pn[r] = ptile[pt_rd[r]];

    // ---- roll the prefetched knot point in BEFORE the stores are issued, so that the wait for its
    //      loads does not also have to drain this step's stores (vmcnt retires in order) ------------
    const double k_store = alive ? k_mine : q_mine;
    cur = nxt;
    // ---- store Kt and [P | p] --------------------------------------------------------------------------
    // Unconditional, branch-free stores: lanes 13..15 duplicate column 12 onto column 12's address,
    // and a failed problem's stores are redirected to its trash record.  With no exec-masked VMEM in
    // the loop hipcc can count the queue exactly and wait for the prefetch with vmcnt(#younger ops)
    // instead of draining the stores too.
    S* __restrict__ ok_ = was_alive ? out + (size_t)k * a.out_ks : trash;
    S* __restrict__ op_ = alive ? out + (size_t)k * a.out_ks : trash;
    ok_[g * 13 + jc] = (S)dpp_col12_dup(k_store);
#pragma unroll
    for (int r = 0; r < 3; ++r) op_[p_off[r]] = (S)Pn[r];
    if (STORE_Q && was_alive) {  // Qxx_, Quu_, Qux_, Qx_, Qu_ are API-visible in the reference
      S* qb = a.qblk + ((size_t)b * N + k) * MF_QB;
#pragma unroll
      for (int r = 0; r < 4; ++r) qb[(g + 4 * r) * 16 + j] = (S)G[r];
      if (g == 0) qb[256 + j] = (S)gv;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) Pt[r] = Pn[r];   // columns 13..15 are exactly 0 by construction
  }
  {
    const double t0 = group4_allreduce(dv0), t1 = group4_allreduce(dv1);
    if (j == 12 && g == 0) {
      a.status[b] = fail_k;   // -1 == TVLQR_SUCCESS, else the failing knot point
      a.delta_V[2 * (size_t)b + 0] = (S)t0;
      a.delta_V[2 * (size_t)b + 1] = (S)t1;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Forward sweep: x_0 = x0 ; u = d - K x ; x+ = f + A x + B u ; y = P x + p  (tvlqr.cpp:208-246).
//
// Lane-per-row with scalar-broadcast operands.  The records are row-major in HBM (Z = [A B] 12x16,
// Kt = [K | -d] 4x13, [P | p] 12x13), so after one coalesced load -> LDS hop every lane reads ITS row
// (16 doubles) and the state vector is broadcast from SGPRs (v_readlane), which makes each product an
// unrolled chain of v_fma_f64 with a scalar operand: no cross-lane reductions at all.
//   lanes  0..15 : row min(l,11) of Z          -> x+[i]   (phase A: A x, phase C: + B u + f)
//   lanes 16..31 : row min(l-16,3) of Kt       -> u[a] = -(K x - d)
//   lanes 32..63 : row min(l&15,11) of [P|p]   -> y[i]
// Spare lanes replicate a neighbour (same value to the same address), so every load, LDS access and
// store in the loop is unconditional (exact vmcnt waits, see the backward kernel).  ~85 instructions
// per knot point instead of ~190 for the DPP-butterfly version: the kernel becomes HBM-bound.
// ------------------------------------------------------------------------------------------------
// LDS image of one knot point: Z with its rows padded to 17 doubles | OUT record | f.  Unpadded, the 12 lanes that
// each read one row of Z (stride 16 doubles = 128 B = all 32 banks) would hit the same bank pair: a 12-way conflict
// on every ds_read_b64; the 13-double rows of Kt / [P|p] are conflict-free as they are.
constexpr int MF_FWD_ZLD = 17;
constexpr int MF_FWD_OUT0 = 12 * MF_FWD_ZLD;          // 204
constexpr int MF_FWD_F0 = MF_FWD_OUT0 + MF_OUT;       // 348
constexpr int MF_FWD_LDS = MF_FWD_F0 + 12;

struct Mfma16FwdRegs {   // one knot point's coalesced loads (7 doubles / lane)
  double z[3], o[3], f;
};
template <typename S>
__device__ __forceinline__ void mfma16_fwd_load(Mfma16FwdRegs& r, const S* __restrict__ rec,
                                                const S* __restrict__ orec, int lane) {
#pragma unroll
  for (int c = 0; c < 3; ++c) r.z[c] = ld_stream(&rec[MF_OFF_Z + c * 64 + lane]);
#pragma unroll
  for (int c = 0; c < 2; ++c) r.o[c] = ld_stream(&orec[c * 64 + lane]);
  r.o[2] = ld_stream(&orec[128 + (lane & 15)]);
  r.f = ld_stream(&rec[MF_OFF_F + (lane < 12 ? lane : 11)]);
}
__device__ __forceinline__ void mfma16_fwd_stage(const Mfma16FwdRegs& r, double* __restrict__ L, int lane) {
#pragma unroll
  for (int c = 0; c < 3; ++c) L[c * 4 * MF_FWD_ZLD + (lane >> 4) * MF_FWD_ZLD + (lane & 15)] = r.z[c];   // row 4c + lane/16
#pragma unroll
  for (int c = 0; c < 2; ++c) L[MF_FWD_OUT0 + c * 64 + lane] = r.o[c];
  L[MF_FWD_OUT0 + 128 + (lane & 15)] = r.o[2];
  L[MF_FWD_F0 + (lane < 12 ? lane : 11)] = r.f;
}
// fp32 storage: the two records of a knot point (DYN 816 B, OUT 576 B, both 16-byte multiples at 16-byte-aligned addresses)
// as ONE 16-byte-per-lane load each -- 51 and 36 lanes' worth, the other lanes repeat the last chunk -- instead of seven
// 4-byte-per-lane loads: the sweep is paid per vector-memory instruction and per request size, not per byte.
// The LDS image built from them is the same, element for element.
typedef float mf_f32x4 __attribute__((ext_vector_type(4)));
struct Mfma16FwdRegsF32 {
  mf_f32x4 dyn, out;
};
__device__ __forceinline__ void mfma16_fwd_load(Mfma16FwdRegsF32& r, const float* __restrict__ rec, const float* __restrict__ orec,
                                                int lane) {
  static_assert(MF_DYN % 4 == 0 && MF_OUT % 4 == 0 && MF_OFF_F % 4 == 0, "16-byte chunks never straddle a block");
  const int cd = lane < MF_DYN / 4 ? lane : MF_DYN / 4 - 1, co = lane < MF_OUT / 4 ? lane : MF_OUT / 4 - 1;
  r.dyn = *reinterpret_cast<const mf_f32x4*>(rec + 4 * cd);
  r.out = *reinterpret_cast<const mf_f32x4*>(orec + 4 * co);
}
__device__ __forceinline__ void mfma16_fwd_stage(const Mfma16FwdRegsF32& r, double* __restrict__ L, int lane) {
  const int cd = lane < MF_DYN / 4 ? lane : MF_DYN / 4 - 1, co = lane < MF_OUT / 4 ? lane : MF_OUT / 4 - 1;
  const int e = 4 * cd;                       // elements e .. e + 3 of the DYN record: a row of Z (16 wide) or f
  const int base = e < MF_OFF_F ? (e >> 4) * MF_FWD_ZLD + (e & 15) : MF_FWD_F0 + (e - MF_OFF_F);
#pragma unroll
  for (int q = 0; q < 4; ++q) L[base + q] = (double)r.dyn[q];
#pragma unroll
  for (int q = 0; q < 4; ++q) L[MF_FWD_OUT0 + 4 * co + q] = (double)r.out[q];
}
template <typename S>
struct Mfma16FwdRing { using type = Mfma16FwdRegs; };
template <>
struct Mfma16FwdRing<float> { using type = Mfma16FwdRegsF32; };
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  union { double d; int i[2]; } in, out;
  in.d = v;
  out.i[0] = __builtin_amdgcn_readlane(in.i[0], src_lane);
  out.i[1] = __builtin_amdgcn_readlane(in.i[1], src_lane);
  return out.d;
}

template <typename S, int DEPTH>
__global__ __launch_bounds__(64) void mfma16_forward_kernel(Mfma16Args<S> a) {
  __shared__ __attribute__((aligned(16))) double lds[MF_FWD_LDS + 4];
  __shared__ double xu_lds[16];   // x_k | u_k: broadcast reads (fp32 storage: the sweep is issue-bound, 32 v_readlane cost more)
  const int lane = threadIdx.x;
  const int b = mf_problem(blockIdx.x, a.batch);
  if (b >= a.batch) return;
  const int N = a.N;
  const S* __restrict__ in = a.in + (size_t)b * a.in_bs;
  const S* __restrict__ out = a.out + (size_t)b * a.out_bs;
  S* __restrict__ xuy = a.xuy + (size_t)b * a.xuy_bs;
  S* __restrict__ trash = a.trash + (size_t)b * MF_OUT;
  // roles
  const int grp = lane >> 4;
  const int sub = lane & 15;
  const int row = (grp == 1) ? (sub < 4 ? sub : 3) : (sub < 12 ? sub : 11);
  const int row_base = (grp == 0) ? MF_FWD_ZLD * row : MF_FWD_OUT0 + 13 * (grp == 1 ? row : 0);
  // LDS addresses of this lane's row: linear for the rows of Z and Kt; the rows of [P | p] are gathered from the
  // packed upper triangle (entries left of the diagonal are the transposed ones) and the p block
  int ra[13];
#pragma unroll
  for (int jj = 0; jj < 12; ++jj) ra[jj] = (grp >= 2) ? MF_FWD_OUT0 + MF_OFF_P + mf_sym(row, jj) : row_base + jj;
  ra[12] = (grp >= 2) ? MF_FWD_OUT0 + MF_OFF_p + row : row_base + 12;
  const int out_off = (grp == 0) ? row : ((grp == 1) ? 24 + row : 12 + row);   // x | y | u inside a record
  const bool is_x = (grp == 0), is_u = (grp == 1);

  double xcur = (double)a.x0[(size_t)b * 12 + row];   // meaningful in group 0: x_k[row]
  // Register ring: knot point k + DEPTH is requested while knot point k computes, so DEPTH records
  // (DEPTH x 3.3 KB per wave) are in flight -- the read-dominated sweep needs that to fill HBM.
  typename Mfma16FwdRing<S>::type ring[DEPTH];
#pragma unroll
  for (int dd = 0; dd < DEPTH; ++dd) {
    const int kk = (dd < N) ? dd : N - 1;
    mfma16_fwd_load(ring[dd], in + (size_t)kk * a.in_ks, out + (size_t)kk * a.out_ks, lane);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): drain before the loop (see the backward kernel)
  mfma16_fwd_stage(ring[0], lds, lane);
  __syncthreads();

  const int Npad = ((N + DEPTH - 1) / DEPTH) * DEPTH;
  for (int k0 = 0; k0 < Npad; k0 += DEPTH) {
#pragma unroll
    for (int dd = 0; dd < DEPTH; ++dd) {
      const int k = k0 + dd;
      const bool live = k < N;     // padding steps (N not a multiple of DEPTH) compute on a clamped record
      {   // slot dd held record k (already staged): refill it with record k + DEPTH (clamped, branch-free)
        const int kn = (k + DEPTH < N) ? k + DEPTH : N - 1;
        mfma16_fwd_load(ring[dd], in + (size_t)kn * a.in_ks, out + (size_t)kn * a.out_ks, lane);
      }
      // this lane's row (16 doubles) and, for the x+ rows, f[row]
      double rd[16];
#pragma unroll
      for (int j = 0; j < 13; ++j) rd[j] = lds[ra[j]];
#pragma unroll
      for (int j = 13; j < 16; ++j) rd[j] = lds[row_base + j];   // columns of B: meaningful in the rows of Z only
      const double fi = lds[MF_FWD_F0 + row];
      // x_k broadcast from lanes 0..11 through SGPRs
      double xs[12];
      if constexpr (sizeof(S) == 4) {
        if (lane < 12) xu_lds[lane] = xcur;
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): one wave, in-order LDS: the write is visible to the reads below
#pragma unroll
        for (int j = 0; j < 12; ++j) xs[j] = xu_lds[j];
      } else {
#pragma unroll
        for (int j = 0; j < 12; ++j) xs[j] = readlane_f64(xcur, j);
      }
      // phase A: sum_j row[j] x[j]   (A x | K x | P x depending on the lane's role)
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 12; ++j) acc = __builtin_fma(rd[j], xs[j], acc);
      // phase B: the affine column.  u = -(K x - d) ; y = P x + p
      const double aff = acc + rd[12];
      const double uval = -aff;
      double us[4];
      if constexpr (sizeof(S) == 4) {
        if (lane >= 16 && lane < 20) xu_lds[lane - 4] = uval;
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
        for (int c = 0; c < 4; ++c) us[c] = xu_lds[12 + c];
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) us[c] = readlane_f64(uval, 16 + c);
      }
      // phase C: x+ = A x + B u + f
      double xn = acc + fi;
#pragma unroll
      for (int c = 0; c < 4; ++c) xn = __builtin_fma(rd[12 + c], us[c], xn);
      // store x_k | y_k | u_k (every lane stores; replicas write identical values)
      const double sval = is_x ? xcur : (is_u ? uval : aff);
      __syncthreads();                       // all reads of this knot point's LDS image are done
      mfma16_fwd_stage(ring[(dd + 1) % DEPTH], lds, lane);   // record k+1: requested DEPTH-1 steps ago
      S* __restrict__ o = live ? xuy + (size_t)k * a.xuy_ks : trash;
      o[out_off] = (S)sval;
      __syncthreads();
      xcur = live ? xn : xcur;
    }
  }
  // terminal knot point: x_N, y_N = P_N x_N + p_N (tvlqr.cpp:238-246), u slot zeroed
  {
    const S* __restrict__ on = a.outn + (size_t)b * MF_TERM;
    double xs[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) xs[j] = readlane_f64(xcur, j);
    double acc = (double)on[row * 13 + 12];
#pragma unroll
    for (int j = 0; j < 12; ++j) acc = __builtin_fma((double)on[row * 13 + j], xs[j], acc);
    const double sval = is_x ? xcur : (is_u ? 0.0 : acc);
    xuy[(size_t)N * a.xuy_ks + out_off] = (S)sval;
  }
}

// ---- MFMA layout self-test: D = A(16x4) B(4x16) + C with the layout this file assumes ----------
__global__ void mfma16_selftest_kernel(const double* A, const double* B, const double* C, double* D) {
  const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
  f64x4 c;
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = C[(g + 4 * r) * 16 + j];
  // A[i][k] row-major 16x4 ; B[k][j] row-major 4x16
  f64x4 d = mfma_f64_16x16x4(A[j * 4 + g], B[g * 16 + j], c);
#pragma unroll
  for (int r = 0; r < 4; ++r) D[(g + 4 * r) * 16 + j] = d[r];
}

}  // namespace altro_hip
