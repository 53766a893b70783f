// tvlqr_mfma16_f32.hip -- plan MFMA16 in pure fp32: the (12, 4) backward sweep on v_mfma_f32_16x16x4_f32
// (BASELINE.json configs[4]: "quadrotor-sized n=12, m=4, N=512, fp32").
//
// Same algebra, same HBM records (plain row-major blocks, see tvlqr_mfma16.hip) and the same branch-free
// structure as the fp64 kernel; what changes is the accumulator layout.  The f32 16x16x4 tile puts
//        lane l, register r   <->   row 4 (l >> 4) + r,  column l & 15
// (fp64: row (l >> 4) + 4 r).  A-operand lane (g, i) supplies A[i][k = g] and B-operand lane (g, j) supplies
// B[k = g][j] in both, so a register r of a D-layout tile is, read as an A or B operand, the slice "rows
// 4 g + r".  Summing over r = 0..3 and g = 0..3 therefore walks a K index of 16 rows in the order
// 4 g + r -- any order is fine for a contraction as long as both operands use it:
//        D1 = sum_r  mfma(A = [P'|t] reg r,  B = Z rows 4g+r)          rows 12..15 of [P'|t] are zero
//        G  = sum_r  mfma(A = Z rows 4g+r,   B = D1 reg r) + [Q H^T; H R]      Z rows 12..15 are zero
// so D1 still feeds the second product straight from registers.  Row 12 of D1 (the gradient part Z^T t)
// and rows 12..15 of G ([Qux | Quu]) now live in lane group 3 instead of in register 3.
// 10 MFMAs of 32 cycles per knot point (fp64: 8 of 64); the Cholesky / substitutions run in fp32 on the
// full-rate VALU.  Parity: fp32 tolerances (tests/test_gpu_parity.py), stated there.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tvlqr_mfma16.hip"

namespace altro_hip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// lanes 13..15 of every 16-lane row take lane 12's value (see dpp_col12_dup)
__device__ __forceinline__ float dpp_col12_dup_f32(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x00, 0xf, 0x8, false));
}
__device__ __forceinline__ float group4_allreduce_f32(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// 1/sqrt(x): v_rsq_f32 seed + one Newton step
__device__ __forceinline__ float rsqrt_nr_f32(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  const float e = __builtin_fmaf(-x, y * y, 1.0f);
  return __builtin_fmaf(0.5f * y, e, y);
}

struct Mfma16KnotF32 {   // one knot point's inputs, in registers (13 floats / lane)
  float z[4], q[4], qr, f[4];
};

// Unconditional, branch-free loads (clamped addresses for lanes without an element, see the fp64 kernel)
template <bool HAS_F>
__device__ __forceinline__ void mfma16_load_knot_f32(Mfma16KnotF32& kn, const float* __restrict__ rec,
                                                     const float* __restrict__ crec, int zoff, const int (&qo)[4],
                                                     int foff, int j) {
#pragma unroll
  for (int r = 0; r < 4; ++r) kn.z[r] = rec[MF_OFF_Z + zoff + r * 16];
#pragma unroll
  for (int r = 0; r < 4; ++r) kn.q[r] = crec[qo[r]];
  kn.qr = crec[MF_OFF_QR + j];
  if (HAS_F) {
#pragma unroll
    for (int r = 0; r < 4; ++r) kn.f[r] = rec[MF_OFF_F + foff + r];
  } else {
    kn.f[0] = kn.f[1] = kn.f[2] = kn.f[3] = 0.0f;
  }
}

// DEPTH = knot points requested ahead (register ring, loop unrolled DEPTH times so that no ring slot is ever
// copied).  fp32 records are half the bytes of fp64 ones, so one knot point ahead (what the fp64 kernel does)
// leaves only ~7 MB in flight on the whole chip -- less than latency x bandwidth; three keep HBM busy.
template <bool HAS_F, int DEPTH>
__global__ __launch_bounds__(64, 4) void mfma16_backward_f32_kernel(Mfma16Args<float> a) {
  // lds[0..63] = [Qux | Quu] (row r, col j), lds[64..79] = [Qx | Qu], lds[80] = 0 (the "zero slot")
  // lds[82 + ...] = dump area: lanes outside group 3 write their (unused) G rows there, so that the exchange
  // has no exec-masked region -- any branch in this loop degrades hipcc's s_waitcnt placement for the ring
  __shared__ __attribute__((aligned(16))) float lds[64 + 16 + 2 + 5 * 64];
  __shared__ float ptile[12 * 17];   // the new [P | p] tile, written row-major and read back mirrored (symmetry note, tvlqr_mfma16.hip)
  const int lane = threadIdx.x;
  const int j = lane & 15, g = lane >> 4;
  const int b = mf_problem(blockIdx.x, a.batch);
  if (b >= a.batch) return;
  if (a.active && !a.active[b]) return;
  const int N = a.N;
  const bool g3 = (g == 3);          // lane group 3 holds rows 12..15: no rows of Z / Q / P
  if (lane < 2) lds[80 + lane] = 0.0f;
  int rhs_idx[4];   // column j of Qt = [Qux | Qu]; zero for the padding columns 13..15
#pragma unroll
  for (int r = 0; r < 4; ++r) rhs_idx[r] = (j < 12) ? (r * 16 + j) : ((j == 12) ? (64 + 12 + r) : 80);
  const int qmine_idx = (j < 12) ? lane : ((j == 12) ? (64 + 12 + g) : 80);   // Qt[g][j]
  int qx_idx[4];    // column 12 of [Qxx | Qx]: Qx[4g + r]; zero elsewhere
#pragma unroll
  for (int r = 0; r < 4; ++r) qx_idx[r] = (j == 12 && !g3) ? (64 + 4 * g + r) : 80;
  // loop-invariant element offsets of this lane's loads
  const int jq = (j < 12) ? j : 11;
  const int gz = g3 ? 0 : g;                                   // group 3 re-reads group 0's rows and zeroes them
  const int zoff = (4 * gz) * 16 + j;                          // Z[4g + r][j], r -> + 16 r
  int qo[4];   // G-tile init: Q[4g + r][j] gathered from the packed upper triangle | [H R] rows (group 3)
#pragma unroll
  for (int r = 0; r < 4; ++r) qo[r] = g3 ? (MF_OFF_HR + j + 16 * r) : (MF_OFF_Q + mf_sym(4 * g + r, jq));
  const int foff = 4 * gz;
  const float* __restrict__ in = a.in + (size_t)b * a.in_bs;
  const float* __restrict__ cin = a.cin + (size_t)b * a.cin_bs;
  float* __restrict__ out = a.out + (size_t)b * a.out_bs;
  const bool col_ok = (j <= 12);
  const int jc = col_ok ? j : 12;
  float* __restrict__ trash = a.trash + (size_t)b * MF_OUT;
  // where this lane's [P | p] registers (rows 4g + r, column j) go inside an OUT record (see the fp64 kernel)
  int p_off[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * gz + r;
    p_off[r] = (j == 12) ? MF_OFF_p + row : ((j < 12 && j >= row) ? MF_OFF_P + mf_sym(row, j) : MF_OFF_PAD + (j > 12 ? 1 : 0));
  }
  int pt_wr[4], pt_rd[4];     // the exchange that keeps the carried P symmetric (rows 4 gz + r; group 3 shadows group 0's slots)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * gz + r;
    pt_wr[r] = row * 17 + j;
    pt_rd[r] = (j < 12 && j < row) ? j * 17 + row : pt_wr[r];
  }
  const int w_idx = g3 ? j : 82 + lane, w_stride = g3 ? 16 : 64, gv_idx = g3 ? 64 + j : 82 + 4 * 64 + lane;
  const float g_keep = (j < 12 && !g3) ? 1.0f : 0.0f;          // lanes whose G registers are entries of Qxx

  // terminal cost-to-go: tile [P | p], rows 4g + r
  float Pt[4];
  {
    const float* term = a.term + (size_t)b * MF_TERM;
    float* on = a.outn + (size_t)b * MF_TERM;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = 0.0f;
      if (!g3) {
        if (j < 12) v = term[(4 * g + r) * 12 + j];
        else if (j == 12) v = term[144 + 4 * g + r];
        if (col_ok) on[(4 * g + r) * 13 + j] = v;
      }
      Pt[r] = v;
    }
  }
  float dv0 = 0.0f, dv1 = 0.0f;
  int fail_k = -1;
  const float reg = (float)(a.reg_pp ? a.reg_pp[b] : a.reg);

  Mfma16KnotF32 ring[DEPTH];
#pragma unroll
  for (int dd = 0; dd < DEPTH; ++dd) {
    const size_t kk = (N - 1 - dd > 0) ? N - 1 - dd : 0;
    mfma16_load_knot_f32<HAS_F>(ring[dd], in + kk * a.in_ks, cin + kk * a.cin_ks, zoff, qo, foff, j);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): drain before the loop (see the fp64 kernel)

  const int Npad = ((N + DEPTH - 1) / DEPTH) * DEPTH;
  for (int s0 = 0; s0 < Npad; s0 += DEPTH) {
#pragma unroll
   for (int dd = 0; dd < DEPTH; ++dd) {
    const int step = s0 + dd;
    const bool live = step < N;          // padding steps (N not a multiple of DEPTH) recompute knot point 0 into the trash
    const int k = live ? N - 1 - step : 0;
    const Mfma16KnotF32 cur = ring[dd];
    {   // slot dd is consumed: refill it with the record DEPTH knot points further down (clamped, branch-free)
      const size_t kp = (k - DEPTH > 0) ? k - DEPTH : 0;
      mfma16_load_knot_f32<HAS_F>(ring[dd], in + kp * a.in_ks, cin + kp * a.cin_ks, zoff, qo, foff, j);
    }
    float z[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = g3 ? 0.0f : cur.z[r];

    // ---- D1 = [P'|t]^T Z : rows 0..11 = P'^T Z, row 12 (group 3, register 0) = t^T Z ----------------
    f32x4 D1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < 4; ++r) D1 = mfma_f32_16x16x4(Pt[r], z[r], D1);

    // ---- G = [Q H^T; H R] + Z^T D1 ; rows 12..15 ([Qux | Quu]) are group 3's registers ---------------
    f32x4 G = {cur.q[0], cur.q[1], cur.q[2], cur.q[3]};
#pragma unroll
    for (int r = 0; r < 4; ++r) G = mfma_f32_16x16x4(z[r], D1[r], G);

    // ---- gradient [Qx; Qu] = [q; r] + Z^T t (+ Z^T P' f): valid in group 3 ---------------------------
    float gv = D1[0] + cur.qr;
    if (HAS_F) {   // sum_i f[i] D1[i][j] over rows i = 4g + r < 12
      float s = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) s = __builtin_fmaf(g3 ? 0.0f : cur.f[r], D1[r], s);
      gv += group4_allreduce_f32(s);
    }

    // ---- LDS exchange -----------------------------------------------------------------------------------
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) lds[w_idx + r * w_stride] = G[r];   // group 3: [Qux | Quu], row r, col j
    lds[gv_idx] = gv;                                              // group 3: [Qx | Qu]
    __syncthreads();
    const float a00 = lds[0 * 16 + 12];
    const float a10 = lds[1 * 16 + 12], a11 = lds[1 * 16 + 13];
    const float a20 = lds[2 * 16 + 12], a21 = lds[2 * 16 + 13], a22 = lds[2 * 16 + 14];
    const float a30 = lds[3 * 16 + 12], a31 = lds[3 * 16 + 13], a32 = lds[3 * 16 + 14], a33 = lds[3 * 16 + 15];
    float rhs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rhs[r] = lds[rhs_idx[r]];
    float quu_row[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) quu_row[c] = lds[g * 16 + 12 + c];
    f32x4 Pn;   // accumulator init: column j of [Qxx | Qx], rows 4g + r (zero in group 3)
#pragma unroll
    for (int r = 0; r < 4; ++r) Pn[r] = __builtin_fmaf(g_keep, G[r], lds[qx_idx[r]]);   // qx is 0 wherever G is kept
    const float q_mine = lds[qmine_idx];

    // ---- Cholesky of Quu + reg I (lower; tvlqr.cpp:159-164), reciprocal pivots only -------------------
    const float x0 = a00 + reg;
    const float i0 = rsqrt_nr_f32(x0);
    const float l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
    const float x1 = (a11 + reg) - l10 * l10;
    const float i1 = rsqrt_nr_f32(x1);
    const float l21 = (a21 - l20 * l10) * i1, l31 = (a31 - l30 * l10) * i1;
    const float x2 = (a22 + reg) - l20 * l20 - l21 * l21;
    const float i2 = rsqrt_nr_f32(x2);
    const float l32 = (a32 - l30 * l20 - l31 * l21) * i2;
    const float x3 = (a33 + reg) - l30 * l30 - l31 * l31 - l32 * l32;
    const float i3 = rsqrt_nr_f32(x3);
    const bool fail = !(x0 > 0.0f) || !(x1 > 0.0f) || !(x2 > 0.0f) || !(x3 > 0.0f);
    const bool was_alive = (fail_k < 0) && live;
    fail_k = (was_alive && fail) ? k : fail_k;
    const bool alive = (fail_k < 0) && live;
    // ---- Kt[:, j] = (L L^T)^-1 Qt[:, j] ------------------------------------------------------------------
    const float y0 = rhs[0] * i0;
    const float y1 = (rhs[1] - l10 * y0) * i1;
    const float y2 = (rhs[2] - l20 * y0 - l21 * y1) * i2;
    const float y3 = (rhs[3] - l30 * y0 - l31 * y1 - l32 * y2) * i3;
    const float k3 = y3 * i3;
    const float k2 = (y2 - l32 * k3) * i2;
    const float k1 = (y1 - l21 * k2 - l31 * k3) * i1;
    const float k0 = (y0 - l10 * k1 - l20 * k2 - l30 * k3) * i0;
    const float k_mine = (g == 0) ? k0 : (g == 1) ? k1 : (g == 2) ? k2 : k3;  // Kt[g][j]
    const float qk = quu_row[0] * k0 + quu_row[1] * k1 + quu_row[2] * k2 + quu_row[3] * k3;
    const float w_mine = qk - q_mine;
    dv0 = alive ? __builtin_fmaf(-k_mine, q_mine, dv0) : dv0;
    dv1 = alive ? __builtin_fmaf(0.5f * k_mine, qk, dv1) : dv1;

    // ---- [P | p] = [Qxx | Qx] + Kt^T W - Qt^T Kt -------------------------------------------------------------
    Pn = mfma_f32_16x16x4(k_mine, w_mine, Pn);
    Pn = mfma_f32_16x16x4(q_mine, -k_mine, Pn);
    // the carried P is the stored P: lower triangle <- mirrored upper triangle (group 3 holds no rows: it stays out)
    if (!g3) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ptile[pt_wr[r]] = Pn[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) Pn[r] = g3 ? Pn[r] : ptile[pt_rd[r]];

    const float k_store = alive ? k_mine : q_mine;
    // ---- stores: branch-free; group 3 (no rows of [P | p]) and failed problems write to the trash record --
    float* __restrict__ ok_ = was_alive ? out + (size_t)k * a.out_ks : trash;
    float* __restrict__ op_ = (alive && !g3) ? out + (size_t)k * a.out_ks : trash;
    ok_[g * 13 + jc] = dpp_col12_dup_f32(k_store);
#pragma unroll
    for (int r = 0; r < 4; ++r) op_[p_off[r]] = Pn[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) Pt[r] = !live ? Pt[r] : (g3 ? 0.0f : Pn[r]);   // rows 12..15 are not part of [P | p]
   }
  }
  {
    const float t0 = group4_allreduce_f32(dv0), t1 = group4_allreduce_f32(dv1);
    if (j == 12 && g == 0) {
      a.status[b] = fail_k;
      a.delta_V[2 * (size_t)b + 0] = t0;
      a.delta_V[2 * (size_t)b + 1] = t1;
    }
  }
}

}  // namespace altro_hip
