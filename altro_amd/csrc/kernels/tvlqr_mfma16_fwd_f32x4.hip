// tvlqr_mfma16_fwd_f32x4.hip -- plan MFMA16, fp32 records, forward sweep with FOUR problems per wavefront
// (BASELINE.json configs[4]; the backward twin is tvlqr_mfma16_f32x4.hip).
//
//   x_0 = x0 ; u = d - K x ; x+ = f + A x + B u ; y = P x + p          (tvlqr.cpp:208-246)
//
// Why.  One problem per wave (mfma16_forward_kernel on fp32 records) is bound by instruction issue, not by HBM: 16384
// waves x 512 knot points x ~120 instructions, of which only 28 of 64 lanes do arithmetic.  Here a wave
// owns four ADJACENT problems -- one contiguous run of each [k][b] slab -- and
//   * the records arrive as full-width 16-byte-per-lane sweeps (4 for the DYN records, 3 for the OUT records of the four
//     problems) and leave as one (the four x | y | u records): 8 vector-memory instructions per four knot points
//     instead of 12, none of them half empty;
//   * every lane works: lane (p, r) = (lane / 16, lane % 16) owns, for problem p, row r of Z = [A B] (r < 12: x+[r])
//     and -- in the same instruction slots -- row r of [P | p] (r < 12: y[r]) or row r - 12 of Kt = [K | -d] (r >= 12:
//     u[r - 12]); ~150 instructions per wave and knot point for FOUR problems;
//   * the LDS image keeps Z's rows 16-byte aligned and padded to 20 floats, so a row is four conflict-free ds_read_b128.
// Arithmetic is fp32 (v_fma_f32) like the backward sweep of this variant (ALTRO_HIP_F32_PURE; tolerance 2e-5 relative
// through 512 knot points, tests/test_gpu_parity.py); fp32 storage with fp64 arithmetic keeps mfma16_forward_kernel.
// Requires batch % 4 == 0 (the launcher falls back otherwise).  Masking is done by the buffer hardware.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tvlqr_mfma16_f32x4.hip"

namespace altro_hip {

// ---- LDS image of one wave (float indices) ----------------------------------------------------------------------------
constexpr int MFF_ZLD = 20;                        // row stride of Z: 16 + 4 (80 bytes: rows r = 0..15 start in 16 different
                                                   //  4-bank groups, so 16 lanes' ds_read_b128 of a column block do not collide)
constexpr int MFF_F = 12 * MFF_ZLD;                // f (12) behind the rows of Z
constexpr int MFF_OUT = MFF_F + 12;                // the OUT record as it is (144)
constexpr int MFF_PIMG = MFF_OUT + MF_OUT;         // 396 floats per problem (1584 bytes: a multiple of 16)
constexpr int MFF_XS = 4 * MFF_PIMG;               // x_k of the four problems: [p][12]
constexpr int MFF_US = MFF_XS + 48;                // u_k: [p][4]
constexpr int MFF_XUY = MFF_US + 16;               // the four x | y | u records being assembled: [p][28]
constexpr int MFF_DUMP = MFF_XUY + 112;            // 4 floats where lanes without a chunk write
constexpr int MFF_LDS = MFF_DUMP + 4 + 12;         // (+ slack so that the last row's replicated reads stay inside)
static_assert(MFF_PIMG % 4 == 0 && MFF_OUT % 4 == 0 && MFF_XS % 4 == 0 && MFF_XUY % 4 == 0, "16-byte aligned blocks");

struct MffRaw {          // one knot point's DYN and OUT records of the four problems: 7 sweeps of 16 bytes per lane
  f32x4 dyn[4], out[3];
};
__device__ __forceinline__ void mff_load(MffRaw& rw, const float* __restrict__ in4, const float* __restrict__ out4, int lane) {
  const __amdgpu_buffer_rsrc_t ri = mfq_window(in4, 4 * MF_DYN * 4), ro = mfq_window(out4, 4 * MF_OUT * 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) rw.dyn[i] = mfq_ld4(ri, (uint32_t)((i * 64 + lane) * 16));     // beyond 4 x 816 bytes: zeros
#pragma unroll
  for (int i = 0; i < 3; ++i) rw.out[i] = mfq_ld4(ro, (uint32_t)((i * 64 + lane) * 16));     // beyond 4 x 576 bytes: zeros
}

// DEPTH = knot points requested ahead (register ring of raw records, 28 registers per slot); WAVES = waves per SIMD the
// register budget is cut for (C4: 4096 waves = four per SIMD).
template <int DEPTH, int WAVES>
__global__ __launch_bounds__(64, WAVES) void mfma16_forward_f32x4_kernel(Mfma16Args<float> a) {
  __shared__ __attribute__((aligned(16))) float lds[MFF_LDS];
  const int lane = threadIdx.x;
  const int p = lane >> 4, r = lane & 15;
  const int nquad = a.batch >> 2;
  const int quad = mf_problem(blockIdx.x, nquad);
  if (quad >= nquad) return;
  const int b0 = 4 * quad;
  const int N = a.N;
  const bool zrow = r < 12;                      // this lane owns x+[r] and y[r]; otherwise u[r - 12]
  const int rz = zrow ? r : 11;

  // ---- loop-invariant LDS addresses ------------------------------------------------------------------------------------
  int dst_dyn[4], dst_out[3];                    // where this lane's 16 bytes of each sweep go
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int g = i * 64 + lane, pp = g / (MF_DYN / 4), c = g % (MF_DYN / 4);     // 51 chunks per DYN record: 48 of Z, 3 of f
    dst_dyn[i] = pp < 4 ? pp * MFF_PIMG + (c < 48 ? (c >> 2) * MFF_ZLD + (c & 3) * 4 : MFF_F + (c - 48) * 4) : MFF_DUMP;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int g = i * 64 + lane, pp = g / (MF_OUT / 4), c = g % (MF_OUT / 4);     // 36 chunks per OUT record
    dst_out[i] = pp < 4 ? pp * MFF_PIMG + MFF_OUT + c * 4 : MFF_DUMP;
  }
  const int img = p * MFF_PIMG;
  const int zr = img + rz * MFF_ZLD;             // this lane's row of Z (16 floats, four aligned chunks)
  int ra[13];                                    // this lane's row of [P | p] (gathered from the packed triangle) or of Kt
#pragma unroll
  for (int j = 0; j < 12; ++j) ra[j] = img + MFF_OUT + (zrow ? MF_OFF_P + mf_sym(r, j) : (r - 12) * 13 + j);
  ra[12] = img + MFF_OUT + (zrow ? MF_OFF_p + r : (r - 12) * 13 + 12);
  const int xs_r = MFF_XS + p * 12, us_r = MFF_US + p * 4;
  const int xw = MFF_XS + p * 12 + rz;                                   // where x+[r] goes for the next step (rows only)
  const int uw = zrow ? MFF_DUMP : MFF_US + p * 4 + (r - 12);
  const int rec_x = zrow ? MFF_XUY + p * 28 + r : MFF_DUMP + 1;          // x | y | u inside the record being assembled
  const int rec_y = zrow ? MFF_XUY + p * 28 + 12 + r : MFF_XUY + p * 28 + 24 + (r - 12);   // (the u lanes put u here)
  const uint32_t st_off = lane < 28 ? (uint32_t)(lane * 16) : MFQ_OOB;   // 4 x 28 floats = 28 chunks of 16 bytes
  const int st_r = MFF_XUY + (lane < 28 ? lane : 27) * 4;

  const float* __restrict__ in = a.in + (size_t)b0 * a.in_bs;
  const float* __restrict__ out = a.out + (size_t)b0 * a.out_bs;
  float* __restrict__ xuy = a.xuy + (size_t)b0 * a.xuy_bs;

  float xcur = a.x0[(size_t)(b0 + p) * 12 + rz];
  MffRaw ring[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const size_t kk = (size_t)(d < N ? d : N - 1);
    mff_load(ring[d], in + kk * a.in_ks, out + kk * a.out_ks, lane);
  }
  if (zrow) lds[xw] = xcur;

  auto step = [&](MffRaw& slot, int k, int knext) {
    __syncthreads();   // (one wave per block: a compiler fence; the LDS queue itself is in order)
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&lds[dst_dyn[i]]) = slot.dyn[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4*>(&lds[dst_out[i]]) = slot.out[i];
    mff_load(slot, in + (size_t)knext * a.in_ks, out + (size_t)knext * a.out_ks, lane);   // refill: nothing is copied
    __syncthreads();
    // x_k of this lane's problem, broadcast within its 16 lanes
    float xs[12];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[xs_r + 4 * c]);
#pragma unroll
      for (int q = 0; q < 4; ++q) xs[4 * c + q] = v[q];
    }
    // [P | p] row . x (y[r]) in the row lanes, Kt row . x (K x - d) in the other four
    float s = lds[ra[12]];
#pragma unroll
    for (int j = 0; j < 12; ++j) s = __builtin_fmaf(lds[ra[j]], xs[j], s);
    const float uval = -s;                       // u = -(K x - d)
    lds[uw] = uval;
    __syncthreads();
    const f32x4 us = *reinterpret_cast<const f32x4*>(&lds[us_r]);
    // x+ = A x + B u + f
    float xn = lds[img + MFF_F + rz];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f32x4 z = *reinterpret_cast<const f32x4*>(&lds[zr + 4 * c]);
#pragma unroll
      for (int q = 0; q < 4; ++q) xn = __builtin_fmaf(z[q], xs[4 * c + q], xn);
    }
    {
      const f32x4 z = *reinterpret_cast<const f32x4*>(&lds[zr + 12]);
#pragma unroll
      for (int q = 0; q < 4; ++q) xn = __builtin_fmaf(z[q], us[q], xn);
    }
    // the four x | y | u records, assembled in LDS and stored as one 16-byte-per-lane sweep
    lds[rec_x] = xcur;
    lds[rec_y] = zrow ? s : uval;
    __syncthreads();
    {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[st_r]);
      const __amdgpu_buffer_rsrc_t ro = mfq_window(xuy + (size_t)k * a.xuy_ks, 4 * 28 * 4);
      mfq_st4(ro, st_off, v);
    }
    xcur = xn;
    if (zrow) lds[xw] = xn;
  };

  int k = 0;
  for (; k + DEPTH <= N; k += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int kn = k + d + DEPTH;
      step(ring[d], k + d, __builtin_amdgcn_readfirstlane(kn < N ? kn : N - 1));
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)          // the N % DEPTH knot points left sit in slots 0 ..
    if (k + d < N) step(ring[d], k + d, N - 1);

  // terminal knot point: x_N, y_N = P_N x_N + p_N (tvlqr.cpp:238-246), u slot zeroed
  __syncthreads();
  {
    float xs[12];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[xs_r + 4 * c]);
#pragma unroll
      for (int q = 0; q < 4; ++q) xs[4 * c + q] = v[q];
    }
    const float* __restrict__ on = a.outn + (size_t)(b0 + p) * MF_TERM + rz * 13;
    float s = on[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) s = __builtin_fmaf(on[j], xs[j], s);
    lds[rec_x] = xcur;
    lds[rec_y] = zrow ? s : 0.0f;
    __syncthreads();
    const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[st_r]);
    const __amdgpu_buffer_rsrc_t ro = mfq_window(xuy + (size_t)N * a.xuy_ks, 4 * 28 * 4);
    mfq_st4(ro, st_off, v);
  }
}

}  // namespace altro_hip
