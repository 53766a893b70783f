// tvlqr_mfma16_f32x4.hip -- plan MFMA16, pure fp32, FOUR problems per wavefront (BASELINE.json configs[4]).
//
// Why.  With one problem per wave (tvlqr_mfma16_f32.hip) the (12, 4) sweep in fp32 is NOT bound by HBM:
//   * a wave64 vector-memory instruction occupies the CU's address unit for 16 cycles whatever its width, and that
//     kernel issues 15 of them per knot point, 4 bytes per lane, many lanes masked or duplicated (the gathers from the
//     packed triangle, the padding columns of the stores): 126 M instructions x 16 cycles / 256 CUs = 4.0 ms -- the
//     kernel's whole duration (profiles/r02e_c4pure_pmc.txt);
//   * the 4 x 4 Cholesky and the 13 triangular solves are the same work in all four 16-lane groups of the wave.
// Here a wave owns four ADJACENT problems, whose records are one contiguous run of a [k][b] slab, and
//   * every global access is a full-width 16-byte-per-lane sweep over that run -- 4 + 3 loads and 3 stores per four knot
//     points (10 instead of 60) -- into / out of an LDS image that is simply the four records back to back; the operand
//     fragments (rows 4g + r of Z, the gathers from the packed triangle of Q) are then LDS reads, which cost 2 cycles;
//   * the work is done where it is not redundant:
//       tile domain   (per problem p, lane (g, j) <-> rows 4g + r, column j of a 16 x 16 tile):
//            D1_p = [P'|t]^T Z,  G_p = [Q H^T; H R] + Z^T D1_p             8 x v_mfma_f32_16x16x4_f32 per problem
//       column domain (lane group p <-> problem p, lane j <-> column j of Qt = [Qux | Qu]):
//            Cholesky of Quu_p + reg I, Kt_p[:, j] = Quu^-1 Qt[:, j], W = Quu Kt - Qt    ONCE per wave for 4 problems
//       back in tiles: [P | p] = [Qxx | Qx] + Kt^T W - Qt^T Kt for all four problems at once on the 4-block form
//            v_mfma_f32_16x16x1_4b_f32: block p takes its A / B operand from lane group p -- exactly where the column
//            domain left column i of Kt_p, W_p, Qt_p -- and returns problem p's tile in registers 4p .. 4p+3 in the tile
//            layout (8 instructions for the four problems: rank-1 updates c = 0..3 of the two products).
// Same algebra and the same HBM records as the other MFMA16 kernels (the forward sweep and the iLQR kernels read what
// this one writes); fp32 tolerances in tests/test_gpu_parity.py.  Requires batch % 4 == 0 (the launcher falls back to the
// one-problem kernel otherwise).  Masking is done by the buffer hardware: loads beyond a window return 0, stores beyond
// it are dropped.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tvlqr_mfma16_f32.hip"

namespace altro_hip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma_f32_16x16x1_4b(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c, 0, 0, 0);
}

// buffer window helpers: raw buffer (stride 0), DATA_FORMAT 32; offsets at or above the window's size are out of range
constexpr uint32_t MFQ_OOB = 0x60000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mfq_window(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 mfq_ld4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
typedef unsigned int mfq_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfq_st4(__amdgpu_buffer_rsrc_t r, uint32_t off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mfq_u4, v), r, off, 0, 0);
}
// max(x, 0) of a wave-uniform index, pinned to the scalar unit: hipcc turns the clamp of a loop counter into a
// saturating VALU subtract, which drags the window's base address into VGPRs -- and a buffer resource in VGPRs costs a
// waterfall loop per access.
__device__ __forceinline__ int mfq_clamp0(int x) { return __builtin_amdgcn_readfirstlane(x < 0 ? 0 : x); }

// ---- LDS image of one wave (float indices) ----------------------------------------------------------------------------------
constexpr int MFQ_DYN = 0;                      // the four DYN records back to back (4 x 204), padded to 4 sweeps of 64 x 4
constexpr int MFQ_COST = 1024;                  // the four COST records (4 x 160), padded to 3 sweeps
constexpr int MFQ_OUT = MFQ_COST + 768;         // the four OUT records being assembled (4 x 144), padded to 3 sweeps
constexpr int MFQ_OUT_DUMP = MFQ_OUT + 4 * MF_OUT;   // 192 spare floats of that image: where lanes without an entry write
constexpr int MFQ_QT = MFQ_OUT + 768;           // [p][j][r]  rows 12..15 of G_p (= [Qux | Quu]), column j: 4 x 16 x 4
constexpr int MFQ_QT_DUMP = MFQ_QT + 256;       // [lane][4]  where the lanes outside group 3 put their (unused) G registers
constexpr int MFQ_PS = MFQ_QT + 512;            // [g][j][p]  gradient partial sums of lane (g, j) for the four problems
constexpr int MFQ_GV = MFQ_PS + 256;            // [p][j]     [Qx | Qu] of problem p
constexpr int MFQ_ZERO = MFQ_GV + 64;           // 64 zeros
constexpr int MFQ_LDS = MFQ_ZERO + 64;          // 3456 floats = 13.5 KB

struct MfqRaw {          // one knot point's records of the four problems as they sit in HBM: 7 sweeps of 16 bytes per lane
  f32x4 dyn[4], cost[3];
};
__device__ __forceinline__ void mfq_load(MfqRaw& rw, const float* __restrict__ in4, const float* __restrict__ cin4, int lane) {
  const __amdgpu_buffer_rsrc_t ri = mfq_window(in4, 4 * MF_DYN * 4), rc = mfq_window(cin4, 4 * MF_COST * 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) rw.dyn[i] = mfq_ld4(ri, (uint32_t)((i * 64 + lane) * 16));     // beyond 4 x 816 bytes: zeros
#pragma unroll
  for (int i = 0; i < 3; ++i) rw.cost[i] = mfq_ld4(rc, (uint32_t)((i * 64 + lane) * 16));
}

// per-wave state that lives across knot points
struct MfqState {
  f32x16 Pt;          // tile domain: [P' | t] of the four problems, registers 4p + r <-> rows 4g + r of problem p
  float dv0, dv1;     // column domain (meaningful in lane j == 12 of each group)
  int fail_k;         // column domain: -1 alive, >= 0 knot point of the first failure, -2 problem masked out
};
struct MfqLanes {     // loop-invariant per-lane LDS addresses (float indices) and store offsets
  int z, f, q[4], qr;             // operand fragments inside the DYN / COST images (+ p x record size)
  int qt_w[4], ps_w, gv_w;        // exchange: writes
  int quu_r, ps_r, rhs_r, qx_r;   // exchange: reads
  int k_w, k_stride, p_w[4];      // OUT image: Kt[r][j] of problem g (+ k_stride r), [P | p] entries of a tile (+ p x 144)
  int p_m[4];                     // ... and where the entries BELOW the diagonal read their mirror image back (else -1): the P the
                                  //     recursion carries is the P it stores (see the symmetry note of the fp64 kernel)
  uint32_t st_off[3];             // global offset of this lane's 16 bytes in each of the three store sweeps (or out of window)
  int st_sel[3];                  // bit position in the ballot masks of the problem those 16 bytes belong to, + 64 for the Kt part
};

// ---- tile domain, front: records -> LDS image -> operand fragments; D1_p = [P'|t]^T Z_p, G_p = C_p + Z_p^T D1_p, and the
//      gradient partial sums.  Consumes the ring slot (the caller refills it right after). ------------------------------------------
template <bool HAS_F>
__device__ __forceinline__ void mfq_front(float* __restrict__ lds, const MfqState& st, const MfqRaw& rw, const MfqLanes& L,
                                          int lane, bool g3, f32x16& Gall, f32x4& ps) {
  __syncthreads();   // (one wave per block: a compiler fence; the LDS queue itself is in order)
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&lds[MFQ_DYN + (i * 64 + lane) * 4]) = rw.dyn[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4*>(&lds[MFQ_COST + (i * 64 + lane) * 4]) = rw.cost[i];
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float z[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = lds[L.z + p * MF_DYN + r * 16];     // Z[4g + r][j]
      z[r] = g3 ? 0.0f : v;                               // rows 12..15 of the K index do not exist
    }
    f32x4 G;
#pragma unroll
    for (int r = 0; r < 4; ++r) G[r] = lds[L.q[r] + p * MF_COST];   // Q[4g + r][j] from the packed triangle | [H R] rows
    const float qr = lds[L.qr + p * MF_COST];
    f32x4 D1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < 4; ++r) D1 = mfma_f32_16x16x4(st.Pt[4 * p + r], z[r], D1);
#pragma unroll
    for (int r = 0; r < 4; ++r) G = mfma_f32_16x16x4(z[r], D1[r], G);
    // [Qx; Qu][j] = [q; r][j] + (Z^T t)[j] + sum_i f[i] D1[i][j]: row 12 of D1 and [q r] sit in group 3 (register 0),
    // the f-weighted rows in groups 0..2
    float s = D1[0] + qr;
    if (HAS_F) {
      const f32x4 f = *reinterpret_cast<const f32x4*>(&lds[L.f + p * MF_DYN]);   // f[4g .. 4g+3] (group 3: not used)
      float sf = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) sf = __builtin_fmaf(f[r], D1[r], sf);
      s = g3 ? s : sf;
    } else {
      s = g3 ? s : 0.0f;
    }
    ps[p] = s;
#pragma unroll
    for (int r = 0; r < 4; ++r) Gall[4 * p + r] = G[r];
  }
}

// ---- the rest of one knot point: column domain, cost-to-go update, the OUT records ----------------------------------------------
__device__ __forceinline__ void mfq_back(const Mfma16Args<float>& a, float* __restrict__ lds, MfqState& st, const f32x16& Gall,
                                         const f32x4& ps, int k, int b0, int lane, float reg, float g_keep, const MfqLanes& L) {
  // tile -> column domain
#pragma unroll
  for (int p = 0; p < 4; ++p)
    *reinterpret_cast<f32x4*>(&lds[L.qt_w[p]]) = f32x4{Gall[4 * p], Gall[4 * p + 1], Gall[4 * p + 2], Gall[4 * p + 3]};
  *reinterpret_cast<f32x4*>(&lds[L.ps_w]) = ps;
  __syncthreads();
  // lane (p = g, j): Quu_p by columns (the same in the 16 lanes of the group), the gradient entry of column j
  const f32x4 c0 = *reinterpret_cast<const f32x4*>(&lds[L.quu_r]);
  const f32x4 c1 = *reinterpret_cast<const f32x4*>(&lds[L.quu_r + 4]);
  const f32x4 c2 = *reinterpret_cast<const f32x4*>(&lds[L.quu_r + 8]);
  const f32x4 c3 = *reinterpret_cast<const f32x4*>(&lds[L.quu_r + 12]);
  const float gvj = (lds[L.ps_r] + lds[L.ps_r + 64]) + (lds[L.ps_r + 128] + lds[L.ps_r + 192]);
  lds[L.gv_w] = gvj;
  // ---- Cholesky of Quu + reg I (lower triangle, tvlqr.cpp:159-164), reciprocal pivots only ------------------------------
  const float x0 = c0[0] + reg;
  const float i0 = rsqrt_nr_f32(x0);
  const float l10 = c0[1] * i0, l20 = c0[2] * i0, l30 = c0[3] * i0;
  const float x1 = (c1[1] + reg) - l10 * l10;
  const float i1 = rsqrt_nr_f32(x1);
  const float l21 = (c1[2] - l20 * l10) * i1, l31 = (c1[3] - l30 * l10) * i1;
  const float x2 = (c2[2] + reg) - l20 * l20 - l21 * l21;
  const float i2 = rsqrt_nr_f32(x2);
  const float l32 = (c2[3] - l30 * l20 - l31 * l21) * i2;
  const float x3 = (c3[3] + reg) - l30 * l30 - l31 * l31 - l32 * l32;
  const float i3 = rsqrt_nr_f32(x3);
  const bool fail = !(x0 > 0.0f) || !(x1 > 0.0f) || !(x2 > 0.0f) || !(x3 > 0.0f);
  const bool was_alive = st.fail_k == -1;
  st.fail_k = (was_alive && fail) ? k : st.fail_k;
  const bool alive = st.fail_k == -1;
  __syncthreads();
  // column j of Qt = [Qux | Qu]: rows 12..15 of G (j < 12), Qu (j == 12), zero in the padding columns
  const f32x4 rhs = *reinterpret_cast<const f32x4*>(&lds[L.rhs_r]);
  // tile domain again: column 12 of [Qxx | Qx] for the accumulator init (zero wherever a G register is kept)
  f32x4 qx[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) qx[p] = *reinterpret_cast<const f32x4*>(&lds[L.qx_r + 16 * p]);
  // ---- Kt[:, j] = (L L^T)^-1 Qt[:, j] (tvlqr.cpp:165-166; column 12 gives -d) -----------------------------------------
  const float y0 = rhs[0] * i0;
  const float y1 = (rhs[1] - l10 * y0) * i1;
  const float y2 = (rhs[2] - l20 * y0 - l21 * y1) * i2;
  const float y3 = (rhs[3] - l30 * y0 - l31 * y1 - l32 * y2) * i3;
  float kt[4];
  kt[3] = y3 * i3;
  kt[2] = (y2 - l32 * kt[3]) * i2;
  kt[1] = (y1 - l21 * kt[2] - l31 * kt[3]) * i1;
  kt[0] = (y0 - l10 * kt[1] - l20 * kt[2] - l30 * kt[3]) * i0;
  // (Quu Kt)[:, j] with the UNregularised Quu (tvlqr.cpp:174), W = Quu Kt - Qt
  float qk[4], w[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    qk[r] = c0[r] * kt[0] + c1[r] * kt[1] + c2[r] * kt[2] + c3[r] * kt[3];
    w[r] = qk[r] - rhs[r];
  }
  // expected decrease (tvlqr.cpp:189-191) from column 12: Kt = -d, Qt = Qu
  {
    float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { d0 = __builtin_fmaf(-kt[r], rhs[r], d0); d1 = __builtin_fmaf(0.5f * kt[r], qk[r], d1); }
    st.dv0 += alive ? d0 : 0.0f;
    st.dv1 += alive ? d1 : 0.0f;
  }
  // ---- [P | p] = [Qxx | Qx] + Kt^T W - Qt^T Kt, four problems per instruction ---------------------------------------------
  f32x16 Pn;
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) Pn[4 * p + r] = __builtin_fmaf(g_keep, Gall[4 * p + r], qx[p][r]);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    Pn = mfma_f32_16x16x1_4b(kt[c], w[c], Pn);
    Pn = mfma_f32_16x16x1_4b(rhs[c], -kt[c], Pn);
  }
  // ---- assemble the four OUT records in LDS, then three full-width store sweeps ------------------------------------------------
  // Kt (column domain); the failing knot point itself stores Qt: K_k = Qux, d_k = -Qu left unsolved (tvlqr.cpp:162-164)
#pragma unroll
  for (int r = 0; r < 4; ++r) lds[L.k_w + L.k_stride * r] = alive ? kt[r] : rhs[r];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) lds[L.p_w[r] + p * MF_OUT] = Pn[4 * p + r];
  __syncthreads();
  // the carried P is the stored P: entries below the diagonal <- their mirror image in the records just assembled
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float mv = lds[(L.p_m[r] >= 0 ? L.p_m[r] : MFQ_ZERO) + (L.p_m[r] >= 0 ? p * MF_OUT : 0)];
      Pn[4 * p + r] = L.p_m[r] >= 0 ? mv : Pn[4 * p + r];
    }
  {
    // which problems may store: Kt of those alive BEFORE this knot point, [P | p] of those still alive after it
    const unsigned long long was_mask = __ballot(was_alive), alive_mask = __ballot(alive);
    const __amdgpu_buffer_rsrc_t ro = mfq_window(a.out + ((size_t)b0 * a.out_bs + (size_t)k * a.out_ks), 4 * MF_OUT * 4);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[MFQ_OUT + (i * 64 + lane) * 4]);
      const unsigned long long m = (L.st_sel[i] & 64) ? was_mask : alive_mask;
      const bool ok = (m >> (L.st_sel[i] & 63)) & 1ull;
      mfq_st4(ro, ok ? L.st_off[i] : MFQ_OOB, v);
    }
  }
  st.Pt = Pn;   // rows 12..15 (group 3) hold finite leftovers: they only ever meet the zero rows 12..15 of Z
}

// DEPTH = knot points requested ahead (register ring of raw records, 28 registers per slot; a slot is refilled as soon as
// its knot point has been staged into LDS, so no slot is ever copied); WAVES = waves per SIMD the register budget is cut
// for.  One wave alternates an MFMA-heavy phase (40 matrix instructions per four knot points) with a VALU / LDS-heavy
// one and cannot overlap the two -- the recursion is serial -- so the overlap comes from the second wave on the SIMD.
template <bool HAS_F, int DEPTH, int WAVES>
__global__ __launch_bounds__(64, WAVES) void mfma16_backward_f32x4_kernel(Mfma16Args<float> a) {
  __shared__ __attribute__((aligned(16))) float lds[MFQ_LDS];
  const int lane = threadIdx.x;
  const int j = lane & 15, g = lane >> 4;
  const int nquad = a.batch >> 2;
  const int quad = mf_problem(blockIdx.x, nquad);
  if (quad >= nquad) return;
  const int b0 = 4 * quad;
  const int N = a.N;
  const bool g3 = (g == 3);
  // problems the batched solver has masked out behave like failed ones (no stores), but keep their status
  int masked = 0;
  if (a.active) {
    masked = a.active[b0 + g] ? 0 : 1;                       // column domain: lane group g <-> problem b0 + g
    if (__ballot(!masked) == 0ull) return;                   // the whole quad is inactive
  }
  for (int i = lane; i < MFQ_LDS; i += 64) lds[i] = 0.0f;    // (the zero block and the images' padding stay zero)
  const float g_keep = (j < 12 && !g3) ? 1.0f : 0.0f;        // lanes whose G registers are entries of Qxx
  const float reg = (float)(a.reg_pp ? a.reg_pp[b0 + g] : a.reg);

  // ---- loop-invariant LDS addresses / store offsets ----------------------------------------------------------------------------
  MfqLanes L;
  const int jq = (j < 12) ? j : 11;
  const int gz = g3 ? 0 : g;                                              // group 3 reads group 0's rows and zeroes them
  L.z = MFQ_DYN + MF_OFF_Z + (4 * gz) * 16 + j;                           // Z[4g + r][j], r -> + 16
  L.f = MFQ_DYN + MF_OFF_F + 4 * gz;
#pragma unroll
  for (int r = 0; r < 4; ++r)   // G-tile init: Q[4g + r][j] from the packed upper triangle | [H R] rows (group 3)
    L.q[r] = MFQ_COST + (g3 ? (MF_OFF_HR + j + 16 * r) : (MF_OFF_Q + mf_sym(4 * g + r, jq)));
  L.qr = MFQ_COST + MF_OFF_QR + j;
#pragma unroll
  for (int p = 0; p < 4; ++p) L.qt_w[p] = g3 ? (MFQ_QT + p * 64 + j * 4) : (MFQ_QT_DUMP + lane * 4);
  L.ps_w = MFQ_PS + (g * 16 + j) * 4;
  L.gv_w = MFQ_GV + g * 16 + j;
  L.quu_r = MFQ_QT + g * 64 + 12 * 4;                       // columns 12..15 of problem g: 16 consecutive floats
  L.ps_r = MFQ_PS + j * 4 + g;                              // + 64 g' : partial sums of lane (g', j) for problem g
  L.rhs_r = (j < 12) ? (MFQ_QT + g * 64 + j * 4) : ((j == 12) ? (MFQ_GV + g * 16 + 12) : MFQ_ZERO);
  L.qx_r = (j == 12 && !g3) ? (MFQ_GV + 4 * g) : MFQ_ZERO;   // + 16 p : Qx_p[4g .. 4g+3]
  // Kt[r][j] of problem g goes to float r * 13 + j of its record; the padding columns write into the image's spare tail
  L.k_w = (j <= 12) ? (MFQ_OUT + g * MF_OUT + j) : (MFQ_OUT_DUMP + lane);
  L.k_stride = (j <= 12) ? 13 : 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * g + r;
    const bool has = !g3 && (j == 12 || (j < 12 && j >= row));
    // entries below the diagonal, the padding columns and group 3 go to each record's two spare slots (nobody reads them)
    L.p_w[r] = MFQ_OUT + (has ? ((j == 12) ? (MF_OFF_p + row) : (MF_OFF_P + mf_sym(row, j))) : (MF_OFF_PAD + (lane & 1)));
    L.p_m[r] = (!g3 && j < 12 && j < row) ? MFQ_OUT + MF_OFF_P + mf_sym(row, j) : -1;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int q4 = i * 64 + lane;                            // this lane's 16 bytes: float4 number q4 of the four OUT records
    const int prob = q4 / (MF_OUT / 4), within = q4 % (MF_OUT / 4);
    L.st_off[i] = (prob < 4) ? (uint32_t)(q4 * 16) : MFQ_OOB;
    L.st_sel[i] = 16 * (prob < 4 ? prob : 0) + (within < 13 ? 64 : 0);    // Kt = the first 52 floats = 13 float4 of a record
  }

  // ---- terminal cost-to-go: [P_N | p_N] = [Q_N | q_N] (tvlqr.cpp:81-90) -> tiles, and OUTN --------------------------------------
  MfqState st;
  st.dv0 = st.dv1 = 0.0f;
  st.fail_k = masked ? -2 : -1;
  {
    const unsigned long long live_mask = __ballot(!masked);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float* term = a.term + (size_t)(b0 + p) * MF_TERM;
      float* on = a.outn + (size_t)(b0 + p) * MF_TERM;
      const bool wr = (live_mask >> (16 * p)) & 1ull;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = 0.0f;
        if (!g3) {
          if (j < 12) v = term[(4 * g + r) * 12 + j];
          else if (j == 12) v = term[144 + 4 * g + r];
          if (j <= 12 && wr) on[(4 * g + r) * 13 + j] = v;
        }
        st.Pt[4 * p + r] = v;
      }
    }
  }
  const float* __restrict__ in = a.in + (size_t)b0 * a.in_bs;
  const float* __restrict__ cin = a.cin + (size_t)b0 * a.cin_bs;

  // ---- the sweep: a DEPTH-deep register ring of raw records (loop unrolled DEPTH times) ------------------------------------------
  MfqRaw ring[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const size_t kk = (size_t)mfq_clamp0(N - 1 - d);
    mfq_load(ring[d], in + kk * a.in_ks, cin + kk * a.cin_ks, lane);
  }
  int k = N - 1;
  f32x16 Gall;
  f32x4 ps;
  for (; k >= DEPTH - 1; k -= DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      mfq_front<HAS_F>(lds, st, ring[d], L, lane, g3, Gall, ps);
      const size_t kn = (size_t)mfq_clamp0(k - d - DEPTH);
      mfq_load(ring[d], in + kn * a.in_ks, cin + kn * a.cin_ks, lane);
      mfq_back(a, lds, st, Gall, ps, k - d, b0, lane, reg, g_keep, L);
    }
  }
  // the N % DEPTH knot points left (k .. 0) sit in slots 0 .. k
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
    if (k - d >= 0) {
      mfq_front<HAS_F>(lds, st, ring[d], L, lane, g3, Gall, ps);
      mfq_back(a, lds, st, Gall, ps, k - d, b0, lane, reg, g_keep, L);
    }

  if (j == 12 && st.fail_k != -2) {
    a.status[b0 + g] = st.fail_k;   // -1 == TVLQR_SUCCESS, else the failing knot point
    a.delta_V[2 * (size_t)(b0 + g) + 0] = st.dv0;
    a.delta_V[2 * (size_t)(b0 + g) + 1] = st.dv1;
  }
}

// ---- layout self-test of the 4-block form: D_b = A_b (16x1) B_b (1x16) + C_b for b = 0..3 -------------------------------------
//   A: lane 16 b + i supplies A_b[i];  B: lane 16 b + j supplies B_b[j];  D: register 4 b + r, lane 16 g + j <-> D_b[4 g + r][j]
__global__ void mfma16_selftest_4b_kernel(const float* A, const float* B, const float* C, float* D) {
  const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
  f32x16 c;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[4 * b + r] = C[b * 256 + (4 * g + r) * 16 + j];
  const f32x16 d = mfma_f32_16x16x1_4b(A[lane], B[lane], c);
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) D[b * 256 + (4 * g + r) * 16 + j] = d[4 * b + r];
}

}  // namespace altro_hip
