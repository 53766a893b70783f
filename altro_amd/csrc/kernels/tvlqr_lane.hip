// tvlqr_lane.hip -- plan LANE: one LANE per problem, batch structure-of-arrays, for small (n, m).
//
// For n <= 6 a whole knot point (P', A, B, Q-blocks, gains) fits in one lane's registers, so 64
// consecutive problems ride one wavefront and every load/store is a unit-stride 512-byte run:
//   IN  [k][e][batch]   e over A(n*n) B(n*m) f(n) Q(n*n) R(m*m) H(m*n) q(n) r(m)   (column-major blocks)
//   TERM[e][batch]      e over Q_N(n*n) q_N(n)
//   OUT [k][e][batch]   e over K(m*n) d(m) P(n*n) p(n) ;  OUTN[e][batch] = P_N, p_N
//   XUY [k][e][batch]   e over x(n) y(n) u(m)
// This is the layout BASELINE.json's north star describes ("batched structure-of-arrays in HBM with
// coalesced loads").  The arithmetic follows src/tvlqr/tvlqr.cpp:92-192 / :208-246 statement by
// statement with index-ordered dot products and no FMA fusion, i.e. the same order as the CPU oracle
// and the GENERIC plan, so fp64 results are bit-identical to both.
//
// These shapes (BASELINE.json configs[2], [3]) are latency-bound, not HBM-bound: N dependent steps of
// a few hundred flops each.  Occupancy comes from the batch; nothing here tries to look like a GEMM.
#pragma once
#include "../fp_contract.h"
#include "../rtc_compat.h"

namespace altro_hip {


template <int n, int m>
struct LaneDims {
  static constexpr int E_IN = 2 * n * n + 2 * n * m + m * m + 2 * n + m;
  static constexpr int E_TERM = n * n + n;
  static constexpr int E_OUT = m * n + m + n * n + n;
  static constexpr int E_XUY = 2 * n + m;
  static constexpr int O_A = 0, O_B = n * n, O_f = O_B + n * m, O_Q = O_f + n, O_R = O_Q + n * n,
                       O_H = O_R + m * m, O_q = O_H + m * n, O_r = O_q + n;
  static constexpr int O_K = 0, O_d = m * n, O_P = O_d + m, O_p = O_P + n * n;
};

template <typename T>
struct LaneArgs {
  const T* in;
  const T* term;
  T* out;
  T* outn;
  const T* x0;   // [n][batch]
  T* xuy;
  T* delta_V;    // [batch][2]
  int* status;   // [batch]
  int N;
  int batch;
  T reg;
  const int* active;   // optional per-problem mask (the batched solver skips problems that have stopped)
  const double* reg_pp;  // optional per-problem regularisation (overrides reg; the solver's retry schedule)
};

// Row access for the SoA layout through buffer instructions: a wave-uniform base (the record of knot point
// k, in a buffer resource held in SGPRs), a wave-uniform element offset e * batch * sizeof(T) in an SGPR and
// ONE 32-bit per-lane VGPR offset.  With plain global loads hipcc materialises a 64-bit VGPR address per
// element (v_lshl_add_u64 / v_mad_u64_u32 per load, the addresses parked in AGPRs), which costs more
// issue slots than the arithmetic of these small shapes.
typedef unsigned int lane_v2u __attribute__((ext_vector_type(2)));
struct LaneBuf {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ explicit LaneBuf(const void* base) {
    // gfx9 raw buffer: stride 0, 2 GiB window, word3 = 0x00020000 (DATA_FORMAT 32)
    r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
  }
};
template <typename T>
__device__ __forceinline__ T lane_ld(const LaneBuf& bf, uint32_t lane_off, uint32_t row_off) {
  if constexpr (sizeof(T) == 8) {
    return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(bf.r, lane_off, row_off, 0));
  } else {
    return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(bf.r, lane_off, row_off, 0));
  }
}
template <typename T>
__device__ __forceinline__ void lane_st(const LaneBuf& bf, uint32_t lane_off, uint32_t row_off, T v) {
  if constexpr (sizeof(T) == 8) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(lane_v2u, v), bf.r, lane_off, row_off, 0);
  } else {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), bf.r, lane_off, row_off, 0);
  }
}

// The value lane S of this lane's quad holds (DPP quad_perm:[S,S,S,S]; all four lanes of the quad must be active).
template <int S, typename T>
__device__ __forceinline__ T quad_bcast(T v) {
  constexpr int ctrl = S | (S << 2) | (S << 4) | (S << 6);
  if constexpr (sizeof(T) == 8) {
    lane_v2u w = __builtin_bit_cast(lane_v2u, v);
    w.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w.x, ctrl, 0xf, 0xf, true);
    w.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w.y, ctrl, 0xf, 0xf, true);
    return __builtin_bit_cast(T, w);
  } else {
    return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true));
  }
}

// Lane l of a quad gets the value of lane {A, B, C, D}[l] of the same quad (DPP quad_perm:[A,B,C,D])
template <int A, int B, int C, int D, typename T>
__device__ __forceinline__ T quad_shuf(T v) {
  constexpr int ctrl = A | (B << 2) | (C << 4) | (D << 6);
  if constexpr (sizeof(T) == 8) {
    lane_v2u w = __builtin_bit_cast(lane_v2u, v);
    w.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w.x, ctrl, 0xf, 0xf, true);
    w.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w.y, ctrl, 0xf, 0xf, true);
    return __builtin_bit_cast(T, w);
  } else {
    return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true));
  }
}

// One knot point's forward-pass operands in registers: A | B | f and K | d | P | p.  The addresses never
// depend on the state, so the record of step k + DEPTH is requested before step k is computed: these
// shapes are latency-bound (one wave per 64 problems, N dependent steps) and an un-prefetched step costs
// several exposed HBM round trips.
template <int n, int m, typename T>
struct LaneFwdRec {
  using D = LaneDims<n, m>;
  static constexpr int NA = n * n + n * m + n;
  T in[NA];
  T out[D::E_OUT];
  // `pin` / `pout`: wave-uniform array bases (first problem of this wave); rowB = batch * sizeof(T)
  __device__ __forceinline__ void load(const T* __restrict__ pin, const T* __restrict__ pout, int64_t B, int k,
                                       uint32_t lane, uint32_t rowB) {
    const LaneBuf bi(pin + (int64_t)k * D::E_IN * B), bo(pout + (int64_t)k * D::E_OUT * B);
#pragma unroll
    for (int e = 0; e < NA; ++e) in[e] = lane_ld<T>(bi, lane, (uint32_t)(D::O_A + e) * rowB);
#pragma unroll
    for (int e = 0; e < D::E_OUT; ++e) out[e] = lane_ld<T>(bo, lane, (uint32_t)e * rowB);
  }
};

// ---- the kernels, in two arithmetic flavours (see tvlqr_lane_body.inc) ---------------------------------------------
// Division and square root of the gain solve.  The exact flavour uses the IEEE operations (what keeps the CPU path's bits and
// what the chain of a small sweep mostly waits for: ~350 cycles per knot point for (2, 1)); the fused flavour
// (ALTRO_HIP_LANE_FUSED, already 1e-12 rather than bit-identical) uses v_rcp_f64 / v_rsq_f64 with two Newton steps and one
// correction -- within an ulp or two of the IEEE result, a fraction of its latency.
template <typename T>
__device__ __forceinline__ T lane_fast_div(T a, T b) {
  if constexpr (sizeof(T) == 8) {
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-b, q, a), r, q);
  } else {
    return a / b;
  }
}
template <typename T>
__device__ __forceinline__ T lane_fast_sqrt(T x) {
  if constexpr (sizeof(T) == 8) {
    if (!(x > 0.0)) return sqrt(x);              // zero, negative, NaN: the library's answer (the caller tests x <= 0 first)
    double y = __builtin_amdgcn_rsq(x);
    y = y * __builtin_fma(__builtin_fma(-x * y, y, 1.0), 0.5, 1.0);
    y = y * __builtin_fma(__builtin_fma(-x * y, y, 1.0), 0.5, 1.0);
    const double s = x * y;
    return __builtin_fma(__builtin_fma(-s, s, x), 0.5 * y, s);
  } else {
    return sqrt(x);
  }
}
ALTRO_FP_REGION_OFF
#define LANE_FN(x) x
#define LANE_DIV(a, b) ((a) / (b))
#define LANE_SQRT(x) sqrt(x)
#include "tvlqr_lane_body.inc"
#include "tvlqr_quad_body.inc"
#include "tvlqr_quad2_body.inc"
#undef LANE_FN
#undef LANE_DIV
#undef LANE_SQRT
ALTRO_FP_REGION_FAST
#define LANE_FN(x) x##_fused
#define LANE_DIV(a, b) lane_fast_div((a), (b))
#define LANE_SQRT(x) lane_fast_sqrt(x)
#include "tvlqr_lane_body.inc"
#include "tvlqr_quad_body.inc"
#include "tvlqr_quad2_body.inc"
#undef LANE_FN
#undef LANE_DIV
#undef LANE_SQRT

ALTRO_FP_REGION_END   // back to the including translation unit's own mode (fp_contract.h)

// ---- layout conversion for plan LANE ------------------------------------------------------------
struct LaneSeg {   // one reference-layout source array -> a run of `len` elements of a SoA record
  const double* p;
  int64_t bs, ks;
  int len;         // elements per knot point in the source (n when a diagonal is expanded to n*n)
  int dst_off;     // first element index inside the destination record
  int diag_n;      // > 0: source holds a diagonal of this size, destination is the dense diag_n^2 block
  int bmod;        // > 0: the source holds only `bmod` distinct problems, tiled over the batch
};

// dst[(k*E + dst_off + e)*batch + b]  <-  src[b*bs + k*ks + e]
template <typename T>
__global__ void lane_pack_kernel(T* dst, int E, LaneSeg s, int nk, int k0, int batch) {
  const int dlen = s.diag_n > 0 ? s.diag_n * s.diag_n : s.len;
  const int64_t total = (int64_t)batch * nk * dlen;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(t % batch);
    const int sb = s.bmod > 0 ? (b % s.bmod) : b;
    const int e = (int)((t / batch) % dlen);
    const int k = (int)(t / ((int64_t)batch * dlen));
    double v;
    if (!s.p) v = 0.0;
    else if (s.diag_n > 0) {
      const int i = e % s.diag_n, j = e / s.diag_n;
      v = (i == j) ? s.p[(int64_t)sb * s.bs + (int64_t)(k0 + k) * s.ks + i] : 0.0;
    } else {
      v = s.p[(int64_t)sb * s.bs + (int64_t)(k0 + k) * s.ks + e];
    }
    dst[((int64_t)k * E + s.dst_off + e) * batch + b] = (T)v;
  }
}

// out[(b*nk + k)*len + e] <- src[(k*E + off + e)*batch + b0 + b]   (reference layout, doubles)
template <typename T>
__global__ void lane_unpack_kernel(double* out, const T* src, const T* src_term, int E, int off,
                                   int E_term, int off_term, int len, int nk, int nk_main, int b0,
                                   int nb, int batch) {
  const int64_t total = (int64_t)nb * nk * len;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(t % len);
    const int k = (int)((t / len) % nk);
    const int b = (int)(t / ((int64_t)len * nk));
    T v;
    if (k < nk_main) v = src[((int64_t)k * E + off + e) * batch + b0 + b];
    else v = src_term[((int64_t)off_term + e) * batch + b0 + b];
    (void)E_term;
    out[t] = (double)v;
  }
}

}  // namespace altro_hip
