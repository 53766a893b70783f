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
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace altro_hip {

#pragma clang fp contract(off)

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

// One knot point of the backward recursion (tvlqr.cpp:92-192) on a record already in registers.
template <int n, int m, typename T>
__device__ __forceinline__ void lane_backward_step(const T (&cur)[LaneDims<n, m>::E_IN], const LaneBuf& bo, uint32_t lane,
                                                   uint32_t rowB, T reg, int k, T (&P)[n * n], T (&p)[n], T& dv0,
                                                   T& dv1, int& fail_k) {
  using D = LaneDims<n, m>;
  T A[n * n], Bm[n * m], f[n], Qxx[n * n], Quu[m * m], Qux[m * n], Qx[n], Qu[m];
#pragma unroll
  for (int e = 0; e < n * n; ++e) A[e] = cur[D::O_A + e];
#pragma unroll
  for (int e = 0; e < n * m; ++e) Bm[e] = cur[D::O_B + e];
#pragma unroll
  for (int e = 0; e < n; ++e) f[e] = cur[D::O_f + e];
#pragma unroll
  for (int e = 0; e < n * n; ++e) Qxx[e] = cur[D::O_Q + e];
#pragma unroll
  for (int e = 0; e < m * m; ++e) Quu[e] = cur[D::O_R + e];
#pragma unroll
  for (int e = 0; e < m * n; ++e) Qux[e] = cur[D::O_H + e];
#pragma unroll
  for (int e = 0; e < n; ++e) Qx[e] = cur[D::O_q + e];
#pragma unroll
  for (int e = 0; e < m; ++e) Qu[e] = cur[D::O_r + e];

  // Qxx_tmp = A^T P' ; Qux_tmp = B^T P' ; Qx_tmp = p' + P' f        (tvlqr.cpp:135,139,147-148)
  T T1[n * n], T2[m * n], t[n];
#pragma unroll
  for (int j = 0; j < n; ++j)
#pragma unroll
    for (int i = 0; i < n; ++i) {
      T s = T(0);
#pragma unroll
      for (int kk = 0; kk < n; ++kk) s += A[kk + i * n] * P[kk + j * n];
      T1[i + j * n] = T(0) + s;
    }
#pragma unroll
  for (int j = 0; j < n; ++j)
#pragma unroll
    for (int i = 0; i < m; ++i) {
      T s = T(0);
#pragma unroll
      for (int kk = 0; kk < n; ++kk) s += Bm[kk + i * n] * P[kk + j * n];
      T2[i + j * m] = T(0) + s;
    }
#pragma unroll
  for (int i = 0; i < n; ++i) {
    T s = T(0);
#pragma unroll
    for (int kk = 0; kk < n; ++kk) s += P[i + kk * n] * f[kk];
    t[i] = p[i] + s;
  }
  // Qxx += Qxx_tmp A ; Quu += Qux_tmp B ; Qux += Qux_tmp A             (tvlqr.cpp:136,140,143)
#pragma unroll
  for (int j = 0; j < n; ++j)
#pragma unroll
    for (int i = 0; i < n; ++i) {
      T s = T(0);
#pragma unroll
      for (int kk = 0; kk < n; ++kk) s += T1[i + kk * n] * A[kk + j * n];
      Qxx[i + j * n] = Qxx[i + j * n] + s;
    }
#pragma unroll
  for (int j = 0; j < m; ++j)
#pragma unroll
    for (int i = 0; i < m; ++i) {
      T s = T(0);
#pragma unroll
      for (int kk = 0; kk < n; ++kk) s += T2[i + kk * m] * Bm[kk + j * n];
      Quu[i + j * m] = Quu[i + j * m] + s;
    }
#pragma unroll
  for (int j = 0; j < n; ++j)
#pragma unroll
    for (int i = 0; i < m; ++i) {
      T s = T(0);
#pragma unroll
      for (int kk = 0; kk < n; ++kk) s += T2[i + kk * m] * A[kk + j * n];
      Qux[i + j * m] = Qux[i + j * m] + s;
    }
  // Qx = q + A^T Qx_tmp ; Qu = r + B^T Qx_tmp                           (tvlqr.cpp:149-152)
#pragma unroll
  for (int i = 0; i < n; ++i) {
    T s = T(0);
#pragma unroll
    for (int kk = 0; kk < n; ++kk) s += A[kk + i * n] * t[kk];
    Qx[i] = Qx[i] + s;
  }
#pragma unroll
  for (int i = 0; i < m; ++i) {
    T s = T(0);
#pragma unroll
    for (int kk = 0; kk < n; ++kk) s += Bm[kk + i * n] * t[kk];
    Qu[i] = Qu[i] + s;
  }
  // gains: K = Qux, d = -Qu, LL^T = Quu + reg I                          (tvlqr.cpp:155-166)
  T K[m * n], d[m], L[m * m];
#pragma unroll
  for (int e = 0; e < m * n; ++e) K[e] = Qux[e];
#pragma unroll
  for (int e = 0; e < m; ++e) d[e] = -Qu[e];
#pragma unroll
  for (int e = 0; e < m * m; ++e) L[e] = Quu[e] + ((e % m == e / m) ? reg : T(0));
  bool fail = false;
#pragma unroll
  for (int kk = 0; kk < m; ++kk) {
    T x = L[kk + kk * m];
#pragma unroll
    for (int j = 0; j < kk; ++j) x -= L[kk + j * m] * L[kk + j * m];
    if (x <= T(0)) fail = true;
    x = sqrt(x);
    L[kk + kk * m] = x;
#pragma unroll
    for (int i = kk + 1; i < m; ++i) {
      T s = L[i + kk * m];
#pragma unroll
      for (int j = 0; j < kk; ++j) s -= L[i + j * m] * L[kk + j * m];
      L[i + kk * m] = s / x;
    }
  }
  const bool was_alive = fail_k < 0;
  if (was_alive && fail) fail_k = k;
  const bool alive = fail_k < 0;
  if (was_alive && !alive) {  // tvlqr.cpp:162-164: K_k = Qux, d_k = -Qu stay unsolved; stop here
#pragma unroll
    for (int e = 0; e < m * n; ++e) lane_st<T>(bo, lane, (uint32_t)(D::O_K + e) * rowB, K[e]);
#pragma unroll
    for (int e = 0; e < m; ++e) lane_st<T>(bo, lane, (uint32_t)(D::O_d + e) * rowB, d[e]);
  }
  if (!alive) return;   // (lane-divergent only when a problem failed: rare)
#pragma unroll
  for (int c = 0; c < n + 1; ++c) {
    T* rhs = (c < n) ? (K + c * m) : d;
#pragma unroll
    for (int i = 0; i < m; ++i) {
      T s = rhs[i];
#pragma unroll
      for (int j = 0; j < i; ++j) s -= L[i + j * m] * rhs[j];
      rhs[i] = s / L[i + i * m];
    }
#pragma unroll
    for (int i = m - 1; i >= 0; --i) {
      T s = rhs[i];
#pragma unroll
      for (int j = i + 1; j < m; ++j) s -= L[j + i * m] * rhs[j];
      rhs[i] = s / L[i + i * m];
    }
  }
  // cost-to-go                                                             (tvlqr.cpp:173-186)
  T U[m * n], V[n * n], w[m];
#pragma unroll
  for (int j = 0; j < n; ++j)
#pragma unroll
    for (int i = 0; i < m; ++i) {
      T s = T(0);
#pragma unroll
      for (int kk = 0; kk < m; ++kk) s += Quu[i + kk * m] * K[kk + j * m];
      U[i + j * m] = T(0) + s;   // Qux_tmp = Quu K
    }
#pragma unroll
  for (int j = 0; j < n; ++j)
#pragma unroll
    for (int i = 0; i < n; ++i) {
      T s = T(0);
#pragma unroll
      for (int kk = 0; kk < m; ++kk) s += K[kk + i * m] * Qux[kk + j * m];
      V[i + j * n] = T(0) + s;   // Qxx_tmp = K^T Qux
    }
#pragma unroll
  for (int i = 0; i < m; ++i) {
    T s = T(0);
#pragma unroll
    for (int kk = 0; kk < m; ++kk) s += Quu[i + kk * m] * d[kk];
    w[i] = T(0) + s;             // Qu_tmp = Quu d
  }
  T Pk[n * n], pk[n];
#pragma unroll
  for (int j = 0; j < n; ++j)
#pragma unroll
    for (int i = 0; i < n; ++i) {
      T s = T(0);
#pragma unroll
      for (int kk = 0; kk < m; ++kk) s += U[kk + i * m] * K[kk + j * m];
      Pk[i + j * n] = Qxx[i + j * n] + s;
    }
#pragma unroll
  for (int e = 0; e < n * n; ++e) Pk[e] -= V[e];
#pragma unroll
  for (int j = 0; j < n; ++j)
#pragma unroll
    for (int i = 0; i < n; ++i) Pk[i + j * n] -= V[j + i * n];
#pragma unroll
  for (int i = 0; i < n; ++i) {
    T s = T(0);
#pragma unroll
    for (int kk = 0; kk < m; ++kk) s += U[kk + i * m] * d[kk];
    T v = Qx[i] + T(-1) * s;
    T s2 = T(0);
#pragma unroll
    for (int kk = 0; kk < m; ++kk) s2 += K[kk + i * m] * Qu[kk];
    v = v + T(-1) * s2;
    T s3 = T(0);
#pragma unroll
    for (int kk = 0; kk < m; ++kk) s3 += Qux[kk + i * m] * d[kk];
    pk[i] = v + s3;
  }
  {  // tvlqr.cpp:189-191
    T s0 = T(0), s1 = T(0);
#pragma unroll
    for (int i = 0; i < m; ++i) s0 += d[i] * Qu[i];
#pragma unroll
    for (int i = 0; i < m; ++i) s1 += d[i] * w[i];
    dv0 += s0;
    dv1 += T(0.5) * s1;
  }
#pragma unroll
  for (int e = 0; e < m * n; ++e) lane_st<T>(bo, lane, (uint32_t)(D::O_K + e) * rowB, K[e]);
#pragma unroll
  for (int e = 0; e < m; ++e) lane_st<T>(bo, lane, (uint32_t)(D::O_d + e) * rowB, d[e]);
#pragma unroll
  for (int e = 0; e < n * n; ++e) { lane_st<T>(bo, lane, (uint32_t)(D::O_P + e) * rowB, Pk[e]); P[e] = Pk[e]; }
#pragma unroll
  for (int e = 0; e < n; ++e) { lane_st<T>(bo, lane, (uint32_t)(D::O_p + e) * rowB, pk[e]); p[e] = pk[e]; }
}

template <int n, int m, typename T>
__global__ __launch_bounds__(64) void lane_backward_kernel(LaneArgs<T> a) {
  using D = LaneDims<n, m>;
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= a.batch) return;
  if (a.active && !a.active[b]) return;
  const int64_t B = a.batch;
  const int N = a.N;
  const uint32_t lane = threadIdx.x * (uint32_t)sizeof(T);
  const uint32_t rowB = (uint32_t)B * (uint32_t)sizeof(T);
  const int64_t b0 = (int64_t)blockIdx.x * 64;
  const T* __restrict__ pin = a.in + b0;
  T* __restrict__ pout = a.out + b0;
  T P[n * n], p[n];
  {   // all loads first, then all stores: interleaved they serialise into n*n + n round trips
    const LaneBuf bt(a.term + b0), bn(a.outn + b0);
#pragma unroll
    for (int e = 0; e < n * n; ++e) P[e] = lane_ld<T>(bt, lane, (uint32_t)e * rowB);
#pragma unroll
    for (int e = 0; e < n; ++e) p[e] = lane_ld<T>(bt, lane, (uint32_t)(n * n + e) * rowB);
#pragma unroll
    for (int e = 0; e < n * n; ++e) lane_st<T>(bn, lane, (uint32_t)e * rowB, P[e]);
#pragma unroll
    for (int e = 0; e < n; ++e) lane_st<T>(bn, lane, (uint32_t)(n * n + e) * rowB, p[e]);
  }
  T dv0 = T(0), dv1 = T(0);
  int fail_k = -1;
  const T reg = a.reg_pp ? (T)a.reg_pp[b] : a.reg;
  // ping-pong: the record of knot point k - 1 is requested before knot point k is computed (addresses do
  // not depend on the recursion).  (6,3) has no registers left for a second record.
  constexpr bool kPrefetch = D::E_IN <= 64;
  T r0[D::E_IN], r1[kPrefetch ? D::E_IN : 1];
  auto load = [&](T* r, int k) {
    const LaneBuf bi(pin + (int64_t)k * D::E_IN * B);
#pragma unroll
    for (int e = 0; e < D::E_IN; ++e) r[e] = lane_ld<T>(bi, lane, (uint32_t)e * rowB);
  };
  if constexpr (kPrefetch) {
    int k = N - 1;
    load(r0, k);
    for (; k >= 1; k -= 2) {
      load(r1, k - 1);
      lane_backward_step<n, m, T>(r0, LaneBuf(pout + (int64_t)k * D::E_OUT * B), lane, rowB, reg, k, P, p, dv0, dv1, fail_k);
      load(r0, k >= 2 ? k - 2 : 0);
      lane_backward_step<n, m, T>(r1, LaneBuf(pout + (int64_t)(k - 1) * D::E_OUT * B), lane, rowB, reg, k - 1, P, p, dv0, dv1, fail_k);
    }
    if (k == 0)
      lane_backward_step<n, m, T>(r0, LaneBuf(pout), lane, rowB, reg, 0, P, p, dv0, dv1, fail_k);
  } else {
    for (int k = N - 1; k >= 0; --k) {
      load(r0, k);
      lane_backward_step<n, m, T>(r0, LaneBuf(pout + (int64_t)k * D::E_OUT * B), lane, rowB, reg, k, P, p, dv0, dv1, fail_k);
    }
  }
  a.status[b] = fail_k;
  a.delta_V[2 * b + 0] = dv0;
  a.delta_V[2 * b + 1] = dv1;
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

template <int n, int m, typename T>
__device__ __forceinline__ void lane_forward_step(const LaneFwdRec<n, m, T>& r, T* __restrict__ orow, uint32_t rowB,
                                                  uint32_t lane, T (&x)[n]) {
  using D = LaneDims<n, m>;
  T u[m], xn[n], y[n];
#pragma unroll
  for (int i = 0; i < m; ++i) {   // u = d - K x
    T s = T(0);
#pragma unroll
    for (int j = 0; j < n; ++j) s += r.out[D::O_K + i + j * m] * x[j];
    u[i] = r.out[D::O_d + i] + T(-1) * s;
  }
#pragma unroll
  for (int i = 0; i < n; ++i) {   // y = P x + p
    T s = T(0);
#pragma unroll
    for (int j = 0; j < n; ++j) s += r.out[D::O_P + i + j * n] * x[j];
    y[i] = (T(0) + s) + r.out[D::O_p + i];
  }
#pragma unroll
  for (int i = 0; i < n; ++i) {   // x+ = f + A x + B u
    T s = T(0);
#pragma unroll
    for (int j = 0; j < n; ++j) s += r.in[i + j * n] * x[j];
    T v = r.in[n * n + n * m + i] + s;
    T s2 = T(0);
#pragma unroll
    for (int j = 0; j < m; ++j) s2 += r.in[n * n + i + j * n] * u[j];
    xn[i] = v + s2;
  }
  const LaneBuf bx(orow);
#pragma unroll
  for (int e = 0; e < n; ++e) lane_st<T>(bx, lane, (uint32_t)e * rowB, x[e]);
#pragma unroll
  for (int e = 0; e < n; ++e) lane_st<T>(bx, lane, (uint32_t)(n + e) * rowB, y[e]);
#pragma unroll
  for (int e = 0; e < m; ++e) lane_st<T>(bx, lane, (uint32_t)(2 * n + e) * rowB, u[e]);
#pragma unroll
  for (int e = 0; e < n; ++e) x[e] = xn[e];
}

template <int n, int m, typename T>
__global__ __launch_bounds__(64) void lane_forward_kernel(LaneArgs<T> a) {
  using D = LaneDims<n, m>;
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= a.batch) return;
  const int64_t B = a.batch;
  const int N = a.N;
  const uint32_t lane = threadIdx.x * (uint32_t)sizeof(T);   // byte offset of this lane inside a 64-problem row
  const int64_t b0 = (int64_t)blockIdx.x * 64;               // wave-uniform first problem of this wave
  const T* __restrict__ pin = a.in + b0;
  const T* __restrict__ pout = a.out + b0;
  T* __restrict__ pxuy = a.xuy + b0;
  T x[n];
#pragma unroll
  for (int e = 0; e < n; ++e) x[e] = a.x0[(int64_t)e * B + b];
  T PN[n * n + n];
#pragma unroll
  for (int e = 0; e < n * n + n; ++e) PN[e] = a.outn[(int64_t)e * B + b];
  // ping-pong: the record of step k + 1 is in flight while step k runs
  LaneFwdRec<n, m, T> r0, r1;
  const int last = N - 1;
  const uint32_t rowB = (uint32_t)B * (uint32_t)sizeof(T);
  r0.load(pin, pout, B, 0, lane, rowB);
  int k = 0;
  for (; k + 1 < N; k += 2) {
    r1.load(pin, pout, B, k + 1, lane, rowB);
    lane_forward_step<n, m, T>(r0, pxuy + (int64_t)k * D::E_XUY * B, rowB, lane, x);
    r0.load(pin, pout, B, k + 2 < N ? k + 2 : last, lane, rowB);
    lane_forward_step<n, m, T>(r1, pxuy + (int64_t)(k + 1) * D::E_XUY * B, rowB, lane, x);
  }
  if (k < N) lane_forward_step<n, m, T>(r0, pxuy + (int64_t)k * D::E_XUY * B, rowB, lane, x);
  {
    T* o = a.xuy + (int64_t)N * D::E_XUY * B + b;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      T s = T(0);
#pragma unroll
      for (int j = 0; j < n; ++j) s += PN[i + j * n] * x[j];
      o[(int64_t)(n + i) * B] = (T(0) + s) + PN[n * n + i];
    }
#pragma unroll
    for (int e = 0; e < n; ++e) o[(int64_t)e * B] = x[e];
#pragma unroll
    for (int e = 0; e < m; ++e) o[(int64_t)(2 * n + e) * B] = T(0);
  }
}

#pragma clang fp contract(fast)

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
