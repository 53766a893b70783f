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
};

template <int n, int m, typename T>
__global__ __launch_bounds__(64) void lane_backward_kernel(LaneArgs<T> a) {
  using D = LaneDims<n, m>;
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= a.batch) return;
  if (a.active && !a.active[b]) return;
  const int64_t B = a.batch;
  const int N = a.N;
  T P[n * n], p[n];
#pragma unroll
  for (int e = 0; e < n * n; ++e) { P[e] = a.term[(int64_t)e * B + b]; a.outn[(int64_t)e * B + b] = P[e]; }
#pragma unroll
  for (int e = 0; e < n; ++e) { p[e] = a.term[(int64_t)(n * n + e) * B + b]; a.outn[(int64_t)(n * n + e) * B + b] = p[e]; }
  T dv0 = T(0), dv1 = T(0);
  int fail_k = -1;
  for (int k = N - 1; k >= 0; --k) {
    const T* in = a.in + (int64_t)k * D::E_IN * B + b;
    T* out = a.out + (int64_t)k * D::E_OUT * B + b;
    T A[n * n], Bm[n * m], f[n], Qxx[n * n], Quu[m * m], Qux[m * n], Qx[n], Qu[m];
#pragma unroll
    for (int e = 0; e < n * n; ++e) A[e] = in[(int64_t)(D::O_A + e) * B];
#pragma unroll
    for (int e = 0; e < n * m; ++e) Bm[e] = in[(int64_t)(D::O_B + e) * B];
#pragma unroll
    for (int e = 0; e < n; ++e) f[e] = in[(int64_t)(D::O_f + e) * B];
#pragma unroll
    for (int e = 0; e < n * n; ++e) Qxx[e] = in[(int64_t)(D::O_Q + e) * B];
#pragma unroll
    for (int e = 0; e < m * m; ++e) Quu[e] = in[(int64_t)(D::O_R + e) * B];
#pragma unroll
    for (int e = 0; e < m * n; ++e) Qux[e] = in[(int64_t)(D::O_H + e) * B];
#pragma unroll
    for (int e = 0; e < n; ++e) Qx[e] = in[(int64_t)(D::O_q + e) * B];
#pragma unroll
    for (int e = 0; e < m; ++e) Qu[e] = in[(int64_t)(D::O_r + e) * B];

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
    for (int e = 0; e < m * m; ++e) L[e] = Quu[e] + ((e % m == e / m) ? a.reg : T(0));
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
      for (int e = 0; e < m * n; ++e) out[(int64_t)(D::O_K + e) * B] = K[e];
#pragma unroll
      for (int e = 0; e < m; ++e) out[(int64_t)(D::O_d + e) * B] = d[e];
    }
    if (!alive) continue;   // (lane-divergent only when a problem failed: rare)
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
    for (int e = 0; e < m * n; ++e) out[(int64_t)(D::O_K + e) * B] = K[e];
#pragma unroll
    for (int e = 0; e < m; ++e) out[(int64_t)(D::O_d + e) * B] = d[e];
#pragma unroll
    for (int e = 0; e < n * n; ++e) { out[(int64_t)(D::O_P + e) * B] = Pk[e]; P[e] = Pk[e]; }
#pragma unroll
    for (int e = 0; e < n; ++e) { out[(int64_t)(D::O_p + e) * B] = pk[e]; p[e] = pk[e]; }
  }
  a.status[b] = fail_k;
  a.delta_V[2 * b + 0] = dv0;
  a.delta_V[2 * b + 1] = dv1;
}

template <int n, int m, typename T>
__global__ __launch_bounds__(64) void lane_forward_kernel(LaneArgs<T> a) {
  using D = LaneDims<n, m>;
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= a.batch) return;
  const int64_t B = a.batch;
  const int N = a.N;
  T x[n];
#pragma unroll
  for (int e = 0; e < n; ++e) x[e] = a.x0[(int64_t)e * B + b];
  for (int k = 0; k < N; ++k) {
    const T* in = a.in + (int64_t)k * D::E_IN * B + b;
    const T* out = a.out + (int64_t)k * D::E_OUT * B + b;
    T* o = a.xuy + (int64_t)k * D::E_XUY * B + b;
    T u[m], xn[n];
#pragma unroll
    for (int i = 0; i < m; ++i) {   // u = d - K x
      T s = T(0);
#pragma unroll
      for (int j = 0; j < n; ++j) s += out[(int64_t)(D::O_K + i + j * m) * B] * x[j];
      u[i] = out[(int64_t)(D::O_d + i) * B] + T(-1) * s;
    }
#pragma unroll
    for (int i = 0; i < n; ++i) {   // y = P x + p
      T s = T(0);
#pragma unroll
      for (int j = 0; j < n; ++j) s += out[(int64_t)(D::O_P + i + j * n) * B] * x[j];
      o[(int64_t)(n + i) * B] = (T(0) + s) + out[(int64_t)(D::O_p + i) * B];
    }
#pragma unroll
    for (int i = 0; i < n; ++i) {   // x+ = f + A x + B u
      T s = T(0);
#pragma unroll
      for (int j = 0; j < n; ++j) s += in[(int64_t)(D::O_A + i + j * n) * B] * x[j];
      T v = in[(int64_t)(D::O_f + i) * B] + s;
      T s2 = T(0);
#pragma unroll
      for (int j = 0; j < m; ++j) s2 += in[(int64_t)(D::O_B + i + j * n) * B] * u[j];
      xn[i] = v + s2;
    }
#pragma unroll
    for (int e = 0; e < n; ++e) o[(int64_t)e * B] = x[e];
#pragma unroll
    for (int e = 0; e < m; ++e) o[(int64_t)(2 * n + e) * B] = u[e];
#pragma unroll
    for (int e = 0; e < n; ++e) x[e] = xn[e];
  }
  {
    T* o = a.xuy + (int64_t)N * D::E_XUY * B + b;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      T s = T(0);
#pragma unroll
      for (int j = 0; j < n; ++j) s += a.outn[(int64_t)(i + j * n) * B + b] * x[j];
      o[(int64_t)(n + i) * B] = (T(0) + s) + a.outn[(int64_t)(n * n + i) * B + b];
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
