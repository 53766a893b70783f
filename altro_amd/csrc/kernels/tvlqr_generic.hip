// tvlqr_generic.hip -- plan GENERIC: one wavefront per problem, knot-point blocks staged in LDS.
//
// The catch-all implementation of the TVLQR pair for ANY (n_k, m_k) (per-knot-point dimensions
// allowed, like the reference's `const int* nx, nu` -- src/tvlqr/tvlqr.h:17-33).  It follows the
// operation order of src/tvlqr/tvlqr.cpp:92-192 (backward) and :208-246 (forward) one statement at
// a time, with index-ordered dot products and no FMA contraction (this file is compiled with
// -ffp-contract=off), so that its fp64 results agree with the CPU oracle to the last bit.  It is the
// correctness anchor for the fast plans and the engine behind the single-problem tvlqr_* drop-in;
// it is NOT the path bench.py measures (that is tvlqr_mfma16.hip).
//
// Mapping: block = 64 threads = 1 wavefront = 1 problem.  Every small product C = op(A) op(B) is
// spread over the lanes by output element (lane e computes C[e], C[e+64], ...); operands live in LDS
// so the broadcast reads are bank-conflict-free or same-address.  The k-recursion is serial
// (P_{k+1} -> P_k), P_{k+1}/p_{k+1} never leave LDS.
#pragma once
#include "../fp_contract.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "generic_arrays.h"

namespace altro_hip {

// Bit-parity with the CPU oracle: no a*b+c -> fma fusion anywhere in this file.
ALTRO_FP_REGION_OFF

constexpr size_t kGenericLdsLimit = 64 * 1024;   // dynamic LDS a launch gets without asking for more; beyond it plan GENERIC works in global memory
constexpr size_t kGenericForwardStageLimit = 10 * 1024;   // batched forward sweep: stage the knot point in LDS up to this much per problem (sixteen waves per CU)
constexpr int kGenericMaxDim = 256;              // n, m accepted by plan GENERIC (a bound on the work blocks, not of the algorithm)

// (enum GArr: kernels/generic_arrays.h)

template <typename T>
struct GenericArgs {
  T* base[G_NUM];            // per-array base pointer
  int64_t bstride[G_NUM];    // elements between consecutive problems
  const int64_t* off;        // [(N+1) * G_NUM] element offset of knot point k inside a problem
  const int* nx;             // [N+1]
  const int* nu;             // [N+1] (nu[N] unused)
  const T* x0;               // [batch][nx[0]]
  int64_t x0_stride;
  T* delta_V;                // [batch][2]
  int* status;               // [batch]
  int N;
  int batch;
  int nmax, mmax;
  T reg;
  int is_diag;
  int store_q;
  int want_y;
  int no_f = 0;              // backward sweep: treat f as zero (altro_hip_batch::ilqr_linear: the expansion of the iLQR loop)
  T* ws = nullptr;           // generic_backward_kernel<T, true>: per-problem work blocks in global memory instead of LDS
  int64_t ws_stride = 0;     // elements between consecutive problems' work blocks (>= generic_backward_lds_bytes / sizeof(T))
};

// e -> (e % d, e / d) for 0 <= e < 2^22, d > 0, rd ~ 1 / d: a float product and a fix-up instead of the integer division sequence
__device__ __forceinline__ void split_index(int e, int d, float rd, int& r, int& q) {
  q = (int)((float)e * rd);
  r = e - q * d;
  if (r < 0) { r += d; --q; }
  else if (r >= d) { r -= d; ++q; }
}

// C(mr x nc) = beta*C + alpha * op(A) op(B); operands column-major in LDS (or global for B/A reads).
// The sum over k is taken in index order, one product at a time (the CPU path's order); the OPERANDS of four consecutive terms are
// fetched together first: the trip count is a run-time value, the loop is not unrolled by the compiler, and with one fetch per
// iteration every term paid a full LDS (or global) round trip (round 5: plan GENERIC's (13, 4) sweep 6.6 -> 5.25 ms, (14, 7) 12.9 -> 10.7,
// the seam's (20, 8) call 605 -> 475 us, (12, 4) 250 -> 225; same bits).
template <typename T>
__device__ __forceinline__ void wave_gemm(int lane, int ta, int tb, int mr, int nc, int kd, T alpha,
                                          const T* A, int lda, const T* B, int ldb, T beta, T* C,
                                          int ldc) {
  const int total = mr * nc;
  const float rmr = __builtin_amdgcn_rcpf((float)mr);
  for (int e = lane; e < total; e += 64) {
    int i, j;
    split_index(e, mr, rmr, i, j);
    const T* pa = ta ? A + i * lda : A + i;
    const T* pb = tb ? B + j : B + j * ldb;
    const int sa = ta ? 1 : lda, sb = tb ? ldb : 1;
    T s = T(0);
    int k = 0;
    for (; k + 4 <= kd; k += 4) {
      const T a0 = pa[(k + 0) * sa], a1 = pa[(k + 1) * sa], a2 = pa[(k + 2) * sa], a3 = pa[(k + 3) * sa];
      const T b0 = pb[(k + 0) * sb], b1 = pb[(k + 1) * sb], b2 = pb[(k + 2) * sb], b3 = pb[(k + 3) * sb];
      s += a0 * b0;
      s += a1 * b1;
      s += a2 * b2;
      s += a3 * b3;
    }
    for (; k < kd; ++k) s += pa[k * sa] * pb[k * sb];
    const T c0 = (beta == T(0)) ? T(0) : beta * C[i + j * ldc];
    C[i + j * ldc] = c0 + alpha * s;
  }
}

// The same product on the matrix cores (plan GENERIC with ALTRO_HIP_GENERIC_MATRIX_CORES; fp64): 16 x 16 output tiles, four terms per
// v_mfma_f64_16x16x4.  Lane l = (g = l / 16, j = l % 16) feeds A[row j][term g] and B[term g][column j] of the tile and receives
// rows g, g + 4, g + 8, g + 12 of column j; entries past the blocks' own dimensions are fed as zeros, so any (mr, nc, kd) takes
// ceil(mr / 16) ceil(nc / 16) ceil(kd / 4) instructions with two LDS reads each -- against mr nc kd / 64 multiply-add pairs with four
// reads per pair in wave_gemm.  The sums are the matrix pipe's (fused, its own order): results agree with wave_gemm's to rounding,
// not bit for bit, which is why the plan's default and the tvlqr_* drop-in keep wave_gemm.
typedef double gen_f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wave_gemm_mfma(int lane, int ta, int tb, int mr, int nc, int kd, double alpha, const double* A, int lda,
                                               const double* B, int ldb, double beta, double* C, int ldc) {
  const int j = lane & 15, g = lane >> 4;
  const int ai = ta ? lda : 1, ak = ta ? 1 : lda;      // op(A)(i, k) = A[i ai + k ak]
  const int bk = tb ? ldb : 1, bj = tb ? 1 : ldb;      // op(B)(k, c) = B[k bk + c bj]
  for (int j0 = 0; j0 < nc; j0 += 16) {
    for (int i0 = 0; i0 < mr; i0 += 16) {
      gen_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
      const bool arow = i0 + j < mr, bcol = j0 + j < nc;
      const double* pa = A + (i0 + j) * ai + g * ak;
      const double* pb = B + (j0 + j) * bj + g * bk;
      for (int k0 = 0; k0 < kd; k0 += 4) {
        const bool kin = k0 + g < kd;
        const double av = (arow && kin) ? pa[k0 * ak] : 0.0;
        const double bv = (bcol && kin) ? pb[k0 * bk] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
      if (bcol) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = i0 + g + 4 * r;
          if (row < mr) {
            double* c = C + row + (j0 + j) * ldc;
            const double c0 = (beta == 0.0) ? 0.0 : beta * *c;
            *c = c0 + alpha * acc[r];
          }
        }
      }
    }
  }
}
template <typename T, bool MF>
__device__ __forceinline__ void wave_gemm_sel(int lane, int ta, int tb, int mr, int nc, int kd, T alpha, const T* A, int lda, const T* B,
                                              int ldb, T beta, T* C, int ldc) {
  if constexpr (MF && sizeof(T) == 8) wave_gemm_mfma(lane, ta, tb, mr, nc, kd, alpha, A, lda, B, ldb, beta, C, ldc);
  else wave_gemm<T>(lane, ta, tb, mr, nc, kd, alpha, A, lda, B, ldb, beta, C, ldc);
}

// Two products with the same number of terms and alpha = 1 in ONE pass when their outputs fit the wave together (lanes 0 .. c1-1 the
// first, c1 .. c1+c2-1 the second): a pass costs a lone wave ~1000 cycles whatever it computes, and a knot point has several products
// of a dozen outputs.  Each output's sum is wave_gemm's, term by term.  More than 64 outputs together: one after the other.
template <typename T>
struct GemmOp { int ta, tb, mr, nc; const T* A; int lda; const T* B; int ldb; T beta; T* C; int ldc; };
template <typename T>
__device__ __forceinline__ void wave_gemm_pair(int lane, int kd, const GemmOp<T>& o1, const GemmOp<T>& o2) {
  const int c1 = o1.mr * o1.nc, c2 = o2.mr * o2.nc;
  if (c1 + c2 > 64) {
    wave_gemm<T>(lane, o1.ta, o1.tb, o1.mr, o1.nc, kd, T(1), o1.A, o1.lda, o1.B, o1.ldb, o1.beta, o1.C, o1.ldc);
    wave_gemm<T>(lane, o2.ta, o2.tb, o2.mr, o2.nc, kd, T(1), o2.A, o2.lda, o2.B, o2.ldb, o2.beta, o2.C, o2.ldc);
    return;
  }
  if (lane >= c1 + c2) return;
  const bool w2 = lane >= c1;
  const int e = w2 ? lane - c1 : lane;
  const int mr = w2 ? o2.mr : o1.mr;
  int i, j;
  split_index(e, mr, __builtin_amdgcn_rcpf((float)mr), i, j);
  const T* pa = w2 ? (o2.ta ? o2.A + i * o2.lda : o2.A + i) : (o1.ta ? o1.A + i * o1.lda : o1.A + i);
  const T* pb = w2 ? (o2.tb ? o2.B + j : o2.B + j * o2.ldb) : (o1.tb ? o1.B + j : o1.B + j * o1.ldb);
  const int sa = w2 ? (o2.ta ? 1 : o2.lda) : (o1.ta ? 1 : o1.lda), sb = w2 ? (o2.tb ? o2.ldb : 1) : (o1.tb ? o1.ldb : 1);
  T* pc = w2 ? o2.C + i + j * o2.ldc : o1.C + i + j * o1.ldc;
  const T beta = w2 ? o2.beta : o1.beta;
  T s = T(0);
  int k = 0;
  for (; k + 4 <= kd; k += 4) {
    const T a0 = pa[(k + 0) * sa], a1 = pa[(k + 1) * sa], a2 = pa[(k + 2) * sa], a3 = pa[(k + 3) * sa];
    const T b0 = pb[(k + 0) * sb], b1 = pb[(k + 1) * sb], b2 = pb[(k + 2) * sb], b3 = pb[(k + 3) * sb];
    s += a0 * b0;
    s += a1 * b1;
    s += a2 * b2;
    s += a3 * b3;
  }
  for (; k < kd; ++k) s += pa[k * sa] * pb[k * sb];
  const T c0 = (beta == T(0)) ? T(0) : beta * *pc;
  *pc = c0 + T(1) * s;
}
template <typename T, bool MF>
__device__ __forceinline__ void wave_gemm_pair_sel(int lane, int kd, const GemmOp<T>& o1, const GemmOp<T>& o2) {
  if constexpr (MF && sizeof(T) == 8) {
    wave_gemm_mfma(lane, o1.ta, o1.tb, o1.mr, o1.nc, kd, 1.0, o1.A, o1.lda, o1.B, o1.ldb, o1.beta, o1.C, o1.ldc);
    wave_gemm_mfma(lane, o2.ta, o2.tb, o2.mr, o2.nc, kd, 1.0, o2.A, o2.lda, o2.B, o2.ldb, o2.beta, o2.C, o2.ldc);
  } else {
    wave_gemm_pair<T>(lane, kd, o1, o2);
  }
}

// dst[0 .. count) <- src: four elements per lane in flight (a copy from global memory pays one round trip per 256 elements, not per 64)
template <typename T>
__device__ __forceinline__ void wave_copy(int lane, T* dst, const T* src, int count) {
  for (int e0 = 0; e0 < count; e0 += 256) {
    T v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + lane + 64 * q;
      v[q] = e < count ? src[e] : T(0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + lane + 64 * q;
      if (e < count) dst[e] = v[q];
    }
  }
}

// sum_i a[i] b[i], i = 0 .. cnt-1, in index order, four operand pairs fetched at a time
template <typename T>
__device__ __forceinline__ T gen_dot(const T* a, const T* b, int cnt) {
  T s = T(0);
  int j = 0;
  for (; j + 4 <= cnt; j += 4) {
    const T a0 = a[j], a1 = a[j + 1], a2 = a[j + 2], a3 = a[j + 3];
    const T b0 = b[j], b1 = b[j + 1], b2 = b[j + 2], b3 = b[j + 3];
    s += a0 * b0;
    s += a1 * b1;
    s += a2 * b2;
    s += a3 * b3;
  }
  for (; j < cnt; ++j) s += a[j] * b[j];
  return s;
}

// ---- the gain solve of small input dimensions in registers --------------------------------------------------------------------------
// Quu + reg I = L L^T (unblocked lower Cholesky, Eigen's llt_inplace order: fail when a pivot is <= 0) for m <= M with L in the
// registers of EVERY lane (the same instructions wave-wide: the cost of one lane doing it, without the LDS round trip per entry
// that one lane walking the matrix in LDS pays), then one right-hand side per lane against those registers.
#define GEN_TRI(i, j) ((i) * ((i) + 1) / 2 + (j))
template <typename T, int M>
__device__ __forceinline__ int chol_regs(const T* sL, int m, T (&L)[M * (M + 1) / 2]) {
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) L[GEN_TRI(i, j)] = i < m ? sL[i + j * m] : (i == j ? T(1) : T(0));
  int fail = 0;
#pragma unroll
  for (int kk = 0; kk < M; ++kk) {
    if (kk < m && !fail) {
      T x = L[GEN_TRI(kk, kk)];
#pragma unroll
      for (int j = 0; j < kk; ++j) x -= L[GEN_TRI(kk, j)] * L[GEN_TRI(kk, j)];
      if (x <= T(0)) {
        fail = 1;
      } else {
        x = sqrt(x);
        L[GEN_TRI(kk, kk)] = x;
#pragma unroll
        for (int i = kk + 1; i < M; ++i) {
          if (i < m) {
            T s = L[GEN_TRI(i, kk)];
#pragma unroll
            for (int j = 0; j < kk; ++j) s -= L[GEN_TRI(i, j)] * L[GEN_TRI(kk, j)];
            L[GEN_TRI(i, kk)] = s / x;
          }
        }
      }
    }
  }
  return fail;
}
template <typename T, int M>
__device__ __forceinline__ void solve_regs(const T (&L)[M * (M + 1) / 2], int m, T* rhs) {
  T v[M];
#pragma unroll
  for (int i = 0; i < M; ++i) v[i] = i < m ? rhs[i] : T(0);
#pragma unroll
  for (int i = 0; i < M; ++i) {
    if (i < m) {
      T s = v[i];
#pragma unroll
      for (int j = 0; j < i; ++j) s -= L[GEN_TRI(i, j)] * v[j];
      v[i] = s / L[GEN_TRI(i, i)];
    }
  }
#pragma unroll
  for (int i = M - 1; i >= 0; --i) {
    if (i < m) {
      T s = v[i];
#pragma unroll
      for (int j = i + 1; j < M; ++j) {
        if (j < m) s -= L[GEN_TRI(j, i)] * v[j];
      }
      v[i] = s / L[GEN_TRI(i, i)];
    }
  }
#pragma unroll
  for (int i = 0; i < M; ++i) {
    if (i < m) rhs[i] = v[i];
  }
}
// x - sum_j a[j sa] b[j sb], j = 0 .. cnt-1, one product at a time in index order; four operand pairs fetched at a time
template <typename T>
__device__ __forceinline__ T gen_dot_sub(T x, const T* a, int sa, const T* b, int sb, int cnt) {
  int j = 0;
  for (; j + 4 <= cnt; j += 4) {
    const T a0 = a[(j + 0) * sa], a1 = a[(j + 1) * sa], a2 = a[(j + 2) * sa], a3 = a[(j + 3) * sa];
    const T b0 = b[(j + 0) * sb], b1 = b[(j + 1) * sb], b2 = b[(j + 2) * sb], b3 = b[(j + 3) * sb];
    x -= a0 * b0;
    x -= a1 * b1;
    x -= a2 * b2;
    x -= a3 * b3;
  }
  for (; j < cnt; ++j) x -= a[j * sa] * b[j * sb];
  return x;
}

// m > 4: the same factorisation column by column with the rows below the pivot spread over the lanes (every lane forms the pivot
// itself: same addresses, same instructions), then one right-hand side per lane against the factor in the work image.
template <typename T>
__device__ __forceinline__ int gain_solve_lanes(int lane, T* sL, T* sK, T* sd, int n, int m) {
  for (int kk = 0; kk < m; ++kk) {
    T x = gen_dot_sub<T>(sL[kk + kk * m], sL + kk, m, sL + kk, m, kk);
    if (x <= T(0)) return 1;                           // (uniform)
    x = sqrt(x);
    for (int i = kk + 1 + lane; i < m; i += 64) {
      const T sv = gen_dot_sub<T>(sL[i + kk * m], sL + i, m, sL + kk, m, kk);
      sL[i + kk * m] = sv / x;
    }
    if (lane == 0) sL[kk + kk * m] = x;
    __syncthreads();
  }
  for (int c = lane; c < n + 1; c += 64) {
    T* rhs = (c < n) ? (sK + c * m) : sd;
    for (int i = 0; i < m; ++i) rhs[i] = gen_dot_sub<T>(rhs[i], sL + i, m, rhs, 1, i) / sL[i + i * m];
    for (int i = m - 1; i >= 0; --i) rhs[i] = gen_dot_sub<T>(rhs[i], sL + (i + 1) + i * m, 1, rhs + i + 1, 1, m - 1 - i) / sL[i + i * m];
  }
  return 0;
}

template <typename T, int M>
__device__ __forceinline__ int gain_solve_regs(int lane, T* sL, T* sK, T* sd, int n, int m, bool keep_factor) {
  T L[M * (M + 1) / 2];
  const int fail = chol_regs<T, M>(sL, m, L);
  __syncthreads();                                     // (every lane has read the matrix before lane 0 writes the factor over it)
  if (keep_factor && lane == 0) {                      // Quu_tmp of the reference holds the factor (as far as it got)
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        if (i < m) sL[i + j * m] = L[GEN_TRI(i, j)];
      }
  }
  if (fail) return 1;
  for (int c = lane; c < n + 1; c += 64) solve_regs<T, M>(L, m, (c < n) ? (sK + c * m) : sd);
  return 0;
}

// A block on its way from global memory to the work image: CAP x 64 elements in registers (every block of a knot point is fetched
// before the first is stored: one round trip for the knot point instead of one per block); what is past that is copied at the point of use
template <typename T, int CAP>
__device__ __forceinline__ void reg_fetch(T (&v)[CAP], const T* src, int count, int lane) {
#pragma unroll
  for (int c = 0; c < CAP; ++c) {
    const int e = lane + 64 * c;
    v[c] = e < count ? src[e] : T(0);
  }
}
template <typename T, int CAP>
__device__ __forceinline__ void reg_put(T* dst, const T (&v)[CAP], const T* src, int count, int lane) {
#pragma unroll
  for (int c = 0; c < CAP; ++c) {
    const int e = lane + 64 * c;
    if (e < count) dst[e] = v[c];
  }
  if (count > 64 * CAP) wave_copy(lane, dst + 64 * CAP, src + 64 * CAP, count - 64 * CAP);
}

// BIG = false: the knot point's blocks live in LDS (what fits 64 KB: n, m up to ~32 in fp64).  BIG = true: the same code on a
// per-problem work block in GLOBAL memory (args.ws) -- any dimensions, as the reference takes them (tvlqr.cpp:92-121 sizes every
// block from nx[k], nu[k]); slow (every operand is a cached global load), which is the point: refusing n = 33 is worse.  A workgroup
// is one wave and __syncthreads() orders its global accesses at workgroup scope, so the phases stay as they are.
// LQ ("late Q"): Qxx has no block of its own -- Q_k is brought in AFTER the products with P' (which then includes Qx_tmp += P' f), into
// the block P' leaves, and Qxx, then P_k, are built there: 3 n^2 + .. elements instead of 4 n^2 + .., for the shapes and batches where
// that puts more problems on a CU (the launch decides; same sums, one global round trip in the middle of the knot point).
template <typename T, bool BIG = false, bool MF = false, bool LQ = false>
__global__ __launch_bounds__(64, BIG ? 1 : 4) void generic_backward_kernel(GenericArgs<T> a) {   // (in LDS: 128 registers, four waves per SIMD -- 4096 problems resident)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  if (b >= a.batch) return;
  T* smem = BIG ? a.ws + (int64_t)b * a.ws_stride : reinterpret_cast<T*>(smem_raw);
  const int nm = a.nmax, mm = a.mmax;
  // LDS carve
  T* sP = smem;                 // P_{k+1}, then P_k in its place     nm*nm
  T* sp = sP + nm * nm;         // p_{k+1}, then p_k                  nm
  T* sA = sp + nm;              // A_k         nm*nm
  T* sB = sA + nm * nm;         // B_k         nm*mm
  T* sf = sB + nm * mm;         // f_k         nm
  T* sQxx = LQ ? sP : sf + nm;  // nm*nm (LQ: the block of P')
  T* sQuu = LQ ? sf + nm : sQxx + nm * nm;     // mm*mm
  T* sQux = sQuu + mm * mm;     // mm*nm
  T* sQx = sQux + mm * nm;      // nm
  T* sQu = sQx + nm;            // mm
  T* sT1 = sQu + mm;            // Qxx_tmp nm*nm
  T* sT2 = sT1 + nm * nm;       // Qux_tmp mm*nm
  T* st = sT2 + mm * nm;        // Qx_tmp  nm
  T* sL = st + nm;              // Quu_tmp mm*mm
  T* sK = sL + mm * mm;         // mm*nm
  T* sd = sK + mm * nm;         // mm
  T* sw = sd + mm;              // Qu_tmp mm

#define GPTR(arr, k) (a.base[arr] + (int64_t)b * a.bstride[arr] + a.off[(int64_t)(k) * G_NUM + arr])
  const int N = a.N;
  T dv0 = T(0), dv1 = T(0);
  // terminal cost-to-go (tvlqr.cpp:81-90)
  {
    const int n = a.nx[N];
    const T* Qn = GPTR(G_Q, N);
    if (a.is_diag) {
      for (int e = lane; e < n * n; e += 64) sP[e] = (e % n == e / n) ? Qn[e % n] : T(0);
    } else {
      wave_copy(lane, sP, Qn, n * n);
    }
    wave_copy(lane, sp, (const T*)GPTR(G_q, N), n);
    __syncthreads();
    wave_copy(lane, GPTR(G_P, N), (const T*)sP, n * n);
    wave_copy(lane, GPTR(G_p, N), (const T*)sp, n);
  }
  __syncthreads();

  for (int k = N - 1; k >= 0; --k) {
    const int n = a.nx[k], m = a.nu[k], n2 = a.nx[k + 1];
    // stage the knot point: all its blocks fetched, then all stored (a lone wave otherwise waits a global round trip per block: 5-8
    // thousand cycles of 37 per knot point at (13, 4), 17 of 57 with 4096 waves in flight)
    {   // (two batches: all eight at once is one round trip less but 32 registers more than the rest of the kernel needs -- three waves per SIMD)
      T rA[4], rB[2], rf[1], rq[1], rr[1];
      const T *gA = GPTR(G_A, k), *gB = GPTR(G_B, k), *gf = GPTR(G_f, k), *gq = GPTR(G_q, k), *gr = GPTR(G_r, k);
      reg_fetch<T, 4>(rA, gA, n2 * n, lane);
      reg_fetch<T, 2>(rB, gB, n2 * m, lane);
      if (!a.no_f) reg_fetch<T, 1>(rf, gf, n2, lane);
      reg_fetch<T, 1>(rq, gq, n, lane);
      reg_fetch<T, 1>(rr, gr, m, lane);
      reg_put<T, 4>(sA, rA, gA, n2 * n, lane);
      reg_put<T, 2>(sB, rB, gB, n2 * m, lane);
      if (a.no_f) { for (int e = lane; e < n2; e += 64) sf[e] = T(0); }   // the iLQR loop's expansion carries no affine term (knotpoint_data.cpp:416)
      else reg_put<T, 1>(sf, rf, gf, n2, lane);
      reg_put<T, 1>(sQx, rq, gq, n, lane);
      reg_put<T, 1>(sQu, rr, gr, m, lane);
    }
    {
      T rQ[4], rR[1], rH[2];
      const T *gQ = GPTR(G_Q, k), *gR = GPTR(G_R, k), *gH = GPTR(G_H, k);
      const int cQ = a.is_diag ? n : n * n, cR = a.is_diag ? m : m * m;
      if (!LQ || a.is_diag) reg_fetch<T, 4>(rQ, gQ, cQ, lane);
      reg_fetch<T, 1>(rR, gR, cR, lane);
      if (!a.is_diag) reg_fetch<T, 2>(rH, gH, m * n, lane);
      if (a.is_diag) {  // tvlqr.cpp:125-128: the diagonals through scratch blocks, then spread (LQ: Q's waits in K's block)
        reg_put<T, 4>(LQ ? sK : sT1, rQ, gQ, n, lane);
        reg_put<T, 1>(sT2, rR, gR, m, lane);
      } else {  // tvlqr.cpp:129-133
        if (!LQ) reg_put<T, 4>(sQxx, rQ, gQ, n * n, lane);
        reg_put<T, 1>(sQuu, rR, gR, m * m, lane);
        reg_put<T, 2>(sQux, rH, gH, m * n, lane);
      }
    }
    if (a.is_diag) {
      __syncthreads();
      const float rn = __builtin_amdgcn_rcpf((float)n), rmf = __builtin_amdgcn_rcpf((float)(m > 0 ? m : 1));
      if (!LQ)
        for (int e = lane; e < n * n; e += 64) { int i, j; split_index(e, n, rn, i, j); sQxx[e] = (i == j) ? sT1[i] : T(0); }
      for (int e = lane; e < m * m; e += 64) { int i, j; split_index(e, m, rmf, i, j); sQuu[e] = (i == j) ? sT2[i] : T(0); }
      for (int e = lane; e < m * n; e += 64) sQux[e] = T(0);
    }
    __syncthreads();
    // Qxx_tmp = A^T P' ; Qux_tmp = B^T P' ; Qx_tmp = p' + P' f     (tvlqr.cpp:135,139,147-148)
    wave_gemm_sel<T, MF>(lane, 1, 0, n, n2, n2, T(1), sA, n2, sP, n2, T(0), sT1, n);
    wave_gemm_sel<T, MF>(lane, 1, 0, m, n2, n2, T(1), sB, n2, sP, n2, T(0), sT2, m);
    wave_copy(lane, st, (const T*)sp, n2);
    if (LQ) {   // the last use of P': Qx_tmp += P' f; then Q_k into its block
      __syncthreads();
      wave_gemm_sel<T, MF>(lane, 0, 0, n2, 1, n2, T(1), sP, n2, sf, n2, T(1), st, n2);
      __syncthreads();
      if (a.is_diag) {
        const float rn = __builtin_amdgcn_rcpf((float)n);
        for (int e = lane; e < n * n; e += 64) { int i, j; split_index(e, n, rn, i, j); sQxx[e] = (i == j) ? sK[i] : T(0); }
      } else {
        wave_copy(lane, sQxx, (const T*)GPTR(G_Q, k), n * n);
      }
    }
    __syncthreads();
    if (LQ) wave_gemm_sel<T, MF>(lane, 0, 0, m, m, n2, T(1), sT2, m, sB, n2, T(1), sQuu, m);        // Quu += Qux_tmp B   (:140)
    else
    wave_gemm_pair_sel<T, MF>(lane, n2, GemmOp<T>{0, 0, n2, 1, sP, n2, sf, n2, T(1), st, n2},     // Qx_tmp += P' f
                              GemmOp<T>{0, 0, m, m, sT2, m, sB, n2, T(1), sQuu, m});        // Quu += Qux_tmp B   (:140)
    wave_gemm_sel<T, MF>(lane, 0, 0, n, n, n2, T(1), sT1, n, sA, n2, T(1), sQxx, n);   // :136
    wave_gemm_sel<T, MF>(lane, 0, 0, m, n, n2, T(1), sT2, m, sA, n2, T(1), sQux, m);   // :143
    __syncthreads();
    if (LQ && a.store_q) wave_copy(lane, GPTR(G_Qxx, k), (const T*)sQxx, n * n);   // (Qxx goes out now: P_k is built over it)
    wave_gemm_pair_sel<T, MF>(lane, n2, GemmOp<T>{1, 0, n, 1, sA, n2, st, n2, T(1), sQx, n},      // Qx += A^T Qx_tmp   (:149-150)
                              GemmOp<T>{1, 0, m, 1, sB, n2, st, n2, T(1), sQu, m});       // Qu += B^T Qx_tmp   (:151-152)
    __syncthreads();
    // gains (tvlqr.cpp:155-166)
    wave_copy(lane, sK, (const T*)sQux, m * n);
    for (int e = lane; e < m; e += 64) sd[e] = -sQu[e];
    {
      const float rmf = __builtin_amdgcn_rcpf((float)(m > 0 ? m : 1));
      for (int e = lane; e < m * m; e += 64) { int i, j; split_index(e, m, rmf, i, j); sL[e] = sQuu[e] + ((i == j) ? a.reg : T(0)); }
    }
    __syncthreads();
    // Quu + reg I = L L^T (fail when a pivot is <= 0: Eigen's llt_inplace) and solveInPlace, one right-hand side per lane
    // (columns of K, then d): in registers for m <= 4, spread over the lanes beyond
    const int failed = m <= 4 ? gain_solve_regs<T, 4>(lane, sL, sK, sd, n, m, a.store_q == 2) : gain_solve_lanes<T>(lane, sL, sK, sd, n, m);
    __syncthreads();
    if (failed) {  // tvlqr.cpp:162-164: return k, leaving K_k = Qux, d_k = -Qu unsolved
      wave_copy(lane, GPTR(G_K, k), (const T*)sK, m * n);
      wave_copy(lane, GPTR(G_d, k), (const T*)sd, m);
      if (a.store_q) {
        if (!LQ) wave_copy(lane, GPTR(G_Qxx, k), (const T*)sQxx, n * n);
        wave_copy(lane, GPTR(G_Quu, k), (const T*)sQuu, m * m);
        wave_copy(lane, GPTR(G_Qux, k), (const T*)sQux, m * n);
        wave_copy(lane, GPTR(G_Qx, k), (const T*)sQx, n);
        wave_copy(lane, GPTR(G_Qu, k), (const T*)sQu, m);
      }
      if (a.store_q == 2) {
        wave_copy(lane, GPTR(G_Qxx_tmp, k), (const T*)sT1, n * n);
        wave_copy(lane, GPTR(G_Quu_tmp, k), (const T*)sL, m * m);
        wave_copy(lane, GPTR(G_Qux_tmp, k), (const T*)sT2, m * n);
        wave_copy(lane, GPTR(G_Qx_tmp, k), (const T*)st, n);
      }
      if (lane == 0) {
        a.status[b] = k;
        a.delta_V[2 * (int64_t)b + 0] = dv0;
        a.delta_V[2 * (int64_t)b + 1] = dv1;
      }
      return;
    }
    // cost-to-go (tvlqr.cpp:173-186)
    wave_gemm_pair_sel<T, MF>(lane, m, GemmOp<T>{0, 0, m, n, sQuu, m, sK, m, T(0), sT2, m},       // Qux_tmp = Quu K
                              GemmOp<T>{0, 0, m, 1, sQuu, m, sd, m, T(0), sw, m});        // Qu_tmp = Quu d     (:189)
    wave_gemm_sel<T, MF>(lane, 1, 0, n, n, m, T(1), sK, m, sQux, m, T(0), sT1, n);  // Qxx_tmp = K^T Qux
    if (!LQ) wave_copy(lane, sP, (const T*)sQxx, n * n);   // P_{k+1}, p_{k+1} are dead from here: P_k, p_k are built in their place
    wave_copy(lane, sp, (const T*)sQx, n);
    if (a.store_q == 2) wave_gemm_sel<T, MF>(lane, 1, 0, n, 1, m, T(1), sK, m, sQu, m, T(0), st, n);  // Qx_tmp = K^T Qu (:176)
    __syncthreads();
    wave_gemm_sel<T, MF>(lane, 1, 0, n, n, m, T(1), sT2, m, sK, m, T(1), sP, n);    // P += (Quu K)^T K
    __syncthreads();
    {   // P -= K^T Qux; P -= (K^T Qux)^T: the same lane owns the element in both statements
      const float rn = __builtin_amdgcn_rcpf((float)n);
      for (int e = lane; e < n * n; e += 64) {
        int i, j;
        split_index(e, n, rn, i, j);
        T v = sP[e];
        v -= sT1[e];
        v -= sT1[j + i * n];
        sP[e] = v;
      }
    }
    if constexpr (MF && sizeof(T) == 8) {
      wave_gemm_sel<T, MF>(lane, 1, 0, n, 1, m, T(-1), sT2, m, sd, m, T(1), sp, n);   // p -= (Quu K)^T d
      __syncthreads();
      wave_gemm_sel<T, MF>(lane, 1, 0, n, 1, m, T(-1), sK, m, sQu, m, T(1), sp, n);   // p -= K^T Qu
      __syncthreads();
      wave_gemm_sel<T, MF>(lane, 1, 0, n, 1, m, T(1), sQux, m, sd, m, T(1), sp, n);   // p += Qux^T d
    } else {   // the three statements in one pass: a lane owns p_i in all of them, and forms its three sums side by side
      for (int i = lane; i < n; i += 64) {
        const T s1 = gen_dot<T>(sT2 + i * m, sd, m), s2 = gen_dot<T>(sK + i * m, sQu, m), s3 = gen_dot<T>(sQux + i * m, sd, m);
        T v = sp[i];
        v = T(1) * v + T(-1) * s1;                       // p -= (Quu K)^T d
        v = T(1) * v + T(-1) * s2;                       // p -= K^T Qu
        v = T(1) * v + T(1) * s3;                        // p += Qux^T d
        sp[i] = v;
      }
    }
    if (lane == 0) {  // tvlqr.cpp:189-191
      const T s0 = gen_dot<T>(sd, sQu, m), s1 = gen_dot<T>(sd, sw, m);
      dv0 += s0;
      dv1 += T(0.5) * s1;
    }
    __syncthreads();
    // write the knot point's results, roll P_k -> P_{k+1}
    wave_copy(lane, GPTR(G_K, k), (const T*)sK, m * n);
    wave_copy(lane, GPTR(G_d, k), (const T*)sd, m);
    wave_copy(lane, GPTR(G_P, k), (const T*)sP, n * n);
    wave_copy(lane, GPTR(G_p, k), (const T*)sp, n);
    if (a.store_q) {
      if (!LQ) wave_copy(lane, GPTR(G_Qxx, k), (const T*)sQxx, n * n);
      wave_copy(lane, GPTR(G_Quu, k), (const T*)sQuu, m * m);
      wave_copy(lane, GPTR(G_Qux, k), (const T*)sQux, m * n);
      wave_copy(lane, GPTR(G_Qx, k), (const T*)sQx, n);
      wave_copy(lane, GPTR(G_Qu, k), (const T*)sQu, m);
    }
    if (a.store_q == 2) {
      wave_copy(lane, GPTR(G_Qxx_tmp, k), (const T*)sT1, n * n);
      wave_copy(lane, GPTR(G_Quu_tmp, k), (const T*)sL, m * m);
      wave_copy(lane, GPTR(G_Qux_tmp, k), (const T*)sT2, m * n);
      wave_copy(lane, GPTR(G_Qx_tmp, k), (const T*)st, n);
      wave_copy(lane, GPTR(G_Qu_tmp, k), (const T*)sw, m);
    }
    __syncthreads();
  }
  if (lane == 0) {
    a.status[b] = -1;
    a.delta_V[2 * (int64_t)b + 0] = dv0;
    a.delta_V[2 * (int64_t)b + 1] = dv1;
  }
}

template <typename T>
inline size_t generic_backward_lds_bytes(int nm, int mm, bool late_q = false) {
  size_t el = (size_t)(late_q ? 3 : 4) * nm * nm + (size_t)4 * nm * mm + (size_t)2 * mm * mm + (size_t)5 * nm + (size_t)3 * mm;   // (the carve of generic_backward_kernel)
  return el * sizeof(T) + 64;
}

// sum_j M[i + j ld] v[j], j = 0 .. cnt-1, in index order; four matrix entries (global memory) and vector entries (LDS) fetched at a time
template <typename T>
__device__ __forceinline__ T gen_row_dot(const T* M, int ld, const T* v, int cnt) {
  T s = T(0);
  int j = 0;
  for (; j + 4 <= cnt; j += 4) {
    const T a0 = M[(j + 0) * ld], a1 = M[(j + 1) * ld], a2 = M[(j + 2) * ld], a3 = M[(j + 3) * ld];
    const T b0 = v[j], b1 = v[j + 1], b2 = v[j + 2], b3 = v[j + 3];
    s += a0 * b0;
    s += a1 * b1;
    s += a2 * b2;
    s += a3 * b3;
  }
  for (; j < cnt; ++j) s += M[j * ld] * v[j];
  return s;
}

inline size_t generic_forward_lds_bytes(int nm, int mm, int want_y, size_t esz) {   // STAGE = true
  size_t el = (size_t)3 * nm + mm + (size_t)nm * nm + (size_t)2 * nm * mm + nm + mm + (want_y ? (size_t)nm * nm + nm : 0);
  return el * esz + 64;
}

// tvlqr_ForwardPass (tvlqr.cpp:197-248): x_0 = x0; u = d - K x; x+ = f + A x + B u; y = P x + p.
// STAGE = true (what fits LDS): the knot point's K, d, A, B, f (P, p) are staged in LDS by coalesced loads and those of knot point
// k + 1 are fetched into registers while knot point k is worked on -- the rollout is one chain of dependent rows, and with the rows
// read from global memory term by term every knot point waited four to eight round trips ((13, 4), 4096 problems: forward sweep
// 1.53 ms; sums and their order unchanged).  STAGE = false: the rows straight from global memory (any size).
template <typename T, bool STAGE = false>
__global__ __launch_bounds__(64) void generic_forward_kernel(GenericArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  if (b >= a.batch) return;
  const int nm = a.nmax, mm = a.mmax;
  T* sx = smem;            // x_k
  T* su = sx + nm;         // u_k
  T* sxn = su + mm;        // x_{k+1}
  T* sA = sxn + nm;        // STAGE: A_k, B_k, f_k, K_k, d_k (, P_k, p_k)
  T* sB = sA + nm * nm;
  T* sf = sB + nm * mm;
  T* sK = sf + nm;
  T* sd = sK + mm * nm;
  T* sPm = sd + mm;
  T* spv = sPm + nm * nm;
#define GPTR(arr, k) (a.base[arr] + (int64_t)b * a.bstride[arr] + a.off[(int64_t)(k) * G_NUM + arr])
  const int N = a.N;
  struct { T A[4], B[2], f[1], K[2], d[1], P[4], p[1]; } kr;
  auto fetch_knot = [&](int k) {
    const int n = a.nx[k], m = a.nu[k], n2 = a.nx[k + 1];
    reg_fetch<T, 4>(kr.A, (const T*)GPTR(G_A, k), n2 * n, lane);
    reg_fetch<T, 2>(kr.B, (const T*)GPTR(G_B, k), n2 * m, lane);
    reg_fetch<T, 1>(kr.f, (const T*)GPTR(G_f, k), n2, lane);
    reg_fetch<T, 2>(kr.K, (const T*)GPTR(G_K, k), m * n, lane);
    reg_fetch<T, 1>(kr.d, (const T*)GPTR(G_d, k), m, lane);
    if (a.want_y) {
      reg_fetch<T, 4>(kr.P, (const T*)GPTR(G_P, k), n * n, lane);
      reg_fetch<T, 1>(kr.p, (const T*)GPTR(G_p, k), n, lane);
    }
  };
  if (STAGE && N > 0) fetch_knot(0);
  wave_copy(lane, sx, a.x0 + (int64_t)b * a.x0_stride, a.nx[0]);
  __syncthreads();
  wave_copy(lane, GPTR(G_x, 0), (const T*)sx, a.nx[0]);
  for (int k = 0; k < N; ++k) {
    const int n = a.nx[k], m = a.nu[k], n2 = a.nx[k + 1];
    // the knot point's rows against x, u (both call sites inline it: the LDS blocks keep ds_read, the global ones global_load)
    auto step = [&](const T* A, const T* B, const T* f, const T* K, const T* d, const T* P, const T* p) {
      for (int i = lane; i < m; i += 64) {  // u = d - K x
        const T s = gen_row_dot<T>(K + i, m, sx, n);
        su[i] = d[i] + T(-1) * s;
      }
      if (a.want_y) {  // y = P x + p   (lanes m.. take it so it overlaps the u chain)
        T* y = GPTR(G_y, k);
        for (int i = lane; i < n; i += 64) {
          const T s = gen_row_dot<T>(P + i, n, sx, n);
          y[i] = (T(0) + s) + p[i];
        }
      }
      __syncthreads();
      for (int i = lane; i < n2; i += 64) {  // x+ = f + A x + B u
        const T s = gen_row_dot<T>(A + i, n2, sx, n);
        T v = f[i] + s;
        const T s2 = gen_row_dot<T>(B + i, n2, su, m);
        sxn[i] = v + s2;
      }
    };
    if (STAGE) {
      reg_put<T, 4>(sA, kr.A, (const T*)GPTR(G_A, k), n2 * n, lane);
      reg_put<T, 2>(sB, kr.B, (const T*)GPTR(G_B, k), n2 * m, lane);
      reg_put<T, 1>(sf, kr.f, (const T*)GPTR(G_f, k), n2, lane);
      reg_put<T, 2>(sK, kr.K, (const T*)GPTR(G_K, k), m * n, lane);
      reg_put<T, 1>(sd, kr.d, (const T*)GPTR(G_d, k), m, lane);
      if (a.want_y) {
        reg_put<T, 4>(sPm, kr.P, (const T*)GPTR(G_P, k), n * n, lane);
        reg_put<T, 1>(spv, kr.p, (const T*)GPTR(G_p, k), n, lane);
      }
      if (k + 1 < N) fetch_knot(k + 1);                // in flight until the next iteration stores them
      __syncthreads();
      step(sA, sB, sf, sK, sd, sPm, spv);
    } else {
      step(GPTR(G_A, k), GPTR(G_B, k), GPTR(G_f, k), GPTR(G_K, k), GPTR(G_d, k), GPTR(G_P, k), GPTR(G_p, k));
    }
    wave_copy(lane, GPTR(G_u, k), (const T*)su, m);
    __syncthreads();
    wave_copy(lane, sx, (const T*)sxn, n2);
    wave_copy(lane, GPTR(G_x, k + 1), (const T*)sxn, n2);
    __syncthreads();
  }
  if (a.want_y) {
    const int n = a.nx[N];
    const T* P = GPTR(G_P, N);
    const T* p = GPTR(G_p, N);
    T* y = GPTR(G_y, N);
    for (int i = lane; i < n; i += 64) {
      const T s = gen_row_dot<T>(P + i, n, sx, n);
      y[i] = (T(0) + s) + p[i];
    }
  }
#undef GPTR
}

ALTRO_FP_REGION_END   // back to the including translation unit's own mode (fp_contract.h)

}  // namespace altro_hip
