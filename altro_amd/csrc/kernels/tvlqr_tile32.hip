// tvlqr_tile32.hip -- plan MFMA32: the fp64 TVLQR pair for 12 < n <= 31, m <= 8, n + m <= 32 (and the n <= 12 shapes with 4 < m <= 8)
// as 2 x 2 tiles of v_mfma_f64_16x16x4.
//
// Replaces tvlqr_BackwardPass / tvlqr_ForwardPass (src/tvlqr/tvlqr.cpp:65-195, :197-248) for the shapes one step past the (12, 4)
// tile -- a 13-state quaternion quadrotor, a 7-joint arm -- which plan GENERIC sweeps at 0.08-0.16 of the HBM roofline (a chain of
// separate LDS products per knot point).  Same formulation as kernels/tvlqr_mfma16.hip, on tiles:
//
//        D1 = [P' | t]^T Z                 rows 0..n-1 = P'^T Z, row n = t^T Z (the gradient part)     T1 x TC x KC instructions
//        G  = [Q H^T; H R] + Z^T D1        lower block triangle: Qxx, Qux, Quu (tvlqr.cpp:135-143)      NGT x KC
//        [P p; p^T .] = [Qxx Qx; Qx^T .] + Kt^T (Quu Kt - Qt) - Qt^T Kt   (tvlqr.cpp:173-186)          NPT x 2 MC
//
// with Z = [A B] (n x (n + m)), t = p' (+ P' f), Kt = [K | -d], Qt = [Qux | Qu] (both m x (n + 1)), KC = ceil(n / 4) terms-of-four,
// T1 = ceil((n + 1) / 16), TC = ceil((n + m) / 16), MC = ceil(m / 4).  The accumulator layout of the f64 instruction (lane l, register
// r <-> row (l >> 4) + 4 r, column l & 15) is the B-operand layout of terms 4 r .. 4 r + 3, so D1 feeds the second product from
// registers.  One wavefront owns one problem and walks the horizon; a knot point's blocks are fetched a knot point ahead by linear
// 8-byte-per-lane buffer loads (lanes past a block's end are out of range: they read zero and their stores are dropped -- no exec
// masks, no branches around vector memory), laid into LDS as the operands' images (Z and Q column-major at a leading dimension of
// 2 x odd: a tile column read hits 32 different bank pairs), and the Cholesky factor of Quu + reg I (tvlqr.cpp:159-164, m <= 8, unit
// padding) lives in every lane's registers, each lane solving the columns of Kt it feeds to the matrix cores.
//
// The arrays are plan GENERIC's (reference layout on the device, kernels/generic_arrays.h): every kernel of that plan's iLQR loop,
// every setter and getter works on a handle of this plan unchanged.  The carried cost-to-go is the stored one: after every step
// the lower triangle of [P p; p^T .] goes through an LDS tile and is read back mirrored.
//
// Parity: sums in the matrix pipe's order and reciprocal-square-root pivots -- K, d <= 1e-8 absolute against the oracle (measured
// ~1e-13), like plan MFMA16; plan GENERIC stays the bit-for-bit plan.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfma16_layout.h"

namespace altro_hip {

typedef double t32_f64x4 __attribute__((ext_vector_type(4)));
typedef unsigned int t32_v2u __attribute__((ext_vector_type(2)));

constexpr int T32_MAX_N = 31, T32_MAX_M = 8;
constexpr unsigned T32_OOB = 0x80000000u;   // a buffer offset past every window: loads return 0, stores are dropped
constexpr int T32_SLD = 33;                 // row pitch of the [Qux Quu] exchange tile

// where the pieces of a knot point live in LDS (element = double), computed on the host (tile32_lds_layout)
struct Tile32Lds {
  int ldz;    // leading dimension of the Z image: 2 x odd, >= 4 ceil(n / 4)
  int ldc;    // ... of the C image: 2 x odd, >= 4 ceil((n + m) / 4)
  int ldp;    // row pitch of the [P p; p^T .] exchange tile: odd, >= n + 1
  int z;      // Z = [A B] column-major, n + m columns; rows n .. ldz - 1 stay zero (they are terms of the products)
  int s;      // [Qux Quu] rows (MP x T32_SLD)
  int c;      // C = [Q .; H R] column-major, n + m columns (no entry past its blocks is ever used), later the exchange tile ((n + 1) x ldp)
  int f;      // f padded to 32 entries, f[n] = 1 (the row of D1 that carries t^T Z rides the same sum)
  int qr;     // [q r] padded to 32
  int gv;     // [Qx Qu]
  int cst;    // 0.0, 1.0
  int total;
};
constexpr __host__ __device__ Tile32Lds tile32_lds_layout(int n, int m) {
  const int kc = (n + 3) / 4, mp = 4 * ((m + 3) / 4), nz = n + m, kz = (nz + 3) / 4;
  Tile32Lds L{};
  L.ldz = 2 * ((2 * kc) | 1);   // 2 x odd >= 4 kc
  L.ldc = 2 * ((2 * kz) | 1);
  L.ldp = (n + 1) | 1;
  int at = 0;
  L.z = at;   at += L.ldz * nz;
  L.s = at;   at += mp * T32_SLD;
  const int cin = L.ldc * nz + 32, pex = (n + 4) * L.ldp;   // (+32 / +3 rows: tile reads past the blocks stay inside the region)
  const int park = m > 4 ? (n >= 16 ? 768 : 256) : 0;      // m > 4: G's tiles wait here while the 8 x 8 factor has the registers
  L.c = at;   at += (cin > pex ? cin : pex) > park ? (cin > pex ? cin : pex) : park;
  L.f = at;   at += 32;
  L.qr = at;  at += 32;
  L.gv = at;  at += 40;   // (column n of Qt reads gv[n .. n + MP - 1]: past n + m they are zeros)
  L.cst = at; at += 2;
  L.total = (at + 1) & ~1;
  return L;
}

struct Tile32Args {
  const double *A, *B, *f, *Q, *R, *H, *q, *r;   // [b][k][block], column-major blocks (plan GENERIC's arrays)
  double *K, *d, *P, *p;
  int64_t bsA, bsB, bsf, bsQ, bsR, bsH, bsq, bsr, bsK, bsd, bsP, bsp;   // elements between problems
  double *x, *u, *y;                             // forward sweep
  int64_t bsx, bsu, bsy;
  const double* x0;                              // [b][n]
  double* delta_V;                               // [b][2]
  int* status;                                   // [b]
  int N, batch, n, m;
  double reg;
  int no_f;                                      // the iLQR loop's expansion: the affine term is not part of it (knotpoint_data.cpp:416)
  const int* active;                             // optional per-problem mask (backward only)
  const double* reg_pp;                          // optional per-problem regularisation
  Tile32Lds L;
};

__device__ __forceinline__ t32_f64x4 t32_mfma(double a, double b, t32_f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// 1 / sqrt(x): v_rsq_f64 seed + one third-order step (kernels/tvlqr_mfma16.hip)
__device__ __forceinline__ double t32_rsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-x, y * y, 1.0);
  const double t = __builtin_fma(0.375, e, 0.5);
  return __builtin_fma(y * e, t, y);
}
__device__ __forceinline__ double t32_group4_allreduce(double v) {   // over lanes l, l ^ 16, l ^ 32, l ^ 48
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// v_i for a run-time i = 0..3 (by value: a pointer into a register array would send the array to scratch memory)
__device__ __forceinline__ double t32_pick4(double v0, double v1, double v2, double v3, int i) {
  const double lo = (i & 1) ? v1 : v0, hi = (i & 1) ? v3 : v2;
  return (i & 2) ? hi : lo;
}

// a 2 GiB raw-buffer window on one problem's stretch of an array; the knot point rides the scalar offset
struct T32Buf {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ explicit T32Buf(const void* base) {
    r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
  }
  __device__ __forceinline__ double ld(unsigned voff, unsigned soff) const {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
  }
  __device__ __forceinline__ void st(unsigned voff, unsigned soff, double v) const {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(t32_v2u, v), r, voff, soff, 0);
  }
};

// ------------------------------------------------------------------------------------------------------------------------
// Backward sweep.  KC, T1, TC, MC as above; WPS = waves per SIMD the register budget is set for.
//
// Loads.  A block with columns of length n moves as whole columns: a load takes cpi = floor(64 / n) of them (lane = entry, so the
// addresses of a load are one linear run), and lane (row, column-in-group) writes LDS at the image's leading dimension: both the
// global and the LDS address of load i are the lane's own offset plus i times a wave-uniform step -- two registers instead of one
// per load.  Everything else a lane needs to know where its operands sit is recomputed every knot point from its (j, g) -- a few
// dozen integer operations beside 22-60 matrix-core instructions -- because hoisted out of the loop (which the compiler does with
// anything loop-invariant) those addresses alone overflow the register file of the larger instantiations.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int t32_count_cols(int len, int cols) { return (cols + 64 / len - 1) / (64 / len); }   // loads that cover `cols` columns of length `len`

struct T32Lanes { int j, g; };
__device__ __forceinline__ T32Lanes t32_launder(int j, int g) {   // the same values, opaque to loop-invariant code motion
  asm volatile("" : "+v"(j), "+v"(g));
  return T32Lanes{j, g};
}

// NS, MS: n and m, compile-time constants -- every address, mask and stride below folds, and a knot point's integer work shrinks from
// ~500 vector instructions (the first, run-time-shape form of this kernel: issue-bound at 0.34 of the HBM roofline) to under 100.
// One instantiation per shape (capi_tile32_*.hip).  LAUNDER: recompute the lane-dependent addresses every knot point instead of
// keeping them in registers (the m > 4 shapes, whose 8 x 8 factor needs the registers).
template <int NS, int MS, int WPS, bool LAUNDER>
__global__ __launch_bounds__(64, WPS) void tile32_backward_kernel(Tile32Args a) {
  constexpr int KC = (NS + 3) / 4, T1 = (NS + 16) / 16, TC = (NS + MS + 15) / 16, MC = (MS + 3) / 4;
  static_assert(NS >= 1 && NS <= T32_MAX_N && MS >= 1 && MS <= T32_MAX_M && NS + MS <= 32, "shape");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int MP = 4 * MC;
  constexpr int NA = t32_count_cols(NS, NS);       // loads per n x n block
  constexpr int NB = t32_count_cols(NS, MS);       // ... per n x m block (columns of length n)
  constexpr int NH = t32_count_cols(MS, NS);       // ... per m x n block (columns of length m)
  constexpr int NGT = TC == 1 ? 1 : 3;             // lower block triangle: (0,0), (1,0), (1,1)
  constexpr int NPT = T1 == 1 ? 1 : 3;
  constexpr int TI[3] = {0, 1, 1}, TJ[3] = {0, 0, 1};

  const int lane = threadIdx.x;
  const int b = mf_problem(blockIdx.x, a.batch);
  if (b >= a.batch) return;
  if (a.active && !a.active[b]) return;   // wave-uniform: the problem has stopped, its outputs stay
  constexpr int n = NS, m = MS, nz = n + m, nn = n * n, nm = n * m;
  const int N = a.N;
  constexpr Tile32Lds L = tile32_lds_layout(NS, MS);
  constexpr int ZERO = L.cst;

  for (int e = lane; e < L.total; e += 64) lds[e] = 0.0;
  __syncthreads();
  if (lane == 0) { lds[L.cst + 1] = 1.0; lds[L.f + n] = 1.0; }

  const T32Buf bA(a.A + (int64_t)b * a.bsA), bB(a.B + (int64_t)b * a.bsB), bF(a.f + (int64_t)b * a.bsf),
      bQ(a.Q + (int64_t)b * a.bsQ), bR(a.R + (int64_t)b * a.bsR), bH(a.H + (int64_t)b * a.bsH), bq(a.q + (int64_t)b * a.bsq),
      br(a.r + (int64_t)b * a.bsr), bK(a.K + (int64_t)b * a.bsK), bd(a.d + (int64_t)b * a.bsd), bP(a.P + (int64_t)b * a.bsP),
      bp(a.p + (int64_t)b * a.bsp);
  constexpr unsigned sA = (unsigned)nn * 8u, sB = (unsigned)nm * 8u, sv = (unsigned)n * 8u, sR = (unsigned)(m * m) * 8u, sr = (unsigned)m * 8u;

  // column groups: cpi columns of length n per load (A, Q, B), cph columns of length m (H)
  constexpr int cpi = 64 / n, cph = 64 / m;
  int lrow = lane % n, lcol = lane / n;            // this lane's entry of a group of columns of length n
  int hrow = lane % m, hcol = lane / m;            // ... of length m
  constexpr int tn = n >> 4;   // the tile column that holds column n

  // ---- terminal cost-to-go: P_N = Q_N, p_N = q_N (tvlqr.cpp:81-90) -----------------------------------------------------------
  double Pt[KC][T1];
  {
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int t = 0; t < T1; ++t) {
        const int row = 4 * c + g, col = 16 * t + j;
        const unsigned oq = (row < n && col < n) ? (unsigned)(row + col * n) * 8u : T32_OOB;
        const unsigned ov = (row < n && col == n) ? (unsigned)row * 8u : T32_OOB;
        Pt[c][t] = bQ.ld(oq, (unsigned)N * sA) + bq.ld(ov, (unsigned)N * sv);
        bP.st(oq, (unsigned)N * sA, Pt[c][t]);
        bp.st(ov, (unsigned)N * sv, Pt[c][t]);
      }
  }

  // The next knot point is fetched in two halves, each while nothing else needs its registers: A | B at the top of a step (they
  // become the Z image at its end), Q | H | R | f | q | r after the step's factorisation and solves (they become the C image after
  // the NEXT step's first product) -- the solves are where the registers are scarce.
  struct KnotZ { double a[NA], b[NB]; };
  struct KnotC { double q[NA], h[NH], r, s; };
  auto fetch_z = [&](KnotZ& kn, int k) {
    if (LAUNDER) asm volatile("" : "+v"(lrow), "+v"(lcol));   // (no hoisting of the masks below: see the head comment)
    const bool lin = lcol < cpi;
    const unsigned uk = (unsigned)k, cstep = (unsigned)(cpi * n) * 8u, lv = (unsigned)(lcol * n + lrow) * 8u;
#pragma unroll
    for (int i = 0; i < NA; ++i) kn.a[i] = bA.ld((lin && i * cpi + lcol < n) ? lv : T32_OOB, uk * sA + (unsigned)i * cstep);
#pragma unroll
    for (int i = 0; i < NB; ++i) kn.b[i] = bB.ld((lin && i * cpi + lcol < m) ? lv : T32_OOB, uk * sB + (unsigned)i * cstep);
  };
  auto fetch_c = [&](KnotC& kn, int k) {
    if (LAUNDER) asm volatile("" : "+v"(lrow), "+v"(lcol), "+v"(hrow), "+v"(hcol));
    const int lane = lcol * n + lrow;
    const bool lin = lcol < cpi, hin = hcol < cph;
    const unsigned uk = (unsigned)k, cstep = (unsigned)(cpi * n) * 8u, hstep = (unsigned)(cph * m) * 8u, lv = (unsigned)lane * 8u;
#pragma unroll
    for (int i = 0; i < NA; ++i) kn.q[i] = bQ.ld((lin && i * cpi + lcol < n) ? lv : T32_OOB, uk * sA + (unsigned)i * cstep);
#pragma unroll
    for (int i = 0; i < NH; ++i) kn.h[i] = bH.ld((hin && i * cph + hcol < n) ? lv : T32_OOB, uk * sB + (unsigned)i * hstep);
    kn.r = bR.ld(lane < m * m ? lv : T32_OOB, uk * sR);
    const unsigned of = (lane < n && !a.no_f) ? lv : T32_OOB;
    const unsigned oq = (lane >= n && lane < 2 * n) ? (unsigned)(lane - n) * 8u : T32_OOB;
    const unsigned orr = (lane >= 2 * n && lane < 2 * n + m) ? (unsigned)(lane - 2 * n) * 8u : T32_OOB;
    kn.s = bF.ld(of, uk * sv) + bq.ld(oq, uk * sv) + br.ld(orr, uk * sr);
  };
  // the images: Z = [A B] at leading dimension ldz; C = [Q .; H R] at ldc (rows 0..n-1 Q, n..n+m-1 H | R)
  auto stage_z = [&](const KnotZ& kn) {
    if (LAUNDER) asm volatile("" : "+v"(lrow), "+v"(lcol));
    const bool lin = lcol < cpi;
    const int zl = L.z + lcol * L.ldz + lrow;
#pragma unroll
    for (int i = 0; i < NA; ++i) if (lin && i * cpi + lcol < n) lds[zl + i * cpi * L.ldz] = kn.a[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) if (lin && i * cpi + lcol < m) lds[zl + (n + i * cpi) * L.ldz] = kn.b[i];
  };
  auto stage_c = [&](const KnotC& kn) {
    if (LAUNDER) asm volatile("" : "+v"(lrow), "+v"(lcol), "+v"(hrow), "+v"(hcol));
    const int lane = lcol * n + lrow;
    const bool lin = lcol < cpi, hin = hcol < cph;
    const int cl = L.c + lcol * L.ldc + lrow;
#pragma unroll
    for (int i = 0; i < NA; ++i) if (lin && i * cpi + lcol < n) lds[cl + i * cpi * L.ldc] = kn.q[i];
#pragma unroll
    for (int i = 0; i < NH; ++i) if (hin && i * cph + hcol < n) lds[L.c + (i * cph + hcol) * L.ldc + n + hrow] = kn.h[i];
    if (lane < m * m) lds[L.c + (n + hcol) * L.ldc + n + hrow] = kn.r;
    const int sml = lane < n ? (a.no_f ? -1 : L.f + lane) : lane < 2 * n ? L.qr + (lane - n) : lane < 2 * n + m ? L.qr + lane - n : -1;
    if (sml >= 0) lds[sml] = kn.s;
  };

  double dv0 = 0.0, dv1 = 0.0;
  int fail_k = -1;
  const double reg = a.reg_pp ? a.reg_pp[b] : a.reg;

  KnotZ nz_;
  KnotC nc_;
  fetch_c(nc_, N - 1);
  fetch_z(nz_, N - 1);
  __syncthreads();
  stage_z(nz_);
  __syncthreads();

  for (int k = N - 1; k >= 0; --k) {
    fetch_z(nz_, k > 0 ? k - 1 : 0);   // the next knot point while this one computes (k == 0 re-reads block 0: harmless)
    const T32Lanes ln = LAUNDER ? t32_launder(lane & 15, lane >> 4) : T32Lanes{lane & 15, lane >> 4};
    const int j = ln.j, g = ln.g;

    // ---- D1 = [P' | t]^T Z -------------------------------------------------------------------------------------------------
    // Z[4 c + g][16 t + j]; a column past n + m - 1 reads the last one (its products land in entries nobody uses)
    double z[KC][TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) {
      const int col = 16 * t + j;
      const int zb = L.z + (col < nz ? col : nz - 1) * L.ldz + g;
#pragma unroll
      for (int c = 0; c < KC; ++c) z[c][t] = lds[zb + 4 * c];
    }
    t32_f64x4 D1[T1][TC];
#pragma unroll
    for (int tr = 0; tr < T1; ++tr)
#pragma unroll
      for (int tc = 0; tc < TC; ++tc) {
        t32_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < KC; ++c) acc = t32_mfma(Pt[c][tr], z[c][tc], acc);
        D1[tr][tc] = acc;
      }
    // ---- this knot point's cost blocks (requested in the middle of the previous step) become the C image ----------------------
    __syncthreads();   // (the exchange tile that overlays it was read back a step ago; the barrier orders those reads before these writes)
    stage_c(nc_);
    __syncthreads();
    // ---- G = [Q H^T; H R] + Z^T D1 (terms 4 c .. 4 c + 3 = register c & 3 of row tile c >> 2; rows past n - 1 of Z are zero) ------
    // the entries right of Quu and below it start from whatever the image holds there: nobody uses them
    t32_f64x4 G[NGT];
#pragma unroll
    for (int t = 0; t < NGT; ++t) {
      const int col = 16 * TJ[t] + j;
      const int cb = L.c + (col < nz ? col : nz - 1) * L.ldc + g + 16 * TI[t];
#pragma unroll
      for (int r = 0; r < 4; ++r) G[t][r] = (16 * TI[t] + 4 * r < nz) ? lds[cb + 4 * r] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < NGT; ++t)
#pragma unroll
      for (int c = 0; c < KC; ++c) G[t] = t32_mfma(z[c][TI[t]], D1[c >> 2][TJ[t]][c & 3], G[t]);
    // ---- gradient [Qx; Qu] = [q; r] + Z^T t + Z^T P' f: the sum over the rows of D1 against f, whose entry n is 1 ------------
    double gv[TC];
#pragma unroll
    for (int tc = 0; tc < TC; ++tc) {
      double s = 0.0;
#pragma unroll
      for (int tr = 0; tr < T1; ++tr)
#pragma unroll
        for (int r = 0; r < 4; ++r) s = __builtin_fma(lds[L.f + 16 * tr + 4 * r + g], D1[tr][tc][r], s);
      gv[tc] = lds[L.qr + 16 * tc + j] + t32_group4_allreduce(s);
    }

    // ---- exchange: the rows n .. n + m - 1 of G (= [Qux Quu]) and the gradient ------------------------------------------------
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NGT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row0 = 16 * TI[t] + 4 * r;               // rows row0 .. row0 + 3 (wave-uniform test first)
        if (row0 + 3 >= n && row0 < nz) {
          const int ar = row0 + g - n;
          if (ar >= 0 && ar < m) lds[L.s + ar * T32_SLD + 16 * TJ[t] + j] = G[t][r];
        }
      }
    if (g == 0) {
#pragma unroll
      for (int tc = 0; tc < TC; ++tc) if (16 * tc + j < nz) lds[L.gv + 16 * tc + j] = gv[tc];
    }
    __syncthreads();

    // (m > 4: the 8 x 8 factor below needs G's registers -- its entries wait in LDS, accumulator layout, where the C image was)
    constexpr bool PARK = MC == 2;
    if (PARK) {
#pragma unroll
      for (int t = 0; t < NPT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[L.c + (4 * t + r) * 64 + lane] = G[t][r];
    }

    // ---- Cholesky of Quu + reg I (lower; a pivot <= 0 fails: tvlqr.cpp:159-164); reciprocal pivots only; unit diagonal past m -----
    double qa[MP][MP], inv[MP];
#pragma unroll
    for (int aa = 0; aa < MP; ++aa)
#pragma unroll
      for (int bb = 0; bb <= aa; ++bb) {
        const double v = lds[(aa < m && bb < m) ? L.s + aa * T32_SLD + n + bb : ZERO];
        qa[aa][bb] = (aa == bb && aa >= m) ? 1.0 : v;
      }
    bool fail = false;
#pragma unroll
    for (int kk = 0; kk < MP; ++kk) {
      double x = qa[kk][kk] + ((kk < m) ? reg : 0.0);
#pragma unroll
      for (int pp = 0; pp < kk; ++pp) x = __builtin_fma(-qa[kk][pp], qa[kk][pp], x);
      fail = fail || !(x > 0.0);
      inv[kk] = t32_rsqrt(x);
#pragma unroll
      for (int ii = kk + 1; ii < MP; ++ii) {
        double s = qa[ii][kk];
#pragma unroll
        for (int pp = 0; pp < kk; ++pp) s = __builtin_fma(-qa[ii][pp], qa[kk][pp], s);
        qa[ii][kk] = s * inv[kk];
      }
    }
    const bool was_alive = (fail_k < 0);
    if (was_alive && fail) fail_k = k;
    const bool alive = (fail_k < 0);

    // ---- Kt[:, col] = (L L^T)^-1 Qt[:, col] for this lane's columns (tvlqr.cpp:165-166), W = Quu Kt - Qt ---------------------
    double km[MC][T1], qm[MC][T1], wm[MC][T1], qk[MC][T1];
#pragma unroll
    for (int t = 0; t < T1; ++t) {
      const int col = 16 * t + j;
      // column col of Qt = [Qux | Qu]: rows of the exchange tile, the gradient for column n, zero past it
      const int rbase = col < n ? L.s + col : (col == n ? L.gv + n : ZERO), rstep = col < n ? T32_SLD : (col == n ? 1 : 0);
      double rhs[MP], y[MP], kt[MP];
#pragma unroll
      for (int aa = 0; aa < MP; ++aa) rhs[aa] = lds[rbase + aa * rstep];
#pragma unroll
      for (int c = 0; c < MC; ++c) qm[c][t] = t32_pick4(rhs[4 * c], rhs[4 * c + 1], rhs[4 * c + 2], rhs[4 * c + 3], g);
#pragma unroll
      for (int aa = 0; aa < MP; ++aa) {
        double s = rhs[aa];
#pragma unroll
        for (int pp = 0; pp < aa; ++pp) s = __builtin_fma(-qa[aa][pp], y[pp], s);
        y[aa] = s * inv[aa];
      }
#pragma unroll
      for (int aa = MP - 1; aa >= 0; --aa) {
        double s = y[aa];
#pragma unroll
        for (int pp = aa + 1; pp < MP; ++pp) s = __builtin_fma(-qa[pp][aa], kt[pp], s);
        kt[aa] = s * inv[aa];
      }
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        km[c][t] = t32_pick4(kt[4 * c], kt[4 * c + 1], kt[4 * c + 2], kt[4 * c + 3], g);
        // row 4 c + g of the unregularised Quu (tvlqr.cpp:174), the entries right of the diagonal from their mirror images
        const int ar = 4 * c + g;
        double s = 0.0;
#pragma unroll
        for (int bb = 0; bb < MP; ++bb) {
          const int hi = ar > bb ? ar : bb, lo = ar > bb ? bb : ar;
          const double quu = lds[(ar < m && bb < m) ? L.s + hi * T32_SLD + n + lo : ZERO];
          s = __builtin_fma(quu, kt[bb], s);
        }
        qk[c][t] = s;
        wm[c][t] = s - qm[c][t];
      }
    }
    // ---- expected decrease (tvlqr.cpp:189-191): in column n Kt = -d, Qt = Qu ------------------------------------------------
    {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        const double kv = (T1 == 2 && tn == 1) ? km[c][T1 - 1] : km[c][0];
        const double qv = (T1 == 2 && tn == 1) ? qm[c][T1 - 1] : qm[c][0];
        const double qkv = (T1 == 2 && tn == 1) ? qk[c][T1 - 1] : qk[c][0];
        s0 = __builtin_fma(-kv, qv, s0);
        s1 = __builtin_fma(0.5 * kv, qkv, s1);
      }
      dv0 = alive ? dv0 + s0 : dv0;
      dv1 = alive ? dv1 + s1 : dv1;
    }

    fetch_c(nc_, k > 0 ? k - 1 : 0);   // the next knot point's cost blocks, now that the solves' registers are free
    // what the registers of the new [P p; p^T .] start from: G's own entry, but Qx in column n and in row n
    t32_f64x4 Pn[NPT];
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
      const int col = 16 * TJ[t] + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * TI[t] + 4 * r + g;
        double v = PARK ? lds[L.c + (4 * t + r) * 64 + lane] : G[t][r];
        if (TJ[t] == T1 - 1 || T1 == 1) {   // column n lives in the last tile column
          const double qx = lds[L.gv + (row < n ? row : n)];
          v = (col == n) ? qx : v;
        }
        if (TI[t] == T1 - 1 || T1 == 1) {   // row n in the last tile row
          const double qx = lds[L.gv + (col < n ? col : n)];
          v = (row == n) ? qx : v;
        }
        Pn[t][r] = v;
      }
    }

    // ---- [P p; p^T .] += Kt^T W - Qt^T Kt ---------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < NPT; ++t)
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        Pn[t] = t32_mfma(km[c][TI[t]], wm[c][TJ[t]], Pn[t]);
        Pn[t] = t32_mfma(qm[c][TI[t]], -km[c][TJ[t]], Pn[t]);
      }
    // ---- the carried cost-to-go is the stored one: lower triangle out, mirrored back in ------------------------------------------
    __syncthreads();   // (every read of the C image this tile overlays is done)
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
      const int col = 16 * TJ[t] + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * TI[t] + 4 * r + g;
        if (row <= n && col <= row) lds[L.c + row * L.ldp + col] = Pn[t][r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T1; ++t) {
      const int col = 16 * t + j;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const int row = 4 * c + g;
        const int hi = row > col ? row : col, lo = row > col ? col : row;
        const double v = lds[L.c + (hi <= n ? hi * L.ldp + lo : 0)];
        Pt[c][t] = (row < n && col <= n) ? v : 0.0;
      }
    }
    const double pv = lds[L.c + n * L.ldp + (lane < n ? lane : 0)];

    // ---- results: K_k, d_k (a failed factorisation leaves Qux, -Qu there: tvlqr.cpp:165 copies before it solves), P_k, p_k ------
    {
      const unsigned uk = (unsigned)k;
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        const int ar = 4 * c + g;
#pragma unroll
        for (int t = 0; t < T1; ++t) {
          const int col = 16 * t + j;
          const unsigned o = (was_alive && ar < m && col < n) ? (unsigned)(ar + col * m) * 8u : T32_OOB;
          bK.st(o, uk * sB, alive ? km[c][t] : qm[c][t]);
        }
        const double dk = (T1 == 2 && tn == 1) ? km[c][T1 - 1] : km[c][0], dq = (T1 == 2 && tn == 1) ? qm[c][T1 - 1] : qm[c][0];
        bd.st((was_alive && ar < m && j == (n & 15)) ? (unsigned)ar * 8u : T32_OOB, uk * sr, alive ? -dk : -dq);
      }
      // P_k[col + row n] = M[row][col]: the transposed position holds the same value and makes the lanes' addresses a unit-stride run
#pragma unroll
      for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int t = 0; t < T1; ++t) {
          const int row = 4 * c + g, col = 16 * t + j;
          bP.st((alive && row < n && col < n) ? (unsigned)(row * n + col) * 8u : T32_OOB, uk * sA, Pt[c][t]);
        }
      bp.st((alive && lane < n) ? (unsigned)lane * 8u : T32_OOB, uk * sv, pv);
    }

    // ---- the next knot point's images ------------------------------------------------------------------------------------------
    __syncthreads();
    stage_z(nz_);
    __syncthreads();
  }
  {
    const double t0 = t32_group4_allreduce(dv0), t1 = t32_group4_allreduce(dv1);
    if ((lane & 15) == (n & 15) && (lane >> 4) == 0) {
      a.status[b] = fail_k;   // -1 = success, else the failing knot point (tvlqr.cpp:163)
      a.delta_V[2 * (size_t)b + 0] = t0;
      a.delta_V[2 * (size_t)b + 1] = t1;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Forward sweep: x_0 = x0; u = d - K x; x+ = f + A x + B u; y = P x + p (tvlqr.cpp:208-246).  One lane per row: lanes 0..n-1 own
// the rows of [A B | f], lanes n..n+m-1 the rows of [K | d], lanes n+m..2n+m-1 the rows of [P | p] (2 n + m <= 64 whenever
// n + m <= 32); the blocks are column-major, so the lanes of a role read consecutive LDS addresses, and x, u are broadcast reads.
// The blocks of knot point k + 1 are fetched (linear buffer loads) while knot point k is worked on.
// NP = 4 ceil(n / 4), MP = 4 ceil(m / 4): the unrolled trip counts; entries past n, m are zero in LDS.
// ------------------------------------------------------------------------------------------------------------------------
struct Tile32FwdLds { int A, B, f, K, d, P, p, x, u, total; };
inline __host__ __device__ Tile32FwdLds tile32_fwd_lds_layout(int n, int m) {
  const int np = 4 * ((n + 3) / 4), mp = 4 * ((m + 3) / 4);
  Tile32FwdLds L;
  int at = 0;
  L.A = at; at += n * np;       // column-major, leading dimension n, np columns (the padding columns stay zero)
  L.B = at; at += n * mp;
  L.f = at; at += n;
  L.K = at; at += m * np;
  L.d = at; at += m;
  L.P = at; at += n * np;
  L.p = at; at += n;
  L.x = at; at += np;
  L.u = at; at += mp;
  L.total = (at + 1) & ~1;
  return L;
}

template <int KC, int MC, int WPS>
__global__ __launch_bounds__(64, WPS) void tile32_forward_kernel(Tile32Args a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int NP = 4 * KC, MP = 4 * MC;
  constexpr int NA = (16 * KC * KC + 63) / 64, NB = (16 * KC * MC + 63) / 64;
  const int lane = threadIdx.x;
  const int b = mf_problem(blockIdx.x, a.batch);
  if (b >= a.batch) return;
  const int n = a.n, m = a.m, N = a.N, nn = n * n, nm = n * m;
  const Tile32FwdLds L = tile32_fwd_lds_layout(n, m);
  for (int e = lane; e < L.total; e += 64) lds[e] = 0.0;
  __syncthreads();

  const T32Buf bA(a.A + (int64_t)b * a.bsA), bB(a.B + (int64_t)b * a.bsB), bF(a.f + (int64_t)b * a.bsf), bK(a.K + (int64_t)b * a.bsK),
      bd(a.d + (int64_t)b * a.bsd), bP(a.P + (int64_t)b * a.bsP), bp(a.p + (int64_t)b * a.bsp), bx(a.x + (int64_t)b * a.bsx),
      bu(a.u + (int64_t)b * a.bsu), by(a.y + (int64_t)b * a.bsy);
  const unsigned sA = (unsigned)nn * 8u, sB = (unsigned)nm * 8u, sv = (unsigned)n * 8u, sr = (unsigned)m * 8u;

  // roles
  const bool is_x = lane < n, is_u = lane >= n && lane < n + m, is_y = lane >= n + m && lane < 2 * n + m;
  const int row = is_x ? lane : is_u ? lane - n : is_y ? lane - n - m : 0;
  const int ld = is_u ? m : n;                                  // leading dimension of this lane's block
  const int mat = (is_x ? L.A : is_u ? L.K : L.P) + row;        // its row: lds[mat + j * ld]
  const int aff = (is_x ? L.f : is_u ? L.d : L.p) + row;        // the affine entry
  const int brow = L.B + (is_x ? row : 0);                      // row of B (x lanes)
  // linear loads and where they land
  unsigned ldA[NA], ldB[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) ldA[i] = 64 * i + lane < nn ? (unsigned)(64 * i + lane) * 8u : T32_OOB;
#pragma unroll
  for (int i = 0; i < NB; ++i) ldB[i] = 64 * i + lane < nm ? (unsigned)(64 * i + lane) * 8u : T32_OOB;
  // f | d | p in the lanes of one register: lanes 0..n-1 f, n..n+m-1 d, n+m..2n+m-1 p -- the role's own affine entry
  const unsigned ldf = is_x ? (unsigned)row * 8u : T32_OOB, ldd = is_u ? (unsigned)row * 8u : T32_OOB, ldp = is_y ? (unsigned)row * 8u : T32_OOB;
  // outputs: x_k | u_k | y_k from the role's lane
  const unsigned stx = is_x ? (unsigned)row * 8u : T32_OOB, stu = is_u ? (unsigned)row * 8u : T32_OOB, sty = is_y ? (unsigned)row * 8u : T32_OOB;

  struct Knot { double a[NA], b[NB], k[NB], p[NA], s; };
  auto fetch = [&](Knot& kn, int k) {
    const unsigned uk = (unsigned)k;
#pragma unroll
    for (int i = 0; i < NA; ++i) kn.a[i] = bA.ld(ldA[i], uk * sA);
#pragma unroll
    for (int i = 0; i < NB; ++i) kn.b[i] = bB.ld(ldB[i], uk * sB);
#pragma unroll
    for (int i = 0; i < NB; ++i) kn.k[i] = bK.ld(ldB[i], uk * sB);
#pragma unroll
    for (int i = 0; i < NA; ++i) kn.p[i] = bP.ld(ldA[i], uk * sA);
    kn.s = bF.ld(ldf, uk * sv) + bd.ld(ldd, uk * sr) + bp.ld(ldp, uk * sv);
  };
  auto stage = [&](const Knot& kn) {
#pragma unroll
    for (int i = 0; i < NA; ++i) if (64 * i + lane < nn) { lds[L.A + 64 * i + lane] = kn.a[i]; lds[L.P + 64 * i + lane] = kn.p[i]; }
#pragma unroll
    for (int i = 0; i < NB; ++i) if (64 * i + lane < nm) { lds[L.B + 64 * i + lane] = kn.b[i]; lds[L.K + 64 * i + lane] = kn.k[i]; }
    if (lane < 2 * n + m) lds[aff] = kn.s;
  };

  double xcur = is_x ? a.x0[(size_t)b * n + row] : 0.0;
  Knot nxt;
  fetch(nxt, 0);
  if (is_x) lds[L.x + row] = xcur;
  __syncthreads();
  stage(nxt);
  __syncthreads();

  for (int k = 0; k < N; ++k) {
    fetch(nxt, k + 1 < N ? k + 1 : N - 1);
    // phase 1: this lane's row against x (A x | K x | P x)
    double acc = 0.0;
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) acc = __builtin_fma(lds[mat + jj * ld], lds[L.x + jj], acc);
    const double af = lds[aff];
    const double uval = af - acc;          // u = d - K x
    const double yval = acc + af;          // y = P x + p
    if (is_u) lds[L.u + row] = uval;
    __syncthreads();
    // phase 2: x+ = A x + B u + f
    double xn = acc + af;
#pragma unroll
    for (int c = 0; c < MP; ++c) xn = __builtin_fma(lds[brow + c * n], lds[L.u + c], xn);
    // outputs of knot point k
    const unsigned uk = (unsigned)k;
    bx.st(stx, uk * sv, xcur);
    bu.st(stu, uk * sr, uval);
    by.st(sty, uk * sv, yval);
    __syncthreads();   // every read of this knot point's images is done
    stage(nxt);
    xcur = xn;
    if (is_x) lds[L.x + row] = xn;
    __syncthreads();
  }
  // terminal knot point: x_N, y_N = P_N x_N + p_N (tvlqr.cpp:238-246)
  {
    double acc = 0.0;
    const double* PN = a.P + (int64_t)b * a.bsP + (int64_t)N * nn;
    const double* pN = a.p + (int64_t)b * a.bsp + (int64_t)N * n;
    if (is_y) {
      for (int jj = 0; jj < n; ++jj) acc = __builtin_fma(PN[row + jj * n], lds[L.x + jj], acc);
      acc += pN[row];
    }
    bx.st(stx, (unsigned)N * sv, xcur);
    by.st(sty, (unsigned)N * sv, acc);
  }
}

}  // namespace altro_hip
