// pack.hip -- layout conversion between the reference layout ([batch][k][column-major block], what
// the reference's callers hold: src/altro/solver/solver.cpp:63-106) and each plan's private device
// layout.  Runs on the device so that broadcast inputs (one A for every k and every problem) are
// expanded in HBM instead of on the host.  Setup/teardown path: never inside the timed sweep.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfma16_layout.h"

namespace altro_hip {

struct SrcArr {
  const double* p;   // device pointer, reference layout, relative to the chunk's first problem
  int64_t bs, ks;    // strides in elements between problems / knot points (0 = broadcast)
  int bmod;          // > 0: the source holds only `bmod` distinct problems, tiled over the batch
};
__device__ __forceinline__ int64_t src_b(const SrcArr& s, int b) { return s.bmod > 0 ? (b % s.bmod) : b; }

// dst[(b0+b)*dst_bs + k*dst_ks + e] = src[b*bs + k*ks + e],  e < block, k < nk, b < nb
template <typename T>
__global__ void expand_copy_kernel(T* dst, int64_t dst_bs, int64_t dst_ks, SrcArr src, int block,
                                   int nk, int b0, int nb) {
  const int64_t total = (int64_t)nb * nk * block;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(t % block);
    const int k = (int)((t / block) % nk);
    const int b = (int)(t / ((int64_t)block * nk));
    dst[(int64_t)(b0 + b) * dst_bs + (int64_t)k * dst_ks + e] =
        (T)src.p[(src.bmod > 0 ? src_b(src, b0 + b) : (int64_t)b) * src.bs + (int64_t)k * src.ks + e];
  }
}

// out[b*bs + k*ks + e] = (double) src[(b0+b)*src_bs + k*src_ks + e]
template <typename T>
__global__ void gather_copy_kernel(double* out, int64_t out_bs, int64_t out_ks, const T* src,
                                   int64_t src_bs, int64_t src_ks, int block, int nk, int b0, int nb) {
  const int64_t total = (int64_t)nb * nk * block;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(t % block);
    const int k = (int)((t / block) % nk);
    const int b = (int)(t / ((int64_t)block * nk));
    out[(int64_t)b * out_bs + (int64_t)k * out_ks + e] =
        (double)src[(int64_t)(b0 + b) * src_bs + (int64_t)k * src_ks + e];
  }
}

// ---- plan MFMA16 ------------------------------------------------------------------------------
struct Mfma16Strides { int64_t in_bs, in_ks, out_bs, out_ks, xuy_bs, xuy_ks, cin_bs, cin_ks; };

enum Mfma16Seg { MSEG_Z = 0, MSEG_F, MSEG_Q, MSEG_HR, MSEG_QR, MSEG_TERM_Q, MSEG_TERM_q };

// One launch fills one segment of IN (or TERM) for problems [b0, b0+nb).  `s0`, `s1` are the one or
// two reference arrays the segment draws from (Z: A,B ; HR: H,R ; QR: q,r).
// The records are always those of the (12, 4) tile; a problem with n <= 12 states and m <= 4 inputs rides them zero-padded:
//   Z = [A 0 | B 0; 0 0 | 0 0], f = [f; 0], Q = diag(Q, 0), H = [H 0; 0 0], q = [q; 0], r = [r; 0]  and  R = diag(R, I),
// so that Quu stays positive definite and its factor is diag(chol(Quu), I): the padding rows of K, d, P, p and the padding
// entries of x, u, y come out exactly 0, and the real entries are the same sums with zero terms added (same bits).
template <typename S>
__global__ void mfma16_pack_kernel(S* in, S* cin, S* term, Mfma16Strides st, int seg, SrcArr s0,
                                   SrcArr s1, int is_diag, int N, int b0, int nb, int n, int m) {
  int len, base;
  switch (seg) {
    case MSEG_Z: len = 192; base = MF_OFF_Z; break;
    case MSEG_F: len = 12; base = MF_OFF_F; break;
    case MSEG_Q: len = MF_TRI; base = MF_OFF_Q; break;
    case MSEG_HR: len = 64; base = MF_OFF_HR; break;
    case MSEG_QR: len = 16; base = MF_OFF_QR; break;
    case MSEG_TERM_Q: len = 144; base = 0; break;
    default: len = 12; base = 144; break;
  }
  const bool terminal = (seg == MSEG_TERM_Q || seg == MSEG_TERM_q);
  const int nk = terminal ? 1 : N;
  const int64_t total = (int64_t)nb * nk * len;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(t % len);
    const int k = terminal ? N : (int)((t / len) % nk);
    const int b = (int)(t / ((int64_t)len * nk));
    const double* p0 = s0.p ? s0.p + src_b(s0, b0 + b) * s0.bs + (int64_t)k * s0.ks : nullptr;
    const double* p1 = s1.p ? s1.p + src_b(s1, b0 + b) * s1.bs + (int64_t)k * s1.ks : nullptr;
    double v = 0.0;
    switch (seg) {
      case MSEG_Z: {  // Zfrag[c][lane] = Z[4c + (lane>>4)][lane&15],  Z = [A B]
        const int c = e / 64, l = e % 64, col = l & 15, row = (l >> 4) + 4 * c;
        if (row < n) {
          if (col < 12) v = col < n ? p0[row + n * col] : 0.0;
          else v = col - 12 < m ? p1[row + n * (col - 12)] : 0.0;
        }
      } break;
      case MSEG_F: v = (p0 && e < n) ? p0[e] : 0.0; break;
      case MSEG_Q: {  // upper triangle of the (symmetric) Q, row-major packed: e <-> (row, jj >= row)
        int row = 0, rem = e;
        while (rem >= 12 - row) { rem -= 12 - row; ++row; }
        const int jj = row + rem;
        if (jj < n) v = is_diag ? (row == jj ? p0[row] : 0.0) : p0[row + n * jj];     // (row <= jj)
      } break;
      case MSEG_TERM_Q: {  // Q_N rows, full: [row][jj]
        const int row = e / 12, jj = e % 12;
        if (row < n && jj < n) v = is_diag ? (row == jj ? p0[row] : 0.0) : p0[row + n * jj];
      } break;
      case MSEG_HR: {  // [g][j] = j < 12 ? H[g][j] : R[g][j-12]
        const int g = e / 16, jj = e % 16;
        if (g >= m) v = (jj == 12 + g) ? 1.0 : 0.0;                                   // padding inputs: R = I
        else if (jj < 12) v = (is_diag || !p0 || jj >= n) ? 0.0 : p0[g + m * jj];
        else if (jj - 12 < m) v = is_diag ? (g == jj - 12 ? p1[g] : 0.0) : p1[g + m * (jj - 12)];
      } break;
      case MSEG_QR: v = (e < 12) ? (e < n ? p0[e] : 0.0) : (e - 12 < m ? p1[e - 12] : 0.0); break;
      default: v = e < n ? p0[e] : 0.0; break;  // MSEG_TERM_q
    }
    if (terminal) term[(int64_t)(b0 + b) * MF_TERM + base + e] = (S)v;
    else if (seg == MSEG_Z || seg == MSEG_F) in[(int64_t)(b0 + b) * st.in_bs + (int64_t)k * st.in_ks + base + e] = (S)v;
    else cin[(int64_t)(b0 + b) * st.cin_bs + (int64_t)k * st.cin_ks + base + e] = (S)v;
  }
}

enum Mfma16Get { MGET_K = 0, MGET_d, MGET_P, MGET_p, MGET_x, MGET_u, MGET_y, MGET_QBLK };

// Reference-layout view of the results for problems [b0, b0+nb): dst is [nb][nk][len].
template <typename S>
__global__ void mfma16_unpack_kernel(double* dst, int what, const S* out, const S* outn,
                                     const S* xuy, const S* qblk, Mfma16Strides st, int N,
                                     int b0, int nb, int n, int m) {
  int len, nk;
  switch (what) {   // (n, m): the problem's own dimensions -- the real entries of the (12, 4) records
    case MGET_K: len = m * n; nk = N; break;
    case MGET_d: len = m; nk = N; break;
    case MGET_P: len = n * n; nk = N + 1; break;
    case MGET_p: len = n; nk = N + 1; break;
    case MGET_x: len = n; nk = N + 1; break;
    case MGET_u: len = m; nk = N; break;
    case MGET_y: len = n; nk = N + 1; break;
    default: len = n * n + m * m + m * n + n + m; nk = N; break;
  }
  const int64_t total = (int64_t)nb * nk * len;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(t % len);
    const int k = (int)((t / len) % nk);
    const int b = b0 + (int)(t / ((int64_t)len * nk));
    const S* o = (k < N) ? out + (int64_t)b * st.out_bs + (int64_t)k * st.out_ks : nullptr;
    const S* pn = outn + (int64_t)b * MF_TERM;   // [P_N p_N] 12x13 (k == N)
    const S* xr = xuy ? xuy + (int64_t)b * st.xuy_bs + (int64_t)k * st.xuy_ks : nullptr;
    double v;
    switch (what) {
      case MGET_K: v = (double)o[(e % m) * 13 + (e / m)]; break;        // K[a + m j] = Kt[a][j]
      case MGET_d: v = -(double)o[e * 13 + 12]; break;                  // d = -Kt[:, 12]
      case MGET_P:   // P[i + n j]: both halves from the stored upper triangle (k < N) / tile[i][j] (k == N)
        v = (k < N) ? (double)o[MF_OFF_P + mf_sym(e % n, e / n)] : (double)pn[(e % n) * 13 + (e / n)];
        break;
      case MGET_p: v = (k < N) ? (double)o[MF_OFF_p + e] : (double)pn[e * 13 + 12]; break;
      case MGET_x: v = xr[e]; break;
      case MGET_y: v = xr[12 + e]; break;
      case MGET_u: v = xr[24 + e]; break;
      default: {  // Qxx(144) | Quu(16) | Qux(48) | Qx(12) | Qu(4), column-major blocks
        const S* q = qblk + ((int64_t)b * N + k) * MF_QB;
        const int oxx = n * n, ouu = oxx + m * m, oux = ouu + m * n, ox = oux + n;
        if (e < oxx) v = q[(e % n) * 16 + (e / n)];
        else if (e < ouu) { int t2 = e - oxx; v = q[(12 + t2 % m) * 16 + 12 + t2 / m]; }
        else if (e < oux) { int t2 = e - ouu; v = q[(12 + t2 % m) * 16 + t2 / m]; }
        else if (e < ox) v = q[256 + (e - oux)];
        else v = q[256 + 12 + (e - ox)];
      } break;
    }
    dst[t] = v;
  }
}

}  // namespace altro_hip
