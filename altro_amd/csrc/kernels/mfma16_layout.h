// mfma16_layout.h -- element offsets of plan MFMA16's HBM records (see tvlqr_mfma16.hip for the layout picture).
//
// The symmetric blocks are stored ONCE: the cost Hessian Q of a knot point and the cost-to-go Hessian P it produces
// travel through HBM as their upper triangles (78 of 144 entries, row-major packed).  The sweep is HBM-bound, so the
// 132 doubles saved per knot point (of 636) are time: tools/membench.hip prices the same access pattern at 1.066 ms
// for the full records and 0.85 ms for the packed ones.  Q_N / P_N (one record per problem) stay full.
#pragma once

namespace altro_hip {

constexpr int MF_N = 12, MF_M = 4;
constexpr int MF_TRI = 78;     // entries of the upper triangle of a 12 x 12 block
constexpr int MF_DYN = 204;    // elements per knot-point dynamics record:  Z 192 | f 12
constexpr int MF_COST = 160;   // elements per knot-point cost record:      triu(Q) 78 | pad 2 | [H R] 64 | [q r] 16
constexpr int MF_IN = MF_DYN + MF_COST;   // 364
constexpr int MF_OUT = 144;    // elements per knot-point output record:    Kt 52 | triu(P) 78 | p 12 | pad 2
constexpr int MF_TERM = 156;   // elements per terminal record (Q_N rows | q_N, and [P_N p_N] 12x13)
constexpr int MF_OFF_Z = 0, MF_OFF_F = 192;                   // inside a dynamics record
constexpr int MF_OFF_Q = 0, MF_OFF_HR = 80, MF_OFF_QR = 144;  // inside a cost record
constexpr int MF_OFF_P = 52, MF_OFF_p = 130, MF_OFF_PAD = 142;   // inside an OUT record, after Kt = [K | -d] 4x13
constexpr int MF_QB = 256 + 16;  // optional Q-block record: G tile (16x16 row-major) | [Qx Qu]

// offset of entry (i, j) of a symmetric 12 x 12 block inside its packed upper triangle (row-major: row i holds
// columns i..11); every record is a multiple of 64 bytes, so no two problems' records share a cache line.
__host__ __device__ constexpr int mf_tri_row(int i) { return 12 * i - (i * (i - 1)) / 2; }
__host__ __device__ constexpr int mf_sym(int i, int j) {
  return i <= j ? mf_tri_row(i) + (j - i) : mf_tri_row(j) + (i - j);
}
static_assert(mf_sym(11, 11) == MF_TRI - 1, "packed triangle size");
static_assert(MF_COST % 8 == 0 && MF_OUT % 8 == 0, "records are whole 64-byte lines");

// XCD-aware block -> problem mapping of the wave-per-problem kernels.  Workgroups are dealt round-robin to the eight
// XCDs, each with its own L2: with problem = blockIdx the records one XCD touches at a time are every eighth record of
// the [k][b] slab.  Here XCD x takes the contiguous eighth [x * chunk, (x + 1) * chunk) of the batch instead, so each
// L2 streams one contiguous run per slab (measured on C1: backward 0.83-0.85 -> 0.80 ms, forward 0.63-0.65 -> 0.61 ms).
// Launch mf_grid(batch) blocks; blocks past the end of a ragged eighth return at once.
constexpr int MF_XCDS = 8;
__host__ __device__ constexpr int mf_grid(int batch) { return MF_XCDS * ((batch + MF_XCDS - 1) / MF_XCDS); }
__host__ __device__ constexpr int mf_problem(int block, int batch) {
  return (block % MF_XCDS) * ((batch + MF_XCDS - 1) / MF_XCDS) + block / MF_XCDS;
}

}  // namespace altro_hip
