// mfma16_layout.h -- element offsets of plan MFMA16's HBM records (see tvlqr_mfma16.hip for the layout picture).
#pragma once

namespace altro_hip {

constexpr int MF_N = 12, MF_M = 4;
constexpr int MF_DYN = 204;    // elements per knot-point dynamics record:  Z 192 | f 12
constexpr int MF_COST = 224;   // elements per knot-point cost record:      Q rows 144 | [H R] 64 | [q r] 16
constexpr int MF_IN = MF_DYN + MF_COST;   // 428 = 2n^2+2nm+m^2+2n+m: every element is algorithmic
constexpr int MF_OUT = 208;    // doubles per knot-point output record
constexpr int MF_TERM = 156;   // doubles per terminal record
constexpr int MF_OFF_Z = 0, MF_OFF_F = 192;                  // inside a dynamics record
constexpr int MF_OFF_Q = 0, MF_OFF_HR = 144, MF_OFF_QR = 208;  // inside a cost record
constexpr int MF_OFF_P = 52;   // inside an OUT record: [P p] after Kt
constexpr int MF_QB = 256 + 16;  // optional Q-block record: G tile (16x16 row-major) | [Qx Qu]

}  // namespace altro_hip
