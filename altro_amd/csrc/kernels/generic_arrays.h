// generic_arrays.h -- the arrays of plan GENERIC (reference layout on the device: per problem, knot point after knot point, every block
// column-major with its own dimensions) and their order in the offset table off[(N + 1) * G_NUM]: off[k * G_NUM + a] is where knot
// point k's block of array a starts inside one problem's stretch of that array.  Shared by the TVLQR kernels (tvlqr_generic.hip) and
// the iLQR loop of this plan (ilqr_generic.hip), which walks the same table so that per-knot-point dimensions need nothing special.
#pragma once

namespace altro_hip {

enum GArr {
  G_A = 0, G_B, G_f, G_Q, G_R, G_H, G_q, G_r,      // inputs
  G_K, G_d, G_P, G_p,                              // outputs
  G_Qxx, G_Quu, G_Qux, G_Qx, G_Qu,                 // optional outputs
  G_Qxx_tmp, G_Quu_tmp, G_Qux_tmp, G_Qx_tmp, G_Qu_tmp,  // the reference's scratch blocks (store_q == 2)
  G_x, G_u, G_y,                                   // forward outputs
  G_NUM
};

}  // namespace altro_hip
