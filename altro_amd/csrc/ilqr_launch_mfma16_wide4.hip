// plan MFMA16's merit / expansion kernels for knot points with up to 4 constraint slots (see the included file)
#define ALTRO_WIDE_SLOTS 4
#include "ilqr_launch_mfma16_wide.inc"
