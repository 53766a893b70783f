// plan MFMA16's merit / expansion kernels for knot points with up to 6 constraint slots (see the included file)
#define ALTRO_WIDE_SLOTS 6
#include "ilqr_launch_mfma16_wide.inc"
