#define T32_UNIT 5
#define T32_UNIT_FN tile32_backward_unit5
#include "tile32_bwd_unit.inc"
