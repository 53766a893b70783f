#define T32_UNIT 6
#define T32_UNIT_FN tile32_backward_unit6
#include "tile32_bwd_unit.inc"
