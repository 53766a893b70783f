#define T32_UNIT 0
#define T32_UNIT_FN tile32_backward_unit0
#include "tile32_bwd_unit.inc"
