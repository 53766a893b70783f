#define T32_UNIT 1
#define T32_UNIT_FN tile32_backward_unit1
#include "tile32_bwd_unit.inc"
