#define T32_UNIT 2
#define T32_UNIT_FN tile32_backward_unit2
#include "tile32_bwd_unit.inc"
