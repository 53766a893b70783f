#define T32_UNIT 3
#define T32_UNIT_FN tile32_backward_unit3
#include "tile32_bwd_unit.inc"
