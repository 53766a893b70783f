#define T32_UNIT 4
#define T32_UNIT_FN tile32_backward_unit4
#include "tile32_bwd_unit.inc"
