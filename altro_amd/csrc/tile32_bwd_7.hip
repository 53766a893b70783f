#define T32_UNIT 7
#define T32_UNIT_FN tile32_backward_unit7
#include "tile32_bwd_unit.inc"
