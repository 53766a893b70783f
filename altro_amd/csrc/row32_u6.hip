#define R32_UNIT 6
#define R32_UNIT_FN row32_merit_unit6
#include "row32_unit.inc"
