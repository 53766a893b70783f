#define R32_UNIT 1
#define R32_UNIT_FN row32_merit_unit1
#include "row32_unit.inc"
