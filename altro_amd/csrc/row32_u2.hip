#define R32_UNIT 2
#define R32_UNIT_FN row32_merit_unit2
#include "row32_unit.inc"
