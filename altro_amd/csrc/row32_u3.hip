#define R32_UNIT 3
#define R32_UNIT_FN row32_merit_unit3
#include "row32_unit.inc"
