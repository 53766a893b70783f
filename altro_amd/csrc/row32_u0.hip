#define R32_UNIT 0
#define R32_UNIT_FN row32_merit_unit0
#include "row32_unit.inc"
