#define R32_UNIT 4
#define R32_UNIT_FN row32_merit_unit4
#include "row32_unit.inc"
