#define R32_UNIT 5
#define R32_UNIT_FN row32_merit_unit5
#include "row32_unit.inc"
