#define R32_UNIT 7
#define R32_UNIT_FN row32_merit_unit7
#include "row32_unit.inc"
