#define FUSED_T double
#define FUSED_G 8
#include "ilqr_fused_unit.inc"
