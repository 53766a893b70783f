#define FUSED_T double
#define FUSED_G 32
#include "ilqr_fused_unit.inc"
