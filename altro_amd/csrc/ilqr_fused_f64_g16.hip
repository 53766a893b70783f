#define FUSED_T double
#define FUSED_G 16
#include "ilqr_fused_unit.inc"
