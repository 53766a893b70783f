#define FUSED_T float
#define FUSED_G 8
#include "ilqr_fused_unit.inc"
