#define FUSED_T float
#define FUSED_G 16
#include "ilqr_fused_unit.inc"
