#define FUSED_T float
#define FUSED_G 32
#include "ilqr_fused_unit.inc"
