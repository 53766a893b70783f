// fp_contract.h -- floating-point contraction regions.
//
// The device functions the solve paths share (models.h, linesearch_sm.h, kernels/al_lane.hip, kernels/ilqr_loop_logic.h,
// kernels/ilqr_lane.hip) must round the same way in every kernel they are inlined into -- the fused solve kernel, the
// launch-sequenced kernels and the three-launch merit evaluation are compared bit for bit -- so they are compiled under
// `contract(on)` (only a syntactic a * b + c becomes an FMA; `fast` contracts across statements, differently per inlining
// context); the TVLQR bodies that repeat the CPU path operation for operation are compiled under `contract(off)`.
//
// `#pragma clang fp` has no push / pop.  A region is therefore closed by re-establishing the TRANSLATION UNIT's own mode,
// which a TU states with ALTRO_TU_FP_CONTRACT before its first include (0 off, 1 on, 2 fast); the default, 2, is what
// hipcc compiles device code with (-ffp-contract=fast), host C++ units say 1 (clang's default for C++).  Regions do not
// nest, and a header never leaves a mode behind that its includer did not ask for (the headers used to end with
// an unconditional contract(fast), which silently switched the rest of any includer -- host units too -- to fast).
#pragma once

#ifndef ALTRO_TU_FP_CONTRACT
#define ALTRO_TU_FP_CONTRACT 2
#endif

#if defined(__clang__)
#define ALTRO_FP_REGION_ON _Pragma("clang fp contract(on)")
#define ALTRO_FP_REGION_OFF _Pragma("clang fp contract(off)")
#define ALTRO_FP_REGION_FAST _Pragma("clang fp contract(fast)")
#if ALTRO_TU_FP_CONTRACT == 0
#define ALTRO_FP_REGION_END _Pragma("clang fp contract(off)")
#elif ALTRO_TU_FP_CONTRACT == 1
#define ALTRO_FP_REGION_END _Pragma("clang fp contract(on)")
#else
#define ALTRO_FP_REGION_END _Pragma("clang fp contract(fast)")
#endif
#else
#define ALTRO_FP_REGION_ON
#define ALTRO_FP_REGION_OFF
#define ALTRO_FP_REGION_FAST
#define ALTRO_FP_REGION_END
#endif
