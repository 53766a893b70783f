// rtc_compat.h -- the system headers of the device code that can also be compiled at run time by hiprtc
// (altro_hip_set_model_source, capi_rtc.hip): hiprtc has the HIP device API and the fixed-width integer types built in and
// no system include path, so under __HIPCC_RTC__ nothing is included.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#else
typedef __hip_internal::int64_t int64_t;
typedef __hip_internal::uint64_t uint64_t;
typedef __hip_internal::int32_t int32_t;
typedef __hip_internal::uint32_t uint32_t;
typedef struct ihipStream_t* hipStream_t;   // (host launcher DECLARATIONS in the shared type headers mention it)
#endif
