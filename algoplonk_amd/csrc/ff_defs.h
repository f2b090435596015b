// Host/device qualifier shared by the arithmetic headers.  The same templates run inside the HIP
// kernels (gfx950) and, for the handful of scalar computations between Fiat-Shamir challenges, on the
// host thread that drives them.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define APK_HD __host__ __device__ __forceinline__
#else
#define APK_HD inline
#endif
