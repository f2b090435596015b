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

// Wave issue priority of the latency-bound kernels (s_setprio, 0..3).  With 16 proofs in flight the counting-sort and
// reduction kernels share their SIMDs with the accumulate loops of other proofs; they issue few instructions but hold wave
// slots and LDS for as long as they are resident, so they are let through first.  Compile-time (the instruction takes an
// immediate): SORT = msm_digits / scans, TAIL = combine / row-column / bit sums, FR = NTT passes and the pointwise Fr kernels.
#ifndef APK_PRIO_SORT
#define APK_PRIO_SORT 0
#endif
#ifndef APK_PRIO_TAIL
#define APK_PRIO_TAIL 0
#endif
#ifndef APK_PRIO_FR
#define APK_PRIO_FR 0
#endif
#if defined(__HIPCC__)
template <int P> __device__ __forceinline__ void wave_priority() { if constexpr (P > 0) __builtin_amdgcn_s_setprio(P); }
#endif
