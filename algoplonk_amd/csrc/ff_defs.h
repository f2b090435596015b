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
// Round 4, variant builds interleaved on one box (tools/ab_libs.sh, BN254 2^17, proofs/s | lone proof ms): all 0: 502.7 | 3.33;
// SORT 3: +0.7 %; TAIL 3: -0.2 %; SORT 3 + TAIL 3: 0; FR 3: +0.5 % | 3.44; FR 2 + SORT 3: +2.1 % | 3.46; FR 3 + SORT 3: +1.9 % | 3.43;
// FR 1 + SORT 2 + TAIL 1: +1.4 %; **FR 2 + SORT 3 + TAIL 3: +1.6 % | 3.28** - the only setting that gains on both, and the default:
// the accumulate loops (priority 0) fill whatever issue slots the short kernels leave.
#ifndef APK_PRIO_SORT
#define APK_PRIO_SORT 3
#endif
#ifndef APK_PRIO_TAIL
#define APK_PRIO_TAIL 3
#endif
#ifndef APK_PRIO_FR
#define APK_PRIO_FR 2
#endif
#if defined(__HIPCC__)
template <int P> __device__ __forceinline__ void wave_priority() { if constexpr (P > 0) __builtin_amdgcn_s_setprio(P); }
#endif
