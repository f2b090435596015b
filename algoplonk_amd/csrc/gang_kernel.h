// One launch for the same kernel of several proofs (gang.h): the element-wise / scan / reduction kernels between the MSMs and NTTs
// of a proof are written as functors (kernels_poly.h `struct FooK { static __device__ void run(args...) }`), launched either alone
//     poly1_kernel<FooK, BOUNDS><<<grid, block, lds, stream>>>(args...)
// or for up to GANG_MAX members at once, blockIdx.z = member, every member with ITS OWN arguments
//     polyG_kernel<FooK, BOUNDS><<<(grid.x, grid.y, members), block, lds, stream>>>(GangPack{member 0's args, member 1's args, ...})
// Same body, same arithmetic, same bytes: a member's instance sees exactly the blockIdx.x / .y, blockDim and arguments it would
// have seen alone.  Small circuits under load are bound by launches, not by work (a 2 us kernel takes ~100 us of a loaded stream's
// time): a gang of four then makes four proofs in the launches of one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gang.h"

namespace apk {

// a by-value argument list as a plain aggregate (kernel arguments must be trivially copyable; std::tuple is not guaranteed to be)
template <class... P> struct Pack {};
template <class H, class... T> struct Pack<H, T...> {
    H head;
    Pack<T...> tail;
};
static inline Pack<> make_pack_impl() { return Pack<>{}; }
template <class H, class... T> static inline Pack<H, T...> make_pack_impl(H h, T... t) { return Pack<H, T...>{h, make_pack_impl(t...)}; }

template <class K, class... Got>
__device__ __forceinline__ void pack_run(const Pack<>&, const Got&... got) { K::run(got...); }
template <class K, class H, class... T, class... Got>
__device__ __forceinline__ void pack_run(const Pack<H, T...>& p, const Got&... got) { pack_run<K>(p.tail, got..., p.head); }

template <class... P> struct GangPack { Pack<P...> m[GANG_MAX]; };

template <class K, int BOUNDS, class... P>
__global__ void __launch_bounds__(BOUNDS) poly1_kernel(P... p) { K::run(p...); }

template <class K, int BOUNDS, class... P>
__global__ void __launch_bounds__(BOUNDS) polyG_kernel(GangPack<P...> g) { pack_run<K>(g.m[blockIdx.z]); }

}  // namespace apk
