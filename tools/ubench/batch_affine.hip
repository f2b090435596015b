// Batch-affine bucket accumulation against the lazy XYZZ mixed addition, MEASURED on the access pattern of the MSM
// (DESIGN.md section 5, round 3): the cost model of the design decision, not a correctness test - the table holds random field
// elements, not curve points, and the affine kernel skips the special cases a real one needs (P = +-Q, infinity).
//
//   xyzz    : what msm_accumulate_kernel does - one lane per work unit of 16 gathered table records, madd_lazy into an XYZZ
//             accumulator (2 012 VALU instructions per addition in the real kernel)
//   affine  : one lane per K independent additions P1 + P2 of gathered records, Montgomery batch inversion inside the lane:
//             pass 1 gathers x1, x2, keeps the prefix products of (x2 - x1) in a [K][lanes] scratch array; ONE inversion;
//             pass 2 gathers both records again, peels the inverses backwards, lambda, x3, y3, stores the 64-byte result.
//             inversion = 0: none (the limit of an infinitely large batch), 1: Fe::inv (the library's Kaliski inverse)
//
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I algoplonk_amd/csrc tools/ubench/batch_affine.hip -o tools/ubench/batch_affine.bin
// run  : tools/ubench/batch_affine.bin            (prints additions per second for both, for one 2^17 MSM's worth of additions
//                                                   and for 8 of them in one launch = the saturated device)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ec.h"

// -DUBENCH_BLS21 (round 6): the regime the round-3 analysis named as the one where batch-affine COULD pay - BLS12-381 (3 771
// instructions per XYZZ addition against ~2 300 for 5 products + 1 square), 2^21 bases x 14 windows of 19 bits = a 2.8 GB table that
// no cache holds, 64-entry units in the XYZZ loop, K = 64 .. 512 additions per lane and inversion.
#ifdef UBENCH_BLS21
using FP = FpBLS12381;
#else
using FP = FpBN254;
#endif
using F = FeU<FP>;
using PT = XYZZ<FP, F>;
using Rec = Affine<FP>;

#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(128) xyzz_kernel(const Rec* __restrict__ table, const uint32_t* __restrict__ idx, uint32_t units, uint32_t unit,
                                                   PT* __restrict__ out) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    PT acc = PT::inf();
    bool flipped = false, unit_z = false;
    for (uint32_t e = 0; e < unit; e++) {
        const uint32_t v = idx[u * unit + e];
        const Rec rec = table[v & 0x7fffffffu];
        acc.madd_lazy(unpack_affine<FP>(rec), (v >> 31) != 0, flipped, unit_z);
    }
    acc.lazy_fix_sign(flipped);
    out[u] = acc;
}

template <int INVERSION>
__global__ void __launch_bounds__(128) affine_kernel(const Rec* __restrict__ table, const uint32_t* __restrict__ idx, uint32_t lanes, uint32_t K,
                                                     F* __restrict__ scratch, Rec* __restrict__ out) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= lanes) return;
    const uint32_t* my = idx + (size_t)u * K * 2;
    F prefix = F::one();
    for (uint32_t k = 0; k < K; k++) {
        const Rec r1 = table[my[2 * k] & 0x7fffffffu], r2 = table[my[2 * k + 1] & 0x7fffffffu];
        const F d = F::template sub_k<2>(F::unpack(r2.x.l), F::unpack(r1.x.l));
        prefix = k ? F::mul_nr(prefix, d) : d;
        scratch[(size_t)k * lanes + u] = prefix;
    }
    F inv = prefix;
    if constexpr (INVERSION == 1) {
        Fe<FP> t;
        F::template canon<4>(prefix).pack(t.l);
        t = Fe<FP>::inv(t);
        inv = F::unpack(t.l);
    }
    for (uint32_t k = K; k-- > 0;) {
        const Rec r1 = table[my[2 * k] & 0x7fffffffu], r2 = table[my[2 * k + 1] & 0x7fffffffu];
        const F x1 = F::unpack(r1.x.l), y1 = F::unpack(r1.y.l), x2 = F::unpack(r2.x.l), y2 = F::unpack(r2.y.l);
        const F d = F::template sub_k<2>(x2, x1);
        F inv_d = inv;
        if (k) { inv_d = F::mul_nr(inv, scratch[(size_t)(k - 1) * lanes + u]); inv = F::mul_nr(inv, d); }
        const F lam = F::mul_nr(F::template sub_k<2>(y2, y1), inv_d);
        const F x3 = F::template sub2_k<4>(F::sqr_nr(lam), x1, x2);        // lambda^2 - x1 - 2 x2 stands in for lambda^2 - x1 - x2 (same cost)
        const F y3 = F::template sub_k<2>(F::mul_nr(lam, F::template sub_k<6>(x1, x3)), y1);
        Rec o;
        F::template canon<4>(x3).pack(o.x.l);
        F::template canon<4>(y3).pack(o.y.l);
        out[(size_t)u * K + k] = o;
    }
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }

int main() {
#ifdef UBENCH_BLS21
    const uint32_t n = 1u << 21, W = 14, table_n = n * W;               // the 2.8 GB windowed table of a 2^21 BLS12-381 context (c = 19)
#else
    const uint32_t n = 1u << 17, W = 16, table_n = n * W;               // the 134 MB windowed table of a 2^17 BN254 context
#endif
    constexpr int NL = FP::N;
    std::vector<Rec> h_table(table_n);
    for (auto& r : h_table) for (int i = 0; i < NL; i++) { r.x.l[i] = rnd() & (i == NL - 1 ? 0x0fffffffu : ~0u); r.y.l[i] = rnd() & (i == NL - 1 ? 0x0fffffffu : ~0u); }
    Rec* d_table; HCHK(hipMalloc(&d_table, sizeof(Rec) * table_n)); HCHK(hipMemcpy(d_table, h_table.data(), sizeof(Rec) * table_n, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; HCHK(hipEventCreate(&e0)); HCHK(hipEventCreate(&e1));
#ifdef UBENCH_BLS21
    for (uint32_t mult : {1u, 3u}) {     // one MSM, a batch of three (29 M / 88 M additions: the device is saturated either way)
#else
    for (uint32_t mult : {1u, 8u}) {
#endif
        const uint64_t adds = (uint64_t)table_n * mult;                   // one MSM = n * W additions
        std::vector<uint32_t> h_idx(adds * 2);
        for (auto& v : h_idx) v = (rnd() % table_n) | (rnd() & 0x80000000u);
        uint32_t* d_idx; HCHK(hipMalloc(&d_idx, 4 * h_idx.size())); HCHK(hipMemcpy(d_idx, h_idx.data(), 4 * h_idx.size(), hipMemcpyHostToDevice));
        void* d_out; HCHK(hipMalloc(&d_out, adds * sizeof(Rec) + (adds / 16 + 64) * sizeof(PT)));
        F* d_scratch; HCHK(hipMalloc(&d_scratch, adds * sizeof(F)));
        auto timeit = [&](auto launch, const char* name, double extra) {
            launch(); HCHK(hipDeviceSynchronize());
            HCHK(hipEventRecord(e0));
            const int reps = 10;
            for (int i = 0; i < reps; i++) launch();
            HCHK(hipEventRecord(e1)); HCHK(hipEventSynchronize(e1));
            float ms; HCHK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            printf("%-44s x%u MSM: %8.3f ms  %7.2f G additions/s%s\n", name, mult, ms, adds / (ms * 1e-3) / 1e9, extra < 0 ? "" : "");
        };
#ifdef UBENCH_BLS21
        const uint32_t unit = 64, units = (uint32_t)(adds / unit);
        timeit([&] { xyzz_kernel<<<(units + 127) / 128, 128>>>(d_table, d_idx, units, unit, (PT*)d_out); }, "xyzz  (madd_lazy, 64 per lane)", -1);
        for (uint32_t K : {32u, 64u, 128u, 256u, 512u}) {
#else
        const uint32_t unit = 16, units = (uint32_t)(adds / unit);
        timeit([&] { xyzz_kernel<<<(units + 127) / 128, 128>>>(d_table, d_idx, units, unit, (PT*)d_out); }, "xyzz  (madd_lazy, 16 per lane)", -1);
        for (uint32_t K : {8u, 16u, 32u, 64u, 256u}) {
#endif
            const uint32_t lanes = (uint32_t)(adds / K);
            char nm[96];
            snprintf(nm, sizeof nm, "affine K=%-3u no inversion (batch -> inf)", K);
            timeit([&] { affine_kernel<0><<<(lanes + 127) / 128, 128>>>(d_table, d_idx, lanes, K, d_scratch, (Rec*)d_out); }, nm, -1);
            snprintf(nm, sizeof nm, "affine K=%-3u Kaliski inverse per lane", K);
            timeit([&] { affine_kernel<1><<<(lanes + 127) / 128, 128>>>(d_table, d_idx, lanes, K, d_scratch, (Rec*)d_out); }, nm, -1);
        }
        HCHK(hipFree(d_idx)); HCHK(hipFree(d_out)); HCHK(hipFree(d_scratch));
    }
    return 0;
}
