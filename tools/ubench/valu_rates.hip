// Micro-benchmark: issue cost (cycles per wave64 instruction, one wave per SIMD and 4 waves per SIMD) of the VALU
// instructions the Montgomery product is built from.  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>

#define REP 64
#define OUTER 256

template <int KIND>
__global__ void k(uint32_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
    uint64_t c0 = a, c1 = b, c2 = a + b, c3 = a * 3u;
    uint32_t m0 = a, m1 = b, m2 = a ^ b, m3 = a + 7;
    long long t0 = clock64();
    for (int o = 0; o < OUTER; o++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (KIND == 0) {  // 4 independent mad_u64_u32 chains
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");
            } else if (KIND == 1) {  // 1 dependent chain
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0"
                             : "+v"(c0) : "v"(a), "v"(b) : "vcc");
            } else if (KIND == 2) {  // v_mov_b32 x4
                asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3));
            } else if (KIND == 3) {  // v_lshl_add_u64 x4 independent
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
            } else if (KIND == 4) {  // mad + addc pair (Comba step) x2 independent accumulators
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_addc_co_u32 %3, vcc, 0, %3, vcc"
                             : "+v"(c0), "+v"(c1), "+v"(m0), "+v"(m1) : "v"(a), "v"(b) : "vcc");
            } else if (KIND == 5) {  // v_mul_lo_u32 + v_mul_hi_u32 independent
                asm volatile("v_mul_lo_u32 %0, %2, %3\n v_mul_hi_u32 %1, %2, %3\n v_mul_lo_u32 %4, %2, %3\n v_mul_hi_u32 %5, %2, %3"
                             : "=v"(m0), "=v"(m1) : "v"(a), "v"(b), "v"(m2), "v"(m3));
            } else if (KIND == 7) {  // the accumulate loop's mix, 75 % 64-bit: 3 dependent mads + 1 add, twice over two accumulators... counted as 4
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_add_u32 %1, %1, %2"
                             : "+v"(c0), "+v"(m0) : "v"(a), "v"(b) : "vcc");
            } else if (KIND == 8) {  // 50 % 64-bit: 2 dependent mads + 2 adds
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_add_u32 %1, %1, %2\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_add_u32 %1, %1, %3"
                             : "+v"(c0), "+v"(m0) : "v"(a), "v"(b) : "vcc");
            } else if (KIND == 6) {  // v_add_co / v_addc chain x4
                asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_addc_co_u32 %2, vcc, %2, %3, vcc\n v_addc_co_u32 %3, vcc, %3, %0, vcc"
                             : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : : "vcc");
            }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(c0 + c1 + c2 + c3) + m0 + m1 + m2 + m3;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1024] = t1 - t0;
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
    uint32_t* d;
    hipMalloc(&d, 1 << 24);
    int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(d, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long cyc;
    hipMemcpy(&cyc, (char*)d + 1024 * 8, 8, hipMemcpyDeviceToHost);
    double n_inst = (double)OUTER * REP * 4;
    printf("%-34s waves/SIMD=%d  clock64 ticks/inst=%.2f  wall ns/inst/wave=%.3f (x%d waves)\n", name, waves_per_simd, cyc / n_inst,
           ms * 1e6 / n_inst, waves_per_simd);
    hipFree(d);
}

// --json: the one figure bench.py's roofline needs from THIS box - wall time per wave instruction and SIMD of a dependent
// v_mad_u64_u32 chain with 4 and with 8 waves per SIMD (the accumulate kernel runs 8 on the 9-limb field) - as one JSON line
template <int KIND>
double ns_per_inst_per_simd(int waves_per_simd) {
    uint32_t* d;
    hipMalloc(&d, 1 << 24);
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 1);
    hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        k<KIND><<<blocks, 256>>>(d, 2 + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double v = ms * 1e6 / ((double)OUTER * REP * 4) / waves_per_simd;
        if (v < best) best = v;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(d);
    return best;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "--json")) {
        int cus = 0;
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
        printf("{\"mad_u64_dependent_ns_per_wave_inst_per_simd_4waves\": %.4f, \"mad_u64_dependent_ns_per_wave_inst_per_simd_8waves\": %.4f, "
               "\"mad_u64_independent_ns_per_wave_inst_per_simd_4waves\": %.4f, \"add_co_chain_ns_per_wave_inst_per_simd_8waves\": %.4f, "
               "\"lshl_add_u64_ns_per_wave_inst_per_simd_8waves\": %.4f, \"mov_b32_ns_per_wave_inst_per_simd_8waves\": %.4f, "
               "\"mix_75pct_mad_ns_per_wave_inst_per_simd\": %.4f, \"mix_50pct_mad_ns_per_wave_inst_per_simd\": %.4f, \"compute_units\": %d}\n",
               ns_per_inst_per_simd<1>(4), ns_per_inst_per_simd<1>(8), ns_per_inst_per_simd<0>(4), ns_per_inst_per_simd<6>(8),
               ns_per_inst_per_simd<3>(8), ns_per_inst_per_simd<2>(8),
               fmin(fmin(ns_per_inst_per_simd<7>(2), ns_per_inst_per_simd<7>(4)), ns_per_inst_per_simd<7>(8)),
               fmin(fmin(ns_per_inst_per_simd<8>(2), ns_per_inst_per_simd<8>(4)), ns_per_inst_per_simd<8>(8)), cus);
        return 0;
    }
    for (int w : {1, 2, 3, 4}) {
        run<0>("v_mad_u64_u32 x4 independent", w);
        run<1>("v_mad_u64_u32 dependent chain", w);
        run<2>("v_mov_b32", w);
        run<3>("v_lshl_add_u64", w);
        run<4>("mad_u64 + addc_co (comba step)", w);
        run<5>("v_mul_lo_u32 / v_mul_hi_u32", w);
        run<6>("v_add_co/v_addc_co chain", w);
    }
    return 0;
}
