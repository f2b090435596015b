// VERDICT r03 item 8, priced: could the Montgomery REDUCTION half of the 9 x 29-bit product (m = T_lo * (-p^-1) mod R', T + m * p:
// two products by CONSTANTS, 90 + 18 of the product's 206 VALU instructions) move to the i8 MFMA pipe as a Toeplitz-matrix x
// lane-batch product, with the VALU pipe keeping the 81 variable x variable partial products?
//
// What the matrix pipe would need around it, per field element and reduction, all of it VALU work:
//   (a) limbs -> bytes: 9 limbs of 29 bits -> 33 byte digits packed four to a register (the MFMA's A / B operand format);
//   (b) the MFMA's i32 column sums (66 columns of weight 2^(8c), each up to 33 * 255^2 < 2^22) -> 18 limbs of 29 bits with
//       carries - this is what the mad chain gets for free from its 64-bit accumulator;
// and that twice (m itself is a product by the constant -p^-1).  This file makes (a) and (b) exact - compile with -S and count
// (tools/ubench/mont_mfma.sh does) - and, on a GPU, measures the i8 MFMA rate alone and beside a v_mad_u64_u32 chain on the
// same SIMDs (does the matrix pipe co-issue, and at what cost to the chain?).
// Build: hipcc --offload-arch=gfx950 -O3 mont_mfma.hip -o mont_mfma      Count: hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only ...
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int L = 9, B = 29, NB = 33, NC = 66;

// (a) nine 29-bit limbs -> nine registers of packed byte digits (36 bytes, the top three zero)
__device__ __forceinline__ void limbs_to_bytes(const uint32_t* l, uint32_t* w) {
    uint64_t buf = 0;
    int have = 0, li = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        while (have < 32 && li < L) { buf |= (uint64_t)l[li++] << have; have += B; }
        w[k] = (uint32_t)buf;
        buf >>= 32;
        have -= 32;
    }
}
// (b) 66 column sums of weight 2^(8c) -> 18 limbs of 29 bits (a 64-bit window slides over the columns; exact)
__device__ __forceinline__ void columns_to_limbs(const int32_t* col, uint32_t* l) {
    uint64_t acc = 0;
    int have = 0, c = 0;
#pragma unroll
    for (int k = 0; k < 2 * L; k++) {
        while (have < B + 8 && c < NC) { acc += (uint64_t)(uint32_t)col[c++] << have; have += 8; }
        l[k] = (uint32_t)acc & ((1u << B) - 1u);
        acc >>= B;
        have -= B;
    }
}

extern "C" __global__ void count_limbs_to_bytes(const uint32_t* in, uint32_t* out) {
    uint32_t l[L], w[9];
    for (int i = 0; i < L; i++) l[i] = in[threadIdx.x * L + i];
    limbs_to_bytes(l, w);
    for (int i = 0; i < 9; i++) out[threadIdx.x * 9 + i] = w[i];
}
extern "C" __global__ void count_columns_to_limbs(const int32_t* in, uint32_t* out) {
    int32_t col[NC];
    uint32_t l[2 * L];
    for (int i = 0; i < NC; i++) col[i] = in[threadIdx.x * NC + i];
    columns_to_limbs(col, l);
    for (int i = 0; i < 2 * L; i++) out[threadIdx.x * 2 * L + i] = l[i];
}
// the loads and stores of the two kernels above alone, to subtract
extern "C" __global__ void count_baseline_9(const uint32_t* in, uint32_t* out) {
    for (int i = 0; i < 9; i++) out[threadIdx.x * 9 + i] = in[threadIdx.x * 9 + i];
}
extern "C" __global__ void count_baseline_66_18(const int32_t* in, uint32_t* out) {
    for (int i = 0; i < 2 * L; i++) out[threadIdx.x * 2 * L + i] = (uint32_t)in[threadIdx.x * NC + 3 * i];   // addressing only
}

// ---- rates --------------------------------------------------------------------------------------------------------------------
typedef int v16i __attribute__((ext_vector_type(16)));
#define REP 256
// mode 0: MFMA only; 1: mad chain only; 2: even waves MFMA, odd waves mads (same SIMD when two waves share it)
__global__ void rates(int mode, uint32_t* out, long long* ticks) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && (wave & 1) == 0);
    const bool do_mad = mode == 1 || (mode == 2 && (wave & 1) == 1);
    v16i acc = {0};
    long a = threadIdx.x * 0x0101010101010101L, b = 0x0203050709020305L;
    uint64_t c0 = threadIdx.x;
    uint32_t x = threadIdx.x * 2654435761u, y = x ^ 0x9e3779b9u;
    long long t0 = clock64();
    for (int r = 0; r < REP; r++) {
        if (do_mfma) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc = __builtin_amdgcn_mfma_i32_32x32x16_i8(a, b, acc, 0, 0, 0);
        }
        if (do_mad) {
#pragma unroll
            for (int i = 0; i < 64; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c0) : "v"(x), "v"(y) : "vcc");
        }
    }
    long long t1 = clock64();
    uint32_t s = (uint32_t)c0;
    for (int i = 0; i < 16; i++) s += (uint32_t)acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

int main() {
    uint32_t* d; long long* t;
    hipMalloc(&d, 1 << 24); hipMalloc(&t, 8);
    const char* names[3] = {"i8 MFMA 32x32x16 only", "v_mad_u64_u32 chain only", "MFMA waves + mad waves on the same SIMDs"};
    for (int wps : {1, 2, 4}) {
        for (int mode = 0; mode < 3; mode++) {
            const int threads = 256 * wps > 1024 ? 1024 : 256 * wps;      // 4 SIMDs per CU: wps waves per SIMD
            const int blocks = 256 * (256 * wps / threads);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            rates<<<blocks, threads>>>(mode, d, t); hipDeviceSynchronize();
            hipEventRecord(e0); rates<<<blocks, threads>>>(mode, d, t); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double waves = (double)blocks * threads / 64;
            double mf = 0, md = 0;
            if (mode == 0) mf = waves; else if (mode == 1) md = waves; else { mf = waves / 2; md = waves / 2; }
            const double macs = mf * REP * 8 * 32.0 * 32 * 16, mads = md * REP * 64 * 64.0;
            printf("%d wave(s)/SIMD  %-44s %8.3f ms   %8.1f T i8-MAC/s   %8.2f T lane-mad/s\n", wps, names[mode], ms, macs / ms / 1e9, mads / ms / 1e9);
        }
    }
    return 0;
}
