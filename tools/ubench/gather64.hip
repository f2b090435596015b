// Micro-benchmark: random gathers of whole 64-byte records (the accumulate kernel's access pattern: 4 x global_load_dwordx4 per
// lane and record) against 128-byte records, from a 1 GiB table (beyond L2 and the 256 MiB Infinity Cache).  If every 64-byte
// gather dragged its 128-byte neighbour along, the useful rate of the 64-byte case would be half that of the 128-byte case.
// Build: hipcc --offload-arch=gfx950 -O3 gather64.hip -o gather64.bin      
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int WORDS16>   // record = WORDS16 x 16 bytes
__global__ void __launch_bounds__(256) gather(const uint4* __restrict__ table, uint32_t nrec, uint32_t per_thread, uint4* __restrict__ out) {
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint4 acc = {0, 0, 0, 0};
    for (uint32_t i = 0; i < per_thread; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t r = (uint32_t)(((uint64_t)x * nrec) >> 32);
        const uint4* p = table + (size_t)r * WORDS16;
#pragma unroll
        for (int w = 0; w < WORDS16; w++) { uint4 v = p[w]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int WORDS16>
void run(const uint4* table, size_t bytes, uint4* out) {
    const uint32_t nrec = (uint32_t)(bytes / (16 * WORDS16));
    const uint32_t blocks = 256 * 8 * 4, per_thread = 256;
    gather<WORDS16><<<blocks, 256>>>(table, nrec, 16, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    gather<WORDS16><<<blocks, 256>>>(table, nrec, per_thread, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double recs = (double)blocks * 256 * per_thread;
    printf("%3d-byte records: %.2f G records/s, %.2f TB/s useful\n", 16 * WORDS16, recs / ms / 1e6, recs * 16 * WORDS16 / ms / 1e9);
}

int main() {
    const size_t bytes = 1ull << 30;
    uint4 *table, *out;
    hipMalloc(&table, bytes);
    hipMalloc(&out, (size_t)256 * 8 * 4 * 256 * 16);
    hipMemset(table, 1, bytes);
    run<4>(table, bytes, out);
    run<8>(table, bytes, out);
    run<2>(table, bytes, out);
    run<4>(table, bytes, out);
    return 0;
}
