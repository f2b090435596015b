// Measurement aid (not part of the library): how fast do 3 x `bytes` of host memory reach the device, by which mechanism?
//   hipMemcpyAsync from hipHostMalloc'ed (default / portable) memory, from hipHostRegister'ed malloc memory, from pageable memory,
//   and a plain copy KERNEL reading the page-locked buffer through its device pointer - each on a non-blocking stream, timed on the
//   host from the first call to the completion of a dependent kernel on ANOTHER stream (the prover's pattern: copy stream -> event ->
//   proving stream).
// usage: h2d_probe [bytes_per_buffer = 4194304]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void touch_kernel(const uint4* p, uint4* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = p[0]; }
__global__ void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : (4u << 20);
    hipStream_t copy, work;
    CK(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&work, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    void *d[3], *dout;
    for (int j = 0; j < 3; j++) CK(hipMalloc(&d[j], bytes));
    CK(hipMalloc(&dout, 64));
    void *h_def[3], *h_port[3], *h_reg[3], *h_page[3];
    for (int j = 0; j < 3; j++) {
        CK(hipHostMalloc(&h_def[j], bytes, hipHostMallocDefault));
        CK(hipHostMalloc(&h_port[j], bytes, hipHostMallocPortable));
        h_reg[j] = aligned_alloc(4096, bytes);
        h_page[j] = aligned_alloc(4096, bytes);
        memset(h_def[j], 1, bytes); memset(h_port[j], 2, bytes); memset(h_reg[j], 3, bytes); memset(h_page[j], 4, bytes);
        CK(hipHostRegister(h_reg[j], bytes, hipHostRegisterPortable));
    }
    struct Case { const char* name; void** h; int mode; };   // mode 0: hipMemcpyAsync, 1: copy kernel through the device pointer
    Case cases[] = {{"hipMemcpyAsync  hipHostMalloc default ", h_def, 0}, {"hipMemcpyAsync  hipHostMalloc portable", h_port, 0},
                    {"hipMemcpyAsync  hipHostRegister        ", h_reg, 0}, {"hipMemcpyAsync  pageable               ", h_page, 0},
                    {"copy kernel     hipHostMalloc default ", h_def, 1}, {"copy kernel     hipHostMalloc portable", h_port, 1},
                    {"copy kernel     hipHostRegister        ", h_reg, 1}};
    for (const Case& c : cases) {
        double best = 1e9, best_issue = 1e9;
        for (int rep = 0; rep < 8; rep++) {
            CK(hipDeviceSynchronize());
            const double t0 = now_ms();
            for (int j = 0; j < 3; j++) {
                if (c.mode == 0) CK(hipMemcpyAsync(d[j], c.h[j], bytes, hipMemcpyHostToDevice, copy));
                else {
                    void* dp = nullptr;
                    CK(hipHostGetDevicePointer(&dp, c.h[j], 0));
                    copy_kernel<<<256, 256, 0, copy>>>((const uint4*)dp, (uint4*)d[j], bytes / 16);
                }
            }
            CK(hipEventRecord(ev, copy));
            CK(hipStreamWaitEvent(work, ev, 0));
            touch_kernel<<<1, 64, 0, work>>>((const uint4*)d[2], (uint4*)dout);
            const double t1 = now_ms();
            CK(hipStreamSynchronize(work));
            const double t2 = now_ms();
            if (t2 - t0 < best) best = t2 - t0;
            if (t1 - t0 < best_issue) best_issue = t1 - t0;
        }
        printf("%s  3 x %zu bytes: %.3f ms to a dependent kernel on another stream (%.1f GB/s), host calls returned after %.3f ms\n", c.name, bytes,
               best, 3.0 * bytes / best / 1e6, best_issue);
    }
    return 0;
}
