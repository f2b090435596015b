// Curve-erased interface between the C-ABI (apk_api.cpp) and the per-curve HIP backends
// (backend_impl.h, instantiated in backend_bn254.hip / backend_bls12381.hip).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/apk.h"

namespace apk {

void set_error(const char* fmt, ...);

// Run-time knob from the environment, clamped to [lo, hi] (a value outside the range can never reach a launch configuration
// or an allocation size: nothing may abort across the C-ABI); `dflt` when the variable is unset or not a number.
int env_int(const char* name, int dflt, int lo, int hi);

// is this apk_ctx* still a live context (created and not yet destroyed)?  apk_api.cpp keeps the registry.
bool ctx_alive(const void* ctx);

struct Backend {
    virtual ~Backend() {}
    virtual int init(const apk_circuit_desc* d) = 0;
    virtual int init_msm_only(int device, const void* bases, uint64_t count, int msm_window) = 0;
    virtual int get_vk(apk_vk* out) = 0;
    virtual int msm(int basis, const void* scalars, uint64_t len, bool on_device, void* out) = 0;
    virtual int msm_batch(int basis, uint32_t count, const void* const* d_scalars, const uint64_t* offsets, const uint64_t* lens, void* out) = 0;
    virtual int set_commit_hook(apk_commit_hook fn, void* user) = 0;
    virtual int dev_copy(void* d, const void* s, size_t bytes) = 0;
    virtual int set_wire_hook(apk_wire_hook fn, void* user) = 0;
    virtual int set_subcoset(int k, int G, apk_gather_hook fn, void* user) = 0;
    virtual int device_ordinal() = 0;
    virtual int msm_window() = 0;         // the signed-digit window width the context's tables were built for
    virtual uint64_t domain_size() = 0;   // n; 0 on an MSM-only context
    virtual int coset_ntt_dev(const void* d_in, uint64_t len, void* d_out) = 0;
    virtual int ntt(int which, int inverse, int coset, void* data) = 0;
    virtual int prove(const void* L, const void* R, const void* O, bool on_device, const void* pub, const void* blinding,
                      const void* const* pi2, apk_proof* out) = 0;
    virtual int dev_alloc(size_t bytes, void** p) = 0;
    virtual int dev_free(void* p) = 0;
    virtual int dev_upload(void* d, const void* s, size_t bytes) = 0;
    virtual int dev_download(void* d, const void* s, size_t bytes) = 0;
    virtual int stats_enable(int enable) = 0;
    virtual int stats_read(apk_stats* out, int reset) = 0;
    virtual int paths_read(apk_path_counts* out, int reset) = 0;
};

int g1_mul_batch_bn254(int device, const void* base, const void* scalars, uint64_t count, void* out);
int g1_mul_batch_bls12381(int device, const void* base, const void* scalars, uint64_t count, void* out);
int g1_decompress_bn254(int device, const uint8_t* in, uint64_t count, void* out);
int g1_decompress_bls12381(int device, const uint8_t* in, uint64_t count, void* out);
int g1_to_lagrange_bn254(int device, const void* points, uint64_t n, void* out);
int g1_to_lagrange_bls12381(int device, const void* points, uint64_t n, void* out);
Backend* make_backend_bn254();
Backend* make_backend_bls12381();

// host-only helpers implemented per curve (no GPU): used by marshal / conversions / hash_fr
int host_fe_from_be(int curve, int field, const uint8_t* be, void* out);
int host_fe_to_be(int curve, int field, const void* in, uint8_t* be);

}  // namespace apk
