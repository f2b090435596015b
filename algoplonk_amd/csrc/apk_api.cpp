// extern "C" surface of libapk (include/apk.h): argument checking, curve dispatch, and the host-only wire
// formats that replace /root/reference/helper.go (MarshalProof :13-24, marshalPlonkBls12381Proof :27-88,
// MarshalPublicInputs :91-110).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <type_traits>

#include <hip/hip_runtime_api.h>

#include "backend.h"
#include "ec.h"
#include "sha256.h"

namespace apk {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

// One hardware queue per concurrently proving HIP stream: ROCm's default of 4 serialises a context's 16 proving streams onto
// 4 queues (measured: -20 % proofs/s at BN254 2^17).  The ROCm runtime reads GPU_MAX_HW_QUEUES once, when it initialises (the
// process's first HIP call), so the library sets it when it is LOADED - before any HIP call made through it - unless the
// host process chose a value itself.  include/apk.h documents the contract for hosts that initialise HIP earlier.
__attribute__((constructor)) static void apk_set_runtime_defaults() { setenv("GPU_MAX_HW_QUEUES", "24", /*overwrite=*/0); }

int env_int(const char* name, int dflt, int lo, int hi) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    char* end = nullptr;
    long x = strtol(v, &end, 10);
    if (end == v) return dflt;
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return (int)x;
}

template <class P>
static void mont_to_be(const void* in, uint8_t* be) {
    Fe<P> m;
    memcpy(&m, in, sizeof m);
    Fe<P> c = Fe<P>::from_mont(m);
    constexpr int N = P::N;
    for (int i = 0; i < N; i++) {
        uint8_t* p = be + 4 * (N - 1 - i);
        p[0] = (uint8_t)(c.l[i] >> 24); p[1] = (uint8_t)(c.l[i] >> 16); p[2] = (uint8_t)(c.l[i] >> 8); p[3] = (uint8_t)c.l[i];
    }
}

template <class P>
static int be_to_mont(const uint8_t* be, void* out) {
    constexpr int N = P::N;
    Fe<P> a;
    for (int i = 0; i < N; i++) {
        const uint8_t* p = be + 4 * (N - 1 - i);
        a.l[i] = (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3];
    }
    // reject non-canonical input (>= modulus)
    bool lt = false;
    for (int i = N - 1; i >= 0; i--) {
        if (a.l[i] != P::mod(i)) { lt = a.l[i] < P::mod(i); break; }
    }
    if (!lt) { set_error("field element is not canonical (>= modulus)"); return APK_ERR_ARG; }
    Fe<P> m = Fe<P>::to_mont(a);
    memcpy(out, &m, sizeof m);
    return APK_OK;
}

int host_fe_to_be(int curve, int field, const void* in, uint8_t* be) {
    if (curve == APK_BN254) { field ? mont_to_be<FpBN254>(in, be) : mont_to_be<FrBN254>(in, be); return APK_OK; }
    if (curve == APK_BLS12_381) { field ? mont_to_be<FpBLS12381>(in, be) : mont_to_be<FrBLS12381>(in, be); return APK_OK; }
    set_error("unsupported curve id %d", curve);
    return APK_ERR_ARG;
}

int host_fe_from_be(int curve, int field, const uint8_t* be, void* out) {
    if (curve == APK_BN254) return field ? be_to_mont<FpBN254>(be, out) : be_to_mont<FrBN254>(be, out);
    if (curve == APK_BLS12_381) return field ? be_to_mont<FpBLS12381>(be, out) : be_to_mont<FrBLS12381>(be, out);
    set_error("unsupported curve id %d", curve);
    return APK_ERR_ARG;
}

// gnark RawBytes(): X||Y big-endian; infinity -> 0x40 then zeros on BLS12-381 (helper.go:35-72; verifier/verifier.go:95-99),
// all zeros on BN254 (the only encoding the BN254 template's ec ops take: templateLogicSigBN254.go:57-61)
static void g1_raw(int curve, const uint8_t* slot, uint8_t* out) {
    const size_t fpb = apk_fp_bytes(curve);
    bool inf = true;
    for (size_t i = 0; i < 2 * fpb; i++) if (slot[i]) { inf = false; break; }
    if (inf) { memset(out, 0, 2 * fpb); if (fpb == 48) out[0] = 0x40; return; }
    host_fe_to_be(curve, 1, slot, out);
    host_fe_to_be(curve, 1, slot + fpb, out + fpb);
}

// ---- host-side execution of the SAME arithmetic templates the kernels use (ff.h / ec.h are host+device):
// lets the CPU-only test tier check the field and curve formulas against the oracle without a GPU.
template <class P, class = void> struct HasUnsat : std::false_type {};
template <class P> struct HasUnsat<P, std::void_t<decltype(P::UL)>> : std::true_type {};

template <class P>
static int fe_op_t(int op, const void* a, const void* b, void* out) {
    using F = Fe<P>;
    F x, y, r;
    memcpy(&x, a, sizeof x);
    if (b) memcpy(&y, b, sizeof y); else y = F::zero();
    switch (op) {
        case 0: r = F::add(x, y); break;
        case 1: r = F::sub(x, y); break;
        case 2: r = F::mul(x, y); break;
        case 3: r = F::inv(x); break;
        case 4: r = F::neg(x); break;
        // 10..13: the same operation carried out in the unsaturated-limb MSM field (ffu.h), converted in and out
        case 10: case 11: case 12: case 13:
            if constexpr (HasUnsat<P>::value) {
                using U = FeU<P>;
                U ux = U::from_fe(x), uy = U::from_fe(y);
                r = (op == 10 ? U::mul(ux, uy) : op == 11 ? U::add(ux, uy) : op == 12 ? U::sub(ux, uy) : U::neg(ux)).to_fe();
                break;
            } else {
                set_error("field has no unsaturated-limb form");
                return APK_ERR_ARG;
            }
        // 14: ten lazy butterfly stages as the NTT tile runs them (kernels_ntt.h): (u, v) <- (u + w v, u - w v + 2p) with the
        // twiddle w = y in the R' radix, no comparison until the final canon<16>; returns u
        case 14:
            if constexpr (HasUnsat<P>::value) {
                using U = FeU<P>;
                F ratio = F::zero();
                ratio.l[0] = 1u << (U::B * U::L - 32 * F::N);          // R'/R (2^5 for the 9 x 29-bit fields)
                const U w = U::unpack(F::mul(y, F::to_mont(ratio)).l);  // y * R'/R: gnark's radix -> R'
                U u = U::unpack(x.l), v = U::unpack(y.l);
                for (int i = 0; i < 10; i++) {
                    const U t = U::mul_nr(w, v);
                    const U nu = U::add_n(u, t);
                    v = U::template sub_k<2>(u, t);
                    u = nu;
                }
                U::template canon<16>(u).pack(r.l);
                break;
            } else {
                set_error("field has no unsaturated-limb form");
                return APK_ERR_ARG;
            }
        default: set_error("unknown field op %d", op); return APK_ERR_ARG;
    }
    memcpy(out, &r, sizeof r);
    return APK_OK;
}

template <class FRP, class FPP>
static int g1_op_t(int op, const void* p, const void* q, void* out) {
    using A = Affine<FPP>;
    using X = XYZZ<FPP>;
    A a, b, r;
    memcpy(&a, p, sizeof a);
    switch (op) {
        case 0: {  // mixed add
            memcpy(&b, q, sizeof b);
            X acc = X::from_affine(a);
            acc.madd(b);
            r = acc.to_affine();
            break;
        }
        case 1: {  // full add through a non-trivial ZZ: (2a - a) + b
            memcpy(&b, q, sizeof b);
            X acc = X::dbl_affine(a);
            acc.madd(a, true);
            X other = X::dbl_affine(b);
            other.madd(b, true);
            acc.add(other);
            r = acc.to_affine();
            break;
        }
        case 2: r = X::dbl(X::from_affine(a)).to_affine(); break;
        case 3: {  // scalar multiplication, q = Fr scalar (Montgomery)
            Fe<FRP> s;
            memcpy(&s, q, sizeof s);
            s = Fe<FRP>::from_mont(s);
            X acc = X::inf();
            for (int w = Fe<FRP>::N - 1; w >= 0; w--)
                for (int bit = 31; bit >= 0; bit--) {
                    acc = X::dbl(acc);
                    if ((s.l[w] >> bit) & 1u) acc.madd(a);
                }
            r = acc.to_affine();
            break;
        }
        case 12: case 13: {  // a fixed chain of signed mixed additions through every special case: lazy (12) / plain (13)
            using XU = XYZZ<FPP, FeU<FPP>>;
            memcpy(&b, q, sizeof b);
            const Affine<FPP, FeU<FPP>> pa = unpack_affine<FPP>(to_table_record<FPP>(a)), pb = unpack_affine<FPP>(to_table_record<FPP>(b)),
                                        pinf = Affine<FPP, FeU<FPP>>::inf();
            // a, 2a (doubling), a, inf (cancellation), b, 2b, 2b+a, 3b+a, 3b, 2b, b, skip, a+b, a+2b, a+3b, 3b, a+3b
            static const int script[17][2] = {{0, 0}, {0, 0}, {0, 1}, {0, 1}, {1, 0}, {1, 0}, {0, 0}, {1, 0}, {0, 1}, {1, 1}, {1, 1},
                                              {2, 0}, {0, 0}, {1, 0}, {1, 0}, {0, 1}, {0, 0}};
            XU acc = XU::inf();
            bool flipped = false, unit_z = false;
            for (const auto& st : script) {
                const auto& pt = st[0] == 0 ? pa : st[0] == 1 ? pb : pinf;
                if (op == 12) acc.madd_lazy(pt, st[1] != 0, flipped, unit_z); else acc.madd(pt, st[1] != 0);
            }
            if (op == 12) { acc.lazy_fix_sign(flipped); acc.canonicalize(); }
            r = to_fe_point<FPP>(acc).to_affine();
            break;
        }
        case 14: {  // lazy full addition / doubling through the special cases: ends at 4p + 6q
            using XU = XYZZ<FPP, FeU<FPP>>;
            memcpy(&b, q, sizeof b);
            const Affine<FPP, FeU<FPP>> pa = unpack_affine<FPP>(to_table_record<FPP>(a)), pb = unpack_affine<FPP>(to_table_record<FPP>(b));
            XU x1 = XU::dbl_lazy(XU::from_affine(pa));          // 2a
            x1.add_lazy(XU::from_affine(pb));                    // 2a + b
            XU x2 = XU::dbl_lazy(XU::dbl_lazy(XU::from_affine(pb)));   // 4b
            x2.add_lazy(x1);                                     // 2a + 5b
            XU x3 = x2;
            x3.add_lazy(x2);                                     // equal operands: 4a + 10b
            XU x4 = x2; x4.lazy_neg();
            x3.add_lazy(x4);                                     // 2a + 5b
            XU x5 = x3; x5.lazy_neg();
            x3.add_lazy(x5);                                     // cancellation: infinity
            x3.add_lazy(XU::inf());
            x3.add_lazy(x1);                                     // 2a + b
            x3.add_lazy(x2);                                     // 4a + 6b
            r = to_fe_point<FPP>(x3).to_affine();
            break;
        }
        case 10: case 11: {  // mixed (10) / full (11) addition in the unsaturated-limb representation
            using XU = XYZZ<FPP, FeU<FPP>>;
            memcpy(&b, q, sizeof b);
            Affine<FPP> ra = to_table_record<FPP>(a), rb = to_table_record<FPP>(b);
            XU acc = op == 10 ? XU::from_affine(unpack_affine<FPP>(ra)) : XU::dbl_affine(unpack_affine<FPP>(ra));
            if (op == 11) acc.madd(unpack_affine<FPP>(ra), true);
            if (op == 10) {
                acc.madd(unpack_affine<FPP>(rb));
            } else {
                XU other = XU::dbl_affine(unpack_affine<FPP>(rb));
                other.madd(unpack_affine<FPP>(rb), true);
                acc.add(other);
            }
            r = to_fe_point<FPP>(acc).to_affine();
            break;
        }
        default: set_error("unknown g1 op %d", op); return APK_ERR_ARG;
    }
    memcpy(out, &r, sizeof r);
    return APK_OK;
}

template <class FPP>
static int g1_sum_t(const void* points, uint64_t count, void* out) {
    using A = Affine<FPP>;
    using X = XYZZ<FPP>;
    X acc = X::inf();
    for (uint64_t i = 0; i < count; i++) {
        A p;
        memcpy(&p, reinterpret_cast<const uint8_t*>(points) + i * sizeof(A), sizeof p);
        acc.madd(p);
    }
    const A r = acc.to_affine();
    memcpy(out, &r, sizeof r);
    return APK_OK;
}

}  // namespace apk

using namespace apk;

struct apk_ctx {
    Backend* be;
    int curve;
};

// Live contexts: a communicator keeps a pointer to the context it was bound to, and host languages with finalizers (the Python
// mirror; a Go host with runtime.SetFinalizer) may destroy the context FIRST.  apk_comm_destroy / apk_comm_bind ask here before they
// take their hooks off a context (comm.cpp clear_ctx_hooks): a destroyed one is simply forgotten.
#include <mutex>
#include <set>
namespace apk {
static std::mutex g_live_mu;
static std::set<const void*>& live_set() { static std::set<const void*> s; return s; }
static void ctx_register(const void* c) { std::lock_guard<std::mutex> g(g_live_mu); live_set().insert(c); }
static void ctx_forget(const void* c) { std::lock_guard<std::mutex> g(g_live_mu); live_set().erase(c); }
bool ctx_alive(const void* c) { std::lock_guard<std::mutex> g(g_live_mu); return live_set().count(c) != 0; }
}

extern "C" {

const char* apk_last_error(void) { return g_err.c_str(); }
int apk_abi_version(void) { return APK_ABI_VERSION; }

int apk_device_count(int* count) {
    if (!count) { set_error("null count"); return APK_ERR_ARG; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return APK_OK; }  // runtime present, no usable GPU
    *count = n;
    return APK_OK;
}

size_t apk_fp_bytes(int curve) { return curve == APK_BN254 ? 32 : curve == APK_BLS12_381 ? 48 : 0; }
size_t apk_g1_bytes(int curve) { return 2 * apk_fp_bytes(curve); }

int apk_ctx_create(const apk_circuit_desc* d, apk_ctx** out) {
    if (!d || !out) { set_error("null argument"); return APK_ERR_ARG; }
    *out = nullptr;
    Backend* be = nullptr;
    if (d->curve == APK_BN254) be = make_backend_bn254();
    else if (d->curve == APK_BLS12_381) be = make_backend_bls12381();
    else { set_error("unsupported curve: %d", d->curve); return APK_ERR_ARG; }  // algoplonk.go:39-41
    int r = be->init(d);
    if (r != APK_OK) { delete be; return r; }
    *out = new apk_ctx{be, d->curve};
    ctx_register(*out);
    return APK_OK;
}

int apk_msm_ctx_create(int curve, int device, const void* bases, uint64_t count, int msm_window, apk_ctx** out) {
    if (!bases || !out) { set_error("null argument"); return APK_ERR_ARG; }
    *out = nullptr;
    Backend* be = nullptr;
    if (curve == APK_BN254) be = make_backend_bn254();
    else if (curve == APK_BLS12_381) be = make_backend_bls12381();
    else { set_error("unsupported curve: %d", curve); return APK_ERR_ARG; }
    int r = be->init_msm_only(device, bases, count, msm_window);
    if (r != APK_OK) { delete be; return r; }
    *out = new apk_ctx{be, curve};
    ctx_register(*out);
    return APK_OK;
}

void apk_ctx_destroy(apk_ctx* ctx) {
    if (!ctx) return;
    ctx_forget(ctx);
    delete ctx->be;
    delete ctx;
}

#define NEED_CTX()                                          \
    if (!ctx) { set_error("null context"); return APK_ERR_ARG; }

int apk_ctx_get_vk(apk_ctx* ctx, apk_vk* out) { NEED_CTX(); if (!out) { set_error("null out"); return APK_ERR_ARG; } return ctx->be->get_vk(out); }
int apk_msm_g1(apk_ctx* ctx, int basis, const void* sc, uint64_t len, void* out) {
    NEED_CTX(); if (!sc || !out) { set_error("null argument"); return APK_ERR_ARG; }
    return ctx->be->msm(basis, sc, len, false, out);
}
int apk_msm_g1_device(apk_ctx* ctx, int basis, const void* sc, uint64_t len, void* out) {
    NEED_CTX(); if (!sc || !out) { set_error("null argument"); return APK_ERR_ARG; }
    return ctx->be->msm(basis, sc, len, true, out);
}
int apk_msm_g1_batch_device(apk_ctx* ctx, int basis, uint32_t count, const void* const* d_scalars, const uint64_t* offsets,
                            const uint64_t* lens, void* out) {
    NEED_CTX();
    if (!d_scalars || !offsets || !lens || !out) { set_error("null argument"); return APK_ERR_ARG; }
    return ctx->be->msm_batch(basis, count, d_scalars, offsets, lens, out);
}
int apk_ctx_set_commit_hook(apk_ctx* ctx, apk_commit_hook hook, void* user) { NEED_CTX(); return ctx->be->set_commit_hook(hook, user); }
int apk_device_copy(apk_ctx* ctx, void* d, const void* s, size_t b) { NEED_CTX(); return ctx->be->dev_copy(d, s, b); }
int apk_ctx_set_wire_hook(apk_ctx* ctx, apk_wire_hook hook, void* user) { NEED_CTX(); return ctx->be->set_wire_hook(hook, user); }
int apk_ctx_set_subcoset(apk_ctx* ctx, int k, int world, apk_gather_hook hook, void* user) { NEED_CTX(); return ctx->be->set_subcoset(k, world, hook, user); }
int apk_coset_ntt_device(apk_ctx* ctx, const void* d_in, uint64_t len, void* d_out) { NEED_CTX(); return ctx->be->coset_ntt_dev(d_in, len, d_out); }
int apk_ntt(apk_ctx* ctx, int which, int inverse, int coset, void* data) {
    NEED_CTX(); if (!data) { set_error("null data"); return APK_ERR_ARG; }
    return ctx->be->ntt(which, inverse, coset, data);
}
int apk_prove(apk_ctx* ctx, const void* L, const void* R, const void* O, const void* pub, const void* bl,
              const void* const* pi2, apk_proof* out) {
    NEED_CTX();
    return ctx->be->prove(L, R, O, false, pub, bl, pi2, out);
}
int apk_prove_device(apk_ctx* ctx, const void* L, const void* R, const void* O, const void* pub, const void* bl,
                     const void* const* pi2, apk_proof* out) {
    NEED_CTX();
    return ctx->be->prove(L, R, O, true, pub, bl, pi2, out);
}
// page-locked host memory for apk_prove's inputs (include/apk.h)
static int host_mem_device(int device) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) { (void)hipGetLastError(); set_error("no HIP device available (%s); libapk has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e)); return APK_ERR_HIP; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return APK_ERR_ARG; }
    e = hipSetDevice(device);
    if (e != hipSuccess) { set_error("hipSetDevice: %s", hipGetErrorString(e)); return APK_ERR_HIP; }
    return APK_OK;
}
int apk_host_alloc(int device, size_t bytes, void** p) {
    if (!p || bytes == 0) { set_error("null argument"); return APK_ERR_ARG; }
    *p = nullptr;
    int rc = host_mem_device(device);
    if (rc != APK_OK) return rc;
    hipError_t e = hipHostMalloc(p, bytes, hipHostMallocPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); *p = nullptr; set_error("hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); return APK_ERR_HIP; }
    return APK_OK;
}
int apk_host_free(void* p) {
    if (!p) return APK_OK;
    hipError_t e = hipHostFree(p);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipHostFree: %s", hipGetErrorString(e)); return APK_ERR_HIP; }
    return APK_OK;
}
int apk_host_register(void* p, size_t bytes) {
    if (!p || bytes == 0) { set_error("null argument"); return APK_ERR_ARG; }
    int rc = host_mem_device(0);
    if (rc != APK_OK) return rc;
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipHostRegister(%zu): %s", bytes, hipGetErrorString(e)); return APK_ERR_HIP; }
    return APK_OK;
}
int apk_host_unregister(void* p) {
    if (!p) return APK_OK;
    hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipHostUnregister: %s", hipGetErrorString(e)); return APK_ERR_HIP; }
    return APK_OK;
}

int apk_device_alloc(apk_ctx* ctx, size_t bytes, void** p) { NEED_CTX(); return ctx->be->dev_alloc(bytes, p); }
int apk_device_free(apk_ctx* ctx, void* p) { NEED_CTX(); return ctx->be->dev_free(p); }
int apk_device_upload(apk_ctx* ctx, void* d, const void* s, size_t b) { NEED_CTX(); return ctx->be->dev_upload(d, s, b); }
int apk_device_download(apk_ctx* ctx, void* d, const void* s, size_t b) { NEED_CTX(); return ctx->be->dev_download(d, s, b); }
int apk_stats_enable(apk_ctx* ctx, int en) { NEED_CTX(); return ctx->be->stats_enable(en); }
int apk_stats_read(apk_ctx* ctx, apk_stats* out, int reset) { NEED_CTX(); if (!out) { set_error("null out"); return APK_ERR_ARG; } return ctx->be->stats_read(out, reset); }
int apk_ctx_msm_window(apk_ctx* ctx) { if (!ctx || !ctx->be) return 0; return ctx->be->msm_window(); }
int apk_paths_read(apk_ctx* ctx, apk_path_counts* out, int reset) { NEED_CTX(); if (!out) { set_error("null out"); return APK_ERR_ARG; } return ctx->be->paths_read(out, reset); }

int apk_g1_mul_batch(int curve, int device, const void* base, const void* scalars, uint64_t count, void* out) {
    if (!base || !scalars || !out) { set_error("null argument"); return APK_ERR_ARG; }
    if (count == 0) return APK_OK;
    if (count >= (1ull << 31)) { set_error("count too large"); return APK_ERR_ARG; }
    if (curve == APK_BN254) return g1_mul_batch_bn254(device, base, scalars, count, out);
    if (curve == APK_BLS12_381) return g1_mul_batch_bls12381(device, base, scalars, count, out);
    set_error("unsupported curve: %d", curve);
    return APK_ERR_ARG;
}

int apk_g1_decompress(int curve, int device, const uint8_t* compressed, uint64_t count, void* out) {
    if (!compressed || !out) { set_error("null argument"); return APK_ERR_ARG; }
    if (count == 0) return APK_OK;
    if (count >= (1ull << 31)) { set_error("count too large"); return APK_ERR_ARG; }
    if (curve == APK_BN254) return g1_decompress_bn254(device, compressed, count, out);
    if (curve == APK_BLS12_381) return g1_decompress_bls12381(device, compressed, count, out);
    set_error("unsupported curve: %d", curve);
    return APK_ERR_ARG;
}

int apk_g1_to_lagrange(int curve, int device, const void* points, uint64_t n, void* out) {
    if (!points || !out) { set_error("null argument"); return APK_ERR_ARG; }
    if (curve == APK_BN254) return g1_to_lagrange_bn254(device, points, n, out);
    if (curve == APK_BLS12_381) return g1_to_lagrange_bls12381(device, points, n, out);
    set_error("unsupported curve: %d", curve);
    return APK_ERR_ARG;
}

int apk_host_fe_op(int curve, int field, int op, const void* a, const void* b, void* out) {
    if (!a || !out) { set_error("null argument"); return APK_ERR_ARG; }
    if (curve == APK_BN254) return field ? fe_op_t<FpBN254>(op, a, b, out) : fe_op_t<FrBN254>(op, a, b, out);
    if (curve == APK_BLS12_381) return field ? fe_op_t<FpBLS12381>(op, a, b, out) : fe_op_t<FrBLS12381>(op, a, b, out);
    set_error("unsupported curve: %d", curve);
    return APK_ERR_ARG;
}

int apk_host_g1_op(int curve, int op, const void* p, const void* q, void* out) {
    if (!p || !out || (op != 2 && !q)) { set_error("null argument"); return APK_ERR_ARG; }
    if (curve == APK_BN254) return g1_op_t<FrBN254, FpBN254>(op, p, q, out);
    if (curve == APK_BLS12_381) return g1_op_t<FrBLS12381, FpBLS12381>(op, p, q, out);
    set_error("unsupported curve: %d", curve);
    return APK_ERR_ARG;
}

int apk_g1_sum(int curve, const void* points, uint64_t count, void* out) {
    if (!out || (count && !points)) { set_error("null argument"); return APK_ERR_ARG; }
    if (curve == APK_BN254) return g1_sum_t<FpBN254>(points, count, out);
    if (curve == APK_BLS12_381) return g1_sum_t<FpBLS12381>(points, count, out);
    set_error("unsupported curve: %d", curve);
    return APK_ERR_ARG;
}

int apk_fe_from_be(int curve, int field, const uint8_t* be, void* out) {
    if (!be || !out) { set_error("null argument"); return APK_ERR_ARG; }
    return host_fe_from_be(curve, field, be, out);
}
int apk_fe_to_be(int curve, int field, const void* in, uint8_t* be) {
    if (!be || !in) { set_error("null argument"); return APK_ERR_ARG; }
    return host_fe_to_be(curve, field, in, be);
}

int apk_marshal_proof(const apk_proof* p, uint8_t* out, size_t cap, size_t* len) {
    if (!p || !out || !len) { set_error("null argument"); return APK_ERR_ARG; }
    const int curve = (int)p->curve;
    const size_t pt = apk_g1_bytes(curve);
    if (!pt) { set_error("unrecognized proof type"); return APK_ERR_ARG; }  // helper.go:21 panics here
    const uint32_t k = p->nb_commitments;
    if (k > APK_MAX_COMMITMENTS) { set_error("too many commitments"); return APK_ERR_ARG; }
    const size_t need = 9 * pt + 6 * 32 + (size_t)k * (32 + pt);
    *len = need;
    if (cap < need) { set_error("buffer too small: need %zu bytes", need); return APK_ERR_ARG; }
    uint8_t* w = out;
    for (int i = 0; i < 3; i++) { g1_raw(curve, p->lro[i], w); w += pt; }            // helper.go:33-37
    for (int i = 0; i < 3; i++) { g1_raw(curve, p->h[i], w); w += pt; }              // :42-45
    for (int i = 1; i < 6; i++) { host_fe_to_be(curve, 0, p->claimed_values[i], w); w += 32; }  // :53-56
    g1_raw(curve, p->z, w); w += pt;                                                  // :59-60
    host_fe_to_be(curve, 0, p->zshift_value, w); w += 32;                            // :63-64
    g1_raw(curve, p->batched_h, w); w += pt;                                          // :67-68
    g1_raw(curve, p->zshift_h, w); w += pt;                                           // :71-72
    for (uint32_t i = 0; i < k; i++) { host_fe_to_be(curve, 0, p->claimed_values[6 + i], w); w += 32; }  // :76-79
    for (uint32_t i = 0; i < k; i++) { g1_raw(curve, p->bsb22[i], w); w += pt; }     // :80-83
    return APK_OK;
}

int apk_marshal_public_inputs(int curve, const void* pub, uint32_t nb_public, uint8_t* out, size_t cap) {
    if ((!pub && nb_public) || !out) { set_error("null argument"); return APK_ERR_ARG; }
    if (!apk_fp_bytes(curve)) { set_error("unsupported curve: %d", curve); return APK_ERR_ARG; }
    if (cap < (size_t)nb_public * 32) { set_error("buffer too small"); return APK_ERR_ARG; }
    for (uint32_t i = 0; i < nb_public; i++) host_fe_to_be(curve, 0, (const uint8_t*)pub + 32 * i, out + 32 * i);
    return APK_OK;
}

int apk_hash_fr(int curve, const void* g1_affine, void* out_fr) {
    if (!g1_affine || !out_fr) { set_error("null argument"); return APK_ERR_ARG; }
    const size_t pt = apk_g1_bytes(curve);
    if (!pt) { set_error("unsupported curve: %d", curve); return APK_ERR_ARG; }
    uint8_t raw[2 * 48];
    g1_raw(curve, (const uint8_t*)g1_affine, raw);
    static const uint8_t dst_prime[12] = {'B', 'S', 'B', '2', '2', '-', 'P', 'l', 'o', 'n', 'k', 0x0b};
    uint8_t b0[32], b1[32], b2[32], zeros[64] = {0}, x[32];
    const uint8_t lib[3] = {0x00, 0x30, 0x00}, one = 1, two = 2;
    Sha256 h;
    h.update(zeros, 64); h.update(raw, pt); h.update(lib, 3); h.update(dst_prime, 12); h.final(b0);
    h.reset(); h.update(b0, 32); h.update(&one, 1); h.update(dst_prime, 12); h.final(b1);
    for (int i = 0; i < 32; i++) x[i] = b0[i] ^ b1[i];
    h.reset(); h.update(x, 32); h.update(&two, 1); h.update(dst_prime, 12); h.final(b2);
    // (int(b1) * 2^128 + int(b2[:16])) mod r
    auto reduce = [&](auto tag) {
        using P = decltype(tag);
        using F = Fe<P>;
        auto load = [](const uint8_t* be) {
            F a;
            for (int i = 0; i < 8; i++) { const uint8_t* q = be + 32 - 4 * (i + 1); a.l[i] = (uint32_t)q[0] << 24 | (uint32_t)q[1] << 16 | (uint32_t)q[2] << 8 | q[3]; }
            return F::to_mont(a);
        };
        uint8_t lo[32] = {0};
        memcpy(lo + 16, b2, 16);
        F t = F::zero();
        t.l[4] = 1;
        F r = load(b1) * F::to_mont(t) + load(lo);
        memcpy(out_fr, &r, sizeof r);
    };
    if (curve == APK_BN254) reduce(FrBN254{}); else reduce(FrBLS12381{});
    return APK_OK;
}

}  // extern "C"
