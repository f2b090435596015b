// BLS12-381 instantiation of the per-curve backend (Fr: 8 x 32-bit limbs, Fp: 12 x 32-bit limbs).
#include "backend_impl.h"
namespace apk {
Backend* make_backend_bls12381() { return new CurveBackend<FrBLS12381, FpBLS12381, APK_BLS12_381>(); }
int g1_mul_batch_bls12381(int device, const void* base, const void* scalars, uint64_t count, void* out) {
    return g1_mul_batch_impl<FrBLS12381, FpBLS12381>(device, base, scalars, count, out);
}
int g1_decompress_bls12381(int device, const uint8_t* in, uint64_t count, void* out) { return g1_decompress_impl<FrBLS12381, FpBLS12381, APK_BLS12_381>(device, in, count, out); }
int g1_to_lagrange_bls12381(int device, const void* points, uint64_t n, void* out) { return g1_to_lagrange_impl<FrBLS12381, FpBLS12381>(device, points, n, out); }
}  // namespace apk
