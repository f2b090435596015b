// BN254 instantiation of the per-curve backend (Fr/Fp: 8 x 32-bit limbs).
#include "backend_impl.h"
namespace apk {
Backend* make_backend_bn254() { return new CurveBackend<FrBN254, FpBN254, APK_BN254>(); }
int g1_mul_batch_bn254(int device, const void* base, const void* scalars, uint64_t count, void* out) {
    return g1_mul_batch_impl<FrBN254, FpBN254>(device, base, scalars, count, out);
}
int g1_decompress_bn254(int device, const uint8_t* in, uint64_t count, void* out) { return g1_decompress_impl<FrBN254, FpBN254, APK_BN254>(device, in, count, out); }
int g1_to_lagrange_bn254(int device, const void* points, uint64_t n, void* out) { return g1_to_lagrange_impl<FrBN254, FpBN254>(device, points, n, out); }
}  // namespace apk
