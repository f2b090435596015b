// BN254 instantiation of the per-curve backend (Fr/Fp: 8 x 32-bit limbs).
#include "backend_impl.h"
namespace apk {
Backend* make_backend_bn254() { return new CurveBackend<FrBN254, FpBN254, APK_BN254>(); }
int g1_mul_batch_bn254(int device, const void* base, const void* scalars, uint64_t count, void* out) {
    return g1_mul_batch_impl<FrBN254, FpBN254>(device, base, scalars, count, out);
}
}  // namespace apk
