// Host-only compile unit of libapk (plain g++, no HIP): apk_verify and the G2 helpers of include/apk.h.  Kept apart from
// apk_api.cpp because the Fp12 tower templates take minutes to optimise and change rarely.
#include <string.h>

#include "backend.h"
#include "verify_host.h"

namespace apk {

template <class FR, class FP, class PP, int CURVE_ID>
static int g2_decompress_t(const uint8_t* in, void* out) {
    G2Aff<FP, PP> q;
    if (g2_decompress<FP, PP, CURVE_ID>(in, q) != APK_OK) { set_error("not a valid compressed G2 point"); return APK_ERR_ARG; }
    // gnark's G2Affine.SetBytes also rejects points outside the order-r subgroup (the twists have large cofactors)
    if (!q.inf && !G2Aff<FP, PP>::template mul<FR>(q, Fe<FR>::modulus()).inf) { set_error("compressed G2 point is not in the prime-order subgroup"); return APK_ERR_ARG; }
    memset(out, 0, 4 * sizeof(Fe<FP>));
    if (!q.inf) { memcpy(out, &q.x, sizeof q.x); memcpy((uint8_t*)out + sizeof q.x, &q.y, sizeof q.y); }
    return APK_OK;
}
template <class FR, class FP, class PP>
static int g2_mul_gen_t(const void* scalar, void* out) {
    Fe<FR> k;
    memcpy(&k, scalar, sizeof k);
    const G2Aff<FP, PP> q = G2Aff<FP, PP>::template mul<FR>(G2Aff<FP, PP>::generator(), Fe<FR>::from_mont(k));
    memset(out, 0, 4 * sizeof(Fe<FP>));
    if (!q.inf) { memcpy(out, &q.x, sizeof q.x); memcpy((uint8_t*)out + sizeof q.x, &q.y, sizeof q.y); }
    return APK_OK;
}
}  // namespace apk

using namespace apk;

extern "C" {

int apk_verify_ex(const apk_verifying_key* vk, const apk_proof* proof, const void* public_inputs, uint32_t nb_public_inputs,
                  apk_verify_trace* trace) {
    if (!vk || !proof || (vk->nb_public && !public_inputs)) { set_error("null argument"); return APK_ERR_ARG; }
    // gnark's plonk.Verify: len(publicWitness) != vk.NbPublicVariables is an error, never a truncation or an over-read
    if (nb_public_inputs != vk->nb_public) {
        set_error("invalid witness size, got %u, expected %u (public)", nb_public_inputs, vk->nb_public);
        return APK_ERR_VERIFY;
    }
    if (vk->curve == APK_BN254) return HostVerifier<FrBN254, FpBN254, PairBN254, APK_BN254>::verify(vk, proof, public_inputs, trace);
    if (vk->curve == APK_BLS12_381) return HostVerifier<FrBLS12381, FpBLS12381, PairBLS12381, APK_BLS12_381>::verify(vk, proof, public_inputs, trace);
    set_error("unsupported curve: %d", vk->curve);
    return APK_ERR_ARG;
}

int apk_verify(const apk_verifying_key* vk, const apk_proof* proof, const void* public_inputs) {
    if (!vk) { set_error("null argument"); return APK_ERR_ARG; }
    return apk_verify_ex(vk, proof, public_inputs, vk->nb_public, nullptr);
}

int apk_g2_decompress(int curve, const uint8_t* compressed, void* out) {
    if (!compressed || !out) { set_error("null argument"); return APK_ERR_ARG; }
    if (curve == APK_BN254) return g2_decompress_t<FrBN254, FpBN254, PairBN254, APK_BN254>(compressed, out);
    if (curve == APK_BLS12_381) return g2_decompress_t<FrBLS12381, FpBLS12381, PairBLS12381, APK_BLS12_381>(compressed, out);
    set_error("unsupported curve: %d", curve);
    return APK_ERR_ARG;
}

int apk_g2_mul_generator(int curve, const void* scalar_fr, void* out) {
    if (!scalar_fr || !out) { set_error("null argument"); return APK_ERR_ARG; }
    if (curve == APK_BN254) return g2_mul_gen_t<FrBN254, FpBN254, PairBN254>(scalar_fr, out);
    if (curve == APK_BLS12_381) return g2_mul_gen_t<FrBLS12381, FpBLS12381, PairBLS12381>(scalar_fr, out);
    set_error("unsupported curve: %d", curve);
    return APK_ERR_ARG;
}

}  // extern "C"
