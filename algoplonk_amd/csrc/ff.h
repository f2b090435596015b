// Prime-field arithmetic in Montgomery form, 32-bit limbs, for gfx950 VALU (v_mad_u64_u32 is the
// workhorse; there is no MFMA use in this code base - modular big-integer arithmetic is not a dense
// contraction).  R = 2^(32N) equals gnark-crypto's 2^(64*N/2), so an element's in-memory bytes are
// exactly gnark's `fr.Element` / `fp.Element` (little-endian limbs, Montgomery form): buffers cross the
// C-ABI without conversion (include/apk.h).
//
// Replaces (on the device) gnark-crypto v0.20.1 ecc/<curve>/{fr,fp} element arithmetic [UPSTREAM, not
// vendored under /root/reference; reached from /root/reference/algoplonk.go:89].
#pragma once
#include <stdint.h>
#include "ff_defs.h"
#include "ff_params.h"
#include "ff_mac.h"

template <class P>
struct Fe {
    static constexpr int N = P::N;
    uint32_t l[N];

    APK_HD static Fe zero() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    APK_HD static Fe one() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::one(i);
        return r;
    }
    APK_HD static Fe modulus() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::mod(i);
        return r;
    }
    APK_HD bool is_zero() const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) acc |= l[i];
        return acc == 0;
    }
    APK_HD bool operator==(const Fe& o) const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) acc |= l[i] ^ o.l[i];
        return acc == 0;
    }
    APK_HD bool operator!=(const Fe& o) const { return !(*this == o); }

    // r = a - p if a >= p else a   (a < 2p)
    APK_HD static Fe reduce_once(const Fe& a) {
        Fe d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)a.l[i] - P::mod(i) - borrow;
            d.l[i] = (uint32_t)t;
            borrow = (uint32_t)(t >> 63);
        }
        Fe r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
        return r;
    }

    APK_HD static Fe add(const Fe& a, const Fe& b) {
        Fe s;
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)a.l[i] + b.l[i] + carry;
            s.l[i] = (uint32_t)t;
            carry = (uint32_t)(t >> 32);
        }
        return reduce_once(s);  // 2p < 2^(32N): no carry out of the top limb
    }

    APK_HD static Fe sub(const Fe& a, const Fe& b) {
        Fe d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)a.l[i] - b.l[i] - borrow;
            d.l[i] = (uint32_t)t;
            borrow = (uint32_t)(t >> 63);
        }
        uint32_t mask = 0u - borrow;
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)d.l[i] + (P::mod(i) & mask) + carry;
            d.l[i] = (uint32_t)t;
            carry = (uint32_t)(t >> 32);
        }
        return d;
    }

    APK_HD static Fe neg(const Fe& a) {
        if (a.is_zero()) return a;
        Fe d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)P::mod(i) - a.l[i] - borrow;
            d.l[i] = (uint32_t)t;
            borrow = (uint32_t)(t >> 63);
        }
        return d;
    }

    APK_HD static Fe dbl(const Fe& a) { return add(a, a); }

    // Montgomery product a*b*R^-1 mod p.
    //
    // Device: product scanning (Comba / "FIPS" Montgomery).  Every 32x32 partial product is ONE v_mad_u64_u32 into a
    // 64-bit column accumulator plus ONE v_addc_co_u32 collecting the carry-out in a third word: 2 VALU instructions
    // per product and no register shuffling.  The operand-scanning C++ below compiles to 128 mads + 120 64-bit adds
    // + 249 v_mov per 8-limb product (measured 1890 cycles per wave-product; tools/ubench/valu_rates.hip gives
    // 4.6 / 4.4 / 2.6 cycles for those three), this form to 128 mads + 128 addc.  Modulus limbs are SGPR operands.
    // Host: portable CIOS with the reduction folded into each outer step (2p < R, so no (N+1)-th limb survives).
#if defined(__HIP_DEVICE_COMPILE__)
    template <int K>
    __device__ __forceinline__ static void cols_lo(const Fe& a, const Fe& b, uint32_t* m, uint64_t& acc, uint32_t& ex) {
        if constexpr (K < N) {
            MacChain<K + 1>::vv(acc, ex, a.l, b.l);                               // sum_{i<=K} a[i] b[K-i]
            if constexpr (K > 0) MacChain<K>::template vs<P, K>(acc, ex, m);      // sum_{i<K} m[i] p[K-i]
            m[K] = (uint32_t)acc * P::INV;
            MacChain<1>::template vs<P, 0>(acc, ex, m + K);                       // + m[K] p[0]: low word becomes 0
            acc = (acc >> 32) | ((uint64_t)ex << 32);
            ex = 0;
            cols_lo<K + 1>(a, b, m, acc, ex);
        }
    }
    template <int K>
    __device__ __forceinline__ static void cols_hi(const Fe& a, const Fe& b, const uint32_t* m, Fe& r, uint64_t& acc, uint32_t& ex) {
        if constexpr (K < 2 * N - 1) {
            constexpr int I0 = K - N + 1, CNT = 2 * N - 1 - K;
            MacChain<CNT>::vv(acc, ex, a.l + I0, b.l + I0);                       // sum_{i>=I0} a[i] b[K-i]
            MacChain<CNT>::template vs<P, N - 1>(acc, ex, m + I0);                // sum_{i>=I0} m[i] p[K-i]
            r.l[K - N] = (uint32_t)acc;
            acc = (acc >> 32) | ((uint64_t)ex << 32);
            ex = 0;
            cols_hi<K + 1>(a, b, m, r, acc, ex);
        }
    }
    __device__ __forceinline__ static Fe mul(const Fe& a, const Fe& b) {
        uint32_t m[N];
        Fe r;
        uint64_t acc = 0;
        uint32_t ex = 0;
        cols_lo<0>(a, b, m, acc, ex);
        cols_hi<N>(a, b, m, r, acc, ex);
        r.l[N - 1] = (uint32_t)acc;
        return reduce_once(r);
    }
#else
    APK_HD static Fe mul(const Fe& a, const Fe& b) {
        uint32_t t[N];
#pragma unroll
        for (int i = 0; i < N; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t bi = b.l[i];
            uint64_t c = 0;
#pragma unroll
            for (int j = 0; j < N; j++) {
                c += (uint64_t)a.l[j] * bi + t[j];
                t[j] = (uint32_t)c;
                c >>= 32;
            }
            const uint32_t top = (uint32_t)c;
            const uint32_t m = t[0] * P::INV;
            c = (uint64_t)m * P::mod(0) + t[0];
            c >>= 32;
#pragma unroll
            for (int j = 1; j < N; j++) {
                c += (uint64_t)m * P::mod(j) + t[j];
                t[j - 1] = (uint32_t)c;
                c >>= 32;
            }
            t[N - 1] = (uint32_t)c + top;
        }
        Fe r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = t[i];
        return reduce_once(r);
    }
#endif

    APK_HD static Fe sqr(const Fe& a) { return mul(a, a); }

    APK_HD static Fe to_mont(const Fe& a) {
        Fe r2;
#pragma unroll
        for (int i = 0; i < N; i++) r2.l[i] = P::r2(i);
        return mul(r2, a);  // r2 < p as the scanned operand keeps the running value < 2p for ANY a < R
    }
    APK_HD static Fe from_mont(const Fe& a) {
        Fe o = zero();
        o.l[0] = 1;
        return mul(a, o);
    }

    // a^e for a little-endian exponent of `words` 32-bit words (not constant time; exponents are public)
    APK_HD static Fe pow(const Fe& a, const uint32_t* e, int words) {
        Fe r = one();
        bool started = false;
        for (int w = words - 1; w >= 0; w--) {
            for (int b = 31; b >= 0; b--) {
                if (started) r = sqr(r);
                if ((e[w] >> b) & 1) {
                    r = started ? mul(r, a) : a;
                    started = true;
                }
            }
        }
        return r;
    }
    APK_HD static Fe pow_u64(const Fe& a, uint64_t e) {
        uint32_t w[2] = {(uint32_t)e, (uint32_t)(e >> 32)};
        return pow(a, w, 2);
    }
    // Fermat inversion a^(p-2); inv(0) = 0.  ~1.5*BITS multiplications: kept as the cross-check for inv().
    APK_HD static Fe inv_fermat(const Fe& a) {
        uint32_t e[N];
#pragma unroll
        for (int i = 0; i < N; i++) e[i] = P::pm2(i);
        if (a.is_zero()) return a;
        return pow(a, e, N);
    }

    // ---- Montgomery inverse (Kaliski 1995): binary extended Euclid on plain limbs, then <= 3 Montgomery
    // products to fix the power of two.  ~2*BITS shift/subtract steps on N limbs, i.e. roughly a tenth of
    // the dependent-instruction chain of Fermat's a^(p-2); that chain, not throughput, is what a lone lane
    // (final affine conversion, batch-inversion seeds) pays for.  inv(0) = 0.
    APK_HD static bool ge_raw(const uint32_t* x, const uint32_t* y) {
        for (int i = N - 1; i >= 0; i--) {
            if (x[i] != y[i]) return x[i] > y[i];
        }
        return true;
    }
    APK_HD static void sub_raw(uint32_t* x, const uint32_t* y) {  // x -= y
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)x[i] - y[i] - borrow;
            x[i] = (uint32_t)t;
            borrow = (uint32_t)(t >> 63);
        }
    }
    APK_HD static void add_raw(uint32_t* x, const uint32_t* y) {  // x += y (no overflow: values < 2p < R)
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t t = (uint64_t)x[i] + y[i] + carry;
            x[i] = (uint32_t)t;
            carry = (uint32_t)(t >> 32);
        }
    }
    APK_HD static void shr1_raw(uint32_t* x) {
#pragma unroll
        for (int i = 0; i < N - 1; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
        x[N - 1] >>= 1;
    }
    APK_HD static void shl1_raw(uint32_t* x) {
#pragma unroll
        for (int i = N - 1; i > 0; i--) x[i] = (x[i] << 1) | (x[i - 1] >> 31);
        x[0] <<= 1;
    }
    APK_HD static Fe inv(const Fe& a) {
        if (a.is_zero()) return a;
        uint32_t u[N], v[N], r[N], s[N];
#pragma unroll
        for (int i = 0; i < N; i++) { u[i] = P::mod(i); v[i] = a.l[i]; r[i] = 0; s[i] = 0; }
        s[0] = 1;
        int k = 0;
        for (;;) {
            uint32_t vz = 0;
#pragma unroll
            for (int i = 0; i < N; i++) vz |= v[i];
            if (vz == 0) break;
            if ((u[0] & 1u) == 0) { shr1_raw(u); shl1_raw(s); }
            else if ((v[0] & 1u) == 0) { shr1_raw(v); shl1_raw(r); }
            else if (!ge_raw(v, u)) { sub_raw(u, v); shr1_raw(u); add_raw(r, s); shl1_raw(s); }   // u > v
            else { sub_raw(v, u); shr1_raw(v); add_raw(s, r); shl1_raw(r); }
            k++;
        }
        // r = a^-1 * 2^k (mod p) up to sign: result = p - r (after one conditional subtraction)
        Fe x;
        {
            uint32_t m[N];
#pragma unroll
            for (int i = 0; i < N; i++) m[i] = P::mod(i);
            if (ge_raw(r, m)) sub_raw(r, m);
            sub_raw(m, r);
#pragma unroll
            for (int i = 0; i < N; i++) x.l[i] = m[i];
        }
        // x = abar^-1 * 2^k with BITS <= k <= 2*BITS.  Bring it to abar^-1 * R (plain inverse of the plain
        // value when the input is in Montgomery form), then one more product by R^2 re-enters Montgomery form.
        constexpr int M = 32 * N;
        Fe r2;
#pragma unroll
        for (int i = 0; i < N; i++) r2.l[i] = P::r2(i);
        if (k <= M) { x = mul(x, r2); k += M; }      // x = abar^-1 * 2^k, now M < k <= 2M
        Fe pw;
        const int e = 2 * M - k;                      // 0 <= e < M
#pragma unroll
        for (int i = 0; i < N; i++) pw.l[i] = (i == (e >> 5)) ? (1u << (e & 31)) : 0u;
        x = mul(x, pw);                               // abar^-1 * 2^(2M) * R^-1 = abar^-1 * R
        return mul(x, r2);                            // (a R)^-1 * R * R = a^-1 * R
    }
};

template <class P> APK_HD Fe<P> operator+(const Fe<P>& a, const Fe<P>& b) { return Fe<P>::add(a, b); }
template <class P> APK_HD Fe<P> operator-(const Fe<P>& a, const Fe<P>& b) { return Fe<P>::sub(a, b); }
template <class P> APK_HD Fe<P> operator*(const Fe<P>& a, const Fe<P>& b) { return Fe<P>::mul(a, b); }
