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

    // Montgomery product a*b*R^-1 mod p, CIOS with the reduction folded into each outer step.  The
    // top limb of every modulus here leaves a spare bit (2p < R), so the running value stays below 2p
    // and no (N+1)-th limb survives an iteration.
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
    // Fermat inversion a^(p-2); inv(0) = 0
    APK_HD static Fe inv(const Fe& a) {
        uint32_t e[N];
#pragma unroll
        for (int i = 0; i < N; i++) e[i] = P::pm2(i);
        if (a.is_zero()) return a;
        return pow(a, e, N);
    }
};

template <class P> APK_HD Fe<P> operator+(const Fe<P>& a, const Fe<P>& b) { return Fe<P>::add(a, b); }
template <class P> APK_HD Fe<P> operator-(const Fe<P>& a, const Fe<P>& b) { return Fe<P>::sub(a, b); }
template <class P> APK_HD Fe<P> operator*(const Fe<P>& a, const Fe<P>& b) { return Fe<P>::mul(a, b); }
