// HOST-only prime-field arithmetic on 64-bit limbs (unsigned __int128 products), same Montgomery radix and the same bytes
// as Fe<P> (ff.h): an Fe<P> of N 32-bit limbs IS an Fe64<P> of N/2 64-bit limbs on a little-endian host, so values move
// between the two by memcpy.  Fe<P>'s portable host path (32-bit limbs) costs ~100 ns (BN254) / ~500 ns (BLS12-381) per
// product; this one ~25 / ~60 ns.  Used where the prover's host thread does curve arithmetic on the critical path of a
// proof: the commitment of the linearised polynomial taken homomorphically from commitments it already holds
// (backend_impl.h round 4), affine conversions of MSM sums, apk_verify.
//
// Same interface as Fe<P> / FeU<P> as far as ec.h's XYZZ<FP, FT> needs it.
#pragma once
#include <stdint.h>
#include <string.h>
#include "ff.h"

template <class P>
struct Fe64 {
    static constexpr int N = P::N / 2;
    static_assert(P::N % 2 == 0, "whole 64-bit limbs");
    uint64_t l[N];
    using u128 = unsigned __int128;

    static constexpr uint64_t mod(int i) { return (uint64_t)P::mod(2 * i) | ((uint64_t)P::mod(2 * i + 1) << 32); }
    // -p^-1 mod 2^64 from the 32-bit constant by one Newton step
    static constexpr uint64_t ninv() {
        const uint64_t x32 = (uint64_t)(uint32_t)(0u - P::INV);          // p^-1 mod 2^32
        const uint64_t x64 = x32 * (2 - mod(0) * x32);                   // p^-1 mod 2^64
        return 0 - x64;
    }
    static Fe64 from(const Fe<P>& a) { Fe64 r; memcpy(r.l, a.l, sizeof r.l); return r; }
    Fe<P> to() const { Fe<P> r; memcpy(r.l, l, sizeof l); return r; }

    static Fe64 zero() { Fe64 r; for (int i = 0; i < N; i++) r.l[i] = 0; return r; }
    static Fe64 one() { return from(Fe<P>::one()); }
    bool is_zero() const { uint64_t a = 0; for (int i = 0; i < N; i++) a |= l[i]; return a == 0; }
    bool operator==(const Fe64& o) const { uint64_t a = 0; for (int i = 0; i < N; i++) a |= l[i] ^ o.l[i]; return a == 0; }
    bool operator!=(const Fe64& o) const { return !(*this == o); }

    // a < 2p -> [0, p)
    static Fe64 reduce_once(const Fe64& a) {
        Fe64 d;
        uint64_t borrow = 0;
        for (int i = 0; i < N; i++) {
            const u128 t = (u128)a.l[i] - mod(i) - borrow;
            d.l[i] = (uint64_t)t;
            borrow = (uint64_t)(t >> 64) & 1;
        }
        return borrow ? a : d;
    }
    static Fe64 add(const Fe64& a, const Fe64& b) {
        Fe64 s;
        uint64_t carry = 0;
        for (int i = 0; i < N; i++) {
            const u128 t = (u128)a.l[i] + b.l[i] + carry;
            s.l[i] = (uint64_t)t;
            carry = (uint64_t)(t >> 64);
        }
        return reduce_once(s);   // 2p < 2^(64N): no carry out (ff_params.h asserts the spare top bit)
    }
    static Fe64 sub(const Fe64& a, const Fe64& b) {
        Fe64 d;
        uint64_t borrow = 0;
        for (int i = 0; i < N; i++) {
            const u128 t = (u128)a.l[i] - b.l[i] - borrow;
            d.l[i] = (uint64_t)t;
            borrow = (uint64_t)(t >> 64) & 1;
        }
        if (borrow) {
            uint64_t carry = 0;
            for (int i = 0; i < N; i++) {
                const u128 t = (u128)d.l[i] + mod(i) + carry;
                d.l[i] = (uint64_t)t;
                carry = (uint64_t)(t >> 64);
            }
        }
        return d;
    }
    static Fe64 neg(const Fe64& a) { return a.is_zero() ? a : sub(zero(), a); }
    static Fe64 dbl(const Fe64& a) { return add(a, a); }

    // CIOS Montgomery product; the modulus leaves a spare top bit, so the running value stays below 2p in N+1 limbs
    static Fe64 mul(const Fe64& a, const Fe64& b) {
        uint64_t t[N + 2];
        for (int i = 0; i < N + 2; i++) t[i] = 0;
        constexpr uint64_t NI = ninv();
        for (int i = 0; i < N; i++) {
            u128 c = 0;
            for (int j = 0; j < N; j++) {
                c += (u128)a.l[j] * b.l[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[N];
            t[N] = (uint64_t)c;
            t[N + 1] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * NI;
            c = (u128)m * mod(0) + t[0];
            c >>= 64;
            for (int j = 1; j < N; j++) {
                c += (u128)m * mod(j) + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[N];
            t[N - 1] = (uint64_t)c;
            t[N] = t[N + 1] + (uint64_t)(c >> 64);
        }
        Fe64 r;
        for (int i = 0; i < N; i++) r.l[i] = t[i];
        return reduce_once(r);   // t[N] == 0 here: the value is below 2p < 2^(64N)
    }
    static Fe64 sqr(const Fe64& a) { return mul(a, a); }
    static Fe64 inv(const Fe64& a) { return from(Fe<P>::inv(a.to())); }
};
template <class P> inline Fe64<P> operator+(const Fe64<P>& a, const Fe64<P>& b) { return Fe64<P>::add(a, b); }
template <class P> inline Fe64<P> operator-(const Fe64<P>& a, const Fe64<P>& b) { return Fe64<P>::sub(a, b); }
template <class P> inline Fe64<P> operator*(const Fe64<P>& a, const Fe64<P>& b) { return Fe64<P>::mul(a, b); }
