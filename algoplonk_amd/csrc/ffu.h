// Base-field arithmetic on UNSATURATED limbs for the MSM point arithmetic (gfx950).
//
// Why: with saturated 32-bit limbs every 32x32 partial product needs a v_mad_u64_u32 (4.6 cycles per wave) AND a
// v_addc_co_u32 (4.0 cycles) to keep the carry out of the 64-bit column accumulator (tools/ubench/valu_rates.hip).
// With UL limbs of UB < 32 bits (BN254: 9 x 29, BLS12-381: 14 x 28) a whole column of 2*UL partial products fits a
// 64-bit accumulator, so the Montgomery product is a bare chain of v_mad_u64_u32 plus one 64-bit shift per column:
// ~35 % fewer issue cycles per product, and plain C++ (the same code runs on the host for the CPU-only tests).
//
// Values are CANONICAL (0 <= x < p, every limb < 2^UB) after every operation, so the point formulas and their
// special-case tests are unchanged.  (A weakly reduced variant - skip the conditional subtraction after products - was
// tried: subtraction then needs a second conditional correction because an operand may exceed p, which eats the gain.)
// The Montgomery radix is R' = 2^(UB*UL), not gnark's R = 2^(32N): FeU values never leave the MSM pipeline (windowed
// tables are internal data; results are converted back with to_fe()).
//
// Replaces (device side) gnark-crypto v0.20.1 ecc/<curve>/fp element arithmetic inside G1Affine.MultiExp
// [UPSTREAM, not vendored; reached from /root/reference/algoplonk.go:89 via kzg.Commit].
#pragma once
#include "ff.h"
#include "ffu_asm.h"

template <class P>
struct FeU {
    static constexpr int L = P::UL;
    static constexpr int B = P::UB;
    static constexpr uint32_t MASK = (1u << B) - 1u;
    static constexpr int N = P::N;  // saturated 32-bit words of the packed form
    uint32_t l[L];

    APK_HD static FeU zero() {
        FeU r;
#pragma unroll
        for (int i = 0; i < L; i++) r.l[i] = 0;
        return r;
    }
    APK_HD static FeU one() {
        FeU r;
#pragma unroll
        for (int i = 0; i < L; i++) r.l[i] = P::uone(i);
        return r;
    }
    APK_HD bool is_zero() const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < L; i++) acc |= l[i];
        return acc == 0;
    }
    APK_HD bool operator==(const FeU& o) const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < L; i++) acc |= l[i] ^ o.l[i];
        return acc == 0;
    }
    APK_HD bool operator!=(const FeU& o) const { return !(*this == o); }

    // x in [0, 2p) with normalised limbs -> [0, p)
    APK_HD static FeU reduce_once(const FeU& a) {
        FeU d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            uint32_t t = a.l[i] - P::umod(i) - borrow;  // wraps when negative: bit 31 set (operands < 2^30)
            borrow = t >> 31;
            d.l[i] = t & MASK;
        }
        FeU r;
#pragma unroll
        for (int i = 0; i < L; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
        return r;
    }

    APK_HD static FeU add(const FeU& a, const FeU& b) {
        FeU s;
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            uint32_t t = a.l[i] + b.l[i] + carry;
            carry = t >> B;
            s.l[i] = t & MASK;
        }
        // a + b < 2p < 2^(B*L): the last carry is zero (top limb keeps its spare bits)
        return reduce_once(s);
    }

    APK_HD static FeU sub(const FeU& a, const FeU& b) {
        FeU d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            uint32_t t = a.l[i] - b.l[i] - borrow;
            borrow = t >> 31;
            d.l[i] = t & MASK;
        }
        // negative: add p back (the wrapped limbs represent a - b + 2^(B*L); adding p and dropping the top carry fixes it)
        const uint32_t m = 0u - borrow;
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            uint32_t t = d.l[i] + (P::umod(i) & m) + carry;
            carry = t >> B;
            d.l[i] = t & MASK;
        }
        return d;
    }

    APK_HD static FeU neg(const FeU& a) {
        if (a.is_zero()) return a;
        FeU d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            uint32_t t = P::umod(i) - a.l[i] - borrow;
            borrow = t >> 31;
            d.l[i] = t & MASK;
        }
        return d;
    }

    APK_HD static FeU dbl(const FeU& a) { return add(a, a); }

    // Montgomery product a*b/R' mod p, product scanning.  Column k collects a_i*b_(k-i) and m_i*p_(k-i): at most 2L
    // products below 2^(2B) plus a carry below 2^(64-B) - no overflow, no carry flag.  Inputs canonical; the result
    // is below p + p*p/R' < 2p and is brought to [0, p) by one conditional subtraction (left out by mul_nr, see the
    // lazy forms below).
    APK_HD static FeU mul_nr(const FeU& a, const FeU& b) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (UMulAsm<P>::available) { FeU r; UMulAsm<P>::mul(r.l, a.l, b.l); return r; }   // one strict mad chain (ffu_asm.h)
#endif
        uint32_t m[L];
        FeU r;
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < L; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::umod(k - i);
            m[k] = ((uint32_t)acc * P::UINV) & MASK;
            acc += (uint64_t)m[k] * P::umod(0);
            acc >>= B;
        }
#pragma unroll
        for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
            for (int i = k - L + 1; i < L; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * P::umod(k - i);
            r.l[k - L] = (uint32_t)acc & MASK;
            acc >>= B;
        }
        r.l[L - 1] = (uint32_t)acc;
        return r;
    }
    APK_HD static FeU mul(const FeU& a, const FeU& b) { return reduce_once(mul_nr(a, b)); }

    // a*a/R': the off-diagonal partial products are taken once against 2a (L*(L+1)/2 mads instead of L*L)
    APK_HD static FeU sqr_nr(const FeU& a) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (UMulAsm<P>::available) { FeU r; UMulAsm<P>::sqr(r.l, a.l); return r; }
#endif
        uint32_t m[L], d[L];
        FeU r;
#pragma unroll
        for (int i = 0; i < L; i++) d[i] = a.l[i] << 1;   // < 2^(B+1): products stay below 2^(2B+1), column sums below 2^63
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < L; k++) {
#pragma unroll
            for (int i = 0; 2 * i < k; i++) acc += (uint64_t)d[i] * a.l[k - i];
            if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::umod(k - i);
            m[k] = ((uint32_t)acc * P::UINV) & MASK;
            acc += (uint64_t)m[k] * P::umod(0);
            acc >>= B;
        }
#pragma unroll
        for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
            for (int i = k - L + 1; 2 * i < k; i++) acc += (uint64_t)d[i] * a.l[k - i];
            if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
            for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * P::umod(k - i);
            r.l[k - L] = (uint32_t)acc & MASK;
            acc >>= B;
        }
        r.l[L - 1] = (uint32_t)acc;
        return r;
    }
    APK_HD static FeU sqr(const FeU& a) { return reduce_once(sqr_nr(a)); }

    // ---- lazy forms: the bucket-accumulation inner loop (ec.h XYZZ::madd_lazy) --------------------------------------
    // R'/p is large (BN254: 169, BLS12-381: 2520), so a Montgomery product of operands below A*p and C*p comes out
    // below p*(1 + A*C*p/R') < 2p without the conditional subtraction as long as A*C <= HEADROOM = R'/p, and a difference can
    // be kept positive by adding a multiple of p instead of testing for a borrow.  Values are then only congruent mod p
    // (still with every limb but the top one below 2^B); canon<K>() brings them back.  Column sums stay below 2^64:
    // 2L products of limbs below 2^B plus L of m*p, or L products with one operand's limbs below 2^(B+1).
    static constexpr uint32_t HEADROOM = (uint32_t)((1ull << B) / ((uint64_t)P::umod(L - 1) + 1));   // <= R'/p; users assert what they need
    static_assert((uint64_t)(3 * L + 1) <= (1ull << (64 - 2 * B)), "column sums of mul2_nr must fit 64 bits");

    // limb i of K*p, normalised (the top limb keeps the excess)
    template <uint32_t K>
    APK_HD static constexpr uint32_t kp(int i) {
        uint64_t carry = 0;
        uint32_t limb = 0;
        for (int j = 0; j <= i; j++) {
            uint64_t t = (uint64_t)K * P::umod(j) + carry;
            limb = j == L - 1 ? (uint32_t)t : (uint32_t)(t & MASK);
            carry = t >> B;
        }
        return limb;
    }

    // a - b + K*p through a signed carry sweep: no comparison, limbs come out normalised.  Needs b <= K*p as values and
    // limbs below 2^30; the result is below a + K*p.
    template <uint32_t K>
    APK_HD static FeU sub_k(const FeU& a, const FeU& b) {
        FeU r;
        int32_t carry = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            int32_t t = (int32_t)(a.l[i] - b.l[i] + kp<K>(i)) + carry;
            carry = t >> B;   // arithmetic shift: a negative limb borrows from the next one
            r.l[i] = i == L - 1 ? (uint32_t)t : ((uint32_t)t & MASK);
        }
        return r;
    }
    // K*p - a
    template <uint32_t K>
    APK_HD static FeU neg_k(const FeU& a) {
        FeU r;
        int32_t carry = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            int32_t t = (int32_t)(kp<K>(i) - a.l[i]) + carry;
            carry = t >> B;
            r.l[i] = i == L - 1 ? (uint32_t)t : ((uint32_t)t & MASK);
        }
        return r;
    }
    // a + b and 3a with an unsigned carry sweep (limbs normalised again; values just add up)
    APK_HD static FeU add_n(const FeU& a, const FeU& b) {
        FeU r;
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            uint32_t t = a.l[i] + b.l[i] + carry;
            carry = t >> B;
            r.l[i] = i == L - 1 ? t : (t & MASK);
        }
        return r;
    }
    APK_HD static FeU triple_n(const FeU& a) {
        FeU r;
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            uint32_t t = (a.l[i] << 1) + a.l[i] + carry;
            carry = t >> B;
            r.l[i] = i == L - 1 ? t : (t & MASK);
        }
        return r;
    }
    // a - b - 2c + K*p, same conventions (needs b + 2c <= K*p)
    template <uint32_t K>
    APK_HD static FeU sub2_k(const FeU& a, const FeU& b, const FeU& c) {
        FeU r;
        int32_t carry = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            int32_t t = (int32_t)(a.l[i] - ((c.l[i] << 1) + b.l[i]) + kp<K>(i)) + carry;
            carry = t >> B;
            r.l[i] = i == L - 1 ? (uint32_t)t : ((uint32_t)t & MASK);
        }
        return r;
    }

    // (a*b + c*d)/R' with ONE Montgomery reduction; below p + (a*b + c*d)/R'
    APK_HD static FeU mul2_nr(const FeU& a, const FeU& b, const FeU& c, const FeU& d) {
        uint32_t m[L];
        FeU r;
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < L; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) {
                acc += (uint64_t)a.l[i] * b.l[k - i];
                acc += (uint64_t)c.l[i] * d.l[k - i];
            }
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::umod(k - i);
            m[k] = ((uint32_t)acc * P::UINV) & MASK;
            acc += (uint64_t)m[k] * P::umod(0);
            acc >>= B;
        }
#pragma unroll
        for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
            for (int i = k - L + 1; i < L; i++) {
                acc += (uint64_t)a.l[i] * b.l[k - i];
                acc += (uint64_t)c.l[i] * d.l[k - i];
            }
#pragma unroll
            for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * P::umod(k - i);
            r.l[k - L] = (uint32_t)acc & MASK;
            acc >>= B;
        }
        r.l[L - 1] = (uint32_t)acc;
        return r;
    }

    // value below 2K*p (K a power of two), limbs normalised -> [0, p)
    template <uint32_t K>
    APK_HD static FeU canon(const FeU& a) {
        FeU d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            uint32_t t = a.l[i] - kp<K>(i) - borrow;
            borrow = t >> 31;
            d.l[i] = i == L - 1 ? t : (t & MASK);
        }
        FeU r;
#pragma unroll
        for (int i = 0; i < L; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
        if constexpr (K > 1) return canon<K / 2>(r);
        else return r;
    }
    // for a value below 2p with normalised limbs: is it 0 mod p?
    APK_HD bool is_zero_mod_p() const {
        uint32_t z = 0, e = 0;
#pragma unroll
        for (int i = 0; i < L; i++) { z |= l[i]; e |= l[i] ^ P::umod(i); }
        return z == 0 || e == 0;
    }

    // ---- packed (saturated, N x 32-bit) <-> limbs.  The packed form is what sits in HBM (tables, results). ----
    APK_HD static FeU unpack(const uint32_t* w) {
        FeU r;
#pragma unroll
        for (int i = 0; i < L; i++) {
            const int bit = i * B, j = bit >> 5, s = bit & 31;
            uint32_t v = j < N ? (w[j] >> s) : 0u;
            if (s + B > 32 && j + 1 < N) v |= w[j + 1] << (32 - s);
            r.l[i] = v & MASK;
        }
        return r;
    }
    APK_HD void pack(uint32_t* w) const {
#pragma unroll
        for (int j = 0; j < N; j++) {
            uint32_t v = 0;
#pragma unroll
            for (int i = 0; i < L; i++) {
                const int lo = i * B - j * 32;  // position of limb i's bit 0 inside word j
                if (lo > -B && lo < 32) v |= lo >= 0 ? (l[i] << lo) : (l[i] >> (-lo));
            }
            w[j] = v;
        }
    }

    // gnark-Montgomery Fe (x*R, R = 2^(32N))  <->  FeU (x*R')
    APK_HD static FeU from_fe(const Fe<P>& x) {
        FeU c;
#pragma unroll
        for (int i = 0; i < L; i++) c.l[i] = P::uconv_in(i);
        return mul(unpack(x.l), c);
    }
    APK_HD Fe<P> to_fe() const {
        FeU c;
#pragma unroll
        for (int i = 0; i < L; i++) c.l[i] = P::uconv_out(i);
        FeU y = mul(*this, c);
        Fe<P> r;
        y.pack(r.l);
        return r;
    }
};

template <class P> APK_HD FeU<P> operator+(const FeU<P>& a, const FeU<P>& b) { return FeU<P>::add(a, b); }
template <class P> APK_HD FeU<P> operator-(const FeU<P>& a, const FeU<P>& b) { return FeU<P>::sub(a, b); }
template <class P> APK_HD FeU<P> operator*(const FeU<P>& a, const FeU<P>& b) { return FeU<P>::mul(a, b); }
