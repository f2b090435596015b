// HOST-side PLONK verifier: the library's mirror of gnark's `plonk.Verify(proof, vk, publicWitness)`, which the reference
// runs right after the prover inside `(*CompiledCircuit).Verify` (/root/reference/algoplonk.go:93) - so that the host-side
// mirror of that API never hands out a `VerifiedProof` it has not verified.  No GPU work: ~25 G1 scalar multiplications and
// one two-pair pairing check, all on the calling thread.
//
// What it checks is pinned by the reference's own verifier templates (the same statement gnark's Verify checks):
//   transcript gamma, beta, alpha, zeta            verifier/templateLogicSigBN254.go:131-140
//   PI(zeta) incl. the BSB22 hash_fr terms         :142-194
//   linearised polynomial: opening + commitment    :195-278
//   gamma', folded commitment / evaluation         :280-321
//   batching of the two openings, pairing check    :322-356      (BLS12-381 twin: templateLogicSigBLS12_381.go)
// The formulas are written from the identity in SURVEY.md App. E.  The oracle (oracle/plonk.py::verify, a transcription of
// the template) is an independent implementation; tests/test_verify_host.py holds the two against each other.
//
// Pairing: plain ate pairing a(Q, P) = f_{T,Q}(P)^((p^12-1)/r), T = |t - 1|, on the tower Fp2 = Fp[u]/(u^2+1),
// Fp6 = Fp2[v]/(v^3 - xi), Fp12 = Fp6[w]/(w^2 - v); affine Miller loop on the twist, no Frobenius maps; the final
// exponentiation is (p^6 - 1) by conjugate/inverse, then the plain power (p^6+1)/r.  Any non-degenerate bilinear pairing
// decides e(A, G2_0) e(B, G2_1) = 1; constants from tools/gen_pairing_params.py (checked numerically there).
#pragma once
#include <mutex>
#include <string.h>
#include <vector>

#include "ec.h"
#include "pairing_params.h"
#include "sha256.h"

namespace apk {

template <class FP, class PP>
struct Fp2 {
    using F = Fe<FP>;
    F c0, c1;
    static Fp2 zero() { return {F::zero(), F::zero()}; }
    static Fp2 one() { return {F::one(), F::zero()}; }
    static Fp2 from_fp(const F& a) { return {a, F::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
    static Fp2 add(const Fp2& a, const Fp2& b) { return {F::add(a.c0, b.c0), F::add(a.c1, b.c1)}; }
    static Fp2 sub(const Fp2& a, const Fp2& b) { return {F::sub(a.c0, b.c0), F::sub(a.c1, b.c1)}; }
    static Fp2 neg(const Fp2& a) { return {F::neg(a.c0), F::neg(a.c1)}; }
    static Fp2 dbl(const Fp2& a) { return add(a, a); }
    static Fp2 mul(const Fp2& a, const Fp2& b) {
        const F t0 = a.c0 * b.c0, t1 = a.c1 * b.c1;
        const F s = F::add(a.c0, a.c1) * F::add(b.c0, b.c1);
        return {F::sub(t0, t1), F::sub(F::sub(s, t0), t1)};
    }
    static Fp2 sqr(const Fp2& a) { return mul(a, a); }
    static Fp2 mul_fp(const Fp2& a, const F& k) { return {a.c0 * k, a.c1 * k}; }
    static Fp2 inv(const Fp2& a) {
        const F n = F::inv(F::add(F::sqr(a.c0), F::sqr(a.c1)));
        return {a.c0 * n, F::neg(a.c1 * n)};
    }
    // a * xi, xi = XI0 + u
    static Fp2 mul_xi(const Fp2& a) {
        F x0a0 = F::zero(), x0a1 = F::zero();
        for (uint32_t i = 0; i < PP::XI0; i++) { x0a0 = F::add(x0a0, a.c0); x0a1 = F::add(x0a1, a.c1); }
        return {F::sub(x0a0, a.c1), F::add(x0a1, a.c0)};
    }
};

template <class FP, class PP>
struct Fp6 {
    using F2 = Fp2<FP, PP>;
    F2 c0, c1, c2;
    static Fp6 zero() { return {F2::zero(), F2::zero(), F2::zero()}; }
    static Fp6 one() { return {F2::one(), F2::zero(), F2::zero()}; }
    static Fp6 add(const Fp6& a, const Fp6& b) { return {F2::add(a.c0, b.c0), F2::add(a.c1, b.c1), F2::add(a.c2, b.c2)}; }
    static Fp6 sub(const Fp6& a, const Fp6& b) { return {F2::sub(a.c0, b.c0), F2::sub(a.c1, b.c1), F2::sub(a.c2, b.c2)}; }
    static Fp6 neg(const Fp6& a) { return {F2::neg(a.c0), F2::neg(a.c1), F2::neg(a.c2)}; }
    static Fp6 mul(const Fp6& a, const Fp6& b) {
        const F2 t0 = F2::mul(a.c0, b.c0), t1 = F2::mul(a.c1, b.c1), t2 = F2::mul(a.c2, b.c2);
        const F2 m12 = F2::sub(F2::sub(F2::mul(F2::add(a.c1, a.c2), F2::add(b.c1, b.c2)), t1), t2);
        const F2 m01 = F2::sub(F2::sub(F2::mul(F2::add(a.c0, a.c1), F2::add(b.c0, b.c1)), t0), t1);
        const F2 m02 = F2::sub(F2::sub(F2::mul(F2::add(a.c0, a.c2), F2::add(b.c0, b.c2)), t0), t2);
        return {F2::add(t0, F2::mul_xi(m12)), F2::add(m01, F2::mul_xi(t2)), F2::add(m02, t1)};
    }
    // a * v
    static Fp6 mul_v(const Fp6& a) { return {F2::mul_xi(a.c2), a.c0, a.c1}; }
    static Fp6 inv(const Fp6& a) {
        const F2 A = F2::sub(F2::sqr(a.c0), F2::mul_xi(F2::mul(a.c1, a.c2)));
        const F2 B = F2::sub(F2::mul_xi(F2::sqr(a.c2)), F2::mul(a.c0, a.c1));
        const F2 Cc = F2::sub(F2::sqr(a.c1), F2::mul(a.c0, a.c2));
        const F2 Fd = F2::add(F2::mul(a.c0, A), F2::mul_xi(F2::add(F2::mul(a.c2, B), F2::mul(a.c1, Cc))));
        const F2 fi = F2::inv(Fd);
        return {F2::mul(A, fi), F2::mul(B, fi), F2::mul(Cc, fi)};
    }
};

template <class FP, class PP>
struct Fp12 {
    using F6 = Fp6<FP, PP>;
    using F2 = Fp2<FP, PP>;
    F6 a0, a1;
    static Fp12 one() { return {F6::one(), F6::zero()}; }
    bool is_one() const {
        return a0.c0 == F2::one() && a0.c1.is_zero() && a0.c2.is_zero() && a1.c0.is_zero() && a1.c1.is_zero() && a1.c2.is_zero();
    }
    static Fp12 mul(const Fp12& x, const Fp12& y) {
        const F6 t0 = F6::mul(x.a0, y.a0), t1 = F6::mul(x.a1, y.a1);
        const F6 c1 = F6::sub(F6::sub(F6::mul(F6::add(x.a0, x.a1), F6::add(y.a0, y.a1)), t0), t1);
        return {F6::add(t0, F6::mul_v(t1)), c1};
    }
    static Fp12 conj(const Fp12& x) { return {x.a0, F6::neg(x.a1)}; }   // x^(p^6)
    static Fp12 inv(const Fp12& x) {
        const F6 t = F6::inv(F6::sub(F6::mul(x.a0, x.a0), F6::mul_v(F6::mul(x.a1, x.a1))));
        return {F6::mul(x.a0, t), F6::neg(F6::mul(x.a1, t))};
    }
};

// affine point of the twist E'(Fp2): y^2 = x^3 + b'; gnark's in-memory G2Affine = X.A0 || X.A1 || Y.A0 || Y.A1 (Montgomery)
template <class FP, class PP>
struct G2Aff {
    using F2 = Fp2<FP, PP>;
    F2 x, y;
    bool inf;
    static F2 btwist() {
        Fe<FP> b0, b1;
        for (int i = 0; i < FP::N; i++) { b0.l[i] = PP::bt0(i); b1.l[i] = PP::bt1(i); }
        return {Fe<FP>::to_mont(b0), Fe<FP>::to_mont(b1)};
    }
    static G2Aff generator() {
        Fe<FP> a, b, c, d;
        for (int i = 0; i < FP::N; i++) { a.l[i] = PP::g2x0(i); b.l[i] = PP::g2x1(i); c.l[i] = PP::g2y0(i); d.l[i] = PP::g2y1(i); }
        return {{Fe<FP>::to_mont(a), Fe<FP>::to_mont(b)}, {Fe<FP>::to_mont(c), Fe<FP>::to_mont(d)}, false};
    }
    bool on_curve() const { return inf || F2::sqr(y) == F2::add(F2::mul(F2::sqr(x), x), btwist()); }
    // slope of the chord / tangent; *this != -o (callers guarantee it)
    static G2Aff add(const G2Aff& p, const G2Aff& q, F2* slope = nullptr) {
        if (p.inf) return q;
        if (q.inf) return p;
        F2 lam;
        if (p.x == q.x) {
            if (!(p.y == q.y) || p.y.is_zero()) return {F2::zero(), F2::zero(), true};
            const F2 xx = F2::sqr(p.x);
            lam = F2::mul(F2::add(F2::dbl(xx), xx), F2::inv(F2::dbl(p.y)));
        } else {
            lam = F2::mul(F2::sub(q.y, p.y), F2::inv(F2::sub(q.x, p.x)));
        }
        if (slope) *slope = lam;
        const F2 x3 = F2::sub(F2::sub(F2::sqr(lam), p.x), q.x);
        return {x3, F2::sub(F2::mul(lam, F2::sub(p.x, x3)), p.y), false};
    }
    template <class FR>
    static G2Aff mul(const G2Aff& p, const Fe<FR>& k_canonical) {
        G2Aff acc{F2::zero(), F2::zero(), true};
        for (int w = Fe<FR>::N - 1; w >= 0; w--)
            for (int b = 31; b >= 0; b--) {
                acc = add(acc, acc);
                if ((k_canonical.l[w] >> b) & 1u) acc = add(acc, p);
            }
        return acc;
    }
};

// f_{T,Q}(P) (vertical lines dropped: they die in the final exponentiation), multiplied into `f`
template <class FP, class PP>
void miller_loop(const Affine<FP>& P, const G2Aff<FP, PP>& Q, Fp12<FP, PP>& f_acc) {
    using F = Fe<FP>;
    using F2 = Fp2<FP, PP>;
    using F12 = Fp12<FP, PP>;
    if (P.is_inf() || Q.inf) return;
    auto line = [&](const F2& lam, const G2Aff<FP, PP>& T) {
        // D-type twist (x, y) -> (x w^2, y w^3):  l(P) = yP - lam xP w + (lam xT - yT) w^3
        // M-type twist (x, y) -> (x / w^2, y / w^3), scaled by w^3 (a factor from a proper subfield):
        //                                         l(P) = (lam xT - yT) - lam xP v + yP v w            (w^2 = v, w^3 = v w)
        const F2 c = F2::sub(F2::mul(lam, T.x), T.y), m = F2::neg(F2::mul_fp(lam, P.x)), yp = F2::from_fp(P.y);
        F12 l{{F2::zero(), F2::zero(), F2::zero()}, {F2::zero(), F2::zero(), F2::zero()}};
        if (PP::TWIST_M) { l.a0.c0 = c; l.a0.c1 = m; l.a1.c1 = yp; }
        else { l.a0.c0 = yp; l.a1.c0 = m; l.a1.c1 = c; }
        return l;
    };
    F12 f = F12::one();
    G2Aff<FP, PP> T = Q;
    for (int bit = PP::ATE_BITS - 2; bit >= 0; bit--) {
        F2 lam;
        const G2Aff<FP, PP> T2 = G2Aff<FP, PP>::add(T, T, &lam);
        f = F12::mul(F12::mul(f, f), line(lam, T));
        T = T2;
        if ((PP::ate(bit >> 5) >> (bit & 31)) & 1u) {
            const G2Aff<FP, PP> TQ = G2Aff<FP, PP>::add(T, Q, &lam);
            f = F12::mul(f, line(lam, T));
            T = TQ;
        }
    }
    (void)sizeof(F);
    f_acc = F12::mul(f_acc, f);
}

template <class FP, class PP>
bool final_exp_is_one(const Fp12<FP, PP>& f) {
    using F12 = Fp12<FP, PP>;
    const F12 g = F12::mul(F12::conj(f), F12::inv(f));     // f^(p^6 - 1)
    F12 acc = F12::one();
    for (int bit = PP::FEXP_BITS - 1; bit >= 0; bit--) {   // ^((p^6 + 1) / r)
        acc = F12::mul(acc, acc);
        if ((PP::fexp(bit >> 5) >> (bit & 31)) & 1u) acc = F12::mul(acc, g);
    }
    return acc.is_one();
}

// e(a0, q0) * e(a1, q1) == 1
template <class FP, class PP>
bool pairing_check2(const Affine<FP>& a0, const G2Aff<FP, PP>& q0, const Affine<FP>& a1, const G2Aff<FP, PP>& q1) {
    Fp12<FP, PP> f = Fp12<FP, PP>::one();
    miller_loop<FP, PP>(a0, q0, f);
    miller_loop<FP, PP>(a1, q1, f);
    return final_exp_is_one<FP, PP>(f);
}

// ---- G2 encodings ------------------------------------------------------------------------------------------------------
// gnark compressed G2 (vk.bin: SURVEY.md App. A.5): X.A1 || X.A0 big-endian, flag bits in the top of byte 0 exactly as for G1
// (BN254 2 bits: 10 smaller y, 11 larger, 01 infinity; BLS12-381 3 bits: 100 / 101 / 110).  "Larger" for Fp2 is gnark's
// LexicographicallyLargest: decided on A1 unless it is zero, then on A0.
template <class FP, class PP, int CURVE_ID>
int g2_decompress(const uint8_t* in, G2Aff<FP, PP>& out) {
    using F = Fe<FP>;
    using F2 = Fp2<FP, PP>;
    constexpr int N = FP::N, NB = 4 * N;
    auto load = [&](const uint8_t* p, bool mask, F& x) {
        for (int w = 0; w < N; w++) {
            const uint8_t* q = p + NB - 4 * (w + 1);
            x.l[w] = (uint32_t)q[0] << 24 | (uint32_t)q[1] << 16 | (uint32_t)q[2] << 8 | q[3];
        }
        if (mask) x.l[N - 1] &= CURVE_ID == 0 ? 0x3fffffffu : 0x1fffffffu;
        const F m = F::modulus();
        for (int w = N - 1; w >= 0; w--)
            if (x.l[w] != m.l[w]) return x.l[w] < m.l[w];
        return false;
    };
    const uint32_t top = in[0];
    bool inf, largest;
    if (CURVE_ID == 0) { const uint32_t f = top >> 6; if (f == 0) return APK_ERR_ARG; inf = f == 1; largest = f == 3; }
    else { const uint32_t f = top >> 5; if (!(f == 4 || f == 5 || f == 6)) return APK_ERR_ARG; inf = f == 6; largest = f == 5; }
    F x1, x0;
    if (!load(in, true, x1) || !load(in + NB, false, x0)) return APK_ERR_ARG;
    if (inf) { if (!x1.is_zero() || !x0.is_zero()) return APK_ERR_ARG; out = {F2::zero(), F2::zero(), true}; return APK_OK; }
    const F2 X{F::to_mont(x0), F::to_mont(x1)};
    const F2 rhs = F2::add(F2::mul(F2::sqr(X), X), G2Aff<FP, PP>::btwist());
    // sqrt in Fp2 through the norm: y = y0 + y1 u with y0^2 = (a0 + sqrt(a0^2 + a1^2)) / 2 (or the other sign), y1 = a1 / (2 y0)
    uint32_t e[N];
    for (int w = 0; w < N; w++) e[w] = FP::sqrt_exp(w);
    auto fsqrt = [&](const F& a, F& r) { r = F::pow(a, e, N); return F::sqr(r) == a; };
    F2 Y;
    if (rhs.c1.is_zero()) {
        F r;
        if (fsqrt(rhs.c0, r)) Y = {r, F::zero()};
        else if (fsqrt(F::neg(rhs.c0), r)) Y = {F::zero(), r};      // (r u)^2 = -r^2
        else return APK_ERR_ARG;
    } else {
        F s;
        if (!fsqrt(F::add(F::sqr(rhs.c0), F::sqr(rhs.c1)), s)) return APK_ERR_ARG;
        F two = F::add(F::one(), F::one());
        const F half = F::inv(two);
        F y0;
        if (!fsqrt(F::add(rhs.c0, s) * half, y0) && !fsqrt(F::sub(rhs.c0, s) * half, y0)) return APK_ERR_ARG;
        Y = {y0, rhs.c1 * F::inv(F::add(y0, y0))};
    }
    if (!(F2::sqr(Y) == rhs)) return APK_ERR_ARG;
    auto fp_large = [&](const F& a) {
        const F c = F::from_mont(a);
        for (int w = N - 1; w >= 0; w--) { const uint32_t h = FP::half(w); if (c.l[w] != h) return c.l[w] > h; }
        return false;
    };
    const bool y_large = Y.c1.is_zero() ? fp_large(Y.c0) : fp_large(Y.c1);
    if (y_large != largest) Y = F2::neg(Y);
    out = {X, Y, false};
    return APK_OK;
}

// ---- the verifier ------------------------------------------------------------------------------------------------------
template <class FRP, class FPP, class PP, int CURVE_ID>
struct HostVerifier {
    using Fr = Fe<FRP>;
    using Fp = Fe<FPP>;
    using Aff = Affine<FPP>;
    using Pt = XYZZ<FPP>;
    using G2 = G2Aff<FPP, PP>;
    static constexpr int FPB = FPP::N * 4;

    static Fr fr_from_be(const uint8_t* be) {     // any 256-bit value, reduced mod r (templateLogicSigBN254.go:137-140)
        Fr a;
        for (int i = 0; i < 8; i++) {
            const uint8_t* p = be + 32 - 4 * (i + 1);
            a.l[i] = (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3];
        }
        return Fr::to_mont(a);
    }
    template <class P>
    static void fe_to_be(const Fe<P>& m, uint8_t* be) {
        const Fe<P> c = Fe<P>::from_mont(m);
        constexpr int N = P::N;
        for (int i = 0; i < N; i++) {
            uint8_t* p = be + 4 * (N - 1 - i);
            p[0] = (uint8_t)(c.l[i] >> 24); p[1] = (uint8_t)(c.l[i] >> 16); p[2] = (uint8_t)(c.l[i] >> 8); p[3] = (uint8_t)c.l[i];
        }
    }
    static void g1_raw(const Aff& p, uint8_t* out) {
        if (p.is_inf()) { memset(out, 0, 2 * FPB); if (FPB == 48) out[0] = 0x40; return; }   // BN254: all zeros (see backend_impl.h g1_raw_bytes)
        fe_to_be<FPP>(p.x, out);
        fe_to_be<FPP>(p.y, out + FPB);
    }
    static Aff load_pt(const uint8_t* slot) { Aff p; memcpy(&p, slot, sizeof p); return p; }
    static Fr load_fr(const uint8_t* slot) { Fr a; memcpy(&a, slot, sizeof a); return a; }
    static bool g1_on_curve(const Aff& p) {
        if (p.is_inf()) return true;
        Fp b = Fp::zero();
        b.l[0] = FPP::CURVE_B;
        return Fp::sqr(p.y) == Fp::sqr(p.x) * p.x + Fp::to_mont(b);
    }
    static bool fr_canonical(const Fr& m) {   // in-memory Montgomery limbs must be below r
        const Fr q = Fr::modulus();
        for (int w = Fr::N - 1; w >= 0; w--)
            if (m.l[w] != q.l[w]) return m.l[w] < q.l[w];
        return false;
    }
    static Pt smul(const Aff& p, const Fr& k_mont) {
        const Fr k = Fr::from_mont(k_mont);
        Pt acc = Pt::inf();
        bool started = false;
        for (int w = Fr::N - 1; w >= 0; w--)
            for (int b = 31; b >= 0; b--) {
                if (started) acc = Pt::dbl(acc);
                if ((k.l[w] >> b) & 1u) { acc.madd(p); started = true; }
            }
        return acc;
    }
    struct Transcript {
        Sha256 h;
        explicit Transcript(const char* name) { h.update(name, strlen(name)); }
        void bytes(const uint8_t* p, size_t n) { h.update(p, n); }
        void point(const Aff& p) { uint8_t b[2 * FPB]; g1_raw(p, b); h.update(b, 2 * FPB); }
        void scalar(const Fr& s) { uint8_t b[32]; fe_to_be<FRP>(s, b); h.update(b, 32); }
        void done(uint8_t out[32]) { h.final(out); }
    };
    static Fr hash_fr(const Aff& p) {   // gnark fr.Hash(msg, "BSB22-Plonk", 1) as the verifier recomputes it (:386-397)
        static const uint8_t dst_prime[12] = {'B', 'S', 'B', '2', '2', '-', 'P', 'l', 'o', 'n', 'k', 0x0b};
        uint8_t raw[2 * FPB], b0[32], b1[32], b2[32], zeros[64] = {0}, x[32];
        g1_raw(p, raw);
        const uint8_t lib[3] = {0x00, 0x30, 0x00}, one = 1, two = 2;
        Sha256 h;
        h.update(zeros, 64); h.update(raw, 2 * FPB); h.update(lib, 3); h.update(dst_prime, 12); h.final(b0);
        h.reset(); h.update(b0, 32); h.update(&one, 1); h.update(dst_prime, 12); h.final(b1);
        for (int i = 0; i < 32; i++) x[i] = b0[i] ^ b1[i];
        h.reset(); h.update(x, 32); h.update(&two, 1); h.update(dst_prime, 12); h.final(b2);
        uint8_t lo[32] = {0};
        memcpy(lo + 16, b2, 16);
        Fr t = Fr::zero();
        t.l[4] = 1;
        return fr_from_be(b1) * Fr::to_mont(t) + fr_from_be(lo);
    }

    // [r]P == infinity.  BN254's G1 has cofactor 1; BLS12-381's has not (gnark's SetBytes rejects such points, the AVM's
    // pairing_check fails on them).
    static bool g1_in_subgroup(const Aff& p) {
        if (FPB != 48 || p.is_inf()) return true;
        Pt acc = Pt::inf();
        bool started = false;
        const Fr q = Fr::modulus();
        for (int w = Fr::N - 1; w >= 0; w--)
            for (int b = 31; b >= 0; b--) {
                if (started) acc = Pt::dbl(acc);
                if ((q.l[w] >> b) & 1u) { acc.madd(p); started = true; }
            }
        return acc.to_affine().is_inf();
    }
    static void put_fr(uint8_t* out, const Fr& v) { fe_to_be<FRP>(v, out); }
    static void put_pt(uint8_t* out, const Aff& p) {   // what the AVM's ec ops return: X || Y big-endian, all zero for infinity
        memset(out, 0, APK_G1_MAX_BYTES);
        if (!p.is_inf()) { fe_to_be<FPP>(p.x, out); fe_to_be<FPP>(p.y, out + FPB); }
    }

    static int verify(const apk_verifying_key* vk, const apk_proof* pr, const void* public_inputs, apk_verify_trace* tr) {
        if (tr) memset(tr, 0, sizeof *tr);
        if (vk->n < 2 || (vk->n & (vk->n - 1)) || vk->n > (1ull << 30)) { set_error("verifying key: n must be a power of two"); return APK_ERR_ARG; }
        const uint32_t k = vk->nb_commitments;
        if (k > APK_MAX_COMMITMENTS || pr->nb_commitments != k || pr->curve != (uint32_t)CURVE_ID) {
            set_error("proof does not match the verifying key (curve / number of commitments)");
            return APK_ERR_VERIFY;
        }
        const uint64_t n = vk->n;
        int log_n = 0;
        while ((1ull << log_n) < n) log_n++;
        // domain constants as gnark's fft.NewDomain derives them (= VK Generator / SizeInv / CosetShift, :57,:68)
        Fr root;
        for (int i = 0; i < Fr::N; i++) root.l[i] = FRP::root(i);
        Fr omega = Fr::to_mont(root);
        for (int i = 0; i < FRP::ADICITY - log_n; i++) omega = Fr::sqr(omega);
        Fr nf = Fr::zero();
        nf.l[0] = (uint32_t)n;
        const Fr n_inv = Fr::inv(Fr::to_mont(nf));
        Fr sh = Fr::zero();
        sh.l[0] = FRP::COSET_SHIFT;
        const Fr u = Fr::to_mont(sh);

        // proof and key material; everything the proof supplies is range / curve checked (templateLogicSigBN254.go:110-120)
        const Aff Ql = load_pt(vk->ql), Qr = load_pt(vk->qr), Qm = load_pt(vk->qm), Qo = load_pt(vk->qo), Qk = load_pt(vk->qk);
        const Aff S1 = load_pt(vk->s[0]), S2 = load_pt(vk->s[1]), S3 = load_pt(vk->s[2]);
        Aff Qcp[APK_MAX_COMMITMENTS], Bsb[APK_MAX_COMMITMENTS];
        const Aff L = load_pt(pr->lro[0]), R = load_pt(pr->lro[1]), O = load_pt(pr->lro[2]), Z = load_pt(pr->z);
        const Aff H1 = load_pt(pr->h[0]), H2 = load_pt(pr->h[1]), H3 = load_pt(pr->h[2]);
        const Aff Wz = load_pt(pr->batched_h), Wzw = load_pt(pr->zshift_h);
        std::vector<Aff> pts = {L, R, O, Z, H1, H2, H3, Wz, Wzw};
        for (uint32_t i = 0; i < k; i++) { Qcp[i] = load_pt(vk->qcp[i]); Bsb[i] = load_pt(pr->bsb22[i]); pts.push_back(Bsb[i]); }
        for (const Aff& p : pts) if (!g1_on_curve(p)) { set_error("proof point is not on the curve"); return APK_ERR_VERIFY; }
        for (const Aff& p : pts) if (!g1_in_subgroup(p)) { set_error("proof point is not in the prime-order subgroup"); return APK_ERR_VERIFY; }
        const Fr l_z = load_fr(pr->claimed_values[1]), r_z = load_fr(pr->claimed_values[2]), o_z = load_fr(pr->claimed_values[3]);
        const Fr s1_z = load_fr(pr->claimed_values[4]), s2_z = load_fr(pr->claimed_values[5]), zw_z = load_fr(pr->zshift_value);
        Fr qcp_z[APK_MAX_COMMITMENTS];
        std::vector<Fr> vals = {l_z, r_z, o_z, s1_z, s2_z, zw_z};
        for (uint32_t i = 0; i < k; i++) { qcp_z[i] = load_fr(pr->claimed_values[6 + i]); vals.push_back(qcp_z[i]); }
        const Fr* pub = reinterpret_cast<const Fr*>(public_inputs);
        for (uint32_t i = 0; i < vk->nb_public; i++) vals.push_back(pub[i]);
        for (const Fr& v : vals) if (!fr_canonical(v)) { set_error("scalar is not below the field modulus"); return APK_ERR_VERIFY; }

        // ---- Fiat-Shamir
        uint8_t gamma_raw[32], beta_raw[32], alpha_raw[32], zeta_raw[32];
        {
            Transcript t("gamma");
            t.point(S1); t.point(S2); t.point(S3); t.point(Ql); t.point(Qr); t.point(Qm); t.point(Qo); t.point(Qk);
            for (uint32_t i = 0; i < k; i++) t.point(Qcp[i]);
            for (uint32_t i = 0; i < vk->nb_public; i++) t.scalar(pub[i]);
            t.point(L); t.point(R); t.point(O);
            t.done(gamma_raw);
        }
        { Transcript t("beta"); t.bytes(gamma_raw, 32); t.done(beta_raw); }
        { Transcript t("alpha"); t.bytes(beta_raw, 32); for (uint32_t i = 0; i < k; i++) t.point(Bsb[i]); t.point(Z); t.done(alpha_raw); }
        { Transcript t("zeta"); t.bytes(alpha_raw, 32); t.point(H1); t.point(H2); t.point(H3); t.done(zeta_raw); }
        const Fr gamma = fr_from_be(gamma_raw), beta = fr_from_be(beta_raw), alpha = fr_from_be(alpha_raw), zeta = fr_from_be(zeta_raw);
        if (tr) { put_fr(tr->gamma, gamma); put_fr(tr->beta, beta); put_fr(tr->alpha, alpha); put_fr(tr->zeta, zeta); }

        // ---- PI(zeta) = sum pub_i L_i(zeta) + sum hash_fr([pi2_k]) L_{nbPub + cci_k}(zeta),  L_i(X) = w^i (X^n - 1) / (n (X - w^i))
        const Fr one = Fr::one();
        const Fr zn = Fr::pow_u64(zeta, n);
        const Fr zh = zn - one;                          // zeta^n - 1
        auto lagrange_at_zeta = [&](uint64_t i, bool& ok) {
            const Fr wi = Fr::pow_u64(omega, i);
            const Fr den = zeta - wi;
            if (den.is_zero()) { ok = false; return Fr::zero(); }
            return wi * zh * n_inv * Fr::inv(den);
        };
        bool ok = true;
        Fr pi = Fr::zero();
        for (uint32_t i = 0; i < vk->nb_public; i++) pi = pi + pub[i] * lagrange_at_zeta(i, ok);
        for (uint32_t i = 0; i < k; i++) pi = pi + hash_fr(Bsb[i]) * lagrange_at_zeta((uint64_t)vk->nb_public + vk->commitment_constraint_index[i], ok);
        const Fr lag0 = lagrange_at_zeta(0, ok);
        if (!ok) { set_error("zeta lies on the domain"); return APK_ERR_VERIFY; }   // probability ~ n / r

        // ---- opening of the linearised polynomial the verifier expects (App. E "lin(zeta)")
        const Fr alpha2 = alpha * alpha;
        const Fr perm_z = alpha * zw_z * (l_z + beta * s1_z + gamma) * (r_z + beta * s2_z + gamma) * (o_z + gamma);
        const Fr lin_z = Fr::neg(pi + perm_z - alpha2 * lag0);
        if (tr) { put_fr(tr->pi, pi); put_fr(tr->lin_at_zeta, lin_z); }

        // ---- [lin] (App. E "lin(X)")
        const Fr c_s3 = alpha * beta * zw_z * (l_z + beta * s1_z + gamma) * (r_z + beta * s2_z + gamma);
        const Fr bu = beta * u, bu2 = bu * u;
        const Fr c_z = alpha2 * lag0 - alpha * (l_z + beta * zeta + gamma) * (r_z + bu * zeta + gamma) * (o_z + bu2 * zeta + gamma);
        const Fr zn2 = Fr::pow_u64(zeta, n + 2);
        const Fr mzh = Fr::neg(zh);
        Pt lin = smul(Ql, l_z);
        lin.add(smul(Qr, r_z)); lin.add(smul(Qm, l_z * r_z)); lin.add(smul(Qo, o_z)); lin.madd(Qk);
        for (uint32_t i = 0; i < k; i++) lin.add(smul(Bsb[i], qcp_z[i]));
        lin.add(smul(S3, c_s3)); lin.add(smul(Z, c_z));
        lin.add(smul(H1, mzh)); lin.add(smul(H2, mzh * zn2)); lin.add(smul(H3, mzh * zn2 * zn2));
        const Aff lin_com = lin.to_affine();
        if (tr) put_pt(tr->lin_commitment, lin_com);

        // ---- gamma' and the folded opening at zeta (:280-321)
        uint8_t gk_raw[32];
        {
            Transcript t("gamma");
            t.scalar(zeta);
            t.point(lin_com); t.point(L); t.point(R); t.point(O); t.point(S1); t.point(S2);
            for (uint32_t i = 0; i < k; i++) t.point(Qcp[i]);
            t.scalar(lin_z); t.scalar(l_z); t.scalar(r_z); t.scalar(o_z); t.scalar(s1_z); t.scalar(s2_z);
            for (uint32_t i = 0; i < k; i++) t.scalar(qcp_z[i]);
            t.scalar(zw_z);
            t.done(gk_raw);
        }
        const Fr gk = fr_from_be(gk_raw);
        if (tr) put_fr(tr->gamma_kzg, gk);
        Pt F = Pt::from_affine(lin_com);
        Fr c = lin_z, g = gk;
        const Aff fold_pts[5] = {L, R, O, S1, S2};
        const Fr fold_vals[5] = {l_z, r_z, o_z, s1_z, s2_z};
        for (int i = 0; i < 5; i++) { F.add(smul(fold_pts[i], g)); c = c + g * fold_vals[i]; g = g * gk; }
        for (uint32_t i = 0; i < k; i++) { F.add(smul(Qcp[i], g)); c = c + g * qcp_z[i]; g = g * gk; }
        if (tr) { put_pt(tr->folded_digest, F.to_affine()); put_fr(tr->folded_claim, c); }

        // ---- batch the two openings with verifier-side randomness r' (any value unpredictable to the prover: a hash of
        // everything above; :322-345), then e(F - c G1 + zeta W_z + r' (Z - Zw G1 + w zeta W_zw), G2_0) e(-(W_z + r' W_zw), G2_1) = 1
        uint8_t rr_raw[32];
        {
            Transcript t("random");
            t.bytes(gk_raw, 32); t.point(F.to_affine()); t.point(Z); t.point(Wz); t.point(Wzw); t.scalar(c); t.scalar(zw_z);
            t.done(rr_raw);
        }
        const Fr rr = fr_from_be(rr_raw);
        const Aff G1 = load_pt(vk->g1);
        if (!g1_on_curve(G1) || G1.is_inf()) { set_error("verifying key: Kzg.G1 is not a curve point"); return APK_ERR_ARG; }
        Pt A = F;
        A.add(smul(Z, rr));
        Pt cg = smul(G1, c + rr * zw_z);
        cg.neg_inplace();
        A.add(cg);
        A.add(smul(Wz, zeta));
        A.add(smul(Wzw, rr * zeta * omega));
        Pt B = Pt::from_affine(Wz);
        B.add(smul(Wzw, rr));
        B.neg_inplace();
        G2 g2[2];
        for (int j = 0; j < 2; j++) {
            memcpy(&g2[j].x, vk->g2[j], sizeof(g2[j].x));
            memcpy(&g2[j].y, vk->g2[j] + 2 * FPB, sizeof(g2[j].y));
            g2[j].inf = g2[j].x.is_zero() && g2[j].y.is_zero();
        }
        // The key's own points are constants of the circuit: checked ONCE per key (on the curve / the twist, in the prime-order
        // subgroups - gnark rejects such points when it decodes a key; the twists have large cofactors and the ate Miller loop is
        // a pairing only on the order-r subgroup), remembered by value.  Two G2 scalar multiplications by r per call were most of a
        // small verification's time.
        {
            std::vector<uint8_t> blob;
            auto put = [&](const void* p, size_t nb) { const uint8_t* b = (const uint8_t*)p; blob.insert(blob.end(), b, b + nb); };
            const int cid = CURVE_ID;
            put(&cid, sizeof cid);
            for (const Aff* p : {&Ql, &Qr, &Qm, &Qo, &Qk, &S1, &S2, &S3, &G1}) put(p, sizeof(Aff));
            for (uint32_t i = 0; i < k; i++) put(&Qcp[i], sizeof(Aff));
            put(vk->g2[0], 4 * FPB); put(vk->g2[1], 4 * FPB);
            static std::mutex mu;
            static std::vector<std::vector<uint8_t>> seen;
            bool known = false;
            { std::lock_guard<std::mutex> lk(mu); for (const auto& b : seen) if (b == blob) { known = true; break; } }
            if (!known) {
                std::vector<Aff> kp = {Ql, Qr, Qm, Qo, Qk, S1, S2, S3, G1};
                for (uint32_t i = 0; i < k; i++) kp.push_back(Qcp[i]);
                for (const Aff& p : kp) if (!g1_on_curve(p)) { set_error("verifying key: a G1 point is not on the curve"); return APK_ERR_ARG; }
                for (const Aff& p : kp) if (!g1_in_subgroup(p)) { set_error("verifying key: a G1 point is not in the prime-order subgroup"); return APK_ERR_ARG; }
                for (int j = 0; j < 2; j++) {
                    if (g2[j].inf || !g2[j].on_curve()) { set_error("verifying key: Kzg.G2[%d] is not a point of the twist", j); return APK_ERR_ARG; }
                    if (!G2::template mul<FRP>(g2[j], Fr::modulus()).inf) { set_error("verifying key: Kzg.G2[%d] is not in the prime-order subgroup", j); return APK_ERR_ARG; }
                }
                std::lock_guard<std::mutex> lk(mu);
                if (seen.size() >= 16) seen.erase(seen.begin());
                seen.push_back(std::move(blob));
            }
        }
        if (!pairing_check2<FPP, PP>(A.to_affine(), g2[0], B.to_affine(), g2[1])) {
            set_error("plonk verification failed: pairing check");
            return APK_ERR_VERIFY;
        }
        return APK_OK;
    }
};

}  // namespace apk
