// G1 arithmetic for y^2 = x^3 + b (a = 0; BN254 b=3, BLS12-381 b=4 - the formulas never touch b) in
// extended Jacobian ("XYZZ") coordinates: x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2, infinity <=> ZZ == 0.
// Mixed addition costs 8M+2S, full addition 12M+2S, doubling 6M+3S; no inversions until the final
// affine conversion.  Affine points use gnark's in-memory form {X, Y} in Montgomery limbs with
// (0, 0) as infinity, so SRS buffers cross the C-ABI untouched.
//
// Replaces (device side) gnark-crypto v0.20.1 ecc/<curve>/g1.go point arithmetic that
// G1Affine.MultiExp runs on [UPSTREAM; reached from /root/reference/algoplonk.go:89 via kzg.Commit].
#pragma once
#include "ff.h"
#include "ffu.h"

// FT = field element class: Fe<FP> (saturated, gnark's Montgomery radix - what crosses the C-ABI) or FeU<FP>
// (unsaturated limbs, radix R' - the MSM pipeline's internal form, ffu.h).
template <class FP, class FT = Fe<FP>>
struct Affine {
    FT x, y;
    APK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    APK_HD static Affine inf() { return Affine{FT::zero(), FT::zero()}; }
};

// packed R'-domain words (a table record in HBM) -> limbs; no arithmetic
template <class FP>
APK_HD Affine<FP, FeU<FP>> unpack_affine(const Affine<FP>& rec) {
    return Affine<FP, FeU<FP>>{FeU<FP>::unpack(rec.x.l), FeU<FP>::unpack(rec.y.l)};
}
// gnark-layout affine point -> packed R'-domain record (two domain-conversion products)
template <class FP>
APK_HD Affine<FP> to_table_record(const Affine<FP>& p) {
    Affine<FP> rec;
    FeU<FP>::from_fe(p.x).pack(rec.x.l);
    FeU<FP>::from_fe(p.y).pack(rec.y.l);
    return rec;
}

template <class FP, class FT = Fe<FP>>
struct XYZZ {
    using F = FT;
    using Aff = Affine<FP, FT>;
    F X, Y, ZZ, ZZZ;

    APK_HD bool is_inf() const { return ZZ.is_zero(); }
    APK_HD static XYZZ inf() { return XYZZ{F::one(), F::one(), F::zero(), F::zero()}; }
    APK_HD static XYZZ from_affine(const Aff& p) {
        if (p.is_inf()) return inf();
        return XYZZ{p.x, p.y, F::one(), F::one()};
    }
    APK_HD void neg_inplace() { Y = F::neg(Y); }

    // 2 * (affine p)
    APK_HD static XYZZ dbl_affine(const Aff& p) {
        if (p.is_inf() || p.y.is_zero()) return inf();
        F U = F::dbl(p.y);
        F V = F::sqr(U);
        F W = U * V;
        F S = p.x * V;
        F xx = F::sqr(p.x);
        F M = F::add(F::dbl(xx), xx);
        XYZZ r;
        r.X = F::sub(F::sub(F::sqr(M), S), S);
        r.Y = F::sub(M * F::sub(S, r.X), W * p.y);
        r.ZZ = V;
        r.ZZZ = W;
        return r;
    }

    APK_HD static XYZZ dbl(const XYZZ& p) {
        if (p.is_inf() || p.Y.is_zero()) return inf();
        F U = F::dbl(p.Y);
        F V = F::sqr(U);
        F W = U * V;
        F S = p.X * V;
        F xx = F::sqr(p.X);
        F M = F::add(F::dbl(xx), xx);
        XYZZ r;
        r.X = F::sub(F::sub(F::sqr(M), S), S);
        r.Y = F::sub(M * F::sub(S, r.X), W * p.Y);
        r.ZZ = V * p.ZZ;
        r.ZZZ = W * p.ZZZ;
        return r;
    }

    // this += q (affine).  `negate` flips q's sign (signed-digit buckets).
    APK_HD void madd(const Aff& q_in, bool negate = false) {
        if (q_in.is_inf()) return;
        Aff q = q_in;
        if (negate) q.y = F::neg(q.y);
        if (is_inf()) {
            X = q.x; Y = q.y; ZZ = F::one(); ZZZ = F::one();
            return;
        }
        F U2 = q.x * ZZ;
        F S2 = q.y * ZZZ;
        F Pd = F::sub(U2, X);
        F R = F::sub(S2, Y);
        if (Pd.is_zero()) {
            if (R.is_zero()) { *this = dbl_affine(q); return; }
            *this = inf();
            return;
        }
        F PP = F::sqr(Pd);
        F PPP = Pd * PP;
        F Q = X * PP;
        F X3 = F::sub(F::sub(F::sub(F::sqr(R), PPP), Q), Q);
        F Y3 = F::sub(R * F::sub(Q, X3), Y * PPP);
        X = X3;
        Y = Y3;
        ZZ = ZZ * PP;
        ZZZ = ZZZ * PPP;
    }

    // ---- lazy forms used by the MSM kernels (FT = FeU only; ffu.h "lazy forms") -------------------------------------
    // Same formulas as madd / add / dbl, but no conditional subtraction anywhere: differences add a multiple of p,
    // products skip the final reduction, and Y3 = A*(B - X3) - C*D is taken as -(A*(X3 - B) + C*D) with ONE Montgomery
    // reduction for both products.  Points are then held in the LAZY CLASS - limbs normalised, values only congruent:
    //     X < 5.1p,  Y <= 4p (2p from the one-lane forms),  ZZ, ZZZ < 1.4p,  infinity <=> ZZ is exactly zero
    // which every lazy operation maps into itself (R'/p >= 160: a product of operands below a*p and b*p is below
    // (1 + a*b/160)*p; the bounds of the intermediates are noted per line).  to_fe_point() accepts the class as it is, so
    // the affine result is bit for bit the one the canonical formulas give.

    // this += +-q for a canonical affine q.  Inside the accumulation loop the negation of Y3 is not paid for: each call
    // flips the sign of the point the state stands for (`flipped`) and the next call adds -q instead: -(A) + -(q) = -(A+q).
    // lazy_fix_sign() ends the loop.
    // `unit_z` (in/out): the state is a plain affine point (ZZ = ZZZ = 1, as after the first point of a bucket slice), which
    // saves the four products by ZZ / ZZZ on the second point.
    APK_HD void madd_lazy(const Aff& q, bool negate, bool& flipped, bool& unit_z) {
        static_assert(F::HEADROOM >= 160, "the bounds of the lazy class need R'/p >= 160");
        if (q.is_inf()) return;
        const bool ng = negate != flipped;
        if (is_inf()) {
            X = q.x; Y = negate ? F::neg(q.y) : q.y; ZZ = F::one(); ZZZ = F::one();
            flipped = false;
            unit_z = true;
            return;
        }
        const F qy = ng ? F::template neg_k<1>(q.y) : q.y;
        const F U2 = unit_z ? q.x : F::mul_nr(q.x, ZZ);        // < 1.01
        const F S2 = unit_z ? qy : F::mul_nr(qy, ZZZ);         // < 1.01
        const F Pd = F::template sub_k<6>(U2, X);              // < 7.1
        const F R = F::template sub_k<2>(S2, Y);               // < 3.1
        const F PP = F::sqr_nr(Pd);                            // < 1.4
        if (PP.l[0] == 0u || PP.l[0] == FP::umod(0)) {         // cheap filter; P = 0 mod p <=> PP in {0, p}
            if (PP.is_zero_mod_p()) {
                if (F::template canon<2>(R).is_zero()) {
                    *this = dbl_affine(Aff{q.x, ng ? F::neg(q.y) : q.y});   // state == q: 2q in the same sign frame
                } else {
                    *this = inf();
                    flipped = false;
                }
                unit_z = false;
                return;
            }
        }
        const F PPP = F::mul_nr(Pd, PP);                       // < 1.1
        const F Q = F::mul_nr(X, PP);                          // < 1.1
        const F X3 = F::template sub2_k<4>(F::sqr_nr(R), PPP, Q);   // < 1.1 + 4
        const F T = F::template sub_k<2>(X3, Q);               // < 7.1
        Y = F::mul2_nr(R, T, Y, PPP);                          // = -Y3, < 1.2
        X = X3;
        ZZ = unit_z ? PP : F::mul_nr(ZZ, PP);
        ZZZ = unit_z ? PPP : F::mul_nr(ZZZ, PPP);
        flipped = !flipped;
        unit_z = false;
    }
    APK_HD void lazy_fix_sign(bool flipped) {
        if (flipped) Y = F::template neg_k<2>(Y);
    }
    APK_HD void lazy_neg() { Y = F::template neg_k<4>(Y); }
    // lazy class -> canonical limbs (tests; to_fe_point does not need it)
    APK_HD void canonicalize() {
        X = F::template canon<4>(X);
        Y = F::template canon<4>(Y);
        ZZ = F::template canon<1>(ZZ);
        ZZZ = F::template canon<1>(ZZZ);
    }

    // cheap filter for Y in {0, p, 2p, 3p, 4p}
    APK_HD static bool lazy_y_maybe_zero(const F& y) {
        const uint32_t l0 = y.l[0];
        return l0 == 0u || l0 == FP::umod(0) || l0 == F::template kp<2>(0) || l0 == F::template kp<3>(0) || l0 == F::template kp<4>(0);
    }
    // 2p for p in the lazy class
    APK_HD static XYZZ dbl_lazy(const XYZZ& p) {
        static_assert(F::HEADROOM >= 160, "the bounds of the lazy class need R'/p >= 160");
        if (p.is_inf()) return inf();
        if (lazy_y_maybe_zero(p.Y)) {                          // Y = 0 mod p: a 2-torsion point
            if (F::template canon<4>(p.Y).is_zero()) return inf();
        }
        const F U = F::add_n(p.Y, p.Y);                        // <= 4
        const F V = F::sqr_nr(U);                              // < 1.1
        const F W = F::mul_nr(U, V);                           // < 1.1
        const F S = F::mul_nr(p.X, V);                         // < 1.1
        const F M = F::triple_n(F::sqr_nr(p.X));               // < 3 * 1.17
        XYZZ r;
        r.X = F::template sub2_k<4>(F::sqr_nr(M), F::zero(), S);   // < 1.1 + 4
        const F T = F::template sub_k<2>(r.X, S);              // < 7.1
        r.Y = F::template neg_k<2>(F::mul2_nr(M, T, W, p.Y));  // M*(S - X3) - W*Y
        r.ZZ = F::mul_nr(V, p.ZZ);
        r.ZZZ = F::mul_nr(W, p.ZZZ);
        return r;
    }

    // this += q, both in the lazy class
    APK_HD void add_lazy(const XYZZ& q) {
        static_assert(F::HEADROOM >= 160, "the bounds of the lazy class need R'/p >= 160");
        if (q.is_inf()) return;
        if (is_inf()) { *this = q; return; }
        const F U1 = F::mul_nr(X, q.ZZ);                       // < 1.1
        const F U2 = F::mul_nr(q.X, ZZ);
        const F S1 = F::mul_nr(Y, q.ZZZ);                      // < 1.1
        const F S2 = F::mul_nr(q.Y, ZZZ);
        const F Pd = F::template sub_k<2>(U2, U1);             // < 3.1
        const F R = F::template sub_k<2>(S2, S1);              // < 3.1
        const F PP = F::sqr_nr(Pd);                            // < 1.1
        if (PP.l[0] == 0u || PP.l[0] == FP::umod(0)) {
            if (PP.is_zero_mod_p()) {
                if (F::template canon<2>(R).is_zero()) *this = dbl_lazy(q);
                else *this = inf();
                return;
            }
        }
        const F PPP = F::mul_nr(Pd, PP);                       // < 1.1
        const F Q = F::mul_nr(U1, PP);                         // < 1.1
        const F X3 = F::template sub2_k<4>(F::sqr_nr(R), PPP, Q);   // < 1.1 + 4
        const F T = F::template sub_k<2>(X3, Q);               // < 7.1
        Y = F::template neg_k<2>(F::mul2_nr(R, T, S1, PPP));   // R*(Q - X3) - S1*PPP
        X = X3;
        ZZ = F::mul_nr(F::mul_nr(ZZ, q.ZZ), PP);
        ZZZ = F::mul_nr(F::mul_nr(ZZZ, q.ZZZ), PPP);
    }

#if defined(__HIPCC__)
    // ---- FOUR LANES PER POINT OPERATION (the latency-bound reduction kernels after the bucket accumulation) ----------
    // The quad's lanes (4k .. 4k+3) hold identical copies of the operands; each stage is ONE field product per lane on
    // lane-dependent operands (branch-free masks), the results are broadcast inside the quad with DPP quad_perm moves, the
    // cheap sums are recomputed by every lane.  A full addition is 4 product stages instead of 12M + 2S on one lane, a
    // doubling 3 stages instead of 6M + 3S.  Same lazy class, with Y3 = -(A + B) from two separately reduced products
    // (<= 4p); all four lanes end with the same result.  Must be called by all four lanes of a quad together.
    template <int SRC>
    __device__ __forceinline__ static F quad_bcast(const F& v) {
#if defined(__HIP_DEVICE_COMPILE__)
        F r;
#pragma unroll
        for (int i = 0; i < F::L; i++)
            r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], SRC | (SRC << 2) | (SRC << 4) | (SRC << 6), 0xf, 0xf, true);
        return r;
#else
        return v;
#endif
    }
    __device__ __forceinline__ static F quad_sel(int q, const F& a0, const F& a1, const F& a2, const F& a3) {
        // masks, not ternaries: hipcc turns a chain of ?: on values that live in registers into divergent branches
        const uint32_t m0 = 0u - (uint32_t)(q == 0), m1 = 0u - (uint32_t)(q == 1), m2 = 0u - (uint32_t)(q == 2), m3 = 0u - (uint32_t)(q == 3);
        F r;
#pragma unroll
        for (int i = 0; i < F::L; i++) r.l[i] = (a0.l[i] & m0) | (a1.l[i] & m1) | (a2.l[i] & m2) | (a3.l[i] & m3);
        return r;
    }
    // 2p, p not at infinity (the caller checks; a point with Y = 0 doubles to infinity)
    __device__ __forceinline__ static XYZZ dbl_quad_general(const XYZZ& p, int q) {
        static_assert(F::HEADROOM >= 160, "the bounds of the lazy class need R'/p >= 160");
        const F U = F::add_n(p.Y, p.Y);                                   // <= 8
        const F s1 = quad_sel(q, U, p.X, U, p.X);
        const F m1 = F::mul_nr(s1, s1);                                   // lane 0: V = U^2, lane 1: X^2
        const F V = quad_bcast<0>(m1);
        const F M = F::triple_n(quad_bcast<1>(m1));                       // < 3.6
        const F m2 = F::mul_nr(quad_sel(q, U, p.X, V, M), quad_sel(q, V, V, p.ZZ, M));
        const F W = quad_bcast<0>(m2), S = quad_bcast<1>(m2), MM = quad_bcast<3>(m2);   // U V, X V, M^2
        XYZZ r;
        r.ZZ = quad_bcast<2>(m2);                                         // V ZZ
        r.X = F::template sub2_k<4>(MM, F::zero(), S);
        const F T = F::template sub_k<2>(r.X, S);
        const F m3 = F::mul_nr(quad_sel(q, M, W, W, M), quad_sel(q, T, p.Y, p.ZZZ, T));
        r.Y = F::template neg_k<4>(F::add_n(quad_bcast<0>(m3), quad_bcast<1>(m3)));   // M (S - X3) - W Y
        r.ZZZ = quad_bcast<2>(m3);                                        // W ZZZ
        const bool y0 = lazy_y_maybe_zero(p.Y) && F::template canon<4>(p.Y).is_zero();
        if (y0) r.ZZ = F::zero();                                         // 2-torsion: infinity (ZZ exactly zero)
        return r;
    }
    // general case only: neither operand at infinity, operands not equal / opposite (the caller checks `ok` = quad-uniform)
    __device__ __forceinline__ void add_quad_general(const XYZZ& o, int q, bool& degenerate) {
        const F m1 = F::mul_nr(quad_sel(q, X, o.X, Y, o.Y), quad_sel(q, o.ZZ, ZZ, o.ZZZ, ZZZ));
        const F U1 = quad_bcast<0>(m1), U2 = quad_bcast<1>(m1), S1 = quad_bcast<2>(m1), S2 = quad_bcast<3>(m1);
        const F Pd = F::template sub_k<2>(U2, U1);
        const F R = F::template sub_k<2>(S2, S1);
        const F m2 = F::mul_nr(quad_sel(q, Pd, ZZ, R, ZZZ), quad_sel(q, Pd, o.ZZ, R, o.ZZZ));
        const F PP = quad_bcast<0>(m2), ZZ12 = quad_bcast<1>(m2), RR = quad_bcast<2>(m2), ZZZ12 = quad_bcast<3>(m2);
        degenerate = (PP.l[0] == 0u || PP.l[0] == FP::umod(0)) && PP.is_zero_mod_p();
        const F m3 = F::mul_nr(quad_sel(q, Pd, U1, ZZ12, Pd), PP);
        const F PPP = quad_bcast<0>(m3), Q = quad_bcast<1>(m3);
        const F X3 = F::template sub2_k<4>(RR, PPP, Q);
        const F T = F::template sub_k<2>(X3, Q);
        const F m4 = F::mul_nr(quad_sel(q, R, S1, ZZZ12, R), quad_sel(q, T, PPP, PPP, T));
        const F Y3 = F::template neg_k<4>(F::add_n(quad_bcast<0>(m4), quad_bcast<1>(m4)));
        const F ZZ3 = quad_bcast<2>(m3), ZZZ3 = quad_bcast<2>(m4);
        if (!degenerate) { X = X3; Y = Y3; ZZ = ZZ3; ZZZ = ZZZ3; }   // a degenerate pair is left untouched for the one-lane form
    }
#endif

    // this += q
    APK_HD void add(const XYZZ& q) {
        if (q.is_inf()) return;
        if (is_inf()) { *this = q; return; }
        F U1 = X * q.ZZ;
        F U2 = q.X * ZZ;
        F S1 = Y * q.ZZZ;
        F S2 = q.Y * ZZZ;
        F Pd = F::sub(U2, U1);
        F R = F::sub(S2, S1);
        if (Pd.is_zero()) {
            if (R.is_zero()) { *this = dbl(q); return; }
            *this = inf();
            return;
        }
        F PP = F::sqr(Pd);
        F PPP = Pd * PP;
        F Q = U1 * PP;
        F X3 = F::sub(F::sub(F::sub(F::sqr(R), PPP), Q), Q);
        F Y3 = F::sub(R * F::sub(Q, X3), S1 * PPP);
        X = X3;
        Y = Y3;
        ZZ = (ZZ * q.ZZ) * PP;
        ZZZ = (ZZZ * q.ZZZ) * PPP;
    }

    // only for FT = Fe<FP> (needs a field inversion)
    APK_HD Aff to_affine() const {
        if (is_inf()) return Aff::inf();
        // 1/ZZZ gives both: 1/ZZ = ZZZ^-1 * ZZZ / ZZ ... simpler: two field ops from one inversion
        F zzz_inv = F::inv(ZZZ);
        F zz_inv = F::sqr(zzz_inv * ZZ);  // (ZZ/ZZZ)^2 = 1/Z^2 = 1/ZZ
        return Aff{X * zz_inv, Y * zzz_inv};
    }
};

// MSM-internal point (unsaturated limbs, radix R') -> gnark-radix XYZZ: four domain-conversion products
template <class FP>
APK_HD XYZZ<FP> to_fe_point(const XYZZ<FP, FeU<FP>>& p) {
    return XYZZ<FP>{p.X.to_fe(), p.Y.to_fe(), p.ZZ.to_fe(), p.ZZZ.to_fe()};
}
