// HOST-only: sum_i k_i * P_i for a HANDFUL of G1 points (<= HOST_MSM_MAX) with full-width scalars, on the prover's host
// thread.  Used for the commitment of the linearised polynomial: it is a linear combination, with coefficients the host
// already knows, of commitments the host already holds (VK points, [Z], [H1..3], [pi2]) - the same group element gnark gets
// from kzg.Commit(linearizedPolynomialCanonical) [UPSTREAM; its digest enters gamma', templateLogicSigBN254.go:280-286], so
// the prover does not spend one of its ten size-n MSMs on it.
//
// Straus / interleaved fixed windows: signed 5-bit digits, one table of 1..16 multiples per point (made affine with one shared
// inversion), 5 doublings + one mixed addition per column and window.  Every scalar is split in two ~128-bit halves first (GLV,
// glv_params.h): 2 * count columns of 27 windows instead of count columns of 52 - half the doublings, which is what a thread
// with one or two points spends most of its time on.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include "ec.h"
#include "host_fp.h"
#include "glv_params.h"

namespace apk {

// A few parked host threads for the pieces of host work that sit between two Fiat-Shamir challenges with the GPU idle (the [lin]
// combination): starting threads per call cost more than the 0.16 ms BN254 combination saves; waking parked ones costs ~10 us.
// One job at a time: a caller that finds the pool taken does the work itself.
class HostPool {
public:
    explicit HostPool(int workers) {
        for (int i = 0; i < workers; i++) th_.emplace_back([this, i] { loop(i); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int workers() const { return (int)th_.size(); }
    // fn(0) on the caller, fn(1) .. fn(parts - 1) on the workers; returns once all are done - or false, nothing run, when the pool is in use
    bool run(int parts, const std::function<void(int)>& fn) {
        std::unique_lock<std::mutex> use(use_mu_, std::try_to_lock);
        if (!use.owns_lock() || parts < 2 || parts - 1 > (int)th_.size()) return false;
        { std::lock_guard<std::mutex> g(mu_); job_ = &fn; parts_ = parts; pending_ = parts - 1; gen_++; }
        cv_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [&] { return pending_ == 0; });
        job_ = nullptr;
        return true;
    }
private:
    void loop(int i) {
        uint64_t last = 0;
        for (;;) {
            std::unique_lock<std::mutex> g(mu_);
            cv_.wait(g, [&] { return stop_ || gen_ != last; });
            if (stop_) return;
            last = gen_;
            const bool mine = i + 1 < parts_;
            const std::function<void(int)>* job = job_;
            g.unlock();
            if (mine) {
                (*job)(i + 1);
                std::lock_guard<std::mutex> g2(mu_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_, use_mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* job_ = nullptr;
    int parts_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

constexpr int HOST_MSM_MAX = 16;
constexpr int HOST_MSM_W = 5;

// ---- GLV (glv_params.h): k = k1 + k2 lambda with halves of ~128 bits, phi(x, y) = (beta x, y) = lambda (x, y) -----------------
// The Straus pass then runs over 2 * count columns of half the length: half the doublings, the same additions.  Whatever the
// split, the sum is the same group element (k1 + k2 lambda = k mod r exactly), so the proof bytes do not depend on it.
struct U256 { uint64_t w[4]; };
inline U256 u256_mul_lo(const U256& a, const U256& b) {
    U256 r{{0, 0, 0, 0}};
    for (int i = 0; i < 4; i++) {
        unsigned __int128 carry = 0;
        for (int j = 0; i + j < 4; j++) {
            const unsigned __int128 t = (unsigned __int128)a.w[i] * b.w[j] + r.w[i + j] + carry;
            r.w[i + j] = (uint64_t)t;
            carry = t >> 64;
        }
    }
    return r;
}
inline U256 u256_mul_hi(const U256& a, const U256& b) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        unsigned __int128 carry = 0;
        for (int j = 0; j < 4; j++) {
            const unsigned __int128 v = (unsigned __int128)a.w[i] * b.w[j] + t[i + j] + carry;
            t[i + j] = (uint64_t)v;
            carry = v >> 64;
        }
        t[i + 4] = (uint64_t)carry;
    }
    return U256{{t[4], t[5], t[6], t[7]}};
}
inline U256 u256_sub(const U256& a, const U256& b) {
    U256 r;
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned __int128 t = (unsigned __int128)a.w[i] - b.w[i] - borrow;
        r.w[i] = (uint64_t)t;
        borrow = (uint64_t)(t >> 64) & 1;
    }
    return r;
}
inline U256 u256_neg(const U256& a) { return u256_sub(U256{{0, 0, 0, 0}}, a); }

// k (plain, below r) -> |k1|, |k2| and their signs; false when a half is longer than HALF_BITS (never seen: the caller then
// takes the plain pass).  Arithmetic is modulo 2^256 in two's complement: the true values are far below 2^255.
template <class FPP>
inline bool glv_split(const U256& k, U256& k1, bool& neg1, U256& k2, bool& neg2) {
    using G = GlvParams<FPP>;
    auto ld = [](const uint64_t* p) { return U256{{p[0], p[1], p[2], p[3]}}; };
    U256 c1 = u256_mul_hi(k, ld(G::m1)), c2 = u256_mul_hi(k, ld(G::m2));
    if (G::s1 < 0) c1 = u256_neg(c1);
    if (G::s2 < 0) c2 = u256_neg(c2);
    k1 = u256_sub(u256_sub(k, u256_mul_lo(c1, ld(G::a1))), u256_mul_lo(c2, ld(G::a2)));
    k2 = u256_sub(u256_neg(u256_mul_lo(c1, ld(G::b1))), u256_mul_lo(c2, ld(G::b2)));
    neg1 = (k1.w[3] >> 63) != 0;
    neg2 = (k2.w[3] >> 63) != 0;
    if (neg1) k1 = u256_neg(k1);
    if (neg2) k2 = u256_neg(k2);
    auto fits = [](const U256& x) { return x.w[3] == 0 && (x.w[2] >> (G::HALF_BITS - 128)) == 0; };
    return fits(k1) && fits(k2);
}

// signed HOST_MSM_W-bit digits of a `bits`-bit magnitude, least significant window first, nw + 1 entries (the last takes the carry)
inline void host_signed_digits(const uint64_t* words, int bits, int8_t* dig, int nw) {
    constexpr int T = 1 << (HOST_MSM_W - 1);
    int carry = 0;
    for (int w = 0; w <= nw; w++) {
        int d = carry;
        for (int b = 0; b < HOST_MSM_W; b++) {
            const int bit = w * HOST_MSM_W + b;
            if (bit < bits) d += (int)((words[bit >> 6] >> (bit & 63)) & 1u) << b;
        }
        if (d > T) { d -= 2 * T; carry = 1; } else carry = 0;
        dig[w] = (int8_t)d;
    }
}

// the sum in XYZZ form (no final inversion): host_lincomb adds up the parts of several threads.  glv = false: the plain
// full-length pass (kept for the tests that hold the two against each other).
template <class FRP, class FPP>
XYZZ<FPP, Fe64<FPP>> host_lincomb_xyzz(const Affine<FPP>* pts, const Fe<FRP>* scalars_mont, int count, bool glv = true) {
    using F = Fe64<FPP>;
    using A = Affine<FPP, F>;
    using X = XYZZ<FPP, F>;
    constexpr int T = 1 << (HOST_MSM_W - 1);                       // table entries per point: 1 .. 16
    constexpr int NW = (FRP::BITS + 1 + HOST_MSM_W - 1) / HOST_MSM_W;
    static_assert(Fe<FRP>::N == 8, "256-bit scalars");
    if (count <= 0 || count > HOST_MSM_MAX) return X::inf();
    // tables in XYZZ, then one shared inversion for the affine forms
    std::vector<X> tx((size_t)count * T);
    for (int i = 0; i < count; i++) {
        const A p{F::from(pts[i].x), F::from(pts[i].y)};
        X acc = X::from_affine(p);
        tx[(size_t)i * T] = acc;
        X two = X::dbl_affine(p);
        tx[(size_t)i * T + 1] = two;
        acc = two;
        for (int m = 2; m < T; m++) { acc.madd(p); tx[(size_t)i * T + m] = acc; }
    }
    // batch inversion of the ZZZ (infinity entries - only for a base at infinity - are skipped)
    std::vector<F> pref(tx.size() + 1);
    pref[0] = F::one();
    for (size_t j = 0; j < tx.size(); j++) pref[j + 1] = tx[j].is_inf() ? pref[j] : pref[j] * tx[j].ZZZ;
    F inv = F::inv(pref[tx.size()]);
    std::vector<A> ta(tx.size());
    for (size_t j = tx.size(); j-- > 0;) {
        if (tx[j].is_inf()) { ta[j] = A::inf(); continue; }
        const F zzz_inv = inv * pref[j];
        inv = inv * tx[j].ZZZ;
        const F zz_inv = F::sqr(zzz_inv * tx[j].ZZ);               // (ZZ / ZZZ)^2 = 1 / ZZ
        ta[j] = A{tx[j].X * zz_inv, tx[j].Y * zzz_inv};
    }
    // columns of the Straus pass: a table, its digits, a sign
    struct Column { const A* tab; bool neg; int8_t dig[NW + 2]; };
    std::vector<Column> cols;
    std::vector<A> tb;                                             // phi of the tables: (beta x, y)
    int nw = NW;
    bool split = false;
    if constexpr (GlvParams<FPP>::available) {
        if (glv) {
            constexpr int HB = GlvParams<FPP>::HALF_BITS;
            constexpr int NWG = (HB + 1 + HOST_MSM_W - 1) / HOST_MSM_W;
            cols.resize((size_t)2 * count);
            split = true;
            for (int i = 0; i < count && split; i++) {
                const Fe<FRP> k = Fe<FRP>::from_mont(scalars_mont[i]);
                U256 ku, k1, k2;
                for (int w = 0; w < 4; w++) ku.w[w] = (uint64_t)k.l[2 * w] | ((uint64_t)k.l[2 * w + 1] << 32);
                bool n1, n2;
                if (!glv_split<FPP>(ku, k1, n1, k2, n2)) { split = false; break; }
                cols[2 * i].neg = n1; cols[2 * i + 1].neg = n2;
                host_signed_digits(k1.w, HB, cols[2 * i].dig, NWG);
                host_signed_digits(k2.w, HB, cols[2 * i + 1].dig, NWG);
            }
            if (split) {
                Fe<FPP> bp;
                for (int w = 0; w < Fe<FPP>::N; w++) bp.l[w] = GlvParams<FPP>::beta[w];
                const F beta = F::from(Fe<FPP>::to_mont(bp));
                tb.resize(ta.size());
                for (size_t j = 0; j < ta.size(); j++) tb[j] = ta[j].is_inf() ? ta[j] : A{ta[j].x * beta, ta[j].y};
                for (int i = 0; i < count; i++) { cols[2 * i].tab = &ta[(size_t)i * T]; cols[2 * i + 1].tab = &tb[(size_t)i * T]; }
                nw = NWG;
            }
        }
    }
    if (!split) {
        cols.resize((size_t)count);
        for (int i = 0; i < count; i++) {
            const Fe<FRP> k = Fe<FRP>::from_mont(scalars_mont[i]);
            uint64_t kw[4];
            for (int w = 0; w < 4; w++) kw[w] = (uint64_t)k.l[2 * w] | ((uint64_t)k.l[2 * w + 1] << 32);
            cols[i].tab = &ta[(size_t)i * T]; cols[i].neg = false;
            host_signed_digits(kw, 256, cols[i].dig, NW);
        }
    }
    X acc = X::inf();
    for (int w = nw; w >= 0; w--) {
        if (!acc.is_inf())
            for (int b = 0; b < HOST_MSM_W; b++) acc = X::dbl(acc);
        for (const Column& c : cols) {
            const int d = c.dig[w];
            if (d > 0) acc.madd(c.tab[d - 1], c.neg);
            else if (d < 0) acc.madd(c.tab[-d - 1], !c.neg);
        }
    }
    return acc;
}

// ---- fixed bases: the verifying key's commitments ([Ql][Qr][Qm][Qo][S3]) enter every proof's [lin] with fresh full-width
// coefficients.  Their multiples are tabulated once per context - 2^(8w) * j * P for the 33 byte windows w and j = 1..128, affine -
// so a coefficient costs 33 mixed additions and no doubling (the Straus pass: 15 for the table, 54 additions, its share of 130
// doublings).  33 * 128 * 64 B = 264 KiB per point on BN254, 396 KiB on BLS12-381; built in a few milliseconds at context creation.
template <class FPP>
class HostFixedBase {
  public:
    using F = Fe64<FPP>;
    using A = Affine<FPP, F>;
    using X = XYZZ<FPP, F>;
    static constexpr int WB = 8, T = 1 << (WB - 1), NW = (256 + WB - 1) / WB + 1;   // 33 windows: the last one takes the carry
    int points() const { return npoints_; }
    void build(const Affine<FPP>* pts, int count) {
        npoints_ = count;
        tab_.assign((size_t)count * NW * T, A::inf());
        std::vector<X> row(T);
        std::vector<F> pref(T + 1);
        for (int i = 0; i < count; i++) {
            A base{F::from(pts[i].x), F::from(pts[i].y)};
            for (int w = 0; w < NW; w++) {
                A* out = &tab_[((size_t)i * NW + w) * T];
                if (base.is_inf()) continue;                       // a commitment at infinity (zero polynomial): nothing to add, ever
                row[0] = X::from_affine(base);
                row[1] = X::dbl_affine(base);
                for (int j = 2; j < T; j++) { row[j] = row[j - 1]; row[j].madd(base); }
                // affine forms with one inversion (no entry is at infinity: j * 2^(8w) < r for a point of prime order r)
                pref[0] = F::one();
                for (int j = 0; j < T; j++) pref[j + 1] = pref[j] * row[j].ZZZ;
                F inv = F::inv(pref[T]);
                for (int j = T; j-- > 0;) {
                    const F zzz_inv = inv * pref[j];
                    inv = inv * row[j].ZZZ;
                    const F zz_inv = F::sqr(zzz_inv * row[j].ZZ);
                    out[j] = A{row[j].X * zz_inv, row[j].Y * zzz_inv};
                }
                const X next = X::dbl_affine(out[T - 1]);          // 2 * 128 * base = 2^8 * base
                if (next.is_inf()) { base = A::inf(); continue; }
                const F zzz_inv = F::inv(next.ZZZ);
                const F zz_inv = F::sqr(zzz_inv * next.ZZ);
                base = A{next.X * zz_inv, next.Y * zzz_inv};
            }
        }
    }
    // acc += sum_i k_i P_i over the tabulated points (scalars in gnark's Montgomery form, one per point)
    template <class FRP>
    void accumulate(X& acc, const Fe<FRP>* scalars_mont) const {
        static_assert(Fe<FRP>::N == 8, "256-bit scalars");
        for (int i = 0; i < npoints_; i++) {
            const Fe<FRP> k = Fe<FRP>::from_mont(scalars_mont[i]);
            int carry = 0;
            for (int w = 0; w < NW; w++) {
                int d = carry + (w < 32 ? (int)((k.l[w >> 2] >> (8 * (w & 3))) & 0xffu) : 0);
                if (d > T) { d -= 2 * T; carry = 1; } else carry = 0;
                const A* row = &tab_[((size_t)i * NW + w) * T];
                if (d > 0) acc.madd(row[d - 1]);
                else if (d < 0) acc.madd(row[-d - 1], true);
            }
        }
    }
  private:
    int npoints_ = 0;
    std::vector<A> tab_;
};

// `pool`: the points are dealt to the caller and the pool's parked threads (each runs its own Straus pass - the doublings are
// repeated, the additions and the tables are shared out) and the parts are added.  The combination sits between two
// Fiat-Shamir challenges with the GPU idle: on a lone BLS12-381 2^14 proof it was 0.36 of 2.91 ms on one thread.
// host_lincomb_sum leaves the sum in XYZZ form: the prover adds the part it could compute before the evaluations were known
// (the [H] terms: their coefficients depend on zeta alone) to the rest, then converts once.
template <class FRP, class FPP>
XYZZ<FPP, Fe64<FPP>> host_lincomb_sum(const Affine<FPP>* pts, const Fe<FRP>* scalars_mont, int count, HostPool* pool = nullptr, bool glv = true) {
    using X = XYZZ<FPP, Fe64<FPP>>;
    if (count <= 0 || count > HOST_MSM_MAX) return X::inf();
    if (pool) {
        int parts = pool->workers() + 1;
        if (parts > count) parts = count;
        if (parts >= 2) {
            X part[HOST_MSM_MAX];
            const std::function<void(int)> fn = [&](int t) {
                const int lo = (int)((long)count * t / parts), hi = (int)((long)count * (t + 1) / parts);
                part[t] = host_lincomb_xyzz<FRP, FPP>(pts + lo, scalars_mont + lo, hi - lo, glv);
            };
            if (pool->run(parts, fn)) {
                X acc = part[0];
                for (int t = 1; t < parts; t++) acc.add(part[t]);
                return acc;
            }
        }
    }
    return host_lincomb_xyzz<FRP, FPP>(pts, scalars_mont, count, glv);
}

template <class FPP>
Affine<FPP> host_xyzz_to_affine(const XYZZ<FPP, Fe64<FPP>>& acc) {
    using F = Fe64<FPP>;
    if (acc.is_inf()) return Affine<FPP>::inf();
    const F zzz_inv = F::inv(acc.ZZZ);
    const F zz_inv = F::sqr(zzz_inv * acc.ZZ);
    return Affine<FPP>{(acc.X * zz_inv).to(), (acc.Y * zzz_inv).to()};
}

template <class FRP, class FPP>
Affine<FPP> host_lincomb(const Affine<FPP>* pts, const Fe<FRP>* scalars_mont, int count, HostPool* pool = nullptr) {
    return host_xyzz_to_affine<FPP>(host_lincomb_sum<FRP, FPP>(pts, scalars_mont, count, pool));
}

}  // namespace apk
