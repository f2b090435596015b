// HOST-only: sum_i k_i * P_i for a HANDFUL of G1 points (<= HOST_MSM_MAX) with full-width scalars, on the prover's host
// thread.  Used for the commitment of the linearised polynomial: it is a linear combination, with coefficients the host
// already knows, of commitments the host already holds (VK points, [Z], [H1..3], [pi2]) - the same group element gnark gets
// from kzg.Commit(linearizedPolynomialCanonical) [UPSTREAM; its digest enters gamma', templateLogicSigBN254.go:280-286], so
// the prover does not spend one of its ten size-n MSMs on it.
//
// Straus / interleaved fixed windows: signed 5-bit digits, one table of 1..16 multiples per point (made affine with one shared
// inversion), 5 doublings + `count` mixed additions per window.  ~11 k field products for 11 points: ~0.35 ms (BN254) /
// ~0.85 ms (BLS12-381) on one host core with host_fp.h's 64-bit limbs.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include "ec.h"
#include "host_fp.h"

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

// the sum in XYZZ form (no final inversion): host_lincomb adds up the parts of several threads
template <class FRP, class FPP>
XYZZ<FPP, Fe64<FPP>> host_lincomb_xyzz(const Affine<FPP>* pts, const Fe<FRP>* scalars_mont, int count) {
    using F = Fe64<FPP>;
    using A = Affine<FPP, F>;
    using X = XYZZ<FPP, F>;
    constexpr int T = 1 << (HOST_MSM_W - 1);                       // table entries per point: 1 .. 16
    constexpr int NW = (FRP::BITS + 1 + HOST_MSM_W - 1) / HOST_MSM_W;
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
    // signed digits, least significant window first
    int8_t dig[HOST_MSM_MAX][NW + 1];
    for (int i = 0; i < count; i++) {
        const Fe<FRP> k = Fe<FRP>::from_mont(scalars_mont[i]);
        int carry = 0;
        for (int w = 0; w <= NW; w++) {
            int d = carry;
            for (int b = 0; b < HOST_MSM_W; b++) {
                const int bit = w * HOST_MSM_W + b;
                if (bit < 32 * Fe<FRP>::N) d += (int)((k.l[bit >> 5] >> (bit & 31)) & 1u) << b;
            }
            if (d > T) { d -= 2 * T; carry = 1; } else carry = 0;
            dig[i][w] = (int8_t)d;
        }
    }
    X acc = X::inf();
    for (int w = NW; w >= 0; w--) {
        if (!acc.is_inf())
            for (int b = 0; b < HOST_MSM_W; b++) acc = X::dbl(acc);
        for (int i = 0; i < count; i++) {
            const int d = dig[i][w];
            if (d > 0) acc.madd(ta[(size_t)i * T + d - 1]);
            else if (d < 0) acc.madd(ta[(size_t)i * T - d - 1], true);
        }
    }
    return acc;
}

// `pool`: the points are dealt to the caller and the pool's parked threads (each runs its own Straus pass - the doublings are
// repeated, the additions and the tables are shared out), the parts are added, one inversion.  The combination sits between two
// Fiat-Shamir challenges with the GPU idle: on a lone BLS12-381 2^14 proof it was 0.36 of 2.91 ms on one thread.
template <class FRP, class FPP>
Affine<FPP> host_lincomb(const Affine<FPP>* pts, const Fe<FRP>* scalars_mont, int count, HostPool* pool = nullptr) {
    using F = Fe64<FPP>;
    using X = XYZZ<FPP, F>;
    if (count <= 0 || count > HOST_MSM_MAX) return Affine<FPP>::inf();
    X acc;
    bool done = false;
    if (pool) {
        int parts = pool->workers() + 1;
        if (parts > count / 2) parts = count / 2;
        if (parts >= 2) {
            X part[HOST_MSM_MAX];
            const std::function<void(int)> fn = [&](int t) {
                const int lo = (int)((long)count * t / parts), hi = (int)((long)count * (t + 1) / parts);
                part[t] = host_lincomb_xyzz<FRP, FPP>(pts + lo, scalars_mont + lo, hi - lo);
            };
            if (pool->run(parts, fn)) {
                acc = part[0];
                for (int t = 1; t < parts; t++) acc.add(part[t]);
                done = true;
            }
        }
    }
    if (!done) acc = host_lincomb_xyzz<FRP, FPP>(pts, scalars_mont, count);
    if (acc.is_inf()) return Affine<FPP>::inf();
    const F zzz_inv = F::inv(acc.ZZZ);
    const F zz_inv = F::sqr(zzz_inv * acc.ZZ);
    return Affine<FPP>{(acc.X * zz_inv).to(), (acc.Y * zzz_inv).to()};
}

}  // namespace apk
