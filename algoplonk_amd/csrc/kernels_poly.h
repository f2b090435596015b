// Element-wise / scan / reduction kernels over Fr vectors used between the NTTs and MSMs of one proof.
//
// What they replace (all [UPSTREAM gnark v0.15.0 backend/plonk/<curve>/prove.go + gnark-crypto fr/iop,
// not vendored]; the formulas themselves are pinned by the reference's verifier template, cited per kernel):
//   * grand product Z            SURVEY.md §8a row a8; challenge use /root/reference/verifier/templateLogicSigBN254.go:204-215,232-254
//   * quotient numerator / Z_H   row a7; identity = SURVEY.md App. E (templateLogicSigBN254.go:203-278)
//   * Horner evaluations, folding and (f - f(z))/(X - z)    row a9; order pinned at templateLogicSigBN254.go:289-320
// All are HBM-streaming kernels: 32 B/element/operand, one read of each operand and one write.
#pragma once
#include "ff.h"
#include "ffu.h"

namespace apk {

constexpr int POLY_THREADS = 256;
constexpr int SCAN_PER_THREAD = 8;
constexpr int SCAN_BLOCK = POLY_THREADS * SCAN_PER_THREAD;  // 2048 elements per block

template <class FR>
struct Fr4 { Fe<FR> v[4]; };

// ---- small helpers ------------------------------------------------------------------------------------------
template <class FR>
__global__ void fill_zero_kernel(Fe<FR>* p, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = Fe<FR>::zero();
}

// p(X) += b(X) (X^n - 1), deg b = d-1 <= 2;  p has capacity n+d and holds n coefficients: p[n..n+d) is assigned, not added to
template <class FR>
struct BlindK {
    static __device__ __forceinline__ void run(Fe<FR>* p, uint32_t n, Fr4<FR> b, int d) {
        int i = threadIdx.x;
        if (i < d) {
            p[i] = p[i] - b.v[i];
            p[n + i] = b.v[i];
        }
    }
};
template <class FR>
__global__ void blind_kernel(Fe<FR>* p, uint32_t n, Fr4<FR> b, int d) { BlindK<FR>::run(p, n, b, d); }
// the three wire polynomials in one launch (blockIdx.x = wire)
template <class FR>
struct Blind3 {
    Fe<FR>* p[3];
    Fr4<FR> b[3];
    Fe<FR>* lag[3];   // or null: the Lagrange-basis scalar vector of the wire (n values); b is appended at [n, n + d) - the scalars of
                      // the blinding points [tau^(n+k)] - [tau^k] of the extended Lagrange table (backend_impl.h, round 1)
};
template <class FR>
struct Blind3K {
    static __device__ __forceinline__ void run(Blind3<FR> a, uint32_t n, int d) {
        int i = threadIdx.x;
        Fe<FR>* p = a.p[blockIdx.x];
        if (i < d) {
            const Fe<FR> v = a.b[blockIdx.x].v[i];
            if (p) {
                p[i] = p[i] - v;
                p[n + i] = v;
            }
            if (a.lag[blockIdx.x]) a.lag[blockIdx.x][n + i] = v;
        }
    }
};
template <class FR>
__global__ void blind3_kernel(Blind3<FR> a, uint32_t n, int d) { Blind3K<FR>::run(a, n, d); }

// out[i] = a[i] * b[i]  (b indexed with offset/stride so tables can be reused)
template <class FR>
struct MulK {
    static __device__ __forceinline__ void run(Fe<FR>* out, const Fe<FR>* a, const Fe<FR>* b, uint32_t count) {
        wave_priority<APK_PRIO_FR>();
        uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i < count) out[i] = a[i] * b[i];
    }
};
template <class FR>
__global__ void mul_kernel(Fe<FR>* out, const Fe<FR>* a, const Fe<FR>* b, uint32_t count) { MulK<FR>::run(out, a, b, count); }

// ---- grand product ------------------------------------------------------------------------------------------
// Z[0] = 1, Z[k] = prod_{i<k} num_i / den_i with
//   num_i = prod_j (w_j[i] + beta*u^j*omega^i + gamma),  den_i = prod_j (w_j[i] + beta*S_j[i] + gamma)
// (omega^i comes from the size-n twiddle table: tw[i] for i < n/2, -tw[i-n/2] above).  No per-row inversion:
// a forward product scan of num, a reverse product scan of den and ONE field inversion of the total give
//   Z[k] = (prod_{i<k} num_i) * (prod_{i>=k} den_i) * (prod_i den_i)^-1.
template <class FR>
struct GpTermsK {
    static __device__ __forceinline__ void run(const Fe<FR>* __restrict__ L, const Fe<FR>* __restrict__ R,
                                                                const Fe<FR>* __restrict__ O, const Fe<FR>* __restrict__ S1,
                                                                const Fe<FR>* __restrict__ S2, const Fe<FR>* __restrict__ S3,
                                                                const Fe<FR>* __restrict__ tw, uint32_t n, Fe<FR> beta,
                                                                Fe<FR> gamma, Fe<FR> beta_u, Fe<FR> beta_u2,
                                                                Fe<FR>* __restrict__ num, Fe<FR>* __restrict__ den) {
        wave_priority<APK_PRIO_FR>();
        using Fr = Fe<FR>;
        uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= n) return;
        // On unsaturated limbs, lazily.  beta, beta_u, beta_u2 arrive as 32 x (the product's radix R' = 32 R), so beta * w is exact;
        // the two value x value products of each triple leave num and den BOTH divided by 1024 - which cancels in
        // Z[k] = prod_{i<k} num * prod_{i>=k} den / prod den (k factors, n - k factors, n factors).  Sums below 4 r, products below 2 r.
        using U = FeU<FR>;
        static_assert(U::HEADROOM >= 64, "factors below 4 r");
        auto ld = [](const Fr* p, uint32_t k) { Fr v = p[k]; return U::unpack(v.l); };
        Fr wf = i < n / 2 ? tw[i] : Fr::neg(tw[i - n / 2]);
        const U w = U::unpack(wf.l), gm = U::unpack(gamma.l), b = U::unpack(beta.l), bu = U::unpack(beta_u.l), bu2 = U::unpack(beta_u2.l);
        const U l = U::add_n(ld(L, i), gm), r = U::add_n(ld(R, i), gm), o = U::add_n(ld(O, i), gm);      // < 2
        U nu = U::mul_nr(U::add_n(l, U::mul_nr(b, w)), U::add_n(r, U::mul_nr(bu, w)));                      // factors < 4
        nu = U::mul_nr(nu, U::add_n(o, U::mul_nr(bu2, w)));
        U de = U::mul_nr(U::add_n(l, U::mul_nr(b, ld(S1, i))), U::add_n(r, U::mul_nr(b, ld(S2, i))));
        de = U::mul_nr(de, U::add_n(o, U::mul_nr(b, ld(S3, i))));
        Fr on, od;
        U::template canon<1>(nu).pack(on.l);
        U::template canon<1>(de).pack(od.l);
        num[i] = on;
        den[i] = od;
    }
};
template <class FR>
__global__ void __launch_bounds__(POLY_THREADS) gp_terms_kernel(const Fe<FR>* __restrict__ L, const Fe<FR>* __restrict__ R,
                                                                const Fe<FR>* __restrict__ O, const Fe<FR>* __restrict__ S1,
                                                                const Fe<FR>* __restrict__ S2, const Fe<FR>* __restrict__ S3,
                                                                const Fe<FR>* __restrict__ tw, uint32_t n, Fe<FR> beta,
                                                                Fe<FR> gamma, Fe<FR> beta_u, Fe<FR> beta_u2,
                                                                Fe<FR>* __restrict__ num, Fe<FR>* __restrict__ den) { GpTermsK<FR>::run(L, R, O, S1, S2, S3, tw, n, beta, gamma, beta_u, beta_u2, num, den); }

// ---- generic scans over Fr: op = multiply (grand product) or add (suffix sums for the KZG quotient) ------
struct OpMul { template <class F> __device__ static F apply(const F& a, const F& b) { return a * b; }
               template <class F> __device__ static F identity() { return F::one(); } };
struct OpAdd { template <class F> __device__ static F apply(const F& a, const F& b) { return a + b; }
               template <class F> __device__ static F identity() { return F::zero(); } };

// logical index i -> memory index: forward (i) or reversed (count-1-i) so the same code does suffix scans
__device__ __forceinline__ uint32_t scan_idx(uint32_t i, uint32_t count, bool rev) { return rev ? count - 1 - i : i; }

// phase 1: per-block inclusive scan in place; block totals out
template <class FR, class OP>
__device__ __forceinline__ void scan_block_body(Fe<FR>* __restrict__ data, uint32_t count, bool rev, Fe<FR>* __restrict__ block_tot) {
    using Fr = Fe<FR>;
    __shared__ Fr sm[POLY_THREADS];
    const uint32_t t = threadIdx.x;
    const uint32_t base = blockIdx.x * SCAN_BLOCK + t * SCAN_PER_THREAD;
    Fr v[SCAN_PER_THREAD];
    Fr run = OP::template identity<Fr>();
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        uint32_t i = base + k;
        Fr x = i < count ? data[scan_idx(i, count, rev)] : OP::template identity<Fr>();
        run = OP::apply(run, x);
        v[k] = run;
    }
    sm[t] = run;
    __syncthreads();
    Fr mine = run;
    for (uint32_t d = 1; d < POLY_THREADS; d <<= 1) {
        Fr o = OP::template identity<Fr>();
        if (t >= d) o = sm[t - d];
        __syncthreads();
        mine = OP::apply(o, mine);
        sm[t] = mine;
        __syncthreads();
    }
    Fr excl = t == 0 ? OP::template identity<Fr>() : sm[t - 1];
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        uint32_t i = base + k;
        if (i < count) data[scan_idx(i, count, rev)] = OP::apply(excl, v[k]);
    }
    if (t == POLY_THREADS - 1) block_tot[blockIdx.x] = mine;
}
template <class FR, class OP>
struct ScanBlockK {
    static __device__ __forceinline__ void run(Fe<FR>* __restrict__ data, uint32_t count, bool rev,
                                                                  Fe<FR>* __restrict__ block_tot) {
        wave_priority<APK_PRIO_FR>();
        scan_block_body<FR, OP>(data, count, rev, block_tot);
    }
};
template <class FR, class OP>
__global__ void __launch_bounds__(POLY_THREADS) scan_block_kernel(Fe<FR>* __restrict__ data, uint32_t count, bool rev,
                                                                  Fe<FR>* __restrict__ block_tot) { ScanBlockK<FR, OP>::run(data, count, rev, block_tot); }

// phase 2: single block, exclusive scan of block totals in place (nblocks <= POLY_THREADS * 64)
// Returns (in the last thread) the total over all blocks.
template <class FR, class OP>
__device__ __forceinline__ Fe<FR> scan_totals_body(Fe<FR>* __restrict__ tot, uint32_t nblocks) {
    using Fr = Fe<FR>;
    __shared__ Fr sm[POLY_THREADS];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nblocks + POLY_THREADS - 1) / POLY_THREADS;
    const uint32_t lo = min(t * per, nblocks), hi = min(lo + per, nblocks);
    Fr run = OP::template identity<Fr>();
    for (uint32_t i = lo; i < hi; i++) run = OP::apply(run, tot[i]);
    sm[t] = run;
    __syncthreads();
    Fr mine = run;
    for (uint32_t d = 1; d < POLY_THREADS; d <<= 1) {
        Fr o = OP::template identity<Fr>();
        if (t >= d) o = sm[t - d];
        __syncthreads();
        mine = OP::apply(o, mine);
        sm[t] = mine;
        __syncthreads();
    }
    Fr acc = t == 0 ? OP::template identity<Fr>() : sm[t - 1];
    for (uint32_t i = lo; i < hi; i++) {
        Fr x = tot[i];
        tot[i] = acc;  // exclusive
        acc = OP::apply(acc, x);
    }
    return acc;
}
template <class FR, class OP>
struct ScanTotalsK {
    static __device__ __forceinline__ void run(Fe<FR>* __restrict__ tot, uint32_t nblocks) {
        wave_priority<APK_PRIO_FR>();
        scan_totals_body<FR, OP>(tot, nblocks);
    }
};
template <class FR, class OP>
__global__ void __launch_bounds__(POLY_THREADS) scan_totals_kernel(Fe<FR>* __restrict__ tot, uint32_t nblocks) { ScanTotalsK<FR, OP>::run(tot, nblocks); }

// ---- the grand product's pair of scans: blockIdx.y = 0 forward over num, 1 reverse over den ------------------
template <class FR>
struct GpScan {
    Fe<FR>* data[2];
    Fe<FR>* tot[2];   // tot[1] has one extra slot: the product of all denominators
};
template <class FR>
struct GpScanBlockK {
    static __device__ __forceinline__ void run(GpScan<FR> g, uint32_t count) {
        wave_priority<APK_PRIO_FR>();
        scan_block_body<FR, OpMul>(g.data[blockIdx.y], count, blockIdx.y != 0, g.tot[blockIdx.y]);
    }
};
template <class FR>
__global__ void __launch_bounds__(POLY_THREADS) gp_scan_block_kernel(GpScan<FR> g, uint32_t count) { GpScanBlockK<FR>::run(g, count); }
template <class FR>
struct GpScanTotalsK {
    static __device__ __forceinline__ void run(GpScan<FR> g, uint32_t nblocks, Fe<FR>* __restrict__ total_out) {
        wave_priority<APK_PRIO_FR>();
        Fe<FR> total = scan_totals_body<FR, OpMul>(g.tot[blockIdx.x], nblocks);
        // the product of all denominators: inverted on the HOST (a lone GPU lane needs ~100 us for one Kaliski inversion); total_out
        // is the slot's pinned host buffer seen from the device, or a device word the host copies back
        if (blockIdx.x == 1 && threadIdx.x == POLY_THREADS - 1) *total_out = total;
    }
};
template <class FR>
__global__ void __launch_bounds__(POLY_THREADS) gp_scan_totals_kernel(GpScan<FR> g, uint32_t nblocks, Fe<FR>* __restrict__ total_out) { GpScanTotalsK<FR>::run(g, nblocks, total_out); }
// Z[0] = 1; Z[k] = num_prefix_incl[k-1] * den_suffix_incl[k] / den_total
template <class FR>
struct GpFinishK {
    static __device__ __forceinline__ void run(GpScan<FR> g, uint32_t n, Fe<FR> den_total_inv, Fe<FR>* __restrict__ z) {
        wave_priority<APK_PRIO_FR>();
        using Fr = Fe<FR>;
        uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
        if (k >= n) return;
        if (k == 0) { z[0] = Fr::one(); return; }
        Fr np = g.tot[0][(k - 1) / SCAN_BLOCK] * g.data[0][k - 1];
        Fr ds = g.tot[1][(n - 1 - k) / SCAN_BLOCK] * g.data[1][k];
        z[k] = np * (ds * den_total_inv);
    }
};
template <class FR>
__global__ void __launch_bounds__(POLY_THREADS) gp_finish_kernel(GpScan<FR> g, uint32_t n, Fe<FR> den_total_inv, Fe<FR>* __restrict__ z) { GpFinishK<FR>::run(g, n, den_total_inv, z); }

// phase 3: fold the block prefix in.  `shift` turns the inclusive scan into the exclusive one the grand
// product wants: out[0] = identity, out[i] = inclusive[i-1] (forward scans only).
template <class FR, class OP>
struct ScanApplyK {
    static __device__ __forceinline__ void run(const Fe<FR>* __restrict__ data, uint32_t count, bool rev,
                                                                  const Fe<FR>* __restrict__ block_excl,
                                                                  Fe<FR>* __restrict__ out, int shift) {
        wave_priority<APK_PRIO_FR>();
        using Fr = Fe<FR>;
        uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= count) return;
        Fr v = OP::apply(block_excl[i / SCAN_BLOCK], data[scan_idx(i, count, rev)]);
        if (shift) {
            if (i + 1 < count) out[i + 1] = v;
            if (i == 0) out[0] = OP::template identity<Fr>();
        } else {
            out[scan_idx(i, count, rev)] = v;
        }
    }
};
template <class FR, class OP>
__global__ void __launch_bounds__(POLY_THREADS) scan_apply_kernel(const Fe<FR>* __restrict__ data, uint32_t count, bool rev,
                                                                  const Fe<FR>* __restrict__ block_excl,
                                                                  Fe<FR>* __restrict__ out, int shift) { ScanApplyK<FR, OP>::run(data, count, rev, block_excl, out, shift); }

// ---- quotient numerator on the 4n coset, divided by Z_H ---------------------------------------------------
// gate + alpha*(Z(wX) prod(w_j + beta S_j + gamma) - Z(X) prod(w_j + beta u^j X + gamma)) + alpha^2 L_0 (Z - 1)
// (the linearised twin is templateLogicSigBN254.go:203-278).
constexpr int QK_INJECT_MAX = 6;
template <class FR>
struct QuotientArgs {
    const Fe<FR>*l, *r, *o, *z, *qk;                // per proof, 4n evaluations
    const Fe<FR>*ql, *qr, *qm, *qo, *s1, *s2, *s3;  // per circuit
    const Fe<FR>*x, *l0;                            // coset points, L_0 on the coset
    const Fe<FR>* qcp[2];
    const Fe<FR>* pi2[2];
    int nb_commit;
    // Qk completion without transforms: qk(x) = trace_qk(x) + sum_j delta[j] * L_row[j](x) for the few rows the prover writes
    // (public inputs, BSB22 commitment values); nb_inject = 0 means `qk` already holds the completed column's evaluations
    int nb_inject;
    const Fe<FR>* inj_tab[QK_INJECT_MAX];
    Fe<FR> inj_delta[QK_INJECT_MAX];
    Fe<FR> alpha, beta, gamma, beta_u, beta_u2, alpha2;
    Fe<FR> zh_inv[4];
    uint32_t n4;
    // Sub-coset mode (one proof on G GPUs, backend_impl.h SubCoset): this launch covers the points i = sub_k + G j of the 4n coset,
    // j < n4 (then = 4n / G).  The per-proof vectors (l, r, o, z, pi2) hold those evaluations at index j; the per-circuit tables are
    // read at i.  Z(omega X) is the point i + 4: the same class for G <= 4 (index j + 4 / G), class (sub_k + 4) mod 8 for G = 8
    // (`zs`, index j or j + 1).  sub_glog = 0: the whole coset, as ever.
    const Fe<FR>* zs;
    uint32_t sub_k, sub_glog;
};

// The kernel works on unsaturated limbs (ffu.h) without conditional subtractions, like the NTT tiles: polynomial VALUES stay in
// gnark's radix R, while what multiplies them is handed over in the product's own radix R' = 32 R, so w'(R') * v(R) / R' = w v (R):
//   * the selector tables ql, qr, qo, qcp hold 32 q and qm holds 1024 q (its operand l*r is a product of two values: / 32 more);
//   * beta, beta_u, beta_u2, zh_inv, inj_delta arrive as 32 x, alpha as 2^20 alpha (three value products behind it),
//     alpha2 as 2^10 alpha^2 (one value product behind it); gamma stays in R (it is added, not multiplied).
// Bounds (R'/r >= 71): sums of a few products stay below 16 r, every product of such a sum with a canonical factor below 2 r.
template <class FR>
struct QuotientK {
    static __device__ __forceinline__ void run(QuotientArgs<FR> a, Fe<FR>* __restrict__ out) {
        wave_priority<APK_PRIO_FR>();
        using Fr = Fe<FR>;
        using U = FeU<FR>;
        static_assert(U::HEADROOM >= 64, "lazy sums of up to 16 r times a canonical factor must stay below 2 r");
        const uint32_t jj = blockIdx.x * blockDim.x + threadIdx.x;   // index into the per-proof vectors
        if (jj >= a.n4) return;
        const uint32_t i = (jj << a.sub_glog) + a.sub_k;             // index into the per-circuit tables (the point of the 4n coset)
        auto ld = [](const Fr* p, uint32_t k) { Fr v = p[k]; return U::unpack(v.l); };
        auto cst = [](const Fr& v) { return U::unpack(v.l); };
        const U l = ld(a.l, jj), r = ld(a.r, jj), o = ld(a.o, jj), z = ld(a.z, jj);
        uint32_t is;
        const Fr* zsp = a.z;
        if (a.sub_glog <= 2) is = jj + (4u >> a.sub_glog);
        else { is = jj + (a.sub_k + 4u >= 8u ? 1u : 0u); zsp = a.zs; }
        if (is >= a.n4) is -= a.n4;
        const U zs = ld(zsp, is);
        U gate = U::add_n(U::mul_nr(ld(a.ql, i), l), U::mul_nr(ld(a.qr, i), r));
        gate = U::add_n(gate, U::mul_nr(ld(a.qm, i), U::mul_nr(l, r)));
        gate = U::add_n(gate, U::mul_nr(ld(a.qo, i), o));
        gate = U::add_n(gate, ld(a.qk, i));
        for (int k = 0; k < a.nb_commit; k++) gate = U::add_n(gate, U::mul_nr(ld(a.qcp[k], i), ld(a.pi2[k], jj)));
        for (int j = 0; j < a.nb_inject; j++) gate = U::add_n(gate, U::mul_nr(cst(a.inj_delta[j]), ld(a.inj_tab[j], i)));
        const U gm = cst(a.gamma), beta = cst(a.beta);
        const U lg = U::add_n(l, gm), rg = U::add_n(r, gm), og = U::add_n(o, gm);          // < 2
        const U x = ld(a.x, i);
        U pa = U::mul_nr(zs, U::add_n(lg, U::mul_nr(beta, ld(a.s1, i))));                   // factors < 4, products < 1.1
        pa = U::mul_nr(pa, U::add_n(rg, U::mul_nr(beta, ld(a.s2, i))));
        pa = U::mul_nr(pa, U::add_n(og, U::mul_nr(beta, ld(a.s3, i))));
        U pb = U::mul_nr(z, U::add_n(lg, U::mul_nr(beta, x)));
        pb = U::mul_nr(pb, U::add_n(rg, U::mul_nr(cst(a.beta_u), x)));
        pb = U::mul_nr(pb, U::add_n(og, U::mul_nr(cst(a.beta_u2), x)));
        const U perm = U::mul_nr(cst(a.alpha), U::template sub_k<2>(pa, pb));
        const Fr one_r = Fr::one();
        const U one = U::unpack(one_r.l);
        const U loc = U::mul_nr(cst(a.alpha2), U::mul_nr(ld(a.l0, i), U::template sub_k<1>(z, one)));
        const U num = U::add_n(U::add_n(gate, perm), loc);                                   // < 16
        const U res = U::template canon<1>(U::mul_nr(cst(a.zh_inv[i & 3]), num));
        Fr w;
        res.pack(w.l);
        out[jj] = w;
    }
};
template <class FR>
__global__ void __launch_bounds__(POLY_THREADS) quotient_kernel(QuotientArgs<FR> a, Fe<FR>* __restrict__ out) { QuotientK<FR>::run(a, out); }

// Sub-coset mode, the last log2(G) stages of the inverse 4n transform on every rank: part[k][c'] = omega_4n^(-k c') * (size-m inverse
// transform of rank k's quotient values, unscaled)[c'] for the G classes k (all-gathered), m = 4n / G.  Coefficient c = c' + m t:
//     h_c = u^-c / (4n) * sum_k rho^(-k t) part[k][c'],        rho = omega_4n^m (a primitive G-th root of unity)
// post[c] = u^-c / (4n) in the radix R' (the whole-coset transform's own table), rho_inv[e] = 32 rho^-e (R').
template <class FR>
struct SubMergeArgs { Fe<FR> rho_inv[8]; uint32_t m, glog; };
template <class FR>
__global__ void __launch_bounds__(POLY_THREADS) subcoset_merge_kernel(SubMergeArgs<FR> a, const Fe<FR>* __restrict__ part, const Fe<FR>* __restrict__ post,
                                                                      Fe<FR>* __restrict__ out) {
    wave_priority<APK_PRIO_FR>();
    using Fr = Fe<FR>;
    using U = FeU<FR>;
    static_assert(U::HEADROOM >= 64, "sums of 8 products below 2 r each");
    const uint32_t cp = blockIdx.x * blockDim.x + threadIdx.x;
    if (cp >= a.m) return;
    const uint32_t G = 1u << a.glog;
    U v[8];
    for (uint32_t k = 0; k < G; k++) { Fr x = part[(size_t)k * a.m + cp]; v[k] = U::unpack(x.l); }
    for (uint32_t t = 0; t < G; t++) {
        U acc = v[0];                                           // rho^0
        for (uint32_t k = 1; k < G; k++) {
            const uint32_t e = (k * t) & (G - 1u);
            acc = U::add_n(acc, e ? U::mul_nr(U::unpack(a.rho_inv[e].l), v[k]) : v[k]);
        }
        const size_t c = (size_t)t * a.m + cp;
        Fr pw = post[c];
        const U res = U::template canon<1>(U::mul_nr(U::unpack(pw.l), acc));
        Fr w;
        res.pack(w.l);
        out[c] = w;
    }
}

// p[i] *= c  (setup: selector tables into the quotient kernel's radix)
template <class FR>
__global__ void __launch_bounds__(POLY_THREADS) scale_kernel(Fe<FR>* __restrict__ p, Fe<FR> c, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = p[i] * c;
}

// ---- evaluation: partial[p][block] = sum_i f_p[i] * pw[i] -------------------------------------------------
constexpr int EVAL_MAX = 12;
template <class FR>
struct EvalArgs {
    const Fe<FR>* f[EVAL_MAX];
    const Fe<FR>* pw[EVAL_MAX];   // per polynomial: its table of powers of the evaluation point, or null = the kernel's `pw`
    uint32_t len[EVAL_MAX];
    int count;
};
constexpr int EVAL_PER_THREAD = 8;
constexpr int EVAL_BLOCK = POLY_THREADS * EVAL_PER_THREAD;

template <class FR>
struct EvalPartialK {
    static __device__ __forceinline__ void run(EvalArgs<FR> a, const Fe<FR>* __restrict__ pw,
                                                                    uint32_t nblocks, Fe<FR>* __restrict__ partial) {
        wave_priority<APK_PRIO_FR>();
        using Fr = Fe<FR>;
        __shared__ Fr sm[POLY_THREADS];
        const int p = blockIdx.y;
        const uint32_t t = threadIdx.x;
        const Fr* f = a.f[p];
        const Fr* __restrict__ pwp = a.pw[p] ? a.pw[p] : pw;
        const uint32_t len = a.len[p];
        // the EVAL_PER_THREAD products of a lane on unsaturated limbs: value x value products come out as f pw / 32 (both operands
        // are in gnark's radix R, the product's own radix is R' = 32 R) - the HOST multiplies the few results by 32 (eval_many's caller)
        using U = FeU<FR>;
        static_assert(U::HEADROOM >= 64 && EVAL_PER_THREAD * 2 <= 16, "8 products below 2 r each");
        U accu = U::zero();
        for (int k = 0; k < EVAL_PER_THREAD; k++) {
            uint32_t i = blockIdx.x * EVAL_BLOCK + k * POLY_THREADS + t;
            if (i < len) { Fr a0 = f[i], b0 = pwp[i]; accu = U::add_n(accu, U::mul_nr(U::unpack(a0.l), U::unpack(b0.l))); }
        }
        Fr acc;
        U::template canon<8>(accu).pack(acc.l);
        sm[t] = acc;
        __syncthreads();
        for (uint32_t d = POLY_THREADS / 2; d >= 1; d >>= 1) {
            if (t < d) { acc = acc + sm[t + d]; sm[t] = acc; }
            __syncthreads();
        }
        if (t == 0) partial[p * nblocks + blockIdx.x] = acc;
    }
};
template <class FR>
__global__ void __launch_bounds__(POLY_THREADS) eval_partial_kernel(EvalArgs<FR> a, const Fe<FR>* __restrict__ pw,
                                                                    uint32_t nblocks, Fe<FR>* __restrict__ partial) { EvalPartialK<FR>::run(a, pw, nblocks, partial); }

template <class FR>
struct EvalFinalK {
    static __device__ __forceinline__ void run(const Fe<FR>* __restrict__ partial, uint32_t nblocks,
                                                                  Fe<FR>* __restrict__ result) {
        wave_priority<APK_PRIO_FR>();
        using Fr = Fe<FR>;
        __shared__ Fr sm[POLY_THREADS];
        const int p = blockIdx.x;
        const uint32_t t = threadIdx.x;
        Fr acc = Fr::zero();
        for (uint32_t i = t; i < nblocks; i += POLY_THREADS) acc = acc + partial[p * nblocks + i];
        sm[t] = acc;
        __syncthreads();
        for (uint32_t d = POLY_THREADS / 2; d >= 1; d >>= 1) {
            if (t < d) { acc = acc + sm[t + d]; sm[t] = acc; }
            __syncthreads();
        }
        if (t == 0) result[p] = acc;
    }
};
template <class FR>
__global__ void __launch_bounds__(POLY_THREADS) eval_final_kernel(const Fe<FR>* __restrict__ partial, uint32_t nblocks,
                                                                  Fe<FR>* __restrict__ result) { EvalFinalK<FR>::run(partial, nblocks, result); }

// ---- (omega z)^i from z^i: out[i] = in[i] * omega^(+-i), one product per element where the square-and-multiply walk of
// powers_kernel costs ~6.  twu[j] = omega^j in the radix R' for j < n / 2; omega^(n/2) = -1.  inverse: omega^-i = omega^(n - i mod n).
template <class FR>
struct DerivePowers { const Fe<FR>* in[2]; Fe<FR>* out[2]; int inverse[2]; };
template <class FR>
struct DerivePowersK {
    static __device__ __forceinline__ void run(DerivePowers<FR> a, const Fe<FR>* __restrict__ twu, uint32_t n, uint32_t count) {
        wave_priority<APK_PRIO_FR>();
        using Fr = Fe<FR>;
        using U = FeU<FR>;
        const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= count) return;
        const int p = blockIdx.y;
        uint32_t e = i & (n - 1u);                       // omega^n = 1
        if (a.inverse[p]) e = (n - e) & (n - 1u);
        const bool neg = e >= n / 2;
        Fr w = twu[e & (n / 2 - 1u)], v = a.in[p][i];
        U r = U::template canon<1>(U::mul_nr(U::unpack(w.l), U::unpack(v.l)));
        Fr o;
        r.pack(o.l);
        if (neg) o = Fr::neg(o);
        a.out[p][i] = o;
    }
};
template <class FR>
__global__ void __launch_bounds__(POLY_THREADS) derive_powers_kernel(DerivePowers<FR> a, const Fe<FR>* __restrict__ twu, uint32_t n, uint32_t count) { DerivePowersK<FR>::run(a, twu, n, count); }

// ---- out[i] = sum_k coef[k] * f_k[i]  (linearised polynomial, folded opening polynomial) -------------------
constexpr int LC_MAX = 20;   // folded opening polynomial: 11 + k terms of the linearised polynomial, 5 + k folded ones (k <= 2)
template <class FR>
struct LinCombArgs {
    const Fe<FR>* f[LC_MAX];
    uint32_t len[LC_MAX];
    Fe<FR> coef[LC_MAX];
    int count;
    uint32_t out_len;
};

template <class FR>
struct LincombK {
    static __device__ __forceinline__ void run(LinCombArgs<FR> a, Fe<FR>* __restrict__ out) {
        wave_priority<APK_PRIO_FR>();
        using Fr = Fe<FR>;
        // unsaturated limbs, lazily (like the quotient kernel): the coefficients arrive as 32 c (the product's radix R' = 32 R), every
        // product is below 2 r, up to LC_MAX of them add up below 64 r, ONE canonicalisation at the end - 206 instead of ~300
        // instructions per term
        using U = FeU<FR>;
        static_assert(U::HEADROOM >= 64 && LC_MAX * 2 <= 64, "the sum of LC_MAX products below 2 r each must stay below 64 r");
        uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= a.out_len) return;
        U acc = U::zero();
        for (int k = 0; k < a.count; k++)
            if (i < a.len[k]) { Fr v = a.f[k][i]; acc = U::add_n(acc, U::mul_nr(U::unpack(a.coef[k].l), U::unpack(v.l))); }
        const U res = U::template canon<32>(acc);
        Fr w;
        res.pack(w.l);
        out[i] = w;
    }
};
template <class FR>
__global__ void __launch_bounds__(POLY_THREADS) lincomb_kernel(LinCombArgs<FR> a, Fe<FR>* __restrict__ out) { LincombK<FR>::run(a, out); }

// q[j] = zinv_pw[j+1] * suffix[j+1], j < len-1   where suffix[i] = sum_{k>=i} f[k] z^k
template <class FR>
struct DivFinishK {
    static __device__ __forceinline__ void run(const Fe<FR>* __restrict__ suffix,
                                                                  const Fe<FR>* __restrict__ zinv_pw, uint32_t len,
                                                                  Fe<FR>* __restrict__ q) {
        wave_priority<APK_PRIO_FR>();
        uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
        if (j + 1 >= len) return;
        q[j] = suffix[j + 1] * zinv_pw[j + 1];
    }
};
template <class FR>
__global__ void __launch_bounds__(POLY_THREADS) div_finish_kernel(const Fe<FR>* __restrict__ suffix,
                                                                  const Fe<FR>* __restrict__ zinv_pw, uint32_t len,
                                                                  Fe<FR>* __restrict__ q) { DivFinishK<FR>::run(suffix, zinv_pw, len, q); }

// The quotient is a polynomial of degree < 3(n+2) iff the witness satisfies the circuit: OR-reduce ALL the words of the
// coefficients above it (h[3(n+2) .. 4n)).  A non-zero word stamps the proof's epoch into *flag (atomicMax: no zeroing between
// proofs, nothing is written in the normal case); the host compares the flag with the epoch after the stream sync.
template <int DUMMY>
struct TailNonzeroK {
    static __device__ __forceinline__ void run(const uint4* __restrict__ words, uint32_t count4, uint32_t epoch, uint32_t* __restrict__ flag) {
        uint32_t acc = 0;
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count4; i += gridDim.x * blockDim.x) {
            const uint4 v = words[i];
            acc |= v.x | v.y | v.z | v.w;
        }
        // every writer stores the same word (the slot's epochs only grow and its proofs run one after the other), so a plain store
        // does: `flag` may be host memory, where a device atomic is not a given
        if (__ballot(acc != 0) != 0 && (threadIdx.x & 63) == 0) *reinterpret_cast<volatile uint32_t*>(flag) = epoch;
    }
};
template <int DUMMY>
__global__ void __launch_bounds__(256) tail_nonzero_kernel(const uint4* __restrict__ words, uint32_t count4, uint32_t epoch, uint32_t* __restrict__ flag) { TailNonzeroK<DUMMY>::run(words, count4, epoch, flag); }

// z == 0 fallback: q[j] = f[j+1]
template <class FR>
struct ShiftDownK {
    static __device__ __forceinline__ void run(const Fe<FR>* __restrict__ f, uint32_t len, Fe<FR>* __restrict__ q) {
        uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
        if (j + 1 < len) q[j] = f[j + 1];
    }
};
template <class FR>
__global__ void shift_down_kernel(const Fe<FR>* __restrict__ f, uint32_t len, Fe<FR>* __restrict__ q) { ShiftDownK<FR>::run(f, len, q); }

}  // namespace apk
