// Radix-2 NTT / iNTT over Fr, natural order in -> natural order out, gfx950.
//
// What it replaces: gnark-crypto v0.20.1 ecc/<curve>/fr/fft `Domain.FFT / FFTInverse` (+ coset variants)
// [UPSTREAM, not vendored], reached only through plonk.Prove (/root/reference/algoplonk.go:89).  The
// domain generator and coset shift are the VK's `Generator` / `CosetShift`
// (/root/reference/verifier/templateLogicSigBN254.go:57,68).  SURVEY.md §8a row a6.
//
// Design: decimation-in-time over a bit-reversed gather.  The log2(N) butterfly stages are cut into
// passes of <= NTT_PASS_BITS stages; inside a pass a workgroup owns a tile of NTT_TILE = 2048 elements in
// LDS (72 KiB: 36-byte unsaturated-limb elements) - 2^s "mid" positions x C adjacent groups, so global traffic is C*32-byte contiguous runs -
// and runs its s stages out of LDS.  Each pass is one read + one write of the vector (64 B/element,
// SURVEY.md §8d).  Optional fused pre-/post-multiplication by a table (coset powers, 1/N) and a zero-padded
// short input remove the separate scaling / padding passes.
#pragma once
#include "ff.h"
#include "ffu.h"

namespace apk {

constexpr int NTT_TILE_LOG = 11;  // largest tile: 2048 elements * 32 B = 64 KiB LDS (used for log_n > 19)
constexpr int NTT_PASS_BITS = 10; // max stages per pass (C = 2^(11-s) adjacent groups -> C*32-B contiguous runs)
constexpr int NTT_THREADS = 256;

__device__ __forceinline__ uint32_t bitrev32(uint32_t x, int bits) { return __brev(x) >> (32 - bits); }

// up to NTT_MAX_BATCH same-size transforms per launch (blockIdx.y): the three wire polynomials share launches,
// which triples the workgroup count of these latency-bound small transforms
constexpr int NTT_MAX_BATCH = 4;     // ... of ONE proof
constexpr int NTT_ARGS_MAX = 16;     // ... per launch: a gang of up to four proofs shares its launches (gang.h)
struct NttBatch {
    const void* in[NTT_ARGS_MAX];
    void* out[NTT_ARGS_MAX];
    uint32_t in_len[NTT_ARGS_MAX];   // elements >= in_len read as zero (first pass only)
    void* wide[NTT_ARGS_MAX];        // N unsaturated-limb elements (36 B): what the passes hand to each other
};

struct NttPassArgs {
    int log_n;
    int tile_log;        // log2(elements per workgroup tile)
    int t0, t1;          // stages [t0, t1)
    int first, last;     // first pass gathers bit-reversed + pre-multiplies; last pass post-multiplies
    uint32_t out_len;    // last pass: elements >= out_len are not written
    int radix4;          // two stages per LDS round trip (large transforms; see the kernel)
    int tw_shift;        // the twiddle table belongs to a transform 2^tw_shift times this size (sub-coset transforms: tw[j << tw_shift])
};

// Inside the tile the elements are UNSATURATED-limb values (ffu.h, 9 x 29 bits for both scalar fields) handled lazily:
//   * data stays in gnark's Montgomery radix R = 2^256; the twiddle / scaling TABLES are stored in the radix R' = 2^261 of
//     the carry-free product (w * R'), so  mul_nr(w R', x R) = w x R  needs no domain conversion;
//   * a butterfly is one product without its final subtraction (below 2p), one limb-wise sum and one difference kept
//     positive by adding 2p: no comparisons.  Values grow by at most 2p per stage (below 24p after 10 stages, R'/p >= 71);
//   * between the passes of one transform the elements stay in that form (NttBatch::wide, 36 bytes each, values below
//     (2 + 2 log2 N) p < R'); canonical 8-word elements are read by the first pass and written by the last one only.
// Against the saturated-limb butterfly (128 v_mad_u64_u32 + 128 v_addc per product, two conditional subtractions) this is
// 171 mads + ~150 other instructions.
//
// tw[j] = w^j * R' for j < N/2 (w = omega or omega^-1); pre / post / scale tables likewise in R' form
// TWU (round 6): the twiddle table holds the elements already in the tile's unsaturated-limb form (FeU, 36 bytes: `tw` then points at
// FeU<FR> records) - a butterfly's twiddle costs a 36-byte load instead of a 32-byte load + ~18 instructions of unpacking.
template <class FR, bool TWU = false>
__global__ void __launch_bounds__(NTT_THREADS) ntt_pass_kernel(NttBatch nb, const Fe<FR>* __restrict__ tw,
                                                               const Fe<FR>* __restrict__ pre,   // or null
                                                               const Fe<FR>* __restrict__ post,  // or null
                                                               const Fe<FR>* __restrict__ scale, // or null: one element
                                                               NttPassArgs a) {
    wave_priority<APK_PRIO_FR>();
    using Fr = Fe<FR>;
    using Fu = FeU<FR>;
    static_assert(Fu::HEADROOM >= 64, "values reach (4 + 2 log2 N) p < 64 p over the stages of a transform (N <= 2^29)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Fu* sm = reinterpret_cast<Fu*>(smem_raw);
    const Fu* __restrict__ twu = reinterpret_cast<const Fu*>(tw);
    auto twiddle = [&](uint32_t idx) -> Fu {
        if constexpr (TWU) return twu[idx];
        else { Fr w = tw[idx]; return Fu::unpack(w.l); }
    };
    Fr* __restrict__ out = reinterpret_cast<Fr*>(nb.out[blockIdx.y]);
    const Fr* __restrict__ in = reinterpret_cast<const Fr*>(nb.in[blockIdx.y]);
    Fu* __restrict__ wide = reinterpret_cast<Fu*>(nb.wide[blockIdx.y]);
    const uint32_t in_len = nb.in_len[blockIdx.y];
    const int s = a.t1 - a.t0;
    const int tile_log = a.tile_log;
    const int clog = tile_log - s;  // log2(groups per tile)
    const uint32_t C = 1u << clog;
    const uint32_t tile_elems = 1u << tile_log;
    const uint32_t g0 = blockIdx.x << clog;  // first group id of this tile
    const uint32_t lomask = (1u << a.t0) - 1u;

    const uint32_t NT = blockDim.x;
    // load: element e of the tile = (mid, c) with c fastest
    for (uint32_t e = threadIdx.x; e < tile_elems; e += NT) {
        uint32_t c = e & (C - 1), mid = e >> clog;
        uint32_t g = g0 + c;
        uint32_t idx = ((g >> a.t0) << a.t1) | (mid << a.t0) | (g & lomask);
        Fu v;
        if (a.first) {
            uint32_t src = bitrev32(idx, a.log_n);
            if (src < in_len) {
                Fr raw = in[src];
                v = Fu::unpack(raw.l);
                if (pre) { Fr pw = pre[src]; v = Fu::mul_nr(Fu::unpack(pw.l), v); }
                // a polynomial LONGER than the transform (sub-coset evaluations, backend_impl.h SubCoset: n + 3 coefficients on
                // a coset of 4n / G points) folds onto its low coefficients: x^N = const on the coset, and the constant is
                // already inside `pre`.  At most three terms below 2p each, brought back below p for the stage-0 butterfly.
                if (in_len > (1u << a.log_n)) {
                    for (uint32_t s2 = src + (1u << a.log_n); s2 < in_len; s2 += 1u << a.log_n) {
                        Fr raw2 = in[s2];
                        Fu v2 = Fu::unpack(raw2.l);
                        if (pre) { Fr pw = pre[s2]; v2 = Fu::mul_nr(Fu::unpack(pw.l), v2); }
                        v = Fu::add_n(v, v2);
                    }
                    v = Fu::template canon<4>(v);
                }
            } else {
                v = Fu::zero();
            }
        } else {
            v = wide[idx];
        }
        sm[e] = v;
    }
    __syncthreads();
    int q = 0;
    // Two stages per LDS round trip (radix 4): a lane holds the four elements {mid | b0 2^q | b1 2^(q+1)} of its group in
    // registers, runs stage t on (x0,x1), (x2,x3) and stage t+1 on (y0,y2), (y1,y3).  In a prime field this saves no PRODUCT
    // (the factor i = w^(N/4) is an ordinary element: 4 products per 4 butterflies either way) - it saves what surrounds the
    // products: one of four twiddle loads + unpacks (stage t's two pairs share theirs; stage t+1's are tw[k] and tw[k + N/4]),
    // half the LDS traffic and three quarters of the index arithmetic.  Measured (round 3, one box): BLS12-381 2^21 8.5 -> 7.7 ms
    // of NTT per proof; at 2^17 the saturated rate does not move and a lone proof's transforms get SLOWER (0.45 -> 0.52 ms:
    // half the lanes per tile, and these launches are latency-bound) - so the host asks for it above 2^19 only.
    // padded: the zero-padded 4n transforms feed their first radix-4 step groups with one non-zero element (wave-uniform).
    const bool padded = a.first && in_len < (1u << a.log_n);
    const uint32_t nq = tile_elems >> 2;
    for (; a.radix4 && q + 2 <= s; q += 2) {
        const int t = a.t0 + q;
        for (uint32_t u4 = threadIdx.x; u4 < nq; u4 += NT) {
            const uint32_t c = u4 & (C - 1), pr = u4 >> clog;
            const uint32_t low = pr & ((1u << q) - 1u);
            const uint32_t m00 = ((pr >> q) << (q + 2)) | low;
            const uint32_t g = g0 + c;
            const uint32_t imod = (low << a.t0) | (g & lomask);   // index mod 2^t
            const uint32_t e00 = (m00 << clog) | c, st1 = 1u << (q + clog);
            const uint32_t e01 = e00 + st1, e10 = e00 + 2 * st1, e11 = e01 + 2 * st1;
            Fu x0 = sm[e00], x1 = sm[e01], x2 = sm[e10], x3 = sm[e11];
            if (padded && q == 0 && x1.is_zero() && x2.is_zero() && x3.is_zero()) {
                sm[e01] = x0; sm[e10] = x0; sm[e11] = x0;    // all four outputs are x0
                continue;
            }
            if (t != 0) {   // stage t: exponent imod * N / 2^(t+1); stage 0 has unit twiddles and fresh operands (below 2p)
                const Fu w1 = twiddle((imod << (a.log_n - 1 - t)) << a.tw_shift);
                x1 = Fu::mul_nr(w1, x1);
                x3 = Fu::mul_nr(w1, x3);
            }
            const Fu y0 = Fu::add_n(x0, x1);
            const Fu y1 = Fu::template sub_k<2>(x0, x1);
            Fu y2 = Fu::add_n(x2, x3);
            Fu y3 = Fu::template sub_k<2>(x2, x3);
            // stage t+1: exponents imod * N / 2^(t+2) and that + N/4
            const uint32_t k2 = imod << (a.log_n - 2 - t);
            {
                y3 = Fu::mul_nr(twiddle((k2 + (1u << (a.log_n - 2))) << a.tw_shift), y3);
            }
            if (t != 0) {
                y2 = Fu::mul_nr(twiddle(k2 << a.tw_shift), y2);
                sm[e00] = Fu::add_n(y0, y2);
                sm[e10] = Fu::template sub_k<2>(y0, y2);
            } else {   // w2 = 1: y2 = x2 + x3 is below 4p, not a fresh product
                sm[e00] = Fu::add_n(y0, y2);
                sm[e10] = Fu::template sub_k<4>(y0, y2);
            }
            sm[e01] = Fu::add_n(y1, y3);
            sm[e11] = Fu::template sub_k<2>(y1, y3);
        }
        __syncthreads();
    }
    // remaining stage(s), one butterfly at a time; butterfly id -> (pair index within group, c)
    const uint32_t nbf = tile_elems >> 1;
    for (; q < s; q++) {
        const int t = a.t0 + q;
        for (uint32_t bf = threadIdx.x; bf < nbf; bf += NT) {
            uint32_t c = bf & (C - 1), pr = bf >> clog;
            uint32_t low = pr & ((1u << q) - 1u);
            uint32_t mid0 = ((pr >> q) << (q + 1)) | low;
            uint32_t mid1 = mid0 | (1u << q);
            uint32_t g = g0 + c;
            // twiddle exponent: (index mod 2^t) * N / 2^(t+1)
            uint32_t imod = (low << a.t0) | (g & lomask);
            uint32_t tidx = (imod << (a.log_n - 1 - t)) << a.tw_shift;
            uint32_t e0 = (mid0 << clog) | c, e1 = (mid1 << clog) | c;
            Fu u = sm[e0];
            Fu v = sm[e1];
            // no product in stage 0 (unit twiddles; the operands are fresh, i.e. below 2p as the difference needs) and for zero
            // operands (the first two stages of the zero-padded 4n transforms) - both are wave-uniform in practice
            if (t != 0 && !(padded && v.is_zero())) v = Fu::mul_nr(twiddle(tidx), v);
            sm[e0] = Fu::add_n(u, v);
            sm[e1] = Fu::template sub_k<2>(u, v);
        }
        __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < tile_elems; e += NT) {
        uint32_t c = e & (C - 1), mid = e >> clog;
        uint32_t g = g0 + c;
        uint32_t idx = ((g >> a.t0) << a.t1) | (mid << a.t0) | (g & lomask);
        Fu v = sm[e];
        if (!a.last) { wide[idx] = v; continue; }
        if (idx >= a.out_len) continue;
        if (post) { Fr pw = post[idx]; v = Fu::mul_nr(Fu::unpack(pw.l), v); }
        if (scale) { Fr sc = scale[0]; v = Fu::mul_nr(Fu::unpack(sc.l), v); }
        v = (post || scale) ? Fu::template canon<1>(v) : Fu::template canon<32>(v);
        Fr o;
        v.pack(o.l);
        out[idx] = o;
    }
}

// table[i] (Fe, the radix R' twiddles) -> FeU records for the TWU form of the pass kernel
template <class FR>
__global__ void __launch_bounds__(256) unpack_table_kernel(const Fe<FR>* __restrict__ in, FeU<FR>* __restrict__ out, uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fe<FR> v = in[i];
    out[i] = FeU<FR>::unpack(v.l);
}

// tw[j] = w^j, j < count: each thread exponentiates its block start then walks
constexpr int POWERS_MAX_BATCH = 4;
template <class FR>
struct PowersBatch {
    Fe<FR>* out[POWERS_MAX_BATCH];
    Fe<FR> w[POWERS_MAX_BATCH];
    Fe<FR> scale[POWERS_MAX_BATCH];
};
template <class FR>
struct PowersK {
    static __device__ __forceinline__ void run(PowersBatch<FR> pb, uint32_t count) {
        using Fr = Fe<FR>;
        using U = FeU<FR>;
        constexpr uint32_t PER = 8;
        Fr* __restrict__ out = pb.out[blockIdx.y];
        uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
        uint32_t start = t * PER;
        if (start >= count) return;
        // The whole dependent chain (square-and-multiply to w^start, then PER steps) runs on the carry-free product: w is taken
        // into the product's own Montgomery domain R' (closed under mul), the block's first power comes back to gnark's radix R
        // with one product, and the walk multiplies R-radix values by w R'.
        const U wu = U::from_fe(pb.w[blockIdx.y]);           // w R'
        U r = U::one();                                       // R'
        bool started = false;
        for (int b = 31; b >= 0; b--) {
            if (started) r = U::sqr(r);
            if ((start >> b) & 1u) { r = started ? U::mul(r, wu) : wu; started = true; }
        }
        Fr first = r.to_fe();                                 // w^start R
        U cur = U::mul(U::unpack(first.l), U::from_fe(pb.scale[blockIdx.y]));   // scale w^start, R radix, canonical
        for (uint32_t k = 0; k < PER && start + k < count; k++) {
            Fr o;
            cur.pack(o.l);
            out[start + k] = o;
            cur = U::mul(wu, cur);
        }
    }
};
template <class FR>
__global__ void __launch_bounds__(256) powers_kernel(PowersBatch<FR> pb, uint32_t count) { PowersK<FR>::run(pb, count); }

}  // namespace apk
