// Pippenger multi-scalar multiplication over G1 for a FIXED base set (the KZG SRS), gfx950.
//
// What it replaces: gnark-crypto v0.20.1 `G1Affine.MultiExp` [UPSTREAM, not vendored], reached from
// /root/reference/algoplonk.go:89 (plonk.Prove -> kzg.Commit / kzg.Open) and from
// /root/reference/setup/setup.go:107,149 (plonk.Setup -> 8+k trace commitments).  SURVEY.md §8a row a4.
//
// MI355X-first design (NOT gnark's goroutine-per-window CPU algorithm):
//   * the SRS is fixed per circuit and HBM is 288 GB, so every window's multiple 2^(c*j) * P_i is
//     precomputed once ("windowed point tables").  All W windows then feed ONE set of 2^(c-1) buckets:
//     no per-window reduction, no final double-and-add over windows.
//   * scalars are recoded into signed digits (window widths c or c-1 bits, balanced); (digit, point) pairs are
//     counting-sorted by bucket with LDS-private histograms (slice histogram -> column scan -> bucket scan -> scatter);
//   * bucket accumulation is cut into fixed-size work units (<= MSM_UNIT entries of one bucket per lane) so a
//     skewed bucket cannot serialise a wave; point arithmetic runs on unsaturated limbs (ffu.h: carry-free
//     v_mad_u64_u32 chains); unit partials are merged by 2..16-lane groups with wavefront shuffles;
//   * the weighted bucket sum  sum_k k*B_k  is reduced in two levels (row / column sums of the bucket array, then
//     bit-wise weighted sums of <= 256 elements): 2 additions per bucket, critical path ~log2(#buckets) + c point
//     operations instead of the 2*#buckets of a running sum.
// Several MSMs over the same bases (e.g. [L],[R],[O]) run as one batch: bucket id = msm*NB + bucket.
#pragma once
#include "ec.h"

namespace apk {

constexpr int MSM_MAX_BATCH = 4;        // MSMs per batch of ONE proof (its three wire / quotient commitments + one to spare)
constexpr int MSM_ARGS_MAX = 16;        // MSMs per LAUNCH SEQUENCE: a gang of up to four proofs shares its launches (gang.h); what a
                                        // workspace is sized for is the context's choice (backend_impl.h ws_batch_)
constexpr int MSM_UNIT = 16;        // entries per full accumulation work unit; a run-time value in the kernels (APK_MSM_UNIT).
                                    // With the remainder units sorted, 2^17: 16 -> 360, 24 -> 357, 32 -> 349, 64 -> 328 proofs/s
                                    // (longer units quantise worse over the 1024 SIMDs and halve the lanes of a lone MSM)
constexpr int MSM_UNIT_MIN = 16, MSM_UNIT_MAX = 64;
constexpr int MSM_UNIT_SMALL = 4;                       // shortest unit of a batch that cannot fill the SIMDs at MSM_UNIT_MIN
constexpr uint64_t MSM_SMALL_ENTRIES = 1ull << 20;      // ... and the most entries such a batch has (16 x 65 536 lanes)
constexpr int MSM_COMBINE_LANES = 16;
constexpr uint32_t MSM_HEAVY_UNITS = 512;  // unit partials above which a bucket is ALWAYS merged by a whole workgroup
// ... and below it when the bucket is far above what its lanes are sized for: heavy from max(32, 4 x per_lane x lanes) partials.
// (Round 5: on a sparse input - Lagrange-basis wires of a bit-heavy witness - most buckets hold ONE partial, so the merge runs one
// lane per bucket, and the bucket of the digit 1 with ~500 partials was walked by that one lane: 2.8 ms of a lone BLS12-381 2^14
// proof.  Uniform inputs never get there: their buckets agree with the average.)
__device__ __forceinline__ uint32_t msm_heavy_threshold(int lanes_log, uint32_t per_lane) {
    const uint32_t t = (4u * per_lane) << lanes_log;
    return t < 32u ? 32u : (t > MSM_HEAVY_UNITS ? MSM_HEAVY_UNITS : t);
}

// Signed-digit windows.  Widths differ by at most one bit (c or c-1) so the BITS+1 scalar bits are spread evenly:
// with equal widths the top window can be left with 1-3 significant bits, and every scalar then lands in the same
// two or three buckets (measured: 9x slower bucket merge at c = 12 for uniform scalars).
constexpr int MSM_MAX_WINDOWS = 40;
struct MsmWindows {
    int W;
    uint16_t off[MSM_MAX_WINDOWS + 1];  // first bit of window j; off[W] = BITS + 1
    uint8_t width[MSM_MAX_WINDOWS];
};

struct MsmBatchArgs {
    const void* scalars[MSM_ARGS_MAX];   // device, Fr Montgomery, len[b] elements
    uint32_t len[MSM_ARGS_MAX];
    uint32_t offset[MSM_ARGS_MAX];       // first base index used by msm b (bases [offset, offset+len))
    uint32_t batch;
    uint32_t plain;                      // the table holds the bases themselves, not R^-1 * P: the sort leaves the Montgomery form first, so
                                         // that SMALL values have few non-zero digits (Lagrange-basis wire commitments: backend_impl.h)
};

// ---- signed-digit recoding ---------------------------------------------------------------------------
// canonical scalar limbs -> W digits d_j in [-2^(w_j - 1), 2^(w_j - 1)], s = sum d_j 2^(off_j)  (inside msm_digits_kernel)

// Counting sort of the (digit, point) pairs by bucket WITHOUT global atomics (device-scope atomics leave the XCD's
// L2 and were 25 % of an MSM): a workgroup owns a contiguous slice of the scalars and a private LDS histogram.
//   SCATTER = false : counts[(b*G + g)*nb + k] = entries of slice g of msm b that fall into bucket k
//   SCATTER = true  : LDS cursors start at offsets[b*nb+k] + (exclusive prefix of counts over g); every entry takes
//                     the next slot of its bucket with an LDS atomic and is written to `sorted`
// c = 17 (2^16 buckets) does not fit 32-bit counters into the 160 KB of LDS: from MSM_PACKED_NB buckets up two 16-bit counters
// share a word (a slice holds <= 4096 scalars x 17 windows < 2^16 entries, so a half never overflows into its neighbour);
// the scatter pass then keeps only the entry's RANK inside (slice, bucket) in LDS and adds the two bases - the bucket's global
// offset and the slice's prefix inside the bucket - from global memory (both L2 resident).
#ifndef APK_MSM_PACKED_NB
#define APK_MSM_PACKED_NB 65536
#endif
constexpr uint32_t MSM_PACKED_NB = APK_MSM_PACKED_NB;
constexpr int MSM_DIGITS_THREADS = 1024;  // per sort workgroup: the slice's LDS atomics and scattered stores are latency-bound
// PLAIN (every sort kernel): the table holds the bases themselves, so the scalars leave the Montgomery form before they are recoded
// (MsmBatchArgs::plain).  A template parameter, not a run-time test: as a uniform branch the compiler computed the conversion on
// BOTH paths and selected (msm_part1_kernel 28 -> 40 VGPRs, +44 % time under load at BLS12-381 2^14: round 5 A/B against round 4).
template <class FR, bool SCATTER, bool PLAIN = false>
__global__ void __launch_bounds__(MSM_DIGITS_THREADS) msm_digits_kernel(MsmBatchArgs a, MsmWindows win, uint32_t nb, uint32_t n_max, uint32_t G,
                                                         uint32_t* __restrict__ counts,         // [batch][G][nb]
                                                         const uint32_t* __restrict__ offsets,  // SCATTER only
                                                         uint32_t* __restrict__ sorted) {
    wave_priority<APK_PRIO_SORT>();
    using Fr = Fe<FR>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t* lds = reinterpret_cast<uint32_t*>(smem_raw);
    const uint32_t g = blockIdx.x, b = blockIdx.y;
    uint32_t* row = counts + ((size_t)b * G + g) * nb;
    const bool packed = nb >= MSM_PACKED_NB;   // uniform
    if (packed) { for (uint32_t k = threadIdx.x; k < nb / 2; k += blockDim.x) lds[k] = 0u; }
    else { for (uint32_t k = threadIdx.x; k < nb; k += blockDim.x) lds[k] = SCATTER ? offsets[b * nb + k] + row[k] : 0u; }
    __syncthreads();
    const uint32_t len = a.len[b];
    const uint32_t per = (len + G - 1) / G;
    const uint32_t lo = min(g * per, len), hi = min(lo + per, len);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        // The scalars arrive in gnark's Montgomery form a * R and are used AS THEY ARE: the windowed tables hold the multiples of
        // R^-1 * P_i (msm_table_kernel, once per circuit), so sum (a_i R) * (R^-1 P_i) = sum a_i P_i and neither sort pass pays a
        // Montgomery product per scalar to leave the form (it was ~45 % of the two passes' instructions).
#ifndef APK_MSM_NO_RINV
        Fr s = reinterpret_cast<const Fr*>(a.scalars[b])[i];
        if constexpr (PLAIN) s = Fr::from_mont(s);
#else
        Fr s = Fr::from_mont(reinterpret_cast<const Fr*>(a.scalars[b])[i]);
#endif
        uint32_t carry = 0;
        const uint32_t base_idx = a.offset[b] + i;
        // The windows are consecutive bit fields (off[j+1] = off[j] + width[j]): stream the limbs through a 64-bit buffer and
        // peel windows off its low end - no per-window limb selection (the limbs must stay in registers, so indexing them by
        // a run-time window offset costs a compare-select chain per window).  j is uniform across the wave.
        uint64_t buf = 0;
        int avail = 0, j = 0;
#if defined(APK_DEBUG_SCATTER)
        uint32_t dbg = 0;
#endif
#pragma unroll
        for (int li = 0; li < Fr::N; li++) {
            buf |= (uint64_t)s.l[li] << avail;
            avail += 32;
            while (j < win.W && (avail >= (int)win.width[j] || li == Fr::N - 1)) {   // after the last limb the buffer's zeros serve the top windows
                const int c = win.width[j];
                const uint32_t half = 1u << (c - 1);
                uint32_t d = ((uint32_t)buf & ((1u << c) - 1u)) + carry;
                buf >>= c;
                avail -= c;
                uint32_t neg = 0;
                if (d > half) { d = (1u << c) - d; neg = 1; carry = 1; }  // d in (half, 2^c] -> -(2^c - d)
                else carry = 0;
                if (d != 0) {  // d == 0 also covers the d == 2^c case (digit 0, carry 1)
                    const uint32_t k = d - 1;
                    if (packed) {
                        const uint32_t sh = (k & 1u) * 16u;
                        const uint32_t old = atomicAdd(&lds[k >> 1], 1u << sh);
                        if (SCATTER) {
                            const uint32_t pos = offsets[b * nb + k] + row[k] + ((old >> sh) & 0xffffu);
                            sorted[pos] = ((uint32_t)j * n_max + base_idx) | (neg << 31);
                        }
                    } else if (!SCATTER) {
                        atomicAdd(&lds[k], 1u);
                    } else {
#if !defined(APK_DEBUG_SCATTER)
                        uint32_t pos = atomicAdd(&lds[k], 1u);
                        sorted[pos] = ((uint32_t)j * n_max + base_idx) | (neg << 31);
#elif APK_DEBUG_SCATTER == 1      /* measurement builds: the atomics without the stores ... */
                        dbg ^= atomicAdd(&lds[k], 1u);
#elif APK_DEBUG_SCATTER == 2      /* ... the atomics with stores next to each other ... */
                        dbg ^= atomicAdd(&lds[k], 1u);
                        sorted[((size_t)b * n_max + i) * win.W + j] = ((uint32_t)j * n_max + base_idx) | (neg << 31);
#else                             /* ... the scattered stores without the atomics */
                        sorted[(k * 64u + (i & 63u)) % (nb * 64u) + b * nb * 64u] = ((uint32_t)j * n_max + base_idx) | (neg << 31);
#endif
                    }
                }
                j++;
            }
        }
#if defined(APK_DEBUG_SCATTER)
        if (SCATTER && dbg == 0xdeadbeefu) sorted[i] = dbg;   // keeps the atomics' results alive
#endif
    }
    if (!SCATTER) {
        __syncthreads();
        if (packed) { for (uint32_t k = threadIdx.x; k < nb; k += blockDim.x) row[k] = (lds[k >> 1] >> ((k & 1u) * 16u)) & 0xffffu; }
        else { for (uint32_t k = threadIdx.x; k < nb; k += blockDim.x) row[k] = lds[k]; }
    }
}

// per bucket: exclusive prefix of the slice counts over g (in place) and the bucket total
template <int DUMMY>
__global__ void __launch_bounds__(256) msm_colscan_kernel(uint32_t* __restrict__ counts, uint32_t nb, uint32_t G, uint32_t total_buckets,
                                                          uint32_t* __restrict__ hist) {
    wave_priority<APK_PRIO_SORT>();
    const uint32_t kk = blockIdx.x * blockDim.x + threadIdx.x;  // b*nb + k
    if (kk >= total_buckets) return;
    const uint32_t b = kk / nb, k = kk % nb;
    uint32_t run = 0;
    uint32_t* col = counts + (size_t)b * G * nb + k;
    uint32_t g = 0;
    for (; g + 8 <= G; g += 8) {   // the 8 loads go out together: the loop is latency-bound (one counter per slice)
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = col[(size_t)(g + j) * nb];
#pragma unroll
        for (int j = 0; j < 8; j++) { col[(size_t)(g + j) * nb] = run; run += v[j]; }
    }
    for (; g < G; g++) {
        uint32_t v = col[(size_t)g * nb];
        col[(size_t)g * nb] = run;
        run += v;
    }
    hist[kk] = run;
}

// ---- two-level counting sort (run_msm_body, APK_MSM_SORT2) ----------------------------------------------------------------
// The one-level sort above scatters every (digit, point) pair straight to its bucket: a slice of 2 048 scalars has one entry per
// bucket on average, so its 32 k stores are 32 k isolated 4-byte writes, and tools/knockout.py prices that pass at 8 % of an MSM
// at saturation for 1 % of its instructions.  Here the pairs are first dealt to PARTITIONS of 2^pb_log neighbouring buckets
// (msm_part_kernel: a slice writes its pairs of one partition side by side), then one workgroup per partition counting-sorts its
// ~16 k pairs in LDS order and writes them out (msm_part_sort_kernel).  The second level also produces the per-bucket counts, so
// the column scan goes.  Both levels use the same compact numbering, so a partition's pairs occupy the same index range before
// and after.
// Every scatter of the two levels happens INSIDE an LDS tile, and whole lines leave the tile (a wave store whose lanes hit 64
// different lines costs 64 L2 requests however near the lines are: DESIGN section 5).  Intermediate entries are packed words:
// bits [0, idx_bits) the table index j * n_max + i, bits [idx_bits, idx_bits + pb_log) the bucket's position inside its partition,
// bit 31 the sign.  Round 3 fixed idx_bits = 22 and pb_log = 8 (2^17 bases at 16 windows); round 4 picks them per context
// (MsmPartCfg): the index takes the bits it needs (26 for BLS12-381 2^21 x 16 windows - BASELINE configs[4]) and the partition
// count grows until a partition's entries fit the second level's LDS tile (2 048 partitions of 16 buckets there).
constexpr uint32_t MSM_PART_MAX = 8192;     // partitions per MSM (round 5: 2 048 -> 8 192 for the windows above 17 bits and for 2^24 bases: 2^19 buckets in partitions of 64)
constexpr uint32_t MSM_PART_GMAX = 16384;   // slices per MSM in the first level (2^24 bases in slices of ~2 044 scalars)
constexpr uint32_t MSM_LDS_WORDS = 40960;   // 160 KiB of LDS per workgroup
constexpr uint32_t MSM_PART_TILE = 36864;   // most entries of a partition sorted in LDS (144 KiB); larger (skewed) partitions scatter in HBM
constexpr uint32_t MSM_PART_STAGE = 35584;  // most entries of a slice staged in LDS by the first level (139 KiB) beside its cursors:
// the stage and the two cursor arrays (2 P + 1 words) share the kernel's dynamic LDS - msm_part_stage_max(P) entries fit
__host__ __device__ constexpr uint32_t msm_part_stage_max(uint32_t P) {
    return MSM_LDS_WORDS - 2u * P - 1u - 63u < MSM_PART_STAGE ? MSM_LDS_WORDS - 2u * P - 1u - 63u : MSM_PART_STAGE;
}
constexpr uint32_t MSM_PART_COUNTERS = 1024; // second level: counters per workgroup (wave-private sets while 2^pb_log <= 64)
struct MsmPartCfg {
    uint32_t idx_bits, pb_log;   // idx_bits + pb_log <= 31
    uint32_t P;                  // nb >> pb_log
    uint32_t run_lanes;          // first level's copy-out: lanes per (slice, partition) run (8..64, a power of two >= the mean run)
};

template <class FR, bool SCATTER, bool PLAIN = false>
__global__ void __launch_bounds__(MSM_DIGITS_THREADS) msm_part_kernel(MsmBatchArgs a, MsmWindows win, MsmPartCfg pc, uint32_t nb, uint32_t n_max, uint32_t G,
                                                                     uint32_t* __restrict__ pcounts,         // [batch][G][P]   (!SCATTER: out, SCATTER: in)
                                                                     const uint32_t* __restrict__ runstart,  // [batch][G][P]   (SCATTER: in)
                                                                     uint32_t* __restrict__ tmp,
                                                                     uint32_t stage_cap) {                   // SCATTER: entries the LDS stage holds
    wave_priority<APK_PRIO_SORT>();
    using Fr = Fe<FR>;
    // dynamic LDS: [stage: stage_cap words (0 in the count pass)][cur: P][lstart: P + 1]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t* stage = reinterpret_cast<uint32_t*>(smem_raw);
    uint32_t* cur = stage + (SCATTER ? stage_cap : 0u);
    uint32_t* lstart = cur + pc.P;
    const uint32_t g = blockIdx.x, b = blockIdx.y;
    const uint32_t P = pc.P, pb_log = pc.pb_log, pb_mask = (1u << pc.pb_log) - 1u;
    const size_t row = ((size_t)b * G + g) * P;
    bool staged = false;
    if (SCATTER) {
        // slice-local exclusive prefix of this slice's partition counts (the count pass left them in pcounts): one wave, a
        // contiguous chunk per lane, a shuffle scan over the lanes' sums
        for (uint32_t k = threadIdx.x; k < P; k += blockDim.x) cur[k] = pcounts[row + k];
        __syncthreads();
        if (threadIdx.x < 64) {
            const uint32_t lane = threadIdx.x, per = (P + 63u) / 64u, k0 = lane * per;
            uint32_t sum = 0;
            for (uint32_t i = 0; i < per; i++) if (k0 + i < P) sum += cur[k0 + i];
            uint32_t inc = sum;
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t v = __shfl_up(inc, d, 64);
                if (lane >= d) inc += v;
            }
            uint32_t run = inc - sum;
            for (uint32_t i = 0; i < per; i++) if (k0 + i < P) { lstart[k0 + i] = run; run += cur[k0 + i]; }
            if (lane == 63) lstart[P] = inc;
        }
        __syncthreads();
        staged = lstart[P] <= stage_cap;                     // uniform
        for (uint32_t k = threadIdx.x; k < P; k += blockDim.x) cur[k] = staged ? lstart[k] : runstart[row + k];
    } else {
        for (uint32_t k = threadIdx.x; k < P; k += blockDim.x) cur[k] = 0u;
    }
    __syncthreads();
    const uint32_t len = a.len[b];
    const uint32_t per = (len + G - 1) / G;
    const uint32_t lo = min(g * per, len), hi = min(lo + per, len);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
#ifndef APK_MSM_NO_RINV
        Fr s = reinterpret_cast<const Fr*>(a.scalars[b])[i];      // Montgomery form as it is: the tables hold R^-1 * P
        if constexpr (PLAIN) s = Fr::from_mont(s);                // ... unless they hold P itself
#else
        Fr s = Fr::from_mont(reinterpret_cast<const Fr*>(a.scalars[b])[i]);
#endif
        uint32_t carry = 0;
        const uint32_t base_idx = a.offset[b] + i;
        uint64_t buf = 0;
        int avail = 0, j = 0;
#pragma unroll
        for (int li = 0; li < Fr::N; li++) {
            buf |= (uint64_t)s.l[li] << avail;
            avail += 32;
            while (j < win.W && (avail >= (int)win.width[j] || li == Fr::N - 1)) {
                const int c = win.width[j];
                const uint32_t half = 1u << (c - 1);
                uint32_t d = ((uint32_t)buf & ((1u << c) - 1u)) + carry;
                buf >>= c;
                avail -= c;
                uint32_t neg = 0;
                if (d > half) { d = (1u << c) - d; neg = 1; carry = 1; }
                else carry = 0;
                if (d != 0) {
                    const uint32_t k = d - 1;
                    if (!SCATTER) atomicAdd(&cur[k >> pb_log], 1u);
                    else {
                        const uint32_t pos = atomicAdd(&cur[k >> pb_log], 1u);
                        const uint32_t e = ((uint32_t)j * n_max + base_idx) | ((k & pb_mask) << pc.idx_bits) | (neg << 31);
                        if (staged) stage[pos] = e; else tmp[pos] = e;
                    }
                }
                j++;
            }
        }
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < P; k += blockDim.x) pcounts[row + k] = cur[k];
    } else if (staged) {
        // the stage holds the slice's entries in partition order: copy each partition's run to its place, neighbours together;
        // run_lanes lanes per run (a wave per run at 2^17: ~250 entries; sixteen lanes at 2^21 x 2 048 partitions: ~17 entries)
        __syncthreads();
        const uint32_t L = pc.run_lanes, R = 64u / L;
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, waves = blockDim.x >> 6;
        const uint32_t sub = lane / L, sl = lane % L;
        for (uint32_t pp = wave * R + sub; pp < P; pp += waves * R) {
            const uint32_t from = lstart[pp], cnt = lstart[pp + 1] - from;
            uint32_t* dst = tmp + runstart[row + pp];
            for (uint32_t l = sl; l < cnt; l += L) dst[l] = stage[from + l];
        }
    }
}

// one workgroup: first slot of every (msm, partition, slice) run in the order (msm, partition, slice), and the partition totals.
// Eight lanes share a (msm, partition) pair - each walks an eighth of the slices - so a lane has G/8 loads in flight instead
// of a serial walk over all G (the first version, one lane per pair, took 38 us of a lone batch's 128 us sort).
// For batch * P <= 2048 pairs and a few dozen slices (2^16 .. 2^17 bases); larger sorts take the three launches below.
template <int DUMMY>
__global__ void __launch_bounds__(1024) msm_part_scan_kernel(const uint32_t* __restrict__ pcounts, uint32_t* __restrict__ runstart,
                                                            uint32_t* __restrict__ ptot, uint32_t batch, uint32_t G, uint32_t P) {
    wave_priority<APK_PRIO_SORT>();
    __shared__ uint32_t s_tot[2 * 1024];       // per pair: total, then exclusive prefix
    __shared__ uint32_t s_sum[1024];
    const uint32_t t = threadIdx.x, n = batch * P;          // n <= 2048
    const uint32_t grp = t >> 3, j = t & 7u;
    const uint32_t gper = (G + 7u) / 8u;
    const uint32_t g0 = min(j * gper, G), g1 = min(g0 + gper, G);
    s_tot[t] = 0u;
    s_tot[t + 1024u] = 0u;
    __syncthreads();
    for (uint32_t q = grp; q < 2048u; q += 128u) {          // every lane runs every trip: the shuffles need whole groups
        uint32_t sum = 0;
        if (q < n) {
            const uint32_t b = q / P, p = q % P;
            for (uint32_t g = g0; g < g1; g++) sum += pcounts[((size_t)b * G + g) * P + p];
        }
        sum += __shfl_xor(sum, 4, 8);
        sum += __shfl_xor(sum, 2, 8);
        sum += __shfl_xor(sum, 1, 8);
        if (j == 0) { s_tot[q] = sum; if (q < n) ptot[q] = sum; }
        if (q + 128u >= ((n + 127u) / 128u) * 128u) break;  // uniform: past the last trip that holds a pair
    }
    __syncthreads();
    const uint32_t a0 = s_tot[2 * t], a1 = s_tot[2 * t + 1];
    s_sum[t] = a0 + a1;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = t >= d ? s_sum[t - d] : 0u;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    const uint32_t base = s_sum[t] - (a0 + a1);              // exclusive
    __syncthreads();
    s_tot[2 * t] = base;
    s_tot[2 * t + 1] = base + a0;
    __syncthreads();
    for (uint32_t q = grp; q < 2048u; q += 128u) {
        uint32_t sum = 0;
        uint32_t b = 0, p = 0;
        if (q < n) {
            b = q / P; p = q % P;
            for (uint32_t g = g0; g < g1; g++) sum += pcounts[((size_t)b * G + g) * P + p];
        }
        uint32_t incl = sum;                                  // inclusive scan over the group's 8 lanes
        for (uint32_t d = 1; d < 8; d <<= 1) {
            const uint32_t v = __shfl_up(incl, d, 8);
            if (j >= d) incl += v;
        }
        if (q < n) {
            uint32_t run = s_tot[q] + incl - sum;
            for (uint32_t g = g0; g < g1; g++) {
                const size_t at = ((size_t)b * G + g) * P + p;
                runstart[at] = run;
                run += pcounts[at];
            }
        }
        if (q + 128u >= ((n + 127u) / 128u) * 128u) break;
    }
}

// The same scan for any size (up to MSM_MAX_BATCH * MSM_PART_MAX pairs, ~1 000 slices: BLS12-381 2^21), in three launches whose
// lanes stand for 64 consecutive (msm, partition) pairs - so every load and store is a whole 256-byte row of the counter array -
// and whose waves each walk one of MSM_PART_CHUNKS chunks of the slices:
//   msm_part_tot_kernel   csum[c][q] = entries of pair q in the slices of chunk c
//   msm_part_base_kernel  one workgroup: ptot[q], the exclusive scan over q, csum[c][q] := first slot of pair q's chunk c
//   msm_part_runs_kernel  runstart of every (pair, slice)
constexpr uint32_t MSM_PART_CHUNKS = 8;
template <int DUMMY>
__global__ void __launch_bounds__(256) msm_part_tot_kernel(const uint32_t* __restrict__ pcounts, uint32_t* __restrict__ csum, uint32_t batch,
                                                           uint32_t G, uint32_t P) {
    wave_priority<APK_PRIO_SORT>();
    const uint32_t n = batch * P, q = blockIdx.x * 64u + (threadIdx.x & 63u), c = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (q >= n) return;
    const uint32_t gper = (G + MSM_PART_CHUNKS - 1u) / MSM_PART_CHUNKS, g0 = min(c * gper, G), g1 = min(g0 + gper, G);
    const uint32_t b = q / P, p = q % P;
    const uint32_t* col = pcounts + (size_t)b * G * P + p;
    uint32_t sum = 0, g = g0;
    for (; g + 4 <= g1; g += 4) {      // four rows in flight
        const uint32_t v0 = col[(size_t)g * P], v1 = col[(size_t)(g + 1) * P], v2 = col[(size_t)(g + 2) * P], v3 = col[(size_t)(g + 3) * P];
        sum += v0 + v1 + v2 + v3;
    }
    for (; g < g1; g++) sum += col[(size_t)g * P];
    csum[(size_t)c * n + q] = sum;
}
template <int DUMMY>
__global__ void __launch_bounds__(1024) msm_part_base_kernel(uint32_t* __restrict__ csum, uint32_t* __restrict__ ptot, uint32_t n) {
    wave_priority<APK_PRIO_SORT>();
    __shared__ uint32_t s_sum[1024];
    const uint32_t t = threadIdx.x;
    constexpr uint32_t PER = MSM_MAX_BATCH * MSM_PART_MAX / 1024;      // 8 consecutive pairs per thread
    uint32_t tot[PER], mine = 0;
#pragma unroll
    for (uint32_t i = 0; i < PER; i++) {
        const uint32_t q = t * PER + i;
        uint32_t v = 0;
        if (q < n) for (uint32_t c = 0; c < MSM_PART_CHUNKS; c++) v += csum[(size_t)c * n + q];
        tot[i] = v;
        mine += v;
        if (q < n) ptot[q] = v;
    }
    s_sum[t] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = t >= d ? s_sum[t - d] : 0u;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    uint32_t run = s_sum[t] - mine;
#pragma unroll
    for (uint32_t i = 0; i < PER; i++) {
        const uint32_t q = t * PER + i;
        if (q < n) {
            uint32_t r2 = run;
            for (uint32_t c = 0; c < MSM_PART_CHUNKS; c++) { const uint32_t v = csum[(size_t)c * n + q]; csum[(size_t)c * n + q] = r2; r2 += v; }
        }
        run += tot[i];
    }
}
template <int DUMMY>
__global__ void __launch_bounds__(256) msm_part_runs_kernel(const uint32_t* __restrict__ pcounts, const uint32_t* __restrict__ csum,
                                                            uint32_t* __restrict__ runstart, uint32_t batch, uint32_t G, uint32_t P) {
    wave_priority<APK_PRIO_SORT>();
    const uint32_t n = batch * P, q = blockIdx.x * 64u + (threadIdx.x & 63u), c = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (q >= n) return;
    const uint32_t gper = (G + MSM_PART_CHUNKS - 1u) / MSM_PART_CHUNKS, g0 = min(c * gper, G), g1 = min(g0 + gper, G);
    const uint32_t b = q / P, p = q % P;
    const size_t col = (size_t)b * G * P + p;
    uint32_t run = csum[(size_t)c * n + q], g = g0;
    for (; g + 4 <= g1; g += 4) {
        const uint32_t v0 = pcounts[col + (size_t)g * P], v1 = pcounts[col + (size_t)(g + 1) * P], v2 = pcounts[col + (size_t)(g + 2) * P],
                       v3 = pcounts[col + (size_t)(g + 3) * P];
        runstart[col + (size_t)g * P] = run; run += v0;
        runstart[col + (size_t)(g + 1) * P] = run; run += v1;
        runstart[col + (size_t)(g + 2) * P] = run; run += v2;
        runstart[col + (size_t)(g + 3) * P] = run; run += v3;
    }
    for (; g < g1; g++) { runstart[col + (size_t)g * P] = run; run += pcounts[col + (size_t)g * P]; }
}

// grid (P, batch): counting sort of one partition's entries by bucket inside an LDS tile, whole lines out;
// hist[b*nb + k] = entries of bucket k.  With few buckets per partition (2^pb_log <= 64: the large sorts) every wave counts into
// a set of its own - 1 024 lanes on 16 or 32 counters serialise on the LDS atomics otherwise - and the sets are laid out
// [set][bucket] (a wave's lanes spread over the banks) but scanned bucket-major, so a bucket's entries stay together.
template <int DUMMY>
__global__ void __launch_bounds__(1024) msm_part_sort_kernel(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ runstart,
                                                           const uint32_t* __restrict__ ptot, MsmPartCfg pc, uint32_t G, uint32_t nb,
                                                           uint32_t* __restrict__ hist, uint32_t* __restrict__ sorted,
                                                           uint32_t tile_cap) {              // entries the LDS tile of this launch holds
    wave_priority<APK_PRIO_SORT>();
    __shared__ uint32_t cnt[MSM_PART_COUNTERS], cur[MSM_PART_COUNTERS];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t* tile = reinterpret_cast<uint32_t*>(smem_raw);
    const uint32_t p = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const uint32_t P = pc.P, PB = 1u << pc.pb_log, pb_mask = PB - 1u;
    const uint32_t sets_log = pc.pb_log > 6 ? 0u : min(4u, 10u - pc.pb_log);        // sets x buckets <= 1 024; one shared set from 128 buckets up
    const uint32_t SETS = 1u << sets_log, NC = PB << sets_log;
    const uint32_t set_base = ((t >> 6) & (SETS - 1u)) << pc.pb_log;                 // this wave's counters
    const uint32_t first = runstart[(size_t)b * G * P + p];          // slice 0's run opens the partition
    const uint32_t n = ptot[b * P + p];
    const uint32_t KEEP = ((1u << pc.idx_bits) - 1u) | 0x80000000u;
    if (t < NC) cnt[t] = 0u;
    __syncthreads();
    for (uint32_t i = t; i < n; i += blockDim.x) atomicAdd(&cnt[set_base + ((tmp[first + i] >> pc.idx_bits) & pb_mask)], 1u);
    __syncthreads();
    // thread t < NC stands for (bucket t / SETS, set t % SETS): inclusive scan in that order
    uint32_t mine = 0;
    const uint32_t my_at = ((t & (SETS - 1u)) << pc.pb_log) + (t >> sets_log);
    if (t < NC) { mine = cnt[my_at]; cur[t] = mine; }
    __syncthreads();
    for (uint32_t d = 1; d < NC; d <<= 1) {
        uint32_t v = 0;
        if (t < NC && t >= d) v = cur[t - d];
        __syncthreads();
        if (t < NC) cur[t] += v;
        __syncthreads();
    }
    const bool in_lds = n <= tile_cap;                               // uniform
    uint32_t incl = 0, before = 0;
    if (t < NC) {
        incl = cur[t];
        if ((t & (SETS - 1u)) == SETS - 1u) before = t >= SETS ? cur[t - SETS] : 0u;      // last set of a bucket: the bucket's total
    }
    __syncthreads();
    if (t < NC) {
        if ((t & (SETS - 1u)) == SETS - 1u) hist[(size_t)b * nb + p * PB + (t >> sets_log)] = incl - before;
        cnt[my_at] = (in_lds ? 0u : first) + incl - mine;            // exclusive prefix inside the partition: the set's cursor
    }
    __syncthreads();
    for (uint32_t i = t; i < n; i += blockDim.x) {
        const uint32_t e = tmp[first + i];
        const uint32_t pos = atomicAdd(&cnt[set_base + ((e >> pc.idx_bits) & pb_mask)], 1u);
        if (in_lds) tile[pos] = e & KEEP; else sorted[pos] = e & KEEP;
    }
    if (in_lds) {
        __syncthreads();
        for (uint32_t i = t; i < n; i += blockDim.x) sorted[first + i] = tile[i];
    }
}

// ---- the two levels in TWO launches (round 4; APK_MSM_SORT_FUSED) ------------------------------------------------------------
// The four launches above are count - scan - scatter - sort because the first level's runs are laid out in the order (msm,
// partition, slice), which needs every slice's counts before any slice can place an entry.  Laid out SLICE-MAJOR instead - slice g
// of msm b owns tmp[(b G + g) cap, +cap), its entries in partition order exactly as they sit in its LDS stage - a slice needs
// nobody's counts but its own: ONE launch counts (LDS), scans (one wave), recodes the scalars it still holds in registers and
// scatters into the stage, copies the stage out as one contiguous block, and leaves its run table (P + 1 offsets) and its share
// of the partition totals (one device atomic per partition and slice: ~8 k per MSM against the 2 M entries it sorts).  The second
// level gathers a partition's entries from its G runs (a wave per run) instead of one range.  Launches per batch 7 -> 5, the
// digits are still extracted twice but the scalars are read once, the counter arrays between the launches are gone.
template <class FR, class F>
__device__ __forceinline__ void msm_for_each_digit(const Fe<FR>& s, const MsmWindows& win, F&& f) {
    using Fr = Fe<FR>;
    uint32_t carry = 0;
    uint64_t buf = 0;
    int avail = 0, j = 0;
#pragma unroll
    for (int li = 0; li < Fr::N; li++) {
        buf |= (uint64_t)s.l[li] << avail;
        avail += 32;
        while (j < win.W && (avail >= (int)win.width[j] || li == Fr::N - 1)) {
            const int c = win.width[j];
            const uint32_t half = 1u << (c - 1);
            uint32_t d = ((uint32_t)buf & ((1u << c) - 1u)) + carry;
            buf >>= c;
            avail -= c;
            uint32_t neg = 0;
            if (d > half) { d = (1u << c) - d; neg = 1; carry = 1; }
            else carry = 0;
            if (d != 0) f((uint32_t)j, d - 1, neg);
            j++;
        }
    }
}

// Non-zero signed digits of the CANONICAL values of a batch's scalars (grid (blocks, batch), 256 lanes): what a commitment over
// plain tables would sort and add.  The prover compares it with the uniform expectation (len x W) to decide whether the wire
// polynomials are committed over the Lagrange SRS - where the scalars are the witness values themselves, mostly 0 / 1 / small in
// real circuits - or, as always before round 5, over the canonical SRS after the inverse transform (backend_impl.h, round 1).
template <class FR>
__global__ void __launch_bounds__(256) msm_density_kernel(MsmBatchArgs a, MsmWindows win, uint32_t* __restrict__ out) {
    wave_priority<APK_PRIO_SORT>();
    using Fr = Fe<FR>;
    const uint32_t b = blockIdx.y;
    const Fr* __restrict__ sc = reinterpret_cast<const Fr*>(a.scalars[b]);
    uint32_t cnt = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.len[b]; i += gridDim.x * blockDim.x)
        msm_for_each_digit<FR>(Fr::from_mont(sc[i]), win, [&](uint32_t, uint32_t, uint32_t) { cnt++; });
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_down(cnt, d, 64);
    if ((threadIdx.x & 63u) == 0 && cnt) atomicAdd(out, cnt);
}

constexpr int MSM_PART1_HOLD = 3;   // scalars a lane keeps in registers between the two passes (2 049 per slice / 1 024 lanes)
template <class FR, bool PLAIN = false>
__global__ void __launch_bounds__(MSM_DIGITS_THREADS) msm_part1_kernel(MsmBatchArgs a, MsmWindows win, MsmPartCfg pc, uint32_t n_max, uint32_t G,
                                                                      uint32_t* __restrict__ tmp, uint32_t cap,       // cap entries per slice
                                                                      uint32_t* __restrict__ runtab,                   // [batch][G][P + 1]
                                                                      uint32_t* __restrict__ ptot) {                   // [batch][P], zero on entry
    wave_priority<APK_PRIO_SORT>();
    using Fr = Fe<FR>;
    // dynamic LDS: [stage: cap words][cur: P][lstart: P + 1]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t* stage = reinterpret_cast<uint32_t*>(smem_raw);
    uint32_t* cur = stage + cap;
    uint32_t* lstart = cur + pc.P;
    const uint32_t g = blockIdx.x, b = blockIdx.y;
    const uint32_t P = pc.P, pb_log = pc.pb_log, pb_mask = (1u << pc.pb_log) - 1u;
    for (uint32_t k = threadIdx.x; k < P; k += blockDim.x) cur[k] = 0u;
    __syncthreads();
    const uint32_t len = a.len[b];
    const uint32_t per = (len + G - 1) / G;
    const uint32_t lo = min(g * per, len), hi = min(lo + per, len);
    const Fr* __restrict__ sc = reinterpret_cast<const Fr*>(a.scalars[b]);
    Fr held[MSM_PART1_HOLD];
    const bool hold = (hi - lo) <= (uint32_t)MSM_PART1_HOLD * blockDim.x;    // uniform
    {
        int it = 0;
        for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x, it++) {
#ifndef APK_MSM_NO_RINV
            Fr s = sc[i];
            if constexpr (PLAIN) s = Fr::from_mont(s);
#else
            const Fr s = Fr::from_mont(sc[i]);
#endif
            if (hold) {
#pragma unroll
                for (int h = 0; h < MSM_PART1_HOLD; h++) if (h == it) held[h] = s;    // static register indices
            }
            msm_for_each_digit<FR>(s, win, [&](uint32_t, uint32_t k, uint32_t) { atomicAdd(&cur[k >> pb_log], 1u); });
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {      // exclusive prefix of the partition counts: a contiguous chunk per lane, a shuffle scan over the lanes
        const uint32_t lane = threadIdx.x, chunk = (P + 63u) / 64u, k0 = lane * chunk;
        uint32_t sum = 0;
        for (uint32_t i = 0; i < chunk; i++) if (k0 + i < P) sum += cur[k0 + i];
        uint32_t inc = sum;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t v = __shfl_up(inc, d, 64);
            if (lane >= d) inc += v;
        }
        uint32_t run = inc - sum;
        for (uint32_t i = 0; i < chunk; i++) if (k0 + i < P) { lstart[k0 + i] = run; run += cur[k0 + i]; }
        if (lane == 63) lstart[P] = inc;
    }
    __syncthreads();
    uint32_t* rt = runtab + ((size_t)b * G + g) * (P + 1);
    for (uint32_t k = threadIdx.x; k < P; k += blockDim.x) {
        const uint32_t c = cur[k];
        if (c) atomicAdd(&ptot[b * P + k], c);
        rt[k] = lstart[k];
        cur[k] = lstart[k];
    }
    if (threadIdx.x == 0) rt[P] = lstart[P];
    __syncthreads();
    {
        int it = 0;
        for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x, it++) {
            Fr s;
            if (hold) {
#pragma unroll
                for (int h = 0; h < MSM_PART1_HOLD; h++) if (h == it) s = held[h];
            } else {
#ifndef APK_MSM_NO_RINV
                s = sc[i];
                if constexpr (PLAIN) s = Fr::from_mont(s);
#else
                s = Fr::from_mont(sc[i]);
#endif
            }
            const uint32_t base_idx = a.offset[b] + i;
            msm_for_each_digit<FR>(s, win, [&](uint32_t j, uint32_t k, uint32_t neg) {
                const uint32_t pos = atomicAdd(&cur[k >> pb_log], 1u);
                stage[pos] = (j * n_max + base_idx) | ((k & pb_mask) << pc.idx_bits) | (neg << 31);
            });
        }
    }
    __syncthreads();
    const uint32_t total = lstart[P];
    uint32_t* dst = tmp + ((size_t)b * G + g) * cap;
    for (uint32_t l = threadIdx.x; l < total; l += blockDim.x) dst[l] = stage[l];     // one contiguous block: whole lines
}

// grid (P, batch): the second level over slice-major runs.  ptot_next: the other totals buffer, zeroed here for the next batch.
template <int DUMMY>
__global__ void __launch_bounds__(1024) msm_part_sort_runs_kernel(const uint32_t* __restrict__ tmp, uint32_t cap, const uint32_t* __restrict__ runtab,
                                                                const uint32_t* __restrict__ ptot, uint32_t* __restrict__ ptot_next, MsmPartCfg pc,
                                                                uint32_t G, uint32_t nb, uint32_t* __restrict__ hist, uint32_t* __restrict__ sorted,
                                                                uint32_t tile_cap, uint32_t zero_words /* of ptot_next: the workspace's batch capacity x P */) {
    wave_priority<APK_PRIO_SORT>();
    __shared__ uint32_t cnt[MSM_PART_COUNTERS], cur[MSM_PART_COUNTERS];
    __shared__ uint32_t s_red[16];
    __shared__ uint32_t s_first;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t* tile = reinterpret_cast<uint32_t*>(smem_raw);
    const uint32_t p = blockIdx.x, b = blockIdx.y, t = threadIdx.x, wave = t >> 6, lane = t & 63u;
    const uint32_t P = pc.P, PB = 1u << pc.pb_log, pb_mask = PB - 1u;
    const uint32_t sets_log = pc.pb_log > 6 ? 0u : min(4u, 10u - pc.pb_log);
    const uint32_t SETS = 1u << sets_log, NC = PB << sets_log;
    const uint32_t set_base = (wave & (SETS - 1u)) << pc.pb_log;
    const uint32_t KEEP = ((1u << pc.idx_bits) - 1u) | 0x80000000u;
    // first slot of this partition in `sorted` = the totals of every (msm, partition) before it
    {
        const uint32_t before = b * P + p;
        uint32_t sum = 0;
        for (uint32_t q = t; q < before; q += blockDim.x) sum += ptot[q];
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_down(sum, d, 64);
        if (lane == 0) s_red[wave] = sum;
        if (t < NC) cnt[t] = 0u;
        __syncthreads();
        if (t == 0) { uint32_t f = 0; for (uint32_t w = 0; w < (blockDim.x >> 6); w++) f += s_red[w]; s_first = f; }
        if (p == 0 && b == 0) for (uint32_t q = t; q < zero_words; q += blockDim.x) ptot_next[q] = 0u;
        __syncthreads();
    }
    const uint32_t first = s_first;
    const uint32_t n = ptot[b * P + p];
    const uint32_t waves = blockDim.x >> 6;
    const uint32_t* rt0 = runtab + (size_t)b * G * (P + 1) + p;
    // A partition far above the mean is a SKEWED input (all-equal scalars put a whole MSM's entries of a window into one bucket):
    // its lanes would then queue up on one LDS counter, 131 k atomics in a row (226 us for such a partition, measured).  There
    // the lanes of a wave that hold the same bucket add ONCE (a ballot per distinct bucket: one or two rounds when the input is
    // skewed; never taken for uniform scalars, whose partitions stay within 2 % of the mean).
    const bool skew = n > tile_cap;                                  // uniform across the workgroup
    // count: a wave per run
    for (uint32_t g = wave; g < G; g += waves) {
        const uint32_t r0 = rt0[(size_t)g * (P + 1)], r1 = rt0[(size_t)g * (P + 1) + 1];
        const uint32_t* src = tmp + ((size_t)b * G + g) * cap;
        if (!skew) {
            for (uint32_t l = r0 + lane; l < r1; l += 64u) atomicAdd(&cnt[set_base + ((src[l] >> pc.idx_bits) & pb_mask)], 1u);
        } else {
            for (uint32_t l0 = r0; l0 < r1; l0 += 64u) {             // wave-uniform trip count
                const uint32_t l = l0 + lane;
                const bool act = l < r1;
                const uint32_t key = act ? ((src[l] >> pc.idx_bits) & pb_mask) : 0u;
                uint64_t todo = __ballot(act);
                // only when the wave really is concentrated (half of its lanes on the first lane's bucket); a partition that is
                // large but spread over its buckets (256 distinct scalars) keeps the plain atomics
                {
                    const uint32_t kf = __shfl(key, todo ? __ffsll((unsigned long long)todo) - 1 : 0, 64);
                    if (2 * __popcll(__ballot(act && key == kf)) < __popcll(todo)) {
                        if (act) atomicAdd(&cnt[set_base + key], 1u);
                        todo = 0;
                    }
                }
                while (todo) {
                    const int leader = __ffsll((unsigned long long)todo) - 1;
                    const uint32_t k0 = __shfl(key, leader, 64);
                    const uint64_t same = __ballot(act && key == k0);
                    if ((int)lane == leader) atomicAdd(&cnt[set_base + k0], (uint32_t)__popcll(same));
                    todo &= ~same;
                }
            }
        }
    }
    __syncthreads();
    uint32_t mine = 0;
    const uint32_t my_at = ((t & (SETS - 1u)) << pc.pb_log) + (t >> sets_log);
    if (t < NC) { mine = cnt[my_at]; cur[t] = mine; }
    __syncthreads();
    for (uint32_t d = 1; d < NC; d <<= 1) {
        uint32_t v = 0;
        if (t < NC && t >= d) v = cur[t - d];
        __syncthreads();
        if (t < NC) cur[t] += v;
        __syncthreads();
    }
    const bool in_lds = n <= tile_cap;                               // uniform
    uint32_t incl = 0, before = 0;
    if (t < NC) {
        incl = cur[t];
        if ((t & (SETS - 1u)) == SETS - 1u) before = t >= SETS ? cur[t - SETS] : 0u;
    }
    __syncthreads();
    if (t < NC) {
        if ((t & (SETS - 1u)) == SETS - 1u) hist[(size_t)b * nb + p * PB + (t >> sets_log)] = incl - before;
        cnt[my_at] = (in_lds ? 0u : first) + incl - mine;
    }
    __syncthreads();
    for (uint32_t g = wave; g < G; g += waves) {
        const uint32_t r0 = rt0[(size_t)g * (P + 1)], r1 = rt0[(size_t)g * (P + 1) + 1];
        const uint32_t* src = tmp + ((size_t)b * G + g) * cap;
        if (!skew) {
            for (uint32_t l = r0 + lane; l < r1; l += 64u) {
                const uint32_t e = src[l];
                const uint32_t pos = atomicAdd(&cnt[set_base + ((e >> pc.idx_bits) & pb_mask)], 1u);
                if (in_lds) tile[pos] = e & KEEP; else sorted[pos] = e & KEEP;
            }
        } else {                                                     // skew implies !in_lds: straight to HBM, neighbours together
            for (uint32_t l0 = r0; l0 < r1; l0 += 64u) {
                const uint32_t l = l0 + lane;
                const bool act = l < r1;
                const uint32_t e = act ? src[l] : 0u;
                const uint32_t key = (e >> pc.idx_bits) & pb_mask;
                uint64_t todo = __ballot(act);
                uint32_t pos = 0;
                {
                    const uint32_t kf = __shfl(key, todo ? __ffsll((unsigned long long)todo) - 1 : 0, 64);
                    if (2 * __popcll(__ballot(act && key == kf)) < __popcll(todo)) {
                        if (act) pos = atomicAdd(&cnt[set_base + key], 1u);
                        todo = 0;
                    }
                }
                while (todo) {
                    const int leader = __ffsll((unsigned long long)todo) - 1;
                    const uint32_t k0 = __shfl(key, leader, 64);
                    const uint64_t same = __ballot(act && key == k0);
                    uint32_t base = 0;
                    if ((int)lane == leader) base = atomicAdd(&cnt[set_base + k0], (uint32_t)__popcll(same));
                    base = __shfl(base, leader, 64);
                    if (act && key == k0) pos = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                    todo &= ~same;
                }
                if (act) sorted[pos] = e & KEEP;
            }
        }
    }
    if (in_lds) {
        __syncthreads();
        for (uint32_t i = t; i < n; i += blockDim.x) sorted[first + i] = tile[i];
    }
}

// ---- exclusive scans of the bucket counts: entry offsets and work-unit offsets ------------------------------------------
// A bucket of h entries is cut into floor(h/unit) FULL work units and, if h % unit != 0, one REMAINDER unit.
//   offsets[k]  = sum_{i<k} hist[i]                 (entries)
//   unit_off[k] = sum_{i<k} ceil(hist[i]/unit)      (slots of the per-unit partial sums: a bucket's slots are contiguous)
//   full_off[k] = sum_{i<k} floor(hist[i]/unit)     (lane -> bucket map of the full units)
//   rem_list    = the buckets that have a remainder unit, sorted by remainder length (descending, counting sort): the
//                 accumulate kernel runs them after the full units, so the 64 lanes of a wave loop the same number of times
//                 (interleaved with full units, the remainders idled 6 % of the lanes at unit = 16).
// Three small launches (a single block took 58 us for 49 k buckets): 1024 counters per block -> block totals -> one block
// scans the totals -> blocks add their base.
constexpr int MSM_SCAN_BLOCK = 1024;
constexpr int MSM_SCAN_MAX_BLOCKS = 256;   // MSM_MAX_BATCH * 2^16 buckets / MSM_SCAN_BLOCK
constexpr int MSM_SCAN_ITEMS_MAX = 8;      // consecutive buckets per thread of the local scan: 256 blocks x 1024 x 8 = 2^21 = MSM_MAX_BATCH * 2^19 buckets (c = 20)
// The same counting sort orders ALL buckets by their number of unit partials (merge_list, most partials first; counts from
// MSM_MERGE_BINS - 1 up share the last bin): msm_combine_kernel walks the buckets in that order, so the lanes of a wave merge the
// same number of partials.  In bucket order a wave waited for its bucket with the most partials (3..6 at c = 16: 73 % lane
// efficiency, round 3 PMC) - the merge is 7 % of a proof's instructions.
constexpr int MSM_MERGE_BINS = 16;
constexpr int MSM_BINS = MSM_UNIT_MAX + MSM_MERGE_BINS;   // [0, MSM_UNIT_MAX): remainder lengths, then the unit counts

// one block: exclusive scan of the (<= 1024) block totals in place; grand totals to offsets[total] / unit_off[total] /
// full_off[total]; block_bins[blk][r] becomes the first rem_list (merge_list) position of block blk's buckets with remainder r (unit count r - MSM_UNIT_MAX)
// A device function: its own launch (msm_scan_totals_kernel), or the last workgroup of msm_scan_local_kernel to finish (round 4:
// two scan launches instead of three).
struct MsmScanTotalsLds {
    uint32_t s_bintot[MSM_BINS];
    uint16_t s_bins[MSM_SCAN_MAX_BLOCKS * MSM_BINS];   // per-block remainder / unit-count histograms (<= 1024 each): 40 KB
};
__device__ __forceinline__ void msm_scan_totals_body(uint32_t* __restrict__ block_tot, uint32_t* __restrict__ block_bins, uint32_t nblocks, uint32_t total,
                                                     uint32_t* __restrict__ offsets, uint32_t* __restrict__ unit_off, uint32_t* __restrict__ full_off,
                                                     uint32_t* s_cnt, uint32_t* s_unit, uint32_t* s_full, MsmScanTotalsLds& L) {
    const uint32_t t = threadIdx.x;
    const uint32_t h = t < nblocks ? block_tot[t] : 0u, hu = t < nblocks ? block_tot[nblocks + t] : 0u,
                   hf = t < nblocks ? block_tot[2 * nblocks + t] : 0u;
    s_cnt[t] = h; s_unit[t] = hu; s_full[t] = hf;
    for (uint32_t i = t; i < nblocks * MSM_BINS; i += MSM_SCAN_BLOCK) L.s_bins[i] = (uint16_t)block_bins[i];
    __syncthreads();
    if (t < MSM_BINS) {   // per remainder length / unit count: the total over the blocks
        uint32_t run = 0;
        for (uint32_t blk = 0; blk < nblocks; blk++) run += L.s_bins[blk * MSM_BINS + t];
        L.s_bintot[t] = run;
    }
    __syncthreads();
    for (uint32_t d = 1; d < MSM_SCAN_BLOCK; d <<= 1) {
        uint32_t vc = 0, vu = 0, vf = 0;
        if (t >= d) { vc = s_cnt[t - d]; vu = s_unit[t - d]; vf = s_full[t - d]; }
        __syncthreads();
        s_cnt[t] += vc; s_unit[t] += vu; s_full[t] += vf;
        __syncthreads();
    }
    if (t < nblocks) { block_tot[t] = s_cnt[t] - h; block_tot[nblocks + t] = s_unit[t] - hu; block_tot[2 * nblocks + t] = s_full[t] - hf; }
    if (t == MSM_SCAN_BLOCK - 1) {
        offsets[total] = s_cnt[t]; unit_off[total] = s_unit[t]; full_off[total] = s_full[t];
        // unit partials per NON-EMPTY bucket, behind the bins: the merge picks its lanes per bucket from it (a skewed input has few
        // buckets with many partials; the host only knows the average over all buckets)
        const uint32_t nonempty = total - L.s_bintot[MSM_UNIT_MAX];
        block_bins[nblocks * MSM_BINS] = nonempty ? s_unit[t] / nonempty : 0u;
    }
    if (t < MSM_BINS) {   // longest remainders (most partials) first: base of bin t in its list, then the exclusive prefix over the blocks
        uint32_t run = 0;
        const uint32_t top = t < (uint32_t)MSM_UNIT_MAX ? MSM_UNIT_MAX : MSM_BINS;
        for (uint32_t r = top - 1; r > t; r--) run += L.s_bintot[r];
        for (uint32_t blk = 0; blk < nblocks; blk++) {
            const uint32_t v = L.s_bins[blk * MSM_BINS + t];
            block_bins[blk * MSM_BINS + t] = run;
            run += v;
        }
    }
}
// FUSED = 1: the workgroup that finishes last (an agent-scope counter, reset by that workgroup; no spinning) goes on to run the
// totals step, so the scan is two launches; FUSED = 0: msm_scan_totals_kernel follows as a launch of its own.
template <int FUSED, int ITEMS = 1>   // ITEMS = consecutive buckets per thread: 1 (up to 2^18 buckets, the code of rounds 1-4), 2, 4 or 8
__global__ void __launch_bounds__(MSM_SCAN_BLOCK) msm_scan_local_kernel(const uint32_t* __restrict__ hist, uint32_t total, uint32_t unit,
                                                                         uint32_t* __restrict__ offsets, uint32_t* __restrict__ unit_off,
                                                                         uint32_t* __restrict__ full_off, uint32_t* __restrict__ rem_rank,
                                                                         uint32_t* __restrict__ merge_rank,
                                                                         uint32_t* __restrict__ block_tot /* [3][nblocks] */,
                                                                         uint32_t* __restrict__ block_bins /* [nblocks][MSM_BINS] */, uint32_t nblocks,
                                                                         uint32_t* __restrict__ done) {
    constexpr uint32_t items = ITEMS;
    wave_priority<APK_PRIO_SORT>();
    __shared__ uint32_t s_cnt[MSM_SCAN_BLOCK];
    __shared__ uint32_t s_unit[MSM_SCAN_BLOCK];
    __shared__ uint32_t s_full[MSM_SCAN_BLOCK];
    __shared__ uint32_t s_bins[MSM_BINS];
    const uint32_t t = threadIdx.x, i0 = (blockIdx.x * MSM_SCAN_BLOCK + t) * items;   // this thread's `items` consecutive buckets
    if (t < MSM_BINS) s_bins[t] = 0;
    __syncthreads();
    uint32_t hv[ITEMS], sh = 0, su = 0, sf = 0;
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t i = i0 + (uint32_t)it;
        const bool in = i < total;
        const uint32_t h = in ? hist[i] : 0u, hf = h / unit, rem = h - hf * unit, hu = hf + (rem ? 1u : 0u);
        hv[it] = h;
        if (rem) rem_rank[i] = atomicAdd(&s_bins[rem], 1u);   // rank of this bucket among the block's buckets with the same remainder
        if (in) merge_rank[i] = atomicAdd(&s_bins[MSM_UNIT_MAX + min(hu, (uint32_t)MSM_MERGE_BINS - 1u)], 1u);   // ... with the same number of partials
        sh += h; su += hu; sf += hf;
    }
    s_cnt[t] = sh; s_unit[t] = su; s_full[t] = sf;
    __syncthreads();
    for (uint32_t d = 1; d < MSM_SCAN_BLOCK; d <<= 1) {
        uint32_t vc = 0, vu = 0, vf = 0;
        if (t >= d) { vc = s_cnt[t - d]; vu = s_unit[t - d]; vf = s_full[t - d]; }
        __syncthreads();
        s_cnt[t] += vc; s_unit[t] += vu; s_full[t] += vf;
        __syncthreads();
    }
    {   // exclusive, block-local: the thread's base, then its buckets one after the other
        uint32_t ro = s_cnt[t] - sh, ru = s_unit[t] - su, rf = s_full[t] - sf;
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            const uint32_t i = i0 + (uint32_t)it;
            if (i < total) {
                const uint32_t h = hv[it], hf = h / unit, hu = hf + (h - hf * unit ? 1u : 0u);
                offsets[i] = ro; unit_off[i] = ru; full_off[i] = rf;
                ro += h; ru += hu; rf += hf;
            }
        }
    }
    if (t == MSM_SCAN_BLOCK - 1) {
        block_tot[blockIdx.x] = s_cnt[t]; block_tot[nblocks + blockIdx.x] = s_unit[t]; block_tot[2 * nblocks + blockIdx.x] = s_full[t];
    }
    if (t < MSM_BINS) block_bins[blockIdx.x * MSM_BINS + t] = s_bins[t];
    if constexpr (FUSED != 0) {
        __shared__ MsmScanTotalsLds L;
        __shared__ uint32_t s_last;
        __threadfence();                 // this workgroup's totals and bins, written by several lanes, before the count
        __syncthreads();
        if (t == 0) {
            const uint32_t old = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old == gridDim.x - 1) ? 1u : 0u;
            if (s_last) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        msm_scan_totals_body(block_tot, block_bins, nblocks, total, offsets, unit_off, full_off, s_cnt, s_unit, s_full, L);
    }
}

template <int DUMMY>
__global__ void __launch_bounds__(MSM_SCAN_BLOCK) msm_scan_totals_kernel(uint32_t* __restrict__ block_tot, uint32_t* __restrict__ block_bins,
                                                                          uint32_t nblocks, uint32_t total, uint32_t* __restrict__ offsets,
                                                                          uint32_t* __restrict__ unit_off, uint32_t* __restrict__ full_off) {
    wave_priority<APK_PRIO_SORT>();
    __shared__ uint32_t s_cnt[MSM_SCAN_BLOCK];
    __shared__ uint32_t s_unit[MSM_SCAN_BLOCK];
    __shared__ uint32_t s_full[MSM_SCAN_BLOCK];
    __shared__ MsmScanTotalsLds L;
    msm_scan_totals_body(block_tot, block_bins, nblocks, total, offsets, unit_off, full_off, s_cnt, s_unit, s_full, L);
}

template <int DUMMY>
__global__ void __launch_bounds__(MSM_SCAN_BLOCK) msm_scan_apply_kernel(const uint32_t* __restrict__ block_tot, const uint32_t* __restrict__ block_bins,
                                                                         const uint32_t* __restrict__ hist, const uint32_t* __restrict__ rem_rank,
                                                                         const uint32_t* __restrict__ merge_rank,
                                                                         uint32_t nblocks, uint32_t total, uint32_t unit,
                                                                         uint32_t* __restrict__ offsets, uint32_t* __restrict__ unit_off,
                                                                         uint32_t* __restrict__ full_off, uint32_t* __restrict__ rem_list,
                                                                         uint32_t* __restrict__ merge_list, uint32_t items) {
    wave_priority<APK_PRIO_SORT>();
    const uint32_t i = blockIdx.x * MSM_SCAN_BLOCK + threadIdx.x;     // one bucket per thread here; its scan block held `items` per thread
    if (i >= total) return;
    const uint32_t blk = i / (MSM_SCAN_BLOCK * items);
    offsets[i] += block_tot[blk];
    unit_off[i] += block_tot[nblocks + blk];
    full_off[i] += block_tot[2 * nblocks + blk];
    const uint32_t h = hist[i], hf = h / unit, rem = h - hf * unit, hu = hf + (rem ? 1u : 0u);
    if (rem) rem_list[block_bins[blk * MSM_BINS + rem] + rem_rank[i]] = i;
    merge_list[block_bins[blk * MSM_BINS + MSM_UNIT_MAX + min(hu, (uint32_t)MSM_MERGE_BINS - 1u)] + merge_rank[i]] = i;
}

// ---- bucket accumulation: one lane per work unit ---------------------------------------------------------
// Minimum waves per SIMD asked of the compiler for the 14-limb field: 2 (181 VGPRs).  3 (168 VGPRs, 52 bytes of scratch per lane)
// was measured on one box, interleaved: 1 009-1 024 against 1 040-1 064 proofs/s at BLS12-381 2^14, no difference at 2^21.
#ifndef APK_ACC_WAVES_BLS
#define APK_ACC_WAVES_BLS 2
#endif
#ifndef APK_ACC_THREADS
#define APK_ACC_THREADS 128   // lanes per accumulate workgroup (round 4 variant builds, same box: 64 -> -0.6 %, 256 -> +0.6 % proofs/s: inside the noise)
#endif
template <class FP> struct MsmAcc { static constexpr int MIN_WAVES = FP::N > 8 ? APK_ACC_WAVES_BLS : 1; static constexpr int THREADS = APK_ACC_THREADS; };
template <class FP>
__global__ void __launch_bounds__(APK_ACC_THREADS, MsmAcc<FP>::MIN_WAVES) msm_accumulate_kernel(const Affine<FP>* __restrict__ table,
                                                             const uint32_t* __restrict__ sorted,
                                                             const uint32_t* __restrict__ offsets,
                                                             const uint32_t* __restrict__ unit_off,
                                                             const uint32_t* __restrict__ full_off,
                                                             const uint32_t* __restrict__ rem_list,
                                                             uint32_t total_buckets, uint32_t max_units, uint32_t unit,
                                                             XYZZ<FP, FeU<FP>>* __restrict__ partial) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= max_units) return;
    const uint32_t total_units = unit_off[total_buckets];
    if (u >= total_units) return;
    const uint32_t n_full = full_off[total_buckets];
    uint32_t beg, end, slot;
    if (u < n_full) {
        // full unit: bucket = last k with full_off[k] <= u  (buckets without a full unit have full_off[k] == full_off[k+1])
        uint32_t lo = 0, hi = total_buckets;  // invariant: full_off[lo] <= u < full_off[hi]
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (full_off[mid] <= u) lo = mid; else hi = mid;
        }
        const uint32_t slice = u - full_off[lo];
        beg = offsets[lo] + slice * unit;
        end = beg + unit;
        slot = unit_off[lo] + slice;
    } else {
        // remainder unit (sorted by length): the tail of its bucket's entries
        const uint32_t k = rem_list[u - n_full];
        const uint32_t o0 = offsets[k], o1 = offsets[k + 1];
        const uint32_t nf = (o1 - o0) / unit;
        beg = o0 + nf * unit;
        end = o1;
        slot = unit_off[k] + nf;
    }
    // table records are packed R'-domain words: unpack to unsaturated limbs, accumulate carry-free (ffu.h)
    // (madd_lazy: no conditional subtractions; every XYZZ buffer of the MSM holds points of ec.h's lazy class)
    XYZZ<FP, FeU<FP>> acc = XYZZ<FP, FeU<FP>>::inf();
    bool flipped = false, unit_z = false;
#ifndef APK_ACC_NO_PREFETCH
    // Software pipeline.  An iteration used to start with two DEPENDENT loads (the entry, then its table record: a random
    // 64/96-byte gather) and nothing to do until they landed - hidden only by other waves, of which the 14-limb field has two
    // per SIMD and a lone 2^17 MSM two or three.  Now the record of entry e is in registers while the record of entry e + 1
    // and the index of entry e + 2 are in flight behind the ~2 000 (4 700) instructions of the addition.  Past the end the
    // loads repeat the last entry (no branch in the loop; the values are not used).
    if (beg < end) {
        const uint32_t last = end - 1;
        uint32_t v0 = sorted[beg];
        uint32_t v1 = sorted[beg + 1 <= last ? beg + 1 : last];
        Affine<FP> rec = table[v0 & 0x7fffffffu];
        for (uint32_t e = beg; e < end; e++) {
            const Affine<FP> nxt = table[v1 & 0x7fffffffu];
            const uint32_t v2 = sorted[e + 2 <= last ? e + 2 : last];
            acc.madd_lazy(unpack_affine<FP>(rec), (v0 >> 31) != 0, flipped, unit_z);
            rec = nxt; v0 = v1; v1 = v2;
        }
    }
#else
    for (uint32_t e = beg; e < end; e++) {
        uint32_t v = sorted[e];
        Affine<FP> rec = table[v & 0x7fffffffu];
        acc.madd_lazy(unpack_affine<FP>(rec), (v >> 31) != 0, flipped, unit_z);
    }
#endif
    acc.lazy_fix_sign(flipped);
    partial[slot] = acc;
}

template <class PT>
__device__ __forceinline__ PT shfl_down_point(const PT& p, int delta, int width) {
    PT r;
    constexpr int NW = sizeof(PT) / 4;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&p);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int i = 0; i < NW; i++) dst[i] = __shfl_down(src[i], delta, width);
    return r;
}

// merge a bucket's unit partials: MSM_COMBINE_LANES lanes per bucket, strided sums + shuffle tree.
// ONE launch does both cases: blocks [0, normal_blocks) run this light path, the MSM_HEAVY_BLOCKS blocks behind them the
// whole-workgroup merge of skewed buckets (msm_combine_heavy_body below).
constexpr uint32_t MSM_HEAVY_BLOCKS = 128;
template <class FP>
__device__ __forceinline__ void msm_combine_heavy_body(const XYZZ<FP, FeU<FP>>* __restrict__ partial, const uint32_t* __restrict__ unit_off,
                                                       uint32_t total_buckets, XYZZ<FP, FeU<FP>>* __restrict__ bucket_sum, uint32_t blk, uint32_t nblk,
                                                       uint32_t heavy);

// lanes per bucket of the merge: as many as bring the partials per lane down to `per_lane`, from the bucket scan's own count of
// partials per non-empty bucket (avg_partials; null: the host's choice stands), at most 2^max_log
__device__ __forceinline__ int msm_combine_lanes(int max_log, const uint32_t* __restrict__ avg_partials, uint32_t per_lane) {
    if (!avg_partials) return max_log;
    const uint32_t avg = *avg_partials + 1u;
    int l = 0;
    while (l < max_log && (avg >> l) > per_lane) l++;
    return l;
}
template <class FP>
__global__ void __launch_bounds__(256) msm_combine_kernel(const XYZZ<FP, FeU<FP>>* __restrict__ partial,
                                                          const uint32_t* __restrict__ unit_off,
                                                          const uint32_t* __restrict__ merge_list,   // buckets by number of partials, or null: bucket order
                                                          uint32_t total_buckets,
                                                          int lanes_log,  // lanes per bucket = 2^lanes_log <= MSM_COMBINE_LANES: the most the grid covers
                                                          uint32_t normal_blocks,
                                                          XYZZ<FP, FeU<FP>>* __restrict__ bucket_sum,
                                                          const uint32_t* __restrict__ avg_partials, uint32_t per_lane) {
    wave_priority<APK_PRIO_TAIL>();
    using PT = XYZZ<FP, FeU<FP>>;
    lanes_log = msm_combine_lanes(lanes_log, avg_partials, per_lane);
    const uint32_t heavy = msm_heavy_threshold(lanes_log, per_lane);
    if (blockIdx.x >= normal_blocks) {   // block-uniform
        msm_combine_heavy_body<FP>(partial, unit_off, total_buckets, bucket_sum, blockIdx.x - normal_blocks, gridDim.x - normal_blocks, heavy);
        return;
    }
    if ((uint64_t)blockIdx.x * blockDim.x >= ((uint64_t)total_buckets << lanes_log)) return;   // the grid was sized for the most lanes
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t LANES = 1u << lanes_log;
    const uint32_t slot = gid >> lanes_log;
    const uint32_t lane = gid & (LANES - 1);
    uint32_t k = slot;
    if (merge_list && slot < total_buckets) k = merge_list[slot];
    PT acc = PT::inf();
    uint32_t beg = 0, end = 0;
    if (slot < total_buckets) { beg = unit_off[k]; end = unit_off[k + 1]; }
    if (end - beg > heavy) end = beg;   // skewed bucket: left to the heavy blocks
    for (uint32_t u = beg + lane; u < end; u += LANES) acc.add_lazy(partial[u]);
    // all lanes of the wave take part in every shuffle; groups with nothing to add see infinities
    const uint32_t n_units = end - beg;
    uint64_t need = __ballot(n_units > 1);
    if (need) {
        for (int d = (int)LANES / 2; d >= 1; d >>= 1) {
            PT o = shfl_down_point<PT>(acc, d, (int)LANES);
            if (lane < (uint32_t)d && n_units > (uint32_t)d) acc.add_lazy(o);
        }
    }
    if (slot < total_buckets && lane == 0 && unit_off[k + 1] - unit_off[k] <= heavy) bucket_sum[k] = acc;
}

// Skewed inputs (e.g. a Lagrange-basis commitment of a witness full of ones): a bucket with more than MSM_HEAVY_UNITS unit
// partials gets a whole workgroup - strided sums over 256 lanes, then an LDS tree - instead of a few lanes walking thousands
// of partials.  Each of the (few) blocks scans a slice of the bucket list and only does work for heavy buckets, so the launch
// costs microseconds when there are none (uniform scalars never produce one).
template <class FP>
__device__ __forceinline__ void msm_combine_heavy_body(const XYZZ<FP, FeU<FP>>* __restrict__ partial, const uint32_t* __restrict__ unit_off,
                                                       uint32_t total_buckets, XYZZ<FP, FeU<FP>>* __restrict__ bucket_sum, uint32_t blk, uint32_t nblk,
                                                       uint32_t heavy) {
    using PT = XYZZ<FP, FeU<FP>>;
    __shared__ PT sm[256];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (total_buckets + nblk - 1) / nblk;
    const uint32_t k0 = blk * per, k1 = min(k0 + per, total_buckets);
    // the common case has no heavy bucket at all: every thread looks at its own buckets first (one round of loads instead of
    // a serial walk over the slice), and the block only walks the slice if somebody saw one
    bool any = false;
    for (uint32_t k = k0 + t; k < k1; k += 256) any |= unit_off[k + 1] - unit_off[k] > heavy;
    if (!__syncthreads_or(any)) return;
    for (uint32_t k = k0; k < k1; k++) {
        const uint32_t beg = unit_off[k], end = unit_off[k + 1];
        if (end - beg <= heavy) continue;   // uniform across the block
        PT acc = PT::inf();
        for (uint32_t u = beg + t; u < end; u += 256) acc.add_lazy(partial[u]);
        sm[t] = acc;
        __syncthreads();
        for (uint32_t d = 128; d >= 1; d >>= 1) {
            if (t < d) { PT o = sm[t + d]; acc.add_lazy(o); sm[t] = acc; }
            __syncthreads();
        }
        if (t == 0) bucket_sum[k] = acc;
        __syncthreads();
    }
}

// ---- weighted bucket reduction  S = sum_k k * B_k,  k = idx + 1 -----------------------------------------------------
// Work-efficient two-level form.  Write idx = hi * COLS + lo; then
//     S = COLS * sum_hi hi * R_hi  +  sum_lo (lo + 1) * C_lo,      R_hi = row sums, C_lo = column sums
// (2 additions per bucket instead of (c-1)/2 for a bit-wise reduction over all buckets), and the two small weighted
// sums over <= 256 elements are evaluated bit-wise: sum_b 2^b * (sum of the elements whose weight has bit b set).
// Critical path ~ log2(#buckets) + c point operations; every level is an LDS tree.

// grid (rows + cols, batch): block x < rows sums row x (contiguous), else column x - rows (stride COLS)
template <class FP>
__global__ void __launch_bounds__(256) msm_rowcol_kernel(const XYZZ<FP, FeU<FP>>* __restrict__ bucket_sum, uint32_t nb, uint32_t rows,
                                                         uint32_t cols, XYZZ<FP, FeU<FP>>* __restrict__ rc) {
    wave_priority<APK_PRIO_TAIL>();
    using PT = XYZZ<FP, FeU<FP>>;
    __shared__ PT sm[256];
    const uint32_t m = blockIdx.y, x = blockIdx.x, t = threadIdx.x;
    const PT* src = bucket_sum + (size_t)m * nb;
    PT acc = PT::inf();
    if (x < rows) {
        for (uint32_t lo = t; lo < cols; lo += 256) acc.add_lazy(src[x * cols + lo]);
    } else {
        const uint32_t col = x - rows;
        for (uint32_t hi = t; hi < rows; hi += 256) acc.add_lazy(src[hi * cols + col]);
    }
    sm[t] = acc;
    __syncthreads();
    for (uint32_t d = 128; d >= 1; d >>= 1) {
        if (t < d) { PT o = sm[t + d]; acc.add_lazy(o); sm[t] = acc; }
        __syncthreads();
    }
    if (t == 0) rc[(size_t)m * (rows + cols) + x] = acc;
}

// grid (nbits, 2, batch): which = 0 -> rows part (elements R_1..R_{rows-1}, weight = index), which = 1 -> column part
// (elements C_0..C_{cols-1}, weight = index + 1).  out[(m*2 + which)*32 + bit] = sum of the elements whose weight has `bit`.
template <class FP>
__global__ void __launch_bounds__(256) msm_bitsum_kernel(const XYZZ<FP, FeU<FP>>* __restrict__ rc, uint32_t rows, uint32_t cols,
                                                         XYZZ<FP, FeU<FP>>* __restrict__ bit_partial) {
    wave_priority<APK_PRIO_TAIL>();
    using PT = XYZZ<FP, FeU<FP>>;
    __shared__ PT sm[256];
    const uint32_t bit = blockIdx.x, which = blockIdx.y, m = blockIdx.z, t = threadIdx.x;
    const PT* src = rc + (size_t)m * (rows + cols) + (which ? rows : 0);
    const uint32_t count = which ? cols : rows;
    PT acc = PT::inf();
    for (uint32_t i = t; i < count; i += 256) {
        const uint32_t weight = which ? i + 1 : i;
        if ((weight >> bit) & 1u) acc.add_lazy(src[i]);
    }
    sm[t] = acc;
    __syncthreads();
    for (uint32_t d = 128; d >= 1; d >>= 1) {
        if (t < d) { PT o = sm[t + d]; acc.add_lazy(o); sm[t] = acc; }
        __syncthreads();
    }
    if (t == 0) bit_partial[((size_t)m * 2 + which) * 32 + bit] = acc;
}

// one workgroup of 64 lanes per msm: lane (which, bit) scales its bit-sum by 2^bit (and the row part by COLS = 2^cols_log),
// LDS tree over the 64 lanes, conversion back to gnark's radix and to affine form
template <class FP>
__global__ void __launch_bounds__(64) msm_final_kernel(const XYZZ<FP, FeU<FP>>* __restrict__ bit_partial, uint32_t nbits, int cols_log,
                                                       Affine<FP>* __restrict__ result, XYZZ<FP>* __restrict__ result_xyzz) {
    wave_priority<APK_PRIO_TAIL>();
    using PT = XYZZ<FP, FeU<FP>>;
    __shared__ PT sm[64];
    const uint32_t m = blockIdx.x, t = threadIdx.x;
    const uint32_t which = t >> 5, bit = t & 31;
    PT acc = PT::inf();
    if (bit < nbits) {
        acc = bit_partial[((size_t)m * 2 + which) * 32 + bit];
        const int dbl = (int)bit + (which == 0 ? cols_log : 0);
        for (int i = 0; i < dbl; i++) acc = PT::dbl_lazy(acc);
    }
    sm[t] = acc;
    __syncthreads();
    for (uint32_t d = 32; d >= 1; d >>= 1) {
        if (t < d) { PT o = sm[t + d]; acc.add_lazy(o); sm[t] = acc; }
        __syncthreads();
    }
    if (t == 0) {
        XYZZ<FP> g = to_fe_point<FP>(acc);  // back to gnark's Montgomery radix
        if (result_xyzz) result_xyzz[m] = g;
        if (result) result[m] = g.to_affine();
    }
}

// ---- the three reduction kernels with FOUR LANES PER POINT OPERATION (ec.h add_quad_general / dbl_quad_general) ---------------
// They are chains of dependent point operations on a handful of waves: a lone lane needs ~7 us per addition.  With a quad per
// logical thread (lanes 4t .. 4t+3 hold copies of the same point, LDS holds one copy) an addition is 4 product stages plus
// branch-free operand selects and DPP broadcasts: ~1600 instructions instead of ~3700.  blockDim = 4 * logical threads.
// The four-lane functions handle the general case only and keep no large temporaries (nothing may live in scratch memory):
// infinities are copies, a degenerate pair (P = +-Q) falls back to the one-lane form on every lane.
// one tree level: acc += o
template <class PT>
__device__ __forceinline__ void quad_tree_add(PT& acc, const PT& o, int q) {
    if (acc.is_inf()) acc = o;
    else if (!o.is_inf()) {
        bool degenerate;
        acc.add_quad_general(o, q, degenerate);
        if (degenerate) acc.add_lazy(o);
    }
}

// The same merge with FOUR LANES PER POINT OPERATION (ec.h add_quad_general), for a batch that has the GPU to itself: the merge is
// then a chain of 5 - 8 dependent additions on a few hundred lone waves (BLS12-381 2^14: 105 - 150 us of a 0.5 ms MSM), and a quad
// finishes an addition in 4 product stages instead of 14 products.  1.6 x the VALU instructions: never under load (run_msm_body).
// 64 quads per workgroup; the heavy-bucket blocks behind the light ones are the one-lane form above.
template <class FP>
__global__ void __launch_bounds__(256) msm_combine_quad_kernel(const XYZZ<FP, FeU<FP>>* __restrict__ partial,
                                                               const uint32_t* __restrict__ unit_off,
                                                               const uint32_t* __restrict__ merge_list, uint32_t total_buckets,
                                                               int lanes_log,  // QUADS per bucket = 2^lanes_log <= MSM_COMBINE_LANES: the most the grid covers
                                                               uint32_t normal_blocks,
                                                               XYZZ<FP, FeU<FP>>* __restrict__ bucket_sum,
                                                               const uint32_t* __restrict__ avg_partials, uint32_t per_lane) {
    wave_priority<APK_PRIO_TAIL>();
    using PT = XYZZ<FP, FeU<FP>>;
    lanes_log = msm_combine_lanes(lanes_log, avg_partials, per_lane);
    const uint32_t heavy = msm_heavy_threshold(lanes_log, per_lane);
    if (blockIdx.x >= normal_blocks) {   // block-uniform
        msm_combine_heavy_body<FP>(partial, unit_off, total_buckets, bucket_sum, blockIdx.x - normal_blocks, gridDim.x - normal_blocks, heavy);
        return;
    }
    if ((uint64_t)blockIdx.x * blockDim.x >= (((uint64_t)total_buckets << lanes_log) << 2)) return;
    const uint32_t gq = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;      // the quad's number
    const int q = threadIdx.x & 3;
    const uint32_t LANES = 1u << lanes_log;
    const uint32_t slot = gq >> lanes_log;
    const uint32_t lane = gq & (LANES - 1);
    uint32_t k = slot;
    if (merge_list && slot < total_buckets) k = merge_list[slot];
    PT acc = PT::inf();
    uint32_t beg = 0, end = 0;
    if (slot < total_buckets) { beg = unit_off[k]; end = unit_off[k + 1]; }
    if (end - beg > heavy) end = beg;
    for (uint32_t u = beg + lane; u < end; u += LANES) quad_tree_add(acc, partial[u], q);
    const uint32_t n_units = end - beg;
    uint64_t need = __ballot(n_units > 1);
    if (need) {
        for (int d = (int)LANES / 2; d >= 1; d >>= 1) {
            PT o = shfl_down_point<PT>(acc, d * 4, (int)LANES * 4);
            if (lane < (uint32_t)d && n_units > (uint32_t)d) quad_tree_add(acc, o, q);
        }
    }
    if (slot < total_buckets && lane == 0 && q == 0 && unit_off[k + 1] - unit_off[k] <= heavy) bucket_sum[k] = acc;
}

// Row / column sums for throughput contexts: one lane per operation while a level still has more than 16 additions to do
// (they run side by side at no extra cost), four lanes per operation for the last five levels, where the chain of dependent
// additions is all that is left: the latency of those levels halves for a fraction of a percent more VALU work.
template <class FP>
__global__ void __launch_bounds__(256) msm_rowcol_hybrid_kernel(const XYZZ<FP, FeU<FP>>* __restrict__ bucket_sum, uint32_t nb, uint32_t rows,
                                                                uint32_t cols, XYZZ<FP, FeU<FP>>* __restrict__ rc) {
    wave_priority<APK_PRIO_TAIL>();
    using PT = XYZZ<FP, FeU<FP>>;
    __shared__ PT sm[256];
    const uint32_t m = blockIdx.y, x = blockIdx.x, t = threadIdx.x;
    const PT* src = bucket_sum + (size_t)m * nb;
    PT acc = PT::inf();
    if (x < rows) {
        for (uint32_t lo = t; lo < cols; lo += 256) acc.add_lazy(src[x * cols + lo]);
    } else {
        const uint32_t col = x - rows;
        for (uint32_t hi = t; hi < rows; hi += 256) acc.add_lazy(src[hi * cols + col]);
    }
    sm[t] = acc;
    __syncthreads();
    uint32_t d = 128;
    for (; d > 16; d >>= 1) {
        if (t < d) { PT o = sm[t + d]; acc.add_lazy(o); sm[t] = acc; }
        __syncthreads();
    }
    const uint32_t tq = t >> 2;
    const int q = t & 3;
    PT qa = PT::inf();
    if (tq < 16) qa = sm[tq];
    for (; d >= 1; d >>= 1) {
        if (tq < d) { PT o = sm[tq + d]; quad_tree_add(qa, o, q); if (q == 0) sm[tq] = qa; }
        __syncthreads();
    }
    if (t == 0) rc[(size_t)m * (rows + cols) + x] = qa;
}

// Row / column sums when other proofs keep the GPU busy: SIXTEEN lanes per row or column, each lane adds its 8-16 elements one
// after the other, then a four-level shuffle tree.  The tree kernels above spend most of their instructions on levels where a
// handful of lanes of a wave are live (255 additions cost ~20 k wave-instructions per row); here a wave of four lines does 19
// addition-times for 4 x 255 additions (~14 k per row, ~8 k per column).  The chain is twice as long (19 dependent additions
// instead of 3 + 5 short ones), so a lone proof keeps the tree form - the host picks per batch (run_msm_body).
template <class FP, int LPL>   // LPL = lanes per line (16 or 8): fewer lanes = fewer idle shuffle levels, longer chains
__global__ void __launch_bounds__(256) msm_rowcol_serial_kernel(const XYZZ<FP, FeU<FP>>* __restrict__ bucket_sum, uint32_t nb, uint32_t rows,
                                                                uint32_t cols, XYZZ<FP, FeU<FP>>* __restrict__ rc) {
    wave_priority<APK_PRIO_TAIL>();
    using PT = XYZZ<FP, FeU<FP>>;
    constexpr uint32_t LINES = 256 / LPL;
    const uint32_t m = blockIdx.y;
    const uint32_t line = blockIdx.x * LINES + (threadIdx.x / LPL);   // rows first, then columns; a wave holds 64 / LPL lines of one kind
    const uint32_t j = threadIdx.x % LPL;
    const PT* src = bucket_sum + (size_t)m * nb;
    PT acc = PT::inf();
    if (line < rows) {
        for (uint32_t lo = j; lo < cols; lo += LPL) acc.add_lazy(src[line * cols + lo]);
    } else if (line < rows + cols) {
        const uint32_t col = line - rows;
        for (uint32_t hi = j; hi < rows; hi += LPL) acc.add_lazy(src[hi * cols + col]);
    }
    for (int d = LPL / 2; d >= 1; d >>= 1) {
        PT o = shfl_down_point<PT>(acc, d, LPL);
        if (j < (uint32_t)d) acc.add_lazy(o);
    }
    if (j == 0 && line < rows + cols) rc[(size_t)m * (rows + cols) + line] = acc;
}

// Workgroup size of the four-lane reduction kernels.  256 registers per lane are enough for the 9-limb field at 512 lanes;
// the 14-limb field (BLS12-381: a point is 56 registers, an addition keeps four of them and a product's operands live) spilled
// to scratch memory there, so its workgroups are 256 lanes: one wave per SIMD, the whole 512-entry register file per lane
// (the compiler parks what does not fit the 256 architected VGPRs in accumulation registers, not in memory).
template <class FP> struct MsmQuad { static constexpr uint32_t THREADS = FP::N > 8 ? 256 : 512, LT = THREADS / 4; };

template <class FP>
__global__ void __launch_bounds__(MsmQuad<FP>::THREADS) msm_rowcol_quad_kernel(const XYZZ<FP, FeU<FP>>* __restrict__ bucket_sum, uint32_t nb, uint32_t rows,
                                                              uint32_t cols, XYZZ<FP, FeU<FP>>* __restrict__ rc) {
    wave_priority<APK_PRIO_TAIL>();
    using PT = XYZZ<FP, FeU<FP>>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    PT* sm = reinterpret_cast<PT*>(smem_raw);
    const uint32_t m = blockIdx.y, x = blockIdx.x, t = threadIdx.x >> 2, LT = blockDim.x >> 2;
    const int q = threadIdx.x & 3;
    const PT* src = bucket_sum + (size_t)m * nb;
    PT acc = PT::inf();
    if (x < rows) {
        for (uint32_t lo = t; lo < cols; lo += LT) quad_tree_add(acc, src[x * cols + lo], q);
    } else {
        const uint32_t col = x - rows;
        for (uint32_t hi = t; hi < rows; hi += LT) quad_tree_add(acc, src[hi * cols + col], q);
    }
    if (q == 0) sm[t] = acc;
    __syncthreads();
    for (uint32_t d = LT >> 1; d >= 1; d >>= 1) {
        if (t < d) { PT o = sm[t + d]; quad_tree_add(acc, o, q); if (q == 0) sm[t] = acc; }
        __syncthreads();
    }
    if (t == 0 && q == 0) rc[(size_t)m * (rows + cols) + x] = acc;
}

// final phase as a device function: 64 quads (which, bit) scale their bit-sum by 2^bit (rows part by COLS as well), LDS tree,
// conversion back to gnark's radix.  Needs the first 256 threads of the block (threads beyond idle) and 64 PT of LDS.
template <class FP>
__device__ __forceinline__ void msm_final_quad_body(const XYZZ<FP, FeU<FP>>* __restrict__ bit_partial, uint32_t nbits, int cols_log, uint32_t m,
                                                    XYZZ<FP, FeU<FP>>* sm, XYZZ<FP>* __restrict__ result_xyzz) {
    using PT = XYZZ<FP, FeU<FP>>;
    const uint32_t t = threadIdx.x >> 2;
    const int q = threadIdx.x & 3;
    const bool live = threadIdx.x < 256;
    const uint32_t which = t >> 5, bit = t & 31;
    PT acc = PT::inf();
    int dbl = 0;
    if (live && bit < nbits) {
        acc = bit_partial[((size_t)m * 2 + which) * 32 + bit];
        dbl = (int)bit + (which == 0 ? cols_log : 0);
    }
    const int max_dbl = (int)nbits - 1 + cols_log;
    for (int i = 0; i < max_dbl; i++) {
        if (i < dbl && !acc.is_inf()) acc = PT::dbl_quad_general(acc, q);
    }
    if (live && q == 0) sm[t] = acc;
    __syncthreads();
    for (uint32_t d = 32; d >= 1; d >>= 1) {
        if (live && t < d) { PT o = sm[t + d]; quad_tree_add(acc, o, q); if (q == 0) sm[t] = acc; }
        __syncthreads();
    }
    if (threadIdx.x == 0) result_xyzz[m] = to_fe_point<FP>(acc);   // back to gnark's Montgomery radix; affine conversion on the host
}

template <class FP>
__global__ void __launch_bounds__(256) msm_final_quad_kernel(const XYZZ<FP, FeU<FP>>* __restrict__ bit_partial, uint32_t nbits, int cols_log,
                                                             Affine<FP>* __restrict__ result, XYZZ<FP>* __restrict__ result_xyzz) {
    wave_priority<APK_PRIO_TAIL>();
    using PT = XYZZ<FP, FeU<FP>>;
    __shared__ PT sm[64];
    const uint32_t m = blockIdx.x, t = threadIdx.x >> 2;
    const int q = threadIdx.x & 3;
    const uint32_t which = t >> 5, bit = t & 31;
    PT acc = PT::inf();
    int dbl = 0;
    if (bit < nbits) {
        acc = bit_partial[((size_t)m * 2 + which) * 32 + bit];
        dbl = (int)bit + (which == 0 ? cols_log : 0);
    }
    // every quad runs the same number of loop trips (the longest chain); quads that are done, or hold infinity, idle
    const int max_dbl = (int)nbits - 1 + cols_log;
    for (int i = 0; i < max_dbl; i++) {
        if (i < dbl && !acc.is_inf()) acc = PT::dbl_quad_general(acc, q);
    }
    if (q == 0) sm[t] = acc;
    __syncthreads();
    for (uint32_t d = 32; d >= 1; d >>= 1) {
        if (t < d) { PT o = sm[t + d]; quad_tree_add(acc, o, q); if (q == 0) sm[t] = acc; }
        __syncthreads();
    }
    if (t == 0 && q == 0) {
        XYZZ<FP> g = to_fe_point<FP>(acc);  // back to gnark's Montgomery radix
        if (result_xyzz) result_xyzz[m] = g;
        if (result) result[m] = g.to_affine();
    }
}

// Bit sums AND the final scaling in one launch: the workgroup that finishes the last of an MSM's 2 * nbits bit sums (an
// agent-scope counter, reset by that workgroup for the next MSM) goes on to run the final phase - no spinning, so any number
// of concurrent streams is safe.  `lt` logical threads (quads) do the bit sum; blockDim.x = max(4 * lt, 256); LDS holds
// max(lt, 64) points.
template <class FP>
__global__ void __launch_bounds__(MsmQuad<FP>::THREADS) msm_bitsum_final_quad_kernel(const XYZZ<FP, FeU<FP>>* __restrict__ rc, uint32_t rows, uint32_t cols,
                                                                    uint32_t lt, XYZZ<FP, FeU<FP>>* __restrict__ bit_partial,
                                                                    uint32_t* __restrict__ done_count, int cols_log,
                                                                    XYZZ<FP>* __restrict__ result_xyzz) {
    wave_priority<APK_PRIO_TAIL>();
    using PT = XYZZ<FP, FeU<FP>>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    PT* sm = reinterpret_cast<PT*>(smem_raw);
    __shared__ uint32_t s_last;
    const uint32_t bit = blockIdx.x, which = blockIdx.y, m = blockIdx.z, t = threadIdx.x >> 2, LT = lt;
    const int q = threadIdx.x & 3;
    const bool live = t < LT;
    const PT* src = rc + (size_t)m * (rows + cols) + (which ? rows : 0);
    const uint32_t count = which ? cols : rows;
    PT acc = PT::inf();
    if (live)
        for (uint32_t i = t; i < count; i += LT) {
            const uint32_t weight = which ? i + 1 : i;
            if ((weight >> bit) & 1u) acc.add_lazy(src[i]);      // at most a couple per lane: the first is a copy
        }
    if (live && q == 0) sm[t] = acc;
    __syncthreads();
    for (uint32_t d = LT >> 1; d >= 1; d >>= 1) {
        if (live && t < d) { PT o = sm[t + d]; quad_tree_add(acc, o, q); if (q == 0) sm[t] = acc; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        bit_partial[((size_t)m * 2 + which) * 32 + bit] = acc;
        // release our bit sum, count it; the workgroup that sees the full count acquires everybody's
        const uint32_t old = __hip_atomic_fetch_add(&done_count[m], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (old == 2 * gridDim.x - 1) ? 1u : 0u;
        if (s_last) __hip_atomic_store(&done_count[m], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    msm_final_quad_body<FP>(bit_partial, gridDim.x, cols_log, m, sm, result_xyzz);
}

template <class FP>
__global__ void __launch_bounds__(MsmQuad<FP>::THREADS) msm_bitsum_quad_kernel(const XYZZ<FP, FeU<FP>>* __restrict__ rc, uint32_t rows, uint32_t cols,
                                                              XYZZ<FP, FeU<FP>>* __restrict__ bit_partial) {
    wave_priority<APK_PRIO_TAIL>();
    using PT = XYZZ<FP, FeU<FP>>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    PT* sm = reinterpret_cast<PT*>(smem_raw);
    const uint32_t bit = blockIdx.x, which = blockIdx.y, m = blockIdx.z, t = threadIdx.x >> 2, LT = blockDim.x >> 2;
    const int q = threadIdx.x & 3;
    const PT* src = rc + (size_t)m * (rows + cols) + (which ? rows : 0);
    const uint32_t count = which ? cols : rows;
    PT acc = PT::inf();
    for (uint32_t i = t; i < count; i += LT) {
        const uint32_t weight = which ? i + 1 : i;
        if ((weight >> bit) & 1u) acc.add_lazy(src[i]);      // at most a couple per lane: the first is a copy
    }
    if (q == 0) sm[t] = acc;
    __syncthreads();
    for (uint32_t d = LT >> 1; d >= 1; d >>= 1) {
        if (t < d) { PT o = sm[t + d]; quad_tree_add(acc, o, q); if (q == 0) sm[t] = acc; }
        __syncthreads();
    }
    if (t == 0 && q == 0) bit_partial[((size_t)m * 2 + which) * 32 + bit] = acc;
}

// ---- table construction: table[j*n + i] = 2^(off[j]) * P_i (affine), stored as packed R'-domain records (ffu.h) ----------------------------------------
// `pre` (canonical words, most significant last; nwords = 0: none): every base is multiplied by it first.  The backends pass
// R^-1 mod r, which lets msm_digits_kernel read Montgomery-form scalars without converting them.
struct MsmPreScale { uint32_t l[16]; int nwords; };
template <class FP>
__global__ void __launch_bounds__(256) msm_table_kernel(const Affine<FP>* __restrict__ bases, uint32_t n, MsmWindows win, MsmPreScale pre,
                                                        Affine<FP>* __restrict__ table) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<FP> p = bases[i];
    if (pre.nwords > 0 && !p.is_inf()) {
        XYZZ<FP> acc = XYZZ<FP>::inf();
        for (int w = pre.nwords - 1; w >= 0; w--) {
            const uint32_t word = pre.l[w];   // uniform
            for (int b = 31; b >= 0; b--) {
                acc = XYZZ<FP>::dbl(acc);
                if ((word >> b) & 1u) acc.madd(p);
            }
        }
        p = acc.to_affine();
    }
    table[i] = to_table_record<FP>(p);
    for (int j = 1; j < win.W; j++) {
        XYZZ<FP> q = XYZZ<FP>::dbl_affine(p);
        for (int k = 1; k < win.width[j - 1]; k++) q = XYZZ<FP>::dbl(q);
        p = q.to_affine();
        table[(size_t)j * n + i] = to_table_record<FP>(p);
    }
}

// ---- out[i] = scalars[i] * base : SRS generation from a known tau (gnark test/unsafekzg, setup/setup.go:103) ----
template <class FR, class FP>
__global__ void __launch_bounds__(256) g1_mul_batch_kernel(Affine<FP> base, const Fe<FR>* __restrict__ scalars, uint32_t count,
                                                           Affine<FP>* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fe<FR> s = Fe<FR>::from_mont(scalars[i]);
    XYZZ<FP> acc = XYZZ<FP>::inf();
    for (int w = Fe<FR>::N - 1; w >= 0; w--) {
        for (int b = 31; b >= 0; b--) {
            acc = XYZZ<FP>::dbl(acc);
            if ((s.l[w] >> b) & 1u) acc.madd(base);
        }
    }
    out[i] = acc.to_affine();
}

}  // namespace apk
