// Per-curve backend: circuit context resident in HBM + the PLONK prover driven from one host thread per
// proof.  Instantiated for BN254 and BLS12-381 in backend_bn254.hip / backend_bls12381.hip.
//
// Replaces plonk.Prove (/root/reference/algoplonk.go:89) and the device-relevant part of plonk.Setup
// (/root/reference/setup/setup.go:107,149) [both UPSTREAM gnark v0.15.0, not vendored].  The round
// structure follows SURVEY.md §3.3; what every round must produce is pinned by the verifier template
// (/root/reference/verifier/templateLogicSigBN254.go, lines cited at each step).  GPU-natural schedule,
// not gnark's: everything between two Fiat-Shamir challenges is a chain of launches on one HIP stream with
// all polynomials resident; [Z], [H1..3] and the openings are committed over the CANONICAL SRS, [L][R][O] over the canonical SRS
// after the iNTT or - round 5, when the witness makes it cheaper - over the Lagrange SRS the way gnark does (same group
// elements either way); the quotient is evaluated on ONE 4n coset (4 coset NTTs per proof, the 7 trace polynomials' coset
// evaluations are precomputed per circuit).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <new>
#include <mutex>
#include <string.h>
#include <time.h>
#include <vector>

#include "backend.h"
#include "kernels_msm.h"
#include "kernels_ntt.h"
#include "kernels_poly.h"
#include "gang_kernel.h"
#include "kernels_setup.h"
#include "host_msm.h"
#include "slot_gate.h"
#include "gang.h"
#include "sha256.h"

namespace apk {

#define HIPCHK(x)                                                                                              \
    do {                                                                                                       \
        hipError_t e_ = (x);                                                                                   \
        if (e_ != hipSuccess) {                                                                                \
            set_error("%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__);                        \
            return APK_ERR_HIP;                                                                                \
        }                                                                                                      \
    } while (0)
#define CHK(x)                 \
    do {                       \
        int r_ = (x);          \
        if (r_ != APK_OK) return r_; \
    } while (0)
#define KCHK() HIPCHK(hipGetLastError())

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        HIPCHK(hipMalloc(&p, n));
        bytes = n;
        return APK_OK;
    }
};
template <class T> static inline T* ptr(const DevBuf& b) { return reinterpret_cast<T*>(b.p); }

static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// A slot's page-locked result buffer (the last kernel of a chain writes there through the buffer's device view):
//   [PIN_AFF ..)     this proof's commitments after sync_results(), affine
//   [PIN_XYZZ ..)    MSM sums as they leave the device, XYZZ: up to MSM_ARGS_MAX points - a gang's merged batch lands in its LEAD's
//   [PIN_FR ..)      evaluations / the grand product's total
//   PIN_TAIL, PIN_DENSITY   the quotient's tail flag, the wires' digit count
constexpr size_t PIN_AFF = 0, PIN_XYZZ = 1024, PIN_FR = 4096, PIN_TAIL = 6144, PIN_DENSITY = 6208, PIN_BYTES = 8192;

// kzg.ToLagrangeG1 on device buffers (defined at the end of this file): out[i] = [L_i(tau)]G1 from in[j] = [tau^j]G1
template <class FRP, class FPP>
int g1_to_lagrange_dev(const Affine<FPP>* d_in, uint64_t n, Affine<FPP>* d_out, hipStream_t st = nullptr);

template <class FRP, class FPP, int CURVE_ID>
class CurveBackend : public Backend {
  public:
    using Fr = Fe<FRP>;
    using Fp = Fe<FPP>;
    using Aff = Affine<FPP>;
    using Pt = XYZZ<FPP>;
    using PtU = XYZZ<FPP, FeU<FPP>>;  // MSM-internal points: unsaturated limbs (ffu.h)
    static constexpr int FPB = FPP::N * 4;  // bytes per Fp element

    // ---------------------------------------------------------------------------------------------- host Fr
    static Fr fr_u64(uint64_t v) {
        Fr a = Fr::zero();
        a.l[0] = (uint32_t)v;
        a.l[1] = (uint32_t)(v >> 32);
        return Fr::to_mont(a);
    }
    // 32 big-endian bytes (any 256-bit value) -> Fr Montgomery, reduced mod r
    static Fr fr_from_be(const uint8_t* be) {
        Fr a;
        for (int i = 0; i < 8; i++) {
            const uint8_t* p = be + 32 - 4 * (i + 1);
            a.l[i] = (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3];
        }
        return Fr::to_mont(a);
    }
    template <class P>
    static void fe_to_be(const Fe<P>& m, uint8_t* be) {
        Fe<P> c = Fe<P>::from_mont(m);
        constexpr int N = P::N;
        for (int i = 0; i < N; i++) {
            uint8_t* p = be + 4 * (N - 1 - i);
            p[0] = (uint8_t)(c.l[i] >> 24); p[1] = (uint8_t)(c.l[i] >> 16); p[2] = (uint8_t)(c.l[i] >> 8); p[3] = (uint8_t)c.l[i];
        }
    }
    // gnark Marshal()/RawBytes(): X||Y big-endian.  Infinity: BLS12-381 = 0x40 then zeros (verifier/verifier.go:95-99, the
    // `_fs` constants of templateLogicSigBLS12_381.go:73-84); BN254 = all zeros - the BN254 template feeds ONE constant to the
    // transcript and to the AVM's ec ops (templateLogicSigBN254.go:57-61,131-132), which take only the all-zero encoding: pinned by
    // executing that template (tests/golden/template_verdicts.json, circuits whose [Qk] / [Qm] are the point at infinity).
    static void g1_raw_bytes(const Aff& p, uint8_t* out) {
        if (p.is_inf()) {
            memset(out, 0, 2 * FPB);
            if (FPB == 48) out[0] = 0x40;
            return;
        }
        fe_to_be<FPP>(p.x, out);
        fe_to_be<FPP>(p.y, out + FPB);
    }

    // ---------------------------------------------------------------------------------------------- state
    struct MsmTables {
        DevBuf table;
        uint32_t n_bases = 0;
        bool built = false;
        bool plain = false;   // multiples of the bases themselves (scalars leave the Montgomery form in the sort) instead of R^-1 * P
    };
    // APK_MSM_GRAPH=1: the launch sequence of an MSM batch (10 kernels + the result copy) is captured once per (slot, table,
    // scalar vectors, lengths) into a hipGraph and replayed with one hipGraphLaunch.
    struct GraphKey {
        const void* table; MsmBatchArgs a;
        bool operator==(const GraphKey& o) const { return table == o.table && memcmp(&a, &o.a, sizeof a) == 0; }
    };
    // a recorded launch of a gang member (klaunch / flush_cmds below)
    static constexpr size_t CMD_BLOB = 1024;
    struct Cmd {
        dim3 grid, block;
        uint32_t lds = 0;
        int (*multi)(hipStream_t, const Cmd* const*, int) = nullptr;   // launches this kernel for `count` recorded instances
        alignas(16) unsigned char blob[CMD_BLOB];                       // Pack<P...>: the instance's arguments
    };
    struct Slot {
        hipStream_t stream = nullptr;      // the stream this slot's launches go to: its own, or - as a gang member - its lead's
        hipStream_t own_stream = nullptr;  // one of the context's stream_pool_, held while this slot leads a gang or proves alone (not owned)
        // gang membership of the proof in flight on this slot (gang.h; set by MemberGuard)
        Slot* lead = nullptr;              // whose stream, MSM workspace, transform scratch and XYZZ result area it uses (itself when alone)
        int gang_idx = 0;
        bool in_gang = false;
        uint32_t res_off = 0;              // first point of this proof's pending MSM sums in the lead's XYZZ area
        uint32_t res_write_off = 0;        // (as a lead) where the launch sequence being queued writes its sums in that area
        Gang gang;                         // used when this slot leads
        std::vector<Cmd> cmds;             // launches recorded since the last merge point (as a gang member)
        hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
        size_t index = 0;          // position in slots_ = the slot's number at the gate
        // polynomials
        DevBuf wl, wr, wo;             // L,R,O Lagrange (n)
        DevBuf cl, cr, co, cz;         // blinded canonical (n+3 capacity)
        DevBuf qk_lag, qk_can;         // completed Qk
        DevBuf ratio, zlag, scan_tot;  // grand product
        DevBuf el, er, eo, ez, eqk;    // 4n coset evaluations
        DevBuf quot, hcan;             // quotient evaluations / coefficients (4n)
        DevBuf pw_z, pw_zi, pw_zw, pw_zwi;  // powers of zeta, 1/zeta, omega*zeta, 1/(omega*zeta)   (n+3)
        DevBuf lin, folded, tmp, q1, q2;    // n+3
        DevBuf eval_partial, eval_result;
        DevBuf pi2_can[APK_MAX_COMMITMENTS], epi2[APK_MAX_COMMITMENTS];
        DevBuf scratch_in;  // upload staging for primitives
        DevBuf ntt_wide;    // NTT_MAX_BATCH transforms of 4n unsaturated-limb elements: the NTT's inter-pass form
        // MSM workspace
        DevBuf ptot2;   // fused two-level sort: two buffers of partition totals (one in use, one zeroed for the next batch)
        uint32_t ptot_parity = 0;
        DevBuf sort_tmp, counts, hist, offsets, unit_off, full_off, rem_rank, rem_list, merge_rank, merge_list, scan_blk, sorted, partial, bucket_sum, rowcol, bit_partial, result, result_xyzz, done_count;
        void* h_pinned = nullptr;  // small pinned staging for results: [0,1024) affine, [1024,2048) XYZZ, [2048,4096) scalars
        uint32_t pending_pts = 0;  // MSM sums waiting in h_pinned for their affine conversion (sync_results)
        uint8_t* d_pinned = nullptr;   // the device's view of h_pinned (zero-copy results: the last kernel of a batch writes them there), or null
        std::vector<std::pair<GraphKey, hipGraphExec_t>> graphs;   // APK_MSM_GRAPH
        bool hook_pending = false; // a commitment batch handed to the context's commit hook at the next sync_results()
        int hook_basis = 0;
        MsmBatchArgs hook_args{};
        DevBuf sc_gather;          // sub-coset split: the G parts of the inverse transform (4n elements), all-gathered in place
        DevBuf tail_flag;          // epoch of the last proof whose quotient had a non-zero tail (tail_nonzero_kernel)
        uint32_t epoch = 0;
        // tail filling (a lone proof only, see tail_fill()): the coset transforms that follow a commitment run on `side` from the
        // moment the batch's accumulate kernel is done, beside the reduction tail that would otherwise have the GPU to itself
        hipStream_t side = nullptr;
        hipEvent_t ev_acc = nullptr, ev_side = nullptr;
        hipEvent_t ev_sync = nullptr;   // blocking-wait event (hipEventBlockingSync): the host thread sleeps instead of spinning
        int mark_acc = 0; bool side_pending = false;   // mark_acc: 1 = event behind the accumulate kernel, 2 = in front of the batch
    };

    int curve_ = CURVE_ID;
    int device_ = 0;
    uint32_t n_ = 0, log_n_ = 0, n4_ = 0;
    uint32_t nb_public_ = 0, nb_commit_ = 0;
    uint32_t cci_[APK_MAX_COMMITMENTS] = {0, 0};
    int c_ = 0, W_ = 0;
    MsmWindows win_{};
    MsmPartCfg part_cfg_{};   // two-level sort: bit layout of the packed entries, partition count (choose_window)
    uint32_t msm_G_max_ = 256;
    bool msm_only_ = false;
    std::unique_ptr<HostPool> lc_pool_;   // parked host threads for the [lin] combination of a lone proof (created on first use)
    bool many_slots_ = false;   // a throughput context (more than two proving slots): small MSMs may take the two-level sort under load
    // gangs (gang.h, slot_gate.h): up to gang_cap_ proofs per stream, their MSM / NTT batches in one launch sequence.  A slot's MSM
    // workspace, transform scratch and XYZZ result area are sized for ws_batch_ = MSM_MAX_BATCH x gang_cap_ operands.
    int gang_cap_ = 1;
    uint32_t ws_batch_ = MSM_MAX_BATCH;
    Gang::Launcher gang_launcher_;
    uint32_t msm_bases_ = 0;  // bases the MSM workspaces are sized for
    uint32_t NB_ = 0;
    Fr omega_, omega_inv_, omega4_, omega4_inv_, shift_, shift_inv_, n_inv_, n4_inv_;
    Fr zh_inv_[4];
    // circuit-level device data
    // NTT tables hold w * R' (R' = 2^261, the radix of the tile arithmetic: kernels_ntt.h); tw_n_ = omega^i in gnark's radix
    // is what the grand product and the permutation columns read
    DevBuf tw_n_, twu_n_, twi_n_, tw_4n_, twi_4n_, coset_pre_, coset_post_inv_, scales_;  // scales_: [1/n, 1/(4n)]
    DevBuf twu_n_x_, twi_n_x_, tw_4n_x_, twi_4n_x_;   // the four transform tables once more as FeU records (ntt_pass_kernel<.., TWU>); built with APK_NTT_TWU=1 (default 0: measured neutral)
    bool ntt_twu_ = false;
    DevBuf x4_, l0_4_;
    DevBuf s_lag_[3], ql_c_, qr_c_, qm_c_, qo_c_, qk_c_, s_c_[3], qcp_c_[APK_MAX_COMMITMENTS];
    DevBuf qk_lag_trace_;
    // Qk completion in the quotient kernel (kernels_poly.h QuotientArgs::nb_inject): trace Qk on the coset, the Lagrange
    // polynomials of the written rows on the coset, the trace's own values in those rows
    bool qk_direct_ = false;
    DevBuf eqk_trace_, inj_tab_[QK_INJECT_MAX];
    std::vector<uint32_t> inj_row_;
    std::vector<Fr> inj_trace_val_;
    DevBuf eql_, eqr_, eqm_, eqo_, es_[3], eqcp_[APK_MAX_COMMITMENTS];
    // tab_lag_ (round 5): PLAIN multiples of the n Lagrange-SRS points followed by the three blinding points
    // D_k = [tau^(n+k)]G1 - [tau^k]G1, so that  [f + b(X)(X^n - 1)] = sum f(omega^i) [L_i(tau)] + sum b_k D_k  is ONE MSM over the
    // witness values themselves (gnark commits L, R, O this way: setup/setup.go:124,138 builds the Lagrange SRS for it).  Plain:
    // a witness value 0 / 1 / small then has 0 / 1 / few non-zero digits, where R^-1 tables see a uniform a * R mod r.
    MsmTables tab_can_, tab_lag_;
    int wires_lag_mode_ = -1;                            // APK_WIRES_LAGRANGE at context creation: -1 auto, 0 never, 1 always
    // The extended Lagrange table is REQUIRED only by BSB22 commitments and by a forced Lagrange route (it is then built when the
    // context is created and a failure fails the creation).  In the automatic mode without a caller's Lagrange SRS it is an
    // optimisation some witnesses never use (uniform wires measure >= 90 % density; contexts with hooks commit canonically): the
    // context keeps the canonical SRS on the device (8 / 12 MB at 2^17; the table itself would be 134 / 280 MB and a device iFFT in
    // the exponent) and DERIVES the table the first time a proof's density - or the basis-1 primitive - asks for it, best effort:
    // a failure (out of memory ...) clears the error and leaves the canonical route in place.
    DevBuf srs_keep_;                                    // canonical SRS (n + 3 points) for the lazy derive; released after it
    Aff lag_dk_[3];                                      // D_k = [tau^(n+k)]G1 - [tau^k]G1 (host, at creation)
    std::mutex lag_mu_;
    std::atomic<bool> lag_ready_{false};                 // tab_lag_ is built (release / acquire around the build)
    std::atomic<bool> lag_failed_{false};                // a lazy derive failed: never tried again
    std::atomic<uint32_t> wire_density_pm_{0xffffffffu}; // non-zero digits of the last measured proof's wires, per mille of uniform
    std::atomic<uint32_t> proof_seq_{0};
    Aff vk_pts_[8 + APK_MAX_COMMITMENTS];
    HostFixedBase<FPP> vk_fixed_;   // host tables of [Ql][Qr][Qm][Qo][S3] for the [lin] combination (host_msm.h)
    std::vector<Slot*> slots_;
    // The context's streams: as many as it may run at a time (max 16), whichever of its slots lead.  A stream per SLOT - 32 slots
    // for gangs of two - put 23 hardware queues to work within 150 ms and starved some of them for up to 90 ms (with the spinning
    // waits of that build on top: profiles/r06_gang_sweep.txt).
    std::vector<hipStream_t> stream_pool_;
    SlotGate gate_;            // who proves on which slot; its busy count picks the load-dependent kernel forms (slot_gate.h)
    // Host inputs (apk_prove: the call the cgo shim makes): a caller takes one of `in_sets_` BEFORE it takes a
    // proving slot and sends L, R, O on the context's copy stream - so the upload of a proof that still waits for a slot (32 callers
    // on 16 slots) runs beside the rounds of the proofs in flight, and a proof in flight never has a PCIe transfer in its chain.
    // Pinned sources (apk_host_alloc / apk_host_register) are true asynchronous DMA; pageable ones are pinned on the fly by the
    // runtime inside hipMemcpyAsync on the calling thread, which is waiting for a slot anyway (both 46 GB/s for 3 x 4 MiB on the
    // boxes of this build: tools/ubench/h2d_probe.hip).  ONE copy stream for all sets: PCIe serialises the transfers anyway, and
    // every further stream is a further hardware queue - a stream per set (32 of them beside the 16 proving streams, on 24
    // hardware queues) cost 8 % proofs/s and 2.4 ms of a lone proof's latency (profiles/r06_host_inputs_ab.txt).
    // Created on the first host-input proof.
    struct InputSet {
        DevBuf w[3], pi2[APK_MAX_COMMITMENTS];
        hipStream_t copy = nullptr;     // the context's copy stream (not owned)
        hipEvent_t ready = nullptr;
        size_t index = 0;
    };
    std::vector<InputSet*> in_sets_;
    SlotGate in_gate_;
    hipStream_t copy_stream_ = nullptr;
    std::mutex mu_;
    // intra-proof multi-GPU (apk_ctx_set_commit_hook): the prover's commitments go through the host's hook instead of this GPU
    apk_commit_hook hook_ = nullptr;
    void* hook_user_ = nullptr;
    apk_wire_hook wire_hook_ = nullptr;
    void* wire_hook_user_ = nullptr;
    // Sub-coset split of round 3 (apk_ctx_set_subcoset): this context stands for the points
    // i = k (mod G) of the 4n coset - a coset of m = 4n / G points, u w^k <w^G> (w = omega_4n).
    //   forward   f(u w^k w_m^j) = NTT_m( fold_m( f_i (u w^k)^i ) )        pre[0] = (u w^k)^i;  pre[1]: class (k + 4) mod 8 for Z(omega X) at G = 8
    //   inverse   part_k[c'] = w^(-k c') * (unscaled inverse NTT_m of the quotient values)[c']        post = w^(-k c')
    //   merge     h_(c' + m t) = u^-c / (4n) * sum_k rho^(-k t) part_k[c'],  rho = w^m       (subcoset_merge_kernel, after the all-gather)
    struct SubCoset {
        int k = 0, G = 1, glog = 0;
        apk_gather_hook hook = nullptr;
        void* user = nullptr;
        DevBuf pre[2], post;
        Fr rho_inv[8];
        bool on() const { return G > 1 && hook; }
    } sc_;
    // which forms the load-dependent choices took (apk_paths_read): always counted, relaxed atomics
    enum PathIdx { P_PROOFS, P_MSM_BATCHES, P_SORT2, P_SORT2_LOAD, P_SORT_FUSED, P_LEAN_TAIL, P_ROWCOL_SERIAL, P_COMBINE_QUAD, P_SMALL_UNITS,
                   P_ONE_LAUNCH, P_LAGRANGE_WIRES, P_NTT_SEQ, P_NTT_R4, P_NTT_R4_LOAD, P_TAIL_FILL, P_LINCOMB_POOL, P_UNIT_LOADED, P_HOST_INPUTS, P_GANG_PROOFS, P_GANG_MSM, P_GANG_NTT, P_GANG_KERNELS, P_COUNT };
    std::atomic<uint64_t> paths_[P_COUNT] = {};
    void path(PathIdx i) { paths_[i].fetch_add(1, std::memory_order_relaxed); }
    int paths_read(apk_path_counts* out, int reset) override {
        static_assert(sizeof(apk_path_counts) >= P_COUNT * sizeof(uint64_t), "apk_path_counts holds every counter");
        memset(out, 0, sizeof *out);
        uint64_t* o = reinterpret_cast<uint64_t*>(out);    // the struct's fields are in PathIdx order
        for (int i = 0; i < P_COUNT; i++) o[i] = reset ? paths_[i].exchange(0, std::memory_order_relaxed) : paths_[i].load(std::memory_order_relaxed);
        return APK_OK;
    }
    // stats
    bool stats_on_ = false;
    uint32_t simds_ = 1024;  // SIMDs of the device (4 per CU); set at init
    std::mutex stats_mu_;
    apk_stats stats_{};

    ~CurveBackend() override {
        (void)hipSetDevice(device_);
        for (hipStream_t st : stream_pool_) if (st) (void)hipStreamSynchronize(st);
        if (copy_stream_) (void)hipStreamSynchronize(copy_stream_);
        for (InputSet* is : in_sets_) {
            if (is->ready) (void)hipEventDestroy(is->ready);
            delete is;
        }
        if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
        for (Slot* s : slots_) {
            for (auto& e : s->graphs) (void)hipGraphExecDestroy(e.second);
            if (s->ev0) (void)hipEventDestroy(s->ev0);
            if (s->ev1) (void)hipEventDestroy(s->ev1);
            if (s->ev2) (void)hipEventDestroy(s->ev2);
            if (s->ev3) (void)hipEventDestroy(s->ev3);
            if (s->h_pinned) (void)hipHostFree(s->h_pinned);
            if (s->side) { (void)hipStreamSynchronize(s->side); (void)hipStreamDestroy(s->side); }
            if (s->ev_acc) (void)hipEventDestroy(s->ev_acc);
            if (s->ev_side) (void)hipEventDestroy(s->ev_side);
            if (s->ev_sync) (void)hipEventDestroy(s->ev_sync);
            delete s;
        }
        for (hipStream_t st : stream_pool_) if (st) (void)hipStreamDestroy(st);
    }

    // ---------------------------------------------------------------------------------------------- NTT runner
    // natural in -> natural out; in != out.  which: 0 = size n, 1 = size 4n.  `count` same-size transforms per launch.
    // sub_log > 0: a transform of 1 / 2^sub_log of the size, on the same twiddle table (sub-coset transforms)
    int run_ntt_batch(hipStream_t st, int which, bool inverse, int count, const Fr* const* ins, Fr* const* outs, const uint32_t* in_lens,
                      uint32_t out_len, const Fr* pre, const Fr* post, const Fr* scale, int sub_log = 0) {
        Slot* m = gang_member();
        if (m && m->in_gang && st == m->stream && count <= NTT_MAX_BATCH) {     // a merge point of the gang (gang_launch)
            NttReq nr{};
            nr.member = m; nr.which = which; nr.inverse = inverse; nr.count = count;
            for (int i = 0; i < count; i++) { nr.ins[i] = ins[i]; nr.outs[i] = outs[i]; nr.in_lens[i] = in_lens[i]; }
            nr.out_len = out_len; nr.pre = pre; nr.post = post; nr.scale = scale; nr.sub_log = sub_log;
            GangReq r;
            r.kind = GANG_KIND_NTT; r.args = &nr; r.member = m;
            const int rc = m->lead->gang.meet(m->gang_idx, r, gang_launcher_);
            if (rc != APK_OK) set_error("%s", r.err.c_str());
            return rc;
        }
        return run_ntt_batch_now(st, which, inverse, count, ins, outs, in_lens, out_len, pre, post, scale, sub_log);
    }
    int run_ntt_batch_now(hipStream_t st, int which, bool inverse, int count, const Fr* const* ins, Fr* const* outs, const uint32_t* in_lens,
                          uint32_t out_len, const Fr* pre, const Fr* post, const Fr* scale, int sub_log = 0) {
        if (count < 1 || count > NTT_ARGS_MAX) { set_error("ntt: batch of %d", count); return APK_ERR_ARG; }
        const int log_n = (which ? (int)log_n_ + 2 : (int)log_n_) - sub_log;
        const Fr* tw = which ? (inverse ? ptr<Fr>(twi_4n_) : ptr<Fr>(tw_4n_)) : (inverse ? ptr<Fr>(twi_n_) : ptr<Fr>(twu_n_));
        if (ntt_twu_) tw = which ? (inverse ? ptr<Fr>(twi_4n_x_) : ptr<Fr>(tw_4n_x_)) : (inverse ? ptr<Fr>(twi_n_x_) : ptr<Fr>(twu_n_x_));
        // small transforms are latency-bound: 512-element tiles (18 KiB LDS) give >= 256 workgroups at 2^17.  Large ones were
        // assumed bandwidth-bound (2048-element tiles, fewest passes) until round 3 measured them: at 2^21 / 2^23 the 72 KiB tile
        // leaves two workgroups = 2 waves per SIMD on a CU, the pass runs at 6.9 cycles per VALU instruction and 1.9 TB/s - bound
        // by neither (profiles/r03_pmc_bls12381_2p21_*).  1024-element tiles (36 KiB, 4 workgroups per CU), 9 stages per pass:
        // 6.84 instead of 8.65 ms of NTT per BLS12-381 2^21 proof; 512 / 7 is the same, 2048 with 8..10 stages all 8.6 ms.
        static const int tile_env = env_int("APK_NTT_TILE_LOG", 0, 0, NTT_TILE_LOG);     // 0 = default; clamped to what the LDS tile holds
        static const int stages_env = env_int("APK_NTT_STAGES", 0, 0, NTT_TILE_LOG);
        int tile_log = log_n <= 19 ? 9 : 10;
        int max_s = log_n <= 19 ? 7 : 9;
        if (tile_env) tile_log = tile_env;
        if (stages_env) max_s = stages_env;
        if (tile_log > log_n) tile_log = log_n;
        if (max_s > tile_log) max_s = tile_log;
        const int passes = (log_n + max_s - 1) / max_s;
        bool busy_now = false;     // other proofs in flight on this context (the choice of run_msm_body's lean forms)
        if (slots_.size() > 2 && log_n >= 17 && log_n <= 19) busy_now = gate_.busy() > 1;
        NttBatch nb{};
        for (int i = 0; i < count; i++) { nb.in[i] = ins[i]; nb.out[i] = outs[i]; nb.in_len[i] = in_lens[i]; }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        Slot* owner = nullptr;
        for (Slot* s : slots_) if (s->own_stream == st || (s->side && s->side == st)) owner = s;    // (a gang's stream is its lead's own)
        if (passes > 1) {   // the passes hand each other unsaturated-limb elements through the slot's scratch
            if (!owner || log_n > 29) { set_error("ntt: no workspace for this stream"); return APK_ERR_STATE; }
            for (int i = 0; i < count; i++) nb.wide[i] = ptr<FeU<FRP>>(owner->ntt_wide) + ((size_t)i << log_n);
        }
        Slot* timed = stats_on_ ? owner : nullptr;
        if (timed) { e0 = timed->ev2; e1 = timed->ev3; HIPCHK(hipEventRecord(e0, st)); }
        int t0 = 0;
        for (int p = 0; p < passes; p++) {
            int s = (log_n - t0 + (passes - p) - 1) / (passes - p);
            NttPassArgs a;
            a.log_n = log_n; a.tile_log = tile_log; a.t0 = t0; a.t1 = t0 + s;
            a.first = (p == 0); a.last = (p == passes - 1);
            a.out_len = out_len;
            a.tw_shift = sub_log;
            const dim3 grid(1u << (log_n - tile_log), count);
            const size_t lds = ((size_t)1 << tile_log) * sizeof(FeU<FRP>);
            // radix 4 (two stages per LDS round trip, one four-element group per lane and step) pays above 2^19 only: kernels_ntt.h
            // ... and, with other proofs in flight, from 2^17 up: the instruction saving shows there (BN254 2^17, same box,
            // under the round-4 wave priorities: 508.6 -> 513.7 proofs/s) while a lone proof's latency-bound launches would lose
            // ~0.6 % to the halved lane count.
            static const int r4_env = env_int("APK_NTT_RADIX4", -1, -1, 1);
            a.radix4 = r4_env >= 0 ? r4_env : (log_n > 19 || (log_n >= 17 && busy_now) ? 1 : 0);
            if (p == 0) { path(P_NTT_SEQ); if (a.radix4) { path(P_NTT_R4); if (r4_env < 0 && log_n <= 19) path(P_NTT_R4_LOAD); } }
            static const int thr_env = env_int("APK_NTT_THREADS", 0, 0, NTT_THREADS) & ~63;
            uint32_t threads = NTT_THREADS;
            if (a.radix4) {
                threads = (1u << tile_log) / 4;
                if (threads < 64) threads = 64;
                if (threads > (uint32_t)NTT_THREADS) threads = NTT_THREADS;
            }
            if (thr_env) threads = (uint32_t)thr_env;
            if (ntt_twu_) ntt_pass_kernel<FRP, true><<<grid, threads, lds, st>>>(nb, tw, pre, post, scale, a);
            else ntt_pass_kernel<FRP, false><<<grid, threads, lds, st>>>(nb, tw, pre, post, scale, a);
            KCHK();
            t0 += s;
        }
        if (timed) {
            HIPCHK(hipEventRecord(e1, st));
            HIPCHK(hipEventSynchronize(e1));
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            std::lock_guard<std::mutex> g(stats_mu_);
            stats_.ntt_ms += ms;
            stats_.ntt_elements += ((uint64_t)1 << log_n) * count;
        }
        return APK_OK;
    }
    int run_ntt(hipStream_t st, int which, bool inverse, const Fr* in, Fr* out, uint32_t in_len, uint32_t out_len,
                const Fr* pre, const Fr* post, const Fr* scale, int sub_log = 0) {
        return run_ntt_batch(st, which, inverse, 1, &in, &out, &in_len, out_len, pre, post, scale, sub_log);
    }
    // evaluations of `count` canonical polynomials on the points of the 4n coset this context stands for: all 4n of them, or -
    // sub-coset split - the m = 4n / G points of class k (cls = 0) or of the class Z(omega X) needs at G = 8 (cls = 1)
    int coset_eval(hipStream_t st, int count, const Fr* const* ins, const uint32_t* lens, Fr* const* outs, int cls = 0) {
        if (!sc_.on()) return run_ntt_batch(st, 1, false, count, ins, outs, lens, n4_, ptr<Fr>(coset_pre_), nullptr, nullptr);
        return run_ntt_batch(st, 1, false, count, ins, outs, lens, n4_ >> sc_.glog, ptr<Fr>(sc_.pre[cls]), nullptr, nullptr, sc_.glog);
    }
    int inv_ntt_n(hipStream_t st, const Fr* in, Fr* out) { return run_ntt(st, 0, true, in, out, n_, n_, nullptr, nullptr, ptr<Fr>(scales_)); }
    // evaluations of a canonical polynomial (len coefficients) on the 4n coset
    int coset_ntt_4n(hipStream_t st, const Fr* in, uint32_t len, Fr* out) {
        return run_ntt(st, 1, false, in, out, len, n4_, ptr<Fr>(coset_pre_), nullptr, nullptr);
    }

    // ---------------------------------------------------------------------------------------------- MSM runner
    int build_tables(hipStream_t st, const Aff* d_bases, uint32_t count, MsmTables& T, bool plain = false) {
        CHK(T.table.alloc((size_t)count * W_ * sizeof(Aff)));
        T.n_bases = count;
        T.plain = plain;
        MsmPreScale pre{};
#ifndef APK_MSM_NO_RINV
        if (!plain) {   // R^-1 mod r as a plain integer = from_mont of the integer 1
            Fr one_int{};
            one_int.l[0] = 1u;
            const Fr rinv = Fr::from_mont(one_int);
            static_assert(Fr::N <= 16, "MsmPreScale holds 16 words");
            for (int i = 0; i < Fr::N; i++) pre.l[i] = rinv.l[i];
            pre.nwords = Fr::N;
        }
#endif
        msm_table_kernel<FPP><<<cdiv(count, 256), 256, 0, st>>>(d_bases, count, win_, pre, ptr<Aff>(T.table));
        KCHK();
        T.built = true;
        return APK_OK;
    }

    // tab_lag_ = plain multiples of the n Lagrange-SRS points (given on the host, or derived from d_srs on the device) + D_0..D_2
    int build_lagrange_table(hipStream_t st, const Aff* d_srs, const void* h_lagrange) {
        DevBuf lag;
        CHK(lag.alloc((size_t)(n_ + 3) * sizeof(Aff)));
        if (h_lagrange) HIPCHK(hipMemcpyAsync(lag.p, h_lagrange, (size_t)n_ * sizeof(Aff), hipMemcpyHostToDevice, st));
        else CHK((g1_to_lagrange_dev<FRP, FPP>(d_srs, n_, ptr<Aff>(lag), st)));
        HIPCHK(hipMemcpyAsync(ptr<Aff>(lag) + n_, lag_dk_, sizeof lag_dk_, hipMemcpyHostToDevice, st));
        CHK(build_tables(st, ptr<Aff>(lag), n_ + 3, tab_lag_, /*plain=*/true));
        HIPCHK(hipStreamSynchronize(st));     // `lag` is released on return
        lag_ready_.store(true, std::memory_order_release);
        return APK_OK;
    }
    bool lagrange_ready() const { return lag_ready_.load(std::memory_order_acquire); }
    // can this context commit over the Lagrange SRS at all (now, or after a lazy derive)?
    bool lagrange_possible() const { return lagrange_ready() || (srs_keep_.p && !lag_failed_.load(std::memory_order_relaxed)); }
    // Best effort: true when the table is there on return.  A failed derive clears the HIP error state and is not retried.
    bool ensure_lagrange_table(hipStream_t st) {
        if (lagrange_ready()) return true;
        std::lock_guard<std::mutex> lk(lag_mu_);
        if (lagrange_ready()) return true;
        if (!srs_keep_.p || lag_failed_.load(std::memory_order_relaxed)) return false;
        const int rc = build_lagrange_table(st, ptr<Aff>(srs_keep_), nullptr);
        if (rc != APK_OK) {
            (void)hipGetLastError();
            tab_lag_.table.release();
            tab_lag_.built = false;
            lag_failed_.store(true, std::memory_order_relaxed);
            (void)hipStreamSynchronize(st);
            srs_keep_.release();
            return false;
        }
        srs_keep_.release();
        return true;
    }

    int run_msm(Slot& s, const MsmTables& T, const MsmBatchArgs& a_in, Aff* h_out) {
        // whether the scalars leave the Montgomery form in the sort is a property of the TABLE (R^-1 * P or P): derived here, so
        // that no call site can pair a table with the wrong scalar form
        MsmBatchArgs a = a_in;
        a.plain = T.plain ? 1u : 0u;
        static const int graphs = env_int("APK_MSM_GRAPH", 0, 0, 1);
        if (!graphs || stats_on_) return run_msm_body(s, T, a, h_out);
        GraphKey key{};
        key.table = T.table.p; key.a = a;
        for (uint32_t b = a.batch; b < MSM_ARGS_MAX; b++) { key.a.scalars[b] = nullptr; key.a.len[b] = 0; key.a.offset[b] = 0; }
        for (size_t i = 0; i < s.graphs.size(); i++)
            if (s.graphs[i].first == key) {
                if (i) std::swap(s.graphs[i], s.graphs[0]);          // most recently used first
                HIPCHK(hipGraphLaunch(s.graphs[0].second, s.stream));
                s.pending_pts = a.batch;
                return APK_OK;
            }
        hipGraph_t g = nullptr;
        HIPCHK(hipStreamBeginCapture(s.stream, hipStreamCaptureModeThreadLocal));
        const int rc = run_msm_body(s, T, a, h_out);
        const hipError_t e = hipStreamEndCapture(s.stream, &g);
        if (rc != APK_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
        HIPCHK(e);
        hipGraphExec_t exec = nullptr;
        HIPCHK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
        // bounded: the prover's own four batches hit the same keys every proof; callers with per-call staging pointers
        // (apk_msm_g1_batch_device) would otherwise grow the list without limit.  The evicted exec may still be running on
        // this slot's stream, so the stream is drained before it is destroyed.
        constexpr size_t GRAPH_CACHE = 8;
        if (s.graphs.size() >= GRAPH_CACHE) {
            HIPCHK(hipStreamSynchronize(s.stream));
            (void)hipGraphExecDestroy(s.graphs.back().second);
            s.graphs.pop_back();
        }
        s.graphs.insert(s.graphs.begin(), std::make_pair(key, exec));
        HIPCHK(hipGraphLaunch(exec, s.stream));
        s.pending_pts = a.batch;
        return APK_OK;
    }

    // Measurement aid (tools/knockout.sh, -DAPK_DEBUG_KNOCKOUT builds only): APK_DEBUG_SKIP is a bit mask of MSM phases NOT to
    // launch - 1 count pass + column scan, 32 the three scan launches, 64 scatter pass (the previous batch's sort stays in
    // place), 2 accumulate, 4 merge, 8 row/column sums, 16 bit sums + final.  The results are garbage; what it shows is what each phase costs at saturation.
#ifdef APK_DEBUG_KNOCKOUT
#define APK_PHASE(bit) ((env_int("APK_DEBUG_SKIP", 0, 0, 127) & (bit)) == 0)
#else
#define APK_PHASE(bit) true
#endif
    // results land in slot.result (device) and are copied to h_out (host, batch points) - caller syncs the stream
    int run_msm_body(Slot& s, const MsmTables& T, const MsmBatchArgs& a, Aff* h_out) {
        hipStream_t st = s.stream;
        uint32_t maxlen = 0;
        uint64_t entries = 0;
        if (a.batch == 0 || a.batch > ws_batch_ || a.batch > (uint32_t)MSM_ARGS_MAX) { set_error("msm: a batch of %u exceeds the workspace's %u", a.batch, ws_batch_); return APK_ERR_ARG; }
        for (uint32_t b = 0; b < a.batch; b++) {
            if (a.len[b] > T.n_bases || a.offset[b] > T.n_bases - a.len[b]) { set_error("msm: %u scalars exceed the %u bases", a.len[b], T.n_bases); return APK_ERR_ARG; }
            if (a.len[b] > maxlen) maxlen = a.len[b];
            entries += (uint64_t)a.len[b] * W_;
        }
        const uint32_t total_buckets = a.batch * NB_;
        // Work-unit length: every resident wave of the accumulate kernel loops `unit` times, and the launch lasts as long as the
        // SIMD that was handed the most waves - so pick the length in [16, 18] whose full-unit waves fill the SIMDs in the fewest
        // whole rounds (2^17, c = 15, one MSM: 16 -> 2056 waves = 3 rounds on some of the 1024 SIMDs, 17 -> 1930 waves = 2 rounds).
        // Longer units lose more to fewer resident waves than the round count says (measured: 20..24 are slower than 16).
        static const uint32_t unit_env = (uint32_t)env_int("APK_MSM_UNIT", 0, 0, MSM_UNIT_MAX);
        uint32_t unit = unit_env;
        if (!unit) {
            // Large batches have lanes to spare: with 16-entry units a bucket of a 2^21 MSM (1 024 entries at c = 16) is merged from
            // 64 partial sums, and the merges (msm_combine_kernel: a few lanes per bucket, shuffle trees) use the SIMDs far worse than
            // the accumulate loop does (the product count is the same either way).  Keep >= 8 waves per SIMD of full units and let
            // the unit grow to MSM_UNIT_MAX beyond that
            // (round 3, BLS12-381 2^21 on one box: 16 -> 15.6, 24 -> 16.1, 32 -> 15.9, 48 -> 16.3, 64 -> 16.3 proofs/s).
            const uint64_t lanes_wanted = (uint64_t)simds_ * 64 * 8;
            const uint64_t by_size = entries / lanes_wanted;
            if (by_size >= 24) {
                unit = by_size > (uint64_t)MSM_UNIT_MAX ? (uint32_t)MSM_UNIT_MAX : (uint32_t)by_size;
            } else {
                uint64_t best = ~0ull;
                for (uint32_t u = MSM_UNIT; u <= MSM_UNIT + 2; u++) {
                    const uint64_t full_units = entries / u > total_buckets / 2 ? entries / u - total_buckets / 2 : 1;
                    const uint64_t cost = (uint64_t)cdiv(cdiv(full_units, 64), simds_) * u;
                    if (cost < best) { best = cost; unit = u; }
                }
            }
        }
        if (unit < (uint32_t)MSM_UNIT_MIN) unit = MSM_UNIT_MIN;
        if (unit > (uint32_t)MSM_UNIT_MAX) unit = MSM_UNIT_MAX;
        // Are other proofs keeping the GPU busy?  Then nobody waits for this batch's reduction chain and the instruction-lean
        // forms of the tail kernels win (fewer, longer chains: every lane of a wave does useful additions); a lone proof keeps
        // the short-chain forms.
        bool others_busy = false;
        if (slots_.size() > 2) others_busy = gate_.busy() > 1;
        static const int graphs_on = env_int("APK_MSM_GRAPH", 0, 0, 1);   // a captured batch must not depend on the moment of capture:
        if (graphs_on) others_busy = false;                               // neither its unit nor its kernel forms
        // Under load the units grow: a bucket of 64 entries is then merged from 2 partial sums instead of 4 (the merge's general
        // additions cost 14 products against the accumulate loop's 10), and the other proofs' kernels fill the SIMDs the fewer,
        // longer waves leave.  Same box, BN254, two rounds (tools/sweep_unit_window.sh): 2^16 941.6 / 947.9 -> 952.3 / 962.4
        // proofs/s at 40 entries, 2^17 with 17-bit windows 514.4 / 515.5 -> 530.7 / 530.8 at 40 and 533.2 / 532.1 at 64, 2^18 and
        // 2^19 +0.5 %; a lone proof pays for long units (3.25 -> 3.43 ms at 32), so only with others in flight.
        // (2^15 bases and BLS12-381 2^14 lose 1-2 % with them - too few waves left even for a busy GPU - so from 2^16 bases.)
        static const uint32_t unit_loaded = (uint32_t)env_int("APK_MSM_UNIT_LOADED", 48, 0, MSM_UNIT_MAX);
        static const uint32_t unit_loaded_bases = (uint32_t)env_int("APK_MSM_UNIT_LOADED_BASES", 65536, 0, 1 << 30);
        // Where it pays is where one unit holds a whole bucket (the merge then has nothing to add): 2^16 bases with 16-bit
        // windows 967 / 963 -> 985 / 977, with 15-bit windows (68 entries per bucket) 976 / 972 -> 951 / 950 - so only up to 64
        // entries per bucket on average.
        if (!unit_env && others_busy && unit < unit_loaded && msm_bases_ >= unit_loaded_bases && entries <= 64ull * total_buckets) { unit = unit_loaded; path(P_UNIT_LOADED); }
        // Small batches (a lone 2^14 MSM: 360 k entries) do not even give every SIMD one wave at 16 entries per lane, and a lone
        // wave issues a dependent instruction every ~6.5 cycles: the accumulate launch is then 16 additions long whatever the
        // size (BLS12-381 2^14: 229 of the MSM's 580 us).  Below one wave per SIMD the unit shrinks - down to
        // APK_MSM_UNIT_SMALL entries - so that the lanes fill the SIMDs once; the merge takes more lanes per bucket instead.
        static const uint32_t unit_small = (uint32_t)env_int("APK_MSM_UNIT_SMALL", MSM_UNIT_SMALL, MSM_UNIT_SMALL, MSM_UNIT_MIN);
        // (two waves per SIMD: BLS12-381 2^14 lone proof 3.01 -> 2.90 ms, BN254 2^14 1.84 -> 1.78; four: back to 3.01 - the merge
        // grows as the units shrink.  Only for a batch that has the GPU to itself.)
        static const uint32_t small_waves = (uint32_t)env_int("APK_MSM_SMALL_WAVES", 2, 1, 4);   // waves per SIMD the shrunken units aim at
        if (!unit_env && !others_busy && entries <= MSM_SMALL_ENTRIES * small_waves && entries / unit < (uint64_t)simds_ * 64 * small_waves) {
            uint32_t u = (uint32_t)(entries / ((uint64_t)simds_ * 64 * small_waves));
            if (u < unit_small) u = unit_small;
            if (u < unit) { unit = u; path(P_SMALL_UNITS); }
        }
        const uint32_t max_units = (uint32_t)(entries / unit) + total_buckets;
        if (stats_on_) HIPCHK(hipEventRecord(s.ev0, st));
        if (s.mark_acc == 2) HIPCHK(hipEventRecord(s.ev_acc, st));
        // counting sort by bucket: LDS-private histograms per scalar slice, column scan, bucket scan, scatter
        static const uint32_t slice = (uint32_t)env_int("APK_MSM_SLICE", 2048, 64, 1 << 20);  // scalars per sort workgroup
        // packed 16-bit counters: a slice must stay below 2^16 entries per bucket even when every digit of every scalar agrees
        const uint32_t slice_eff = NB_ >= MSM_PACKED_NB && slice > 3072u ? 3072u : slice;
        uint32_t G = cdiv(maxlen, slice_eff ? slice_eff : 2048u);
        if (G < 1) G = 1;
        if (G > msm_G_max_) G = msm_G_max_;
        dim3 gd(G, a.batch);   // (the two-level sort re-cuts its slices below)
        const size_t lds = digits_lds_bytes();
        static const int dth = env_int("APK_MSM_DIGITS_THREADS", MSM_DIGITS_THREADS, 64, MSM_DIGITS_THREADS) & ~63;   // whole waves, <= the launch bound
        static const int lean_env = env_int("APK_MSM_LEAN_TAIL", -1, -1, 1);
        const bool lean = lean_env >= 0 ? lean_env != 0 : others_busy;
        // two-level sort (kernels_msm.h): partitions of 256 buckets, then a counting sort per partition - the stores of both
        // levels are neighbours of each other instead of 2 M isolated 4-byte writes per MSM
        // Measured (round 3, same box, every scatter of both levels inside an LDS tile, a wave per run in the copy-out): BN254 2^17
        // 468 -> 497 proofs/s (+6 %) and a lone proof 3.57 -> 3.53 ms; 2^16 847 -> 872; 2^15 flat (+1.3 % latency); BN254 2^14
        // 1 746 -> 1 678 and BLS12-381 2^14 1 200 -> 1 140: taken from 2^16 bases up (APK_MSM_SORT2: -1 that rule, 0 never, 1
        // whenever it applies).  A first version with the second level's stores still scattered (inside 64 KiB windows) gained
        // nothing: DESIGN section 5.
        static const int sort2_env = env_int("APK_MSM_SORT2", -1, -1, 1);
        // (a lone single MSM was 0.49 against 0.47 ms with the first version - and 0.468 against 0.473 once the partition scan
        // ran eight lanes per pair and the second-level tile let two partitions share a CU: no exception for it any more)
        // Round 4, with the two-launch form and the wave priorities: under LOAD it also pays from 2^13 bases (same box, proofs/s:
        // BN254 2^13 +1 %, 2^14 +3.5 %, 2^15 +4.6 %, BLS12-381 2^14 +3 %) while a LONE proof there is 2.5 - 4 % slower with it - so
        // below 2^16 bases it follows the load.
        const bool sort2_want = !one_level_ok() || (sort2_env >= 0 ? sort2_env != 0 : (T.n_bases >= 65536u || (others_busy && T.n_bases >= 8192u)));
        // Round 4: the packed entry's layout follows the context (MsmPartCfg, part_cfg_): the table index takes the bits it needs
        // and the partitions shrink until one fits the second level's LDS tile - BLS12-381 2^21 x 16 windows (26 index bits,
        // 2 048 partitions of 16 buckets) sorts in two levels as well.  The first level's slices are cut so that a slice's entries
        // fit its LDS stage.
        const MsmPartCfg& pc = part_cfg_;
        const uint32_t P = pc.P;
        uint32_t G2 = G;
        {
            const uint32_t stage_max = part_stage_max();
            uint32_t sl2 = slice_eff ? slice_eff : 2048u;
            if ((uint64_t)sl2 * W_ > stage_max) sl2 = stage_max / (uint32_t)W_;
            G2 = cdiv(maxlen, sl2);
            if (G2 > 1024u && (uint64_t)cdiv(maxlen, 1024u) * W_ <= stage_max) G2 = 1024u;   // 2^21 + 3 scalars: 1 024 slices of 2 049
            else if (G2 > MSM_PART_GMAX && (uint64_t)cdiv(maxlen, MSM_PART_GMAX) * W_ <= stage_max) G2 = MSM_PART_GMAX;
            if (G2 < 1) G2 = 1;
        }
        static const int small_scan_env = env_int("APK_MSM_PART_SMALL_SCAN", 1, 0, 1);   // 0: always the three-launch scan (tests)
        const bool small_scan = small_scan_env && a.batch * P <= 2048u && G2 <= 128u;
        const uint64_t counts_words = s.counts.bytes / 4;
        const bool sort2 = sort2_want && P >= 4 && s.sort_tmp.p && G2 <= MSM_PART_GMAX && (uint64_t)a.batch * P <= (uint64_t)MSM_MAX_BATCH * MSM_PART_MAX &&
                           (uint64_t)a.batch * G2 * P * 2 + (uint64_t)a.batch * P * (1 + MSM_PART_CHUNKS) <= counts_words;
        if (!sort2 && !one_level_ok()) { set_error("msm: a %d-bit window sorts in two levels only, and this batch does not fit them (%u slices, %u partitions)", c_, G2, P); return APK_ERR_STATE; }
        if (sort2) { G = G2; gd = dim3(G, a.batch); }
        path(P_MSM_BATCHES);
        if (sort2) { path(P_SORT2); if (sort2_env < 0 && T.n_bases < 65536u) path(P_SORT2_LOAD); }
        if (lean) path(P_LEAN_TAIL);
        uint32_t* pcounts = ptr<uint32_t>(s.counts);
        uint32_t* runstart = pcounts + (size_t)a.batch * G * P;
        uint32_t* ptot = runstart + (size_t)a.batch * G * P;
        uint32_t* csum = ptot + (size_t)a.batch * P;
        if (sort2 && !APK_PHASE(1)) {
            // knock-out build, bit 1: the two-level sort is not launched (the previous batch's sorted entries stay in place)
        } else if (sort2) {
            // LDS stage of the first level: a slice's entries (<= slice x W words; slices that do not fit scatter in HBM)
            const uint32_t per_slice = cdiv(maxlen, G);
            uint32_t stage_cap = per_slice * (uint32_t)W_;
            if (stage_cap > part_stage_max()) stage_cap = part_stage_max();
            const size_t cursors_lds = (size_t)(2 * P + 1) * 4;     // behind the stage in the first level's dynamic LDS
            // tile of the second level: the mean partition + 15 % (uniform scalars stay within 2 %); at 2^17 that is 74 KiB, so two
            // workgroups share a CU and the 384 partitions of a three-MSM batch run in one round instead of two.  Partitions
            // above it (skewed scalars) scatter in HBM.
            uint32_t tile_cap = (uint32_t)(entries / ((uint64_t)a.batch * P)) + (uint32_t)(entries / ((uint64_t)a.batch * P)) / 7u + 256u;
            if (tile_cap > MSM_PART_TILE) tile_cap = MSM_PART_TILE;
            // Two launches instead of four (kernels_msm.h "the two levels in TWO launches"): slice-major runs need no scan between
            // the levels.  A slice's entries always fit its stage here (per_slice x W <= MSM_PART_STAGE by the choice of G).
            static const int fused_env = env_int("APK_MSM_SORT_FUSED", 1, 0, 1);
            // Only while a (slice, partition) run is at least a wave long: the second level walks a partition run by run, and at
            // BLS12-381 2^21 (1 024 slices x 1 024 partitions, 32 entries per run, two strided table loads per run) it took
            // 64 ms per 99 launches against the four-launch form's 21 (profiles/r04_kernel_trace_bls12381_2p21.txt, first cut).
            const bool fused = fused_env && !graphs_on /* a replayed capture would reuse one totals buffer */ && stage_cap / P >= 64u &&
                               (uint64_t)per_slice * W_ <= part_stage_max() && s.ptot2.p && (uint64_t)ws_batch_ * P <= (uint64_t)MSM_MAX_BATCH * MSM_PART_MAX &&
                               (uint64_t)a.batch * G * (P + 1) <= counts_words &&
                               (uint64_t)a.batch * G * stage_cap * 4 <= s.sort_tmp.bytes;
            if (fused) {
                path(P_SORT_FUSED);
                uint32_t* pt_cur = ptr<uint32_t>(s.ptot2) + (size_t)(s.ptot_parity & 1u) * MSM_MAX_BATCH * MSM_PART_MAX;
                uint32_t* pt_next = ptr<uint32_t>(s.ptot2) + (size_t)((s.ptot_parity & 1u) ^ 1u) * MSM_MAX_BATCH * MSM_PART_MAX;
                s.ptot_parity ^= 1u;
                if (a.plain) msm_part1_kernel<FRP, true><<<gd, dth, (size_t)stage_cap * 4 + cursors_lds, st>>>(a, win_, pc, T.n_bases, G, ptr<uint32_t>(s.sort_tmp), stage_cap, pcounts, pt_cur);
                else msm_part1_kernel<FRP, false><<<gd, dth, (size_t)stage_cap * 4 + cursors_lds, st>>>(a, win_, pc, T.n_bases, G, ptr<uint32_t>(s.sort_tmp), stage_cap, pcounts, pt_cur);
                KCHK();
                msm_part_sort_runs_kernel<0><<<dim3(P, a.batch), 1024, (size_t)tile_cap * 4, st>>>(
                    ptr<uint32_t>(s.sort_tmp), stage_cap, pcounts, pt_cur, pt_next, pc, G, NB_, ptr<uint32_t>(s.hist), ptr<uint32_t>(s.sorted), tile_cap,
                    ws_batch_ * P);
                KCHK();
            } else {
            if (a.plain) msm_part_kernel<FRP, false, true><<<gd, dth, cursors_lds, st>>>(a, win_, pc, NB_, T.n_bases, G, pcounts, nullptr, nullptr, 0);
            else msm_part_kernel<FRP, false, false><<<gd, dth, cursors_lds, st>>>(a, win_, pc, NB_, T.n_bases, G, pcounts, nullptr, nullptr, 0);
            KCHK();
            if (small_scan) {
                msm_part_scan_kernel<0><<<1, 1024, 0, st>>>(pcounts, runstart, ptot, a.batch, G, P);
                KCHK();
            } else {
                const dim3 sg(cdiv(a.batch * P, 64), MSM_PART_CHUNKS / 4);
                msm_part_tot_kernel<0><<<sg, 256, 0, st>>>(pcounts, csum, a.batch, G, P);
                KCHK();
                msm_part_base_kernel<0><<<1, 1024, 0, st>>>(csum, ptot, a.batch * P);
                KCHK();
                msm_part_runs_kernel<0><<<sg, 256, 0, st>>>(pcounts, csum, runstart, a.batch, G, P);
                KCHK();
            }
            MsmPartCfg pc1 = pc;
            {   // lanes per run of the copy-out: the power of two at or above the mean run, 8..64
                const uint32_t mean_run = stage_cap / P + 1;
                pc1.run_lanes = 8;
                while (pc1.run_lanes < 64 && pc1.run_lanes < mean_run) pc1.run_lanes <<= 1;
            }
            if (a.plain) msm_part_kernel<FRP, true, true><<<gd, dth, (size_t)stage_cap * 4 + cursors_lds, st>>>(a, win_, pc1, NB_, T.n_bases, G, pcounts, runstart, ptr<uint32_t>(s.sort_tmp), stage_cap);
            else msm_part_kernel<FRP, true, false><<<gd, dth, (size_t)stage_cap * 4 + cursors_lds, st>>>(a, win_, pc1, NB_, T.n_bases, G, pcounts, runstart, ptr<uint32_t>(s.sort_tmp), stage_cap);
            KCHK();
            msm_part_sort_kernel<0><<<dim3(P, a.batch), 1024, (size_t)tile_cap * 4, st>>>(ptr<uint32_t>(s.sort_tmp), runstart, ptot, pc, G, NB_,
                                                                                          ptr<uint32_t>(s.hist), ptr<uint32_t>(s.sorted), tile_cap);
            KCHK();
            }
        } else if (APK_PHASE(1)) {
        if (a.plain) msm_digits_kernel<FRP, false, true><<<gd, dth, lds, st>>>(a, win_, NB_, T.n_bases, G, ptr<uint32_t>(s.counts), nullptr, nullptr);
        else msm_digits_kernel<FRP, false, false><<<gd, dth, lds, st>>>(a, win_, NB_, T.n_bases, G, ptr<uint32_t>(s.counts), nullptr, nullptr);
        KCHK();
        msm_colscan_kernel<0><<<cdiv(total_buckets, 256), 256, 0, st>>>(ptr<uint32_t>(s.counts), NB_, G, total_buckets, ptr<uint32_t>(s.hist));
        KCHK();
        }
        if (APK_PHASE(32)) {
            // `items` consecutive buckets per scan thread keep the block count at MSM_SCAN_MAX_BLOCKS (the totals step's LDS) up to 2^21 buckets
            uint32_t items = 1;
            while (items < (uint32_t)MSM_SCAN_ITEMS_MAX && (uint64_t)MSM_SCAN_BLOCK * MSM_SCAN_MAX_BLOCKS * items < total_buckets) items <<= 1;   // 1, 2, 4, 8
            const uint32_t nblk = cdiv(total_buckets, (uint64_t)MSM_SCAN_BLOCK * items);
            uint32_t* blk_tot = ptr<uint32_t>(s.scan_blk);
            uint32_t* blk_bins = blk_tot + 3 * nblk;
            // APK_MSM_SCAN_FUSED=1: two launches - the last workgroup of the local scan runs the totals step.  Built and measured
            // (round 4, same box, interleaved): 506 -> 472 proofs/s at BN254 2^17 and a lone proof 3.36 -> 3.41 ms - every scan
            // workgroup then carries the totals step's 40 KiB of LDS and an agent-scope fence, and the step itself runs behind the
            // slowest of them instead of on an idle CU.  Off; three launches stay.
            static const int scan_fused = env_int("APK_MSM_SCAN_FUSED", 0, 0, 1);
            uint32_t* scan_done = ptr<uint32_t>(s.done_count) + MSM_ARGS_MAX;
#define APK_SCAN_LOCAL(F, I) msm_scan_local_kernel<F, I><<<nblk, MSM_SCAN_BLOCK, 0, st>>>(ptr<uint32_t>(s.hist), total_buckets, unit, ptr<uint32_t>(s.offsets), \
                ptr<uint32_t>(s.unit_off), ptr<uint32_t>(s.full_off), ptr<uint32_t>(s.rem_rank), ptr<uint32_t>(s.merge_rank), blk_tot, blk_bins, nblk, scan_done)
            if (scan_fused && items == 1) {
                APK_SCAN_LOCAL(1, 1);
                KCHK();
            } else {
                if (items == 1) APK_SCAN_LOCAL(0, 1); else if (items == 2) APK_SCAN_LOCAL(0, 2); else if (items == 4) APK_SCAN_LOCAL(0, 4); else APK_SCAN_LOCAL(0, 8);
                KCHK();
                msm_scan_totals_kernel<0><<<1, MSM_SCAN_BLOCK, 0, st>>>(blk_tot, blk_bins, nblk, total_buckets, ptr<uint32_t>(s.offsets),
                                                                         ptr<uint32_t>(s.unit_off), ptr<uint32_t>(s.full_off));
                KCHK();
            }
#undef APK_SCAN_LOCAL
            msm_scan_apply_kernel<0><<<cdiv(total_buckets, MSM_SCAN_BLOCK), MSM_SCAN_BLOCK, 0, st>>>(blk_tot, blk_bins, ptr<uint32_t>(s.hist), ptr<uint32_t>(s.rem_rank),
                                                                      ptr<uint32_t>(s.merge_rank), nblk,
                                                                      total_buckets, unit, ptr<uint32_t>(s.offsets), ptr<uint32_t>(s.unit_off),
                                                                      ptr<uint32_t>(s.full_off), ptr<uint32_t>(s.rem_list), ptr<uint32_t>(s.merge_list), items);
            KCHK();
        }
        if (!sort2 && APK_PHASE(64)) {
        if (a.plain) msm_digits_kernel<FRP, true, true><<<gd, dth, lds, st>>>(a, win_, NB_, T.n_bases, G, ptr<uint32_t>(s.counts), ptr<uint32_t>(s.offsets), ptr<uint32_t>(s.sorted));
        else msm_digits_kernel<FRP, true, false><<<gd, dth, lds, st>>>(a, win_, NB_, T.n_bases, G, ptr<uint32_t>(s.counts), ptr<uint32_t>(s.offsets), ptr<uint32_t>(s.sorted));
        KCHK();
        }
        if (stats_on_) HIPCHK(hipEventRecord(s.ev2, st));
        if (APK_PHASE(2))
        msm_accumulate_kernel<FPP><<<cdiv(max_units, MsmAcc<FPP>::THREADS), MsmAcc<FPP>::THREADS, 0, st>>>(ptr<Aff>(T.table), ptr<uint32_t>(s.sorted), ptr<uint32_t>(s.offsets),
                                                                        ptr<uint32_t>(s.unit_off), ptr<uint32_t>(s.full_off), ptr<uint32_t>(s.rem_list),
                                                                        total_buckets, max_units, unit, ptr<PtU>(s.partial));
        KCHK();
        if (stats_on_) HIPCHK(hipEventRecord(s.ev3, st));
        if (s.mark_acc == 1) HIPCHK(hipEventRecord(s.ev_acc, st));
        // lanes per bucket: ~5 unit partials per lane, so the sequential part and the shuffle tree are balanced; one lane walks up
        // to 32 partials when the GPU has other work (a shuffle level costs every lane of the group an addition, useful or not)
        int lanes_log = 0;
        const uint32_t per_lane = lean && total_buckets >= 32768u ? 32u : 5u;
        {
            const uint64_t upb = (entries / unit) / total_buckets + 1;  // unit partials per bucket (estimate)
            // (only with enough buckets to fill the SIMDs at one lane each: at BLS12-381 2^14 - 6 144 buckets, 11 partials
            // each - one lane per bucket was measured SLOWER under load, 1 137 -> 1 114 proofs/s)
            while ((1u << lanes_log) < MSM_COMBINE_LANES && (upb >> lanes_log) > per_lane) lanes_log++;
        }
        // A batch that has the GPU to itself lets the DEVICE pick the lanes per bucket, from the bucket scan's count of partials per
        // NON-EMPTY bucket: the host's estimate averages over all buckets, and a skewed input (256 distinct scalars: 4 096 buckets of
        // 32 partials, the rest empty) then left one lane walking 32 partials (0.56 of that MSM's 1.07 ms).  The grid covers up to
        // four times the host's lanes; for uniform scalars the two rules agree and the surplus workgroups return at once.
        static const int dyn_env = env_int("APK_MSM_COMBINE_DYN", 1, 0, 1);
        const bool dyn_lanes = dyn_env && !lean && APK_PHASE(32);
        if (dyn_lanes) { lanes_log += 2; if ((1u << lanes_log) > MSM_COMBINE_LANES) lanes_log = 4; }
        uint32_t scan_items = 1;
        while (scan_items < (uint32_t)MSM_SCAN_ITEMS_MAX && (uint64_t)MSM_SCAN_BLOCK * MSM_SCAN_MAX_BLOCKS * scan_items < total_buckets) scan_items <<= 1;
        const uint32_t scan_nblk = cdiv(total_buckets, (uint64_t)MSM_SCAN_BLOCK * scan_items);
        const uint32_t* avg_partials = dyn_lanes ? ptr<uint32_t>(s.scan_blk) + (size_t)(3 + MSM_BINS) * scan_nblk : nullptr;
        {   // light and heavy merge in one launch (the heavy blocks return at once when no bucket is skewed)
            const uint32_t normal_blocks = cdiv((uint64_t)total_buckets << lanes_log, 256);
            // the buckets in the order of their partial counts (merge_list): a wave's lanes run the same number of additions
            static const int sorted_merge = env_int("APK_MSM_SORTED_MERGE", 1, 0, 1);
            // a SMALL batch that has the GPU to itself: four lanes per addition (the merge is a chain of dependent additions on a
            // few hundred lone waves then).  Measured, lone proofs, same box: BLS12-381 2^14 (2 048 buckets) 3.30 -> 3.05 ms and a
            // lone MSM 0.59 -> 0.47 ms; BN254 2^17 (16 384 buckets) 3.39 -> 3.48 ms although its lone MSM gains 4 % - the quads'
            // 1.6 x instructions then compete with the coset transforms that fill the tail - so: up to 4 096 buckets per MSM.
            static const int cq_env = env_int("APK_MSM_COMBINE_QUAD", -1, -1, 1);
            const bool cquad = cq_env >= 0 ? cq_env != 0 : (!lean && !graphs_on && NB_ <= 4096u);
            if (!APK_PHASE(4)) {
            } else if (cquad) {
                path(P_COMBINE_QUAD);
                const uint32_t qblocks = cdiv(((uint64_t)total_buckets << lanes_log) * 4, 256);
                msm_combine_quad_kernel<FPP><<<qblocks + MSM_HEAVY_BLOCKS, 256, 0, st>>>(
                    ptr<PtU>(s.partial), ptr<uint32_t>(s.unit_off), sorted_merge ? ptr<uint32_t>(s.merge_list) : nullptr, total_buckets, lanes_log,
                    qblocks, ptr<PtU>(s.bucket_sum), avg_partials, per_lane);
            } else
            msm_combine_kernel<FPP><<<normal_blocks + MSM_HEAVY_BLOCKS, 256, 0, st>>>(
                ptr<PtU>(s.partial), ptr<uint32_t>(s.unit_off), sorted_merge ? ptr<uint32_t>(s.merge_list) : nullptr, total_buckets, lanes_log,
                normal_blocks, ptr<PtU>(s.bucket_sum), avg_partials, per_lane);
            KCHK();
        }
        // sum_k k*B_k: row/column sums of the bucket array, bit-wise weighted sums of those, final scaling + affine
        const int m_bits = c_ - 1, cols_log = m_bits - m_bits / 2;
        const uint32_t rows = 1u << (m_bits / 2), cols = 1u << cols_log;
        const uint32_t nbits = (uint32_t)cols_log + 1;  // weights <= cols
        // Four lanes per point operation in the three reduction kernels (ec.h add_quad_general / dbl_quad_general): they are
        // chains of dependent point operations on lone waves, and a quad finishes an addition in 4 product stages instead of 14
        // products (2^17: bit sums 55 -> 30 us, final 97 -> 50 us, row/column sums 73 -> 45 us per batch).  The row/column kernel
        // has real work (2 additions per bucket) and pays for the quads with 1.7x its VALU instructions: -3 % proofs/s at
        // saturation, so contexts with more than two slots keep its one-lane form.  APK_MSM_QUAD_TAIL overrides the choice with
        // a bit mask (1 row/column sums, 2 bit sums, 4 final, 8 row/column sums with quads in the last five tree levels only).
        // Logical threads per workgroup: the longer of rows / cols, at most 128 (512 lanes leave each lane 256 registers).
        static const int quad_env = env_int("APK_MSM_QUAD_TAIL", -1, -1, 15);
        const int quad = quad_env >= 0 ? quad_env : (slots_.size() <= 2 ? 7 : 14);
        constexpr uint32_t LT_MAX = MsmQuad<FPP>::LT;   // 128 quads (512 lanes) on the 9-limb field, 64 on the 14-limb one
        const uint32_t lt = (rows > cols ? rows : cols) > LT_MAX ? LT_MAX : (rows > cols ? rows : cols);
        // with other proofs in flight nobody waits for this batch's chain: the instruction-lean sixteen-lane form
        static const int serial_env = env_int("APK_MSM_ROWCOL_SERIAL", -1, -1, 1);
        bool serial = false;
        if (quad_env < 0 && !graphs_on && rows % 4 == 0 && cols % 4 == 0) serial = serial_env >= 0 ? serial_env != 0 : lean;
        if (!APK_PHASE(8)) {
        } else if (serial) {
            path(P_ROWCOL_SERIAL);
            // lanes per line: 16 (19 / 11 addition-times per wave of four rows / columns at c = 16) or 8 (34 / 18 per eight)
            static const int lpl = env_int("APK_MSM_ROWCOL_LANES", 16, 8, 16);
            if (lpl == 8 && rows % 8 == 0 && cols % 8 == 0)
                msm_rowcol_serial_kernel<FPP, 8><<<dim3((rows + cols + 31) / 32, a.batch), 256, 0, st>>>(ptr<PtU>(s.bucket_sum), NB_, rows, cols, ptr<PtU>(s.rowcol));
            else
                msm_rowcol_serial_kernel<FPP, 16><<<dim3((rows + cols + 15) / 16, a.batch), 256, 0, st>>>(ptr<PtU>(s.bucket_sum), NB_, rows, cols, ptr<PtU>(s.rowcol));
        }
        else if (quad & 1)
            msm_rowcol_quad_kernel<FPP><<<dim3(rows + cols, a.batch), 4 * lt, lt * sizeof(PtU), st>>>(ptr<PtU>(s.bucket_sum), NB_, rows, cols, ptr<PtU>(s.rowcol));
        else if (quad & 8)
            msm_rowcol_hybrid_kernel<FPP><<<dim3(rows + cols, a.batch), 256, 0, st>>>(ptr<PtU>(s.bucket_sum), NB_, rows, cols, ptr<PtU>(s.rowcol));
        else
            msm_rowcol_kernel<FPP><<<dim3(rows + cols, a.batch), 256, 0, st>>>(ptr<PtU>(s.bucket_sum), NB_, rows, cols, ptr<PtU>(s.rowcol));
        KCHK();
        if (s.res_write_off + a.batch > (uint32_t)MSM_ARGS_MAX) { set_error("msm: result area overflow"); return APK_ERR_STATE; }
        Pt* const res_out = s.d_pinned ? reinterpret_cast<Pt*>(s.d_pinned + PIN_XYZZ) + s.res_write_off : ptr<Pt>(s.result_xyzz);
        // the sums leave the device in XYZZ form: the one field inversion of the affine conversion takes a lone GPU lane
        // ~100 us and the host a few; sync_results() finishes them into h_out (= the slot's pinned buffer)
        if (!APK_PHASE(16)) {
        } else if ((quad & 6) == 6) {
            // bit sums + final scaling in ONE launch (the last workgroup to finish an MSM's bit sums runs its final phase)
            const uint32_t threads = 4 * lt > 256 ? 4 * lt : 256;
            const size_t lds = (size_t)(lt > 64 ? lt : 64) * sizeof(PtU);
            msm_bitsum_final_quad_kernel<FPP><<<dim3(nbits, 2, a.batch), threads, lds, st>>>(
                ptr<PtU>(s.rowcol), rows, cols, lt, ptr<PtU>(s.bit_partial), ptr<uint32_t>(s.done_count), cols_log, res_out);
            KCHK();
        } else {
            if (quad & 2)
                msm_bitsum_quad_kernel<FPP><<<dim3(nbits, 2, a.batch), 4 * lt, lt * sizeof(PtU), st>>>(ptr<PtU>(s.rowcol), rows, cols, ptr<PtU>(s.bit_partial));
            else
                msm_bitsum_kernel<FPP><<<dim3(nbits, 2, a.batch), 256, 0, st>>>(ptr<PtU>(s.rowcol), rows, cols, ptr<PtU>(s.bit_partial));
            KCHK();
            if (quad & 4)
                msm_final_quad_kernel<FPP><<<a.batch, 256, 0, st>>>(ptr<PtU>(s.bit_partial), nbits, cols_log, nullptr, res_out);
            else
                msm_final_kernel<FPP><<<a.batch, 64, 0, st>>>(ptr<PtU>(s.bit_partial), nbits, cols_log, nullptr, res_out);
            KCHK();
        }
        if (stats_on_) HIPCHK(hipEventRecord(s.ev1, st));
        (void)h_out;
        if (!s.d_pinned) HIPCHK(hipMemcpyAsync(reinterpret_cast<uint8_t*>(s.h_pinned) + PIN_XYZZ + s.res_write_off * sizeof(Pt), s.result_xyzz.p, a.batch * sizeof(Pt), hipMemcpyDeviceToHost, st));
        s.pending_pts = a.batch;
        if (stats_on_) {
            HIPCHK(hipEventSynchronize(s.ev1));
            float tot = 0, acc = 0, srt = 0, tail = 0;
            HIPCHK(hipEventElapsedTime(&tot, s.ev0, s.ev1));
            HIPCHK(hipEventElapsedTime(&acc, s.ev2, s.ev3));
            HIPCHK(hipEventElapsedTime(&srt, s.ev0, s.ev2));
            HIPCHK(hipEventElapsedTime(&tail, s.ev3, s.ev1));
            std::lock_guard<std::mutex> g(stats_mu_);
            stats_.msm_sort_ms += srt;
            stats_.msm_tail_ms += tail;
            stats_.msm_total_ms += tot;
            stats_.msm_batches += 1;
            stats_.msm_accumulate_ms += acc;
            stats_.msm_accumulate_launches += 1;
            for (uint32_t b = 0; b < a.batch; b++) stats_.msm_pairs += a.len[b];
        }
        return APK_OK;
    }

    static Slot*& hook_slot() { static thread_local Slot* p = nullptr; return p; }

    // A commitment batch of the prover.  Without a hook: run_msm on this GPU.  With one (the
    // independent commitments of ONE proof dealt to several GPUs) the batch is parked and handed to the hook at the next
    // sync_results(), after the stream has produced the scalar vectors - the launches queued in between (coset NTTs ...) run
    // on this GPU while the other GPUs commit.
    int commit(Slot& s, const MsmTables& T, int basis, const MsmBatchArgs& a, Aff* h_out) {
        if (!hook_ && s.in_gang) {      // a merge point of the gang: ONE launch sequence for every member's batch (gang_launch)
            MsmReq mr{&s, &T, a};
            GangReq r;
            r.kind = GANG_KIND_MSM; r.args = &mr; r.member = &s;
            const int rc = s.lead->gang.meet(s.gang_idx, r, gang_launcher_);
            if (rc != APK_OK) set_error("%s", r.err.c_str());     // (the launch ran on another member's thread)
            return rc;
        }
        if (!hook_) return run_msm(s, T, a, h_out);
        for (uint32_t b = 0; b < a.batch; b++)
            if (a.len[b] > T.n_bases || a.offset[b] > T.n_bases - a.len[b]) { set_error("msm: %u scalars exceed the %u bases", a.len[b], T.n_bases); return APK_ERR_ARG; }
        s.hook_pending = true; s.hook_basis = basis; s.hook_args = a;
        s.hook_args.plain = T.plain ? 1u : 0u;
        return APK_OK;
    }

    // stream sync + affine conversion of the MSM sums run_msm left in the pinned buffer (results land at h_pinned[0..])
    // ---- tail filling --------------------------------------------------------------------------------------------------------
    // A lone proof spends a fifth of its time in the reduction tails of its four commitment batches (combine, row/column sums,
    // bit sums + final scaling: ~210 us per batch on a few hundred waves), and the launches queued behind a batch - the 4n coset
    // transforms of the polynomials just committed - wait for the tail although they need neither its result nor its SIMDs.  When
    // this proof is the only one in flight they go to a second stream gated on an event recorded right behind the accumulate
    // kernel, so they start when the VALU-bound part of the MSM is over, not before (ungated, round 3's first attempt, they
    // only slowed the accumulate kernel down by what they gained).  With several proofs in flight other proofs fill the tails
    // and a second stream per slot costs throughput (-8 % with 16 slots), so the choice is made per proof.
    int tail_fill(Slot& s) {
        static const int on = env_int("APK_TAIL_FILL", 1, 0, 2);
        static const int graphs = env_int("APK_MSM_GRAPH", 0, 0, 1);
        if (!on || graphs || hook_ || wire_hook_ || stats_on_ || !qk_direct_ || sc_.on()) return 0;
        if (gate_.busy() != 1) return 0;
        if (!s.side) {
            // lowest priority: the tail kernels on the main stream are the critical path, the transforms only have to be done
            // by the time the next round's challenge is known
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (hipStreamCreateWithPriority(&s.side, hipStreamNonBlocking, least) != hipSuccess) { s.side = nullptr; return 0; }
            if (hipEventCreateWithFlags(&s.ev_acc, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&s.ev_side, hipEventDisableTiming) != hipSuccess) return 0;
        }
        return s.ev_acc && s.ev_side ? on : 0;
    }
    int side_begin(Slot& s) { HIPCHK(hipStreamWaitEvent(s.side, s.ev_acc, 0)); return APK_OK; }
    int side_end(Slot& s) { HIPCHK(hipEventRecord(s.ev_side, s.side)); s.side_pending = true; return APK_OK; }

    // Wait for the slot's stream.  hipStreamSynchronize spins on the host (ROCm's default with many CPUs visible): right for a lone
    // proof (its six waits are on the critical path), wrong when 16 proving threads spin at once - on the GPU boxes of this build
    // the container's CPU quota is 16 cores for 256 visible CPUs, and N ranks of an N-GPU node share whatever the node grants.
    // With other proofs in flight the thread therefore sleeps between polls of an event (the wake-up latency hides behind the
    // other proofs).
    int wait_stream(Slot& s) {
        // mode: -1 by load (default: sleep-and-poll with other proofs in flight, spin alone), 0 always spin, 1 hipEventSynchronize on a
        // hipEventBlockingSync event, 2 always sleep-and-poll.
        // Round 6: the "blocking" event wait of rounds 3-5 does NOT give the CPU back on this runtime - a loaded context burned one
        // core per proof in flight (17.6 cores for 16 proofs: bench.py host_cpu_timed_region), i.e. the whole 16-CPU quota of the
        // boxes, and with 32 proofs in flight (gangs) the cgroup froze the process for most of every period.  So with other proofs
        // in flight the thread now polls the event and SLEEPS in between (the wake-up hides behind the other proofs).
        static const int mode = env_int("APK_SYNC_BLOCKING", -1, -1, 2);
        static const int poll_us = env_int("APK_SYNC_POLL_US", 50, 1, 10000);
        const bool loaded = slots_.size() > 2 && gate_.busy() > 2;
        if (s.ev_sync && (mode == 2 || (mode < 0 && loaded))) {
            HIPCHK(hipEventRecord(s.ev_sync, s.stream));
            for (;;) {
                const hipError_t e = hipEventQuery(s.ev_sync);
                if (e == hipSuccess) return APK_OK;
                if (e != hipErrorNotReady) { (void)hipGetLastError(); set_error("hipEventQuery: %s", hipGetErrorString(e)); return APK_ERR_HIP; }
                struct timespec ts = {0, (long)poll_us * 1000L};
                nanosleep(&ts, nullptr);
            }
        }
        if (mode == 1 && s.ev_sync) {
            HIPCHK(hipEventRecord(s.ev_sync, s.stream));
            HIPCHK(hipEventSynchronize(s.ev_sync));
            return APK_OK;
        }
        HIPCHK(hipStreamSynchronize(s.stream));
        return APK_OK;
    }

    int sync_results(Slot& s) {
        if (s.in_gang) {     // the launches this member recorded since its last merge point go out now, with the other members'
            GangReq r;
            r.kind = GANG_KIND_FLUSH; r.member = &s;
            const int rc = s.lead->gang.meet(s.gang_idx, r, gang_launcher_);
            if (rc != APK_OK) { set_error("%s", r.err.c_str()); return rc; }
        }
        CHK(wait_stream(s));
        // whatever the main stream is handed from here on runs behind the side stream's transforms
        if (s.side_pending) { s.side_pending = false; HIPCHK(hipStreamWaitEvent(s.stream, s.ev_side, 0)); }
        if (s.hook_pending) {
            s.hook_pending = false;
            const MsmBatchArgs& a = s.hook_args;
            uint8_t pts[MSM_MAX_BATCH * sizeof(Aff)];
            // the hook's own share of the batch comes back in through apk_msm_g1_batch_device on this thread: it runs on THIS
            // slot (idle now: the stream was just synchronised and the parked batch never touched the MSM workspace) instead
            // of waiting for a free one - with one slot it would wait for itself
            hook_slot() = &s;
            const int rc = hook_(hook_user_, s.hook_basis, a.batch, a.scalars, a.len, pts);
            hook_slot() = nullptr;
            if (rc != APK_OK) { set_error("commit hook failed with %d", rc); return rc == APK_ERR_ARG ? APK_ERR_ARG : APK_ERR_STATE; }
            memcpy(s.h_pinned, pts, a.batch * sizeof(Aff));
            s.pending_pts = 0;
            return APK_OK;
        }
        // (a gang's merged batch left its sums in the LEAD's buffer; this proof's start at res_off)
        const Slot& from = s.lead ? *s.lead : s;
        const Pt* g = reinterpret_cast<const Pt*>(reinterpret_cast<const uint8_t*>(from.h_pinned) + PIN_XYZZ) + s.res_off;
        Aff* o = reinterpret_cast<Aff*>(s.h_pinned);
        // the batch's affine conversions share ONE field inversion (Montgomery's trick on the ZZZ coordinates; an inverse is
        // unique, so the bytes are those of XYZZ::to_affine point by point): 3 of 4 host inversions of ~5 us leave the gap between
        // a batch's results and the next round's launches
        Fp pre[MSM_MAX_BATCH], acc = Fp::one();
        for (uint32_t i = 0; i < s.pending_pts; i++) { pre[i] = acc; if (!g[i].is_inf()) acc = acc * g[i].ZZZ; }
        Fp inv = s.pending_pts ? Fp::inv(acc) : acc;
        for (uint32_t i = s.pending_pts; i-- > 0;) {
            if (g[i].is_inf()) { o[i] = Aff::inf(); continue; }
            const Fp zzz_inv = inv * pre[i];
            inv = inv * g[i].ZZZ;
            const Fp zz_inv = Fp::sqr(zzz_inv * g[i].ZZ);
            o[i] = Aff{g[i].X * zz_inv, g[i].Y * zzz_inv};
        }
        s.pending_pts = 0;
        return APK_OK;
    }

    int alloc_slot(Slot& s) {
        s.lead = &s;
        HIPCHK(hipEventCreate(&s.ev0));
        HIPCHK(hipEventCreate(&s.ev1));
        HIPCHK(hipEventCreate(&s.ev2));
        HIPCHK(hipEventCreate(&s.ev3));
        if (hipEventCreateWithFlags(&s.ev_sync, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); s.ev_sync = nullptr; }
        HIPCHK(hipHostMalloc(&s.h_pinned, PIN_BYTES, hipHostMallocDefault));
        memset(s.h_pinned, 0, PIN_BYTES);   // (the tail-flag word is only ever written by a failing proof)
        // MSM sums and evaluations are a few hundred bytes behind a chain of kernels: the kernel that produces them writes them
        // straight into this (coherent, device-visible) host buffer instead of a device buffer + a copy launch - one launch less
        // per batch, and under load every launch of a proof's chain waits ~0.1 ms for its turn.  APK_ZERO_COPY=0: device buffer + copy.
        static const int zero_copy = env_int("APK_ZERO_COPY", 1, 0, 1);
        if (zero_copy) {
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, s.h_pinned, 0) == hipSuccess) s.d_pinned = static_cast<uint8_t*>(dp);
            else (void)hipGetLastError();
        }
        const size_t fn = (size_t)n_ * sizeof(Fr), fn3 = (size_t)(n_ + 4) * sizeof(Fr), f4 = (size_t)n4_ * sizeof(Fr);
        if (msm_only_) {
            CHK(s.scratch_in.alloc((size_t)msm_bases_ * sizeof(Fr)));
            return alloc_msm_workspace(s, 1);
        }
        CHK(s.wl.alloc(fn3)); CHK(s.wr.alloc(fn3)); CHK(s.wo.alloc(fn3));   // n values + the blinding scalars of a Lagrange-basis commitment
        CHK(s.cl.alloc(fn3)); CHK(s.cr.alloc(fn3)); CHK(s.co.alloc(fn3)); CHK(s.cz.alloc(fn3));
        for (DevBuf* b : {&s.cl, &s.cr, &s.co, &s.cz}) HIPCHK(hipMemset(b->p, 0, fn3));
        CHK(s.qk_lag.alloc(fn)); CHK(s.qk_can.alloc(fn));
        CHK(s.ratio.alloc(fn)); CHK(s.zlag.alloc(fn));
        CHK(s.scan_tot.alloc((size_t)(2 * (cdiv(n_ + 4, SCAN_BLOCK) + 1) + 2) * sizeof(Fr)));
        CHK(s.el.alloc(f4)); CHK(s.er.alloc(f4)); CHK(s.eo.alloc(f4)); CHK(s.ez.alloc(f4)); CHK(s.eqk.alloc(f4));
        CHK(s.quot.alloc(f4)); CHK(s.hcan.alloc(f4));
        CHK(s.pw_z.alloc(fn3)); CHK(s.pw_zi.alloc(fn3)); CHK(s.pw_zw.alloc(fn3)); CHK(s.pw_zwi.alloc(fn3));
        CHK(s.lin.alloc(fn3)); CHK(s.folded.alloc(fn3)); CHK(s.tmp.alloc(fn3)); CHK(s.q1.alloc(fn3)); CHK(s.q2.alloc(fn3));
        CHK(s.eval_partial.alloc((size_t)EVAL_MAX * cdiv(n_ + 4, EVAL_BLOCK) * sizeof(Fr)));
        CHK(s.eval_result.alloc(EVAL_MAX * sizeof(Fr)));
        for (uint32_t k = 0; k < nb_commit_; k++) { CHK(s.pi2_can[k].alloc(fn)); CHK(s.epi2[k].alloc(f4)); }
        CHK(s.scratch_in.alloc(f4));
        CHK(s.tail_flag.alloc(16));
        HIPCHK(hipMemset(s.tail_flag.p, 0, 16));
        CHK(s.ntt_wide.alloc((size_t)NTT_MAX_BATCH * gang_cap_ * n4_ * sizeof(FeU<FRP>)));
        return alloc_msm_workspace(s, ws_batch_);
    }

    // MSM workspace sized for `batch` MSMs over all bases
    int alloc_msm_workspace(Slot& s, uint32_t batch) {
        const uint64_t entries = (uint64_t)batch * msm_bases_ * W_;
        const uint32_t tb = batch * NB_;
        CHK(s.hist.alloc((size_t)(tb + 1) * 4)); CHK(s.offsets.alloc((size_t)(tb + 1) * 4));
        CHK(s.unit_off.alloc((size_t)(tb + 1) * 4));
        CHK(s.scan_blk.alloc((size_t)(3 + MSM_BINS) * (cdiv(tb, MSM_SCAN_BLOCK) + 1) * 4));
        CHK(s.merge_rank.alloc((size_t)(tb + 1) * 4)); CHK(s.merge_list.alloc((size_t)(tb + 1) * 4));
        CHK(s.full_off.alloc((size_t)(tb + 1) * 4)); CHK(s.rem_rank.alloc((size_t)(tb + 1) * 4)); CHK(s.rem_list.alloc((size_t)(tb + 1) * 4));
        {   // counter arrays: [batch][G][NB] of the one-level sort, or what the two levels keep between their launches
            // ([batch][G][P] counts + run starts, partition totals and chunk sums; or the fused form's [batch][G][P + 1] run tables)
            uint32_t sl = part_stage_max() / (uint32_t)W_;       // run_msm_body's slice: APK_MSM_SLICE (2 048) scalars, or what the stage holds
            const uint32_t sl_env = (uint32_t)env_int("APK_MSM_SLICE", 2048, 64, 1 << 20);
            if (sl > sl_env) sl = sl_env;
            if (sl < 1) sl = 1;
            const uint64_t g2max = (uint64_t)cdiv(msm_bases_, sl) + 1u;
            const uint64_t g2 = g2max > MSM_PART_GMAX ? MSM_PART_GMAX : g2max;
            const uint64_t two_level = (uint64_t)batch * g2 * ((uint64_t)part_cfg_.P + 1) * 2 + (uint64_t)batch * part_cfg_.P * (2 + MSM_PART_CHUNKS);
            const uint64_t one_level = one_level_ok() ? (uint64_t)tb * msm_G_max_ : 0;
            CHK(s.counts.alloc((size_t)(one_level > two_level ? one_level : two_level) * 4));
        }
        CHK(s.sorted.alloc(entries * 4));
        if (!one_level_ok() || (env_int("APK_MSM_SORT2", -1, -1, 1) != 0 && part_cfg_.P >= 4 && (msm_bases_ >= 65536u || (many_slots_ && msm_bases_ >= 8192u) || env_int("APK_MSM_SORT2", -1, -1, 1) == 1)))
        {
            CHK(s.sort_tmp.alloc((entries + (uint64_t)batch * MSM_PART_GMAX * (uint64_t)W_ + 4096u) * 4));   // two-level sort: packed entries between the levels (slice-major runs: a few entries of slack per slice)
            CHK(s.ptot2.alloc((size_t)2 * MSM_MAX_BATCH * MSM_PART_MAX * 4));
            HIPCHK(hipMemset(s.ptot2.p, 0, (size_t)2 * MSM_MAX_BATCH * MSM_PART_MAX * 4));
        }
        {   // unit partials: entries / 16 + a remainder per bucket - or, for small batches, entries / MSM_UNIT_SMALL (run_msm_body)
            const uint64_t big = entries / MSM_UNIT_MIN + tb;
            const uint64_t small = (entries < 4 * MSM_SMALL_ENTRIES ? entries : 4 * MSM_SMALL_ENTRIES) / MSM_UNIT_SMALL + tb;
            CHK(s.partial.alloc((big > small ? big : small) * sizeof(PtU)));
        }
        CHK(s.bucket_sum.alloc((size_t)tb * sizeof(PtU)));
        {
            const int m_bits = c_ - 1;
            const uint32_t rows = 1u << (m_bits / 2), cols = 1u << (m_bits - m_bits / 2);
            CHK(s.rowcol.alloc((size_t)batch * (rows + cols) * sizeof(PtU)));
        }
        CHK(s.bit_partial.alloc((size_t)batch * 2 * 32 * sizeof(PtU)));
        CHK(s.result.alloc(MSM_ARGS_MAX * sizeof(Aff)));
        CHK(s.result_xyzz.alloc(MSM_ARGS_MAX * sizeof(Pt)));
        CHK(s.done_count.alloc((MSM_ARGS_MAX + 1) * sizeof(uint32_t)));   // per MSM: bit sums done; + 1: scan workgroups done
        HIPCHK(hipMemset(s.done_count.p, 0, (MSM_ARGS_MAX + 1) * sizeof(uint32_t)));
        return APK_OK;
    }

    // ---- host-input staging sets (see InputSet) ---------------------------------------------------------------------------------
    int ensure_input_sets() {
        std::lock_guard<std::mutex> lk(mu_);
        if (!in_sets_.empty()) return APK_OK;
        // as many sets again as proving slots (callers waiting for a slot have their upload in flight), at most 64 - the most
        // callers a context proves for at a time (16 streams of four)
        static const int sets_env = env_int("APK_INPUT_SETS", 0, 0, 128);
        size_t count = sets_env ? (size_t)sets_env : 2 * slots_.size();
        if (count > 64 && !sets_env) count = 64;
        if (count < 1) count = 1;
        if (!copy_stream_) HIPCHK(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
        std::vector<InputSet*> made;
        int rc = APK_OK;
        for (size_t i = 0; i < count && rc == APK_OK; i++) {
            InputSet* is = new InputSet();
            is->index = i;
            made.push_back(is);
            rc = alloc_input_set(*is);
        }
        if (rc != APK_OK && made.size() > 1) {       // out of memory half way: keep the sets that were completed
            InputSet* last = made.back();
            made.pop_back();
            if (last->ready) (void)hipEventDestroy(last->ready);
            delete last;
            (void)hipGetLastError();
            rc = APK_OK;
        }
        if (rc != APK_OK) { for (InputSet* is : made) delete is; return rc; }
        in_sets_ = made;
        in_gate_.resize(in_sets_.size());
        return APK_OK;
    }
    int alloc_input_set(InputSet& is) {
        const size_t fn3 = (size_t)(n_ + 4) * sizeof(Fr);
        for (int j = 0; j < 3; j++) CHK(is.w[j].alloc(fn3));
        for (uint32_t k = 0; k < nb_commit_; k++) CHK(is.pi2[k].alloc(fn3));
        is.copy = copy_stream_;
        HIPCHK(hipEventCreateWithFlags(&is.ready, hipEventDisableTiming));
        return APK_OK;
    }
    struct InputGuard {
        CurveBackend* b; InputSet* set = nullptr; bool ok = false;
        explicit InputGuard(CurveBackend* b_) : b(b_) {}
        int take() {
            CHK(b->ensure_input_sets());
            set = b->in_sets_[b->in_gate_.acquire().slot];
            return APK_OK;
        }
        ~InputGuard() {
            if (!set) return;
            // an error return may leave copies or kernels in flight that still read the caller's buffers / this set
            if (!ok) (void)hipDeviceSynchronize();
            b->in_gate_.release(set->index);
        }
    };

    // ---- deferred launches of a gang member (gang_kernel.h) -----------------------------------------------------------------------
    // Every element-wise / scan / reduction kernel of the prover goes through klaunch().  Alone, it is an ordinary launch.  As a
    // gang member (with the zero-copy result buffer: nothing but kernels then touches the stream between two merge points) it is
    // RECORDED - grid, block, the arguments by value - and launched at the member's next merge point or stream wait, together with
    // the same kernel of the other members: ONE launch, blockIdx.z = member.
    template <class K, int BOUNDS, class... P>
    static int launch_multi(hipStream_t st, const Cmd* const* cmds, int count) {
        static_assert(sizeof(Pack<P...>) <= CMD_BLOB, "kernel arguments exceed a recorded launch's blob");
        static_assert(sizeof(GangPack<P...>) <= 4096, "a gang's arguments exceed the kernel argument segment");
        if (count == 1) {
            const Pack<P...>& one = *reinterpret_cast<const Pack<P...>*>(cmds[0]->blob);
            GangPack<P...> g{};
            g.m[0] = one;
            polyG_kernel<K, BOUNDS, P...><<<cmds[0]->grid, cmds[0]->block, cmds[0]->lds, st>>>(g);
            KCHK();
            return APK_OK;
        }
        GangPack<P...> g{};
        for (int i = 0; i < count; i++) g.m[i] = *reinterpret_cast<const Pack<P...>*>(cmds[i]->blob);
        dim3 grid = cmds[0]->grid;
        grid.z = (uint32_t)count;
        polyG_kernel<K, BOUNDS, P...><<<grid, cmds[0]->block, cmds[0]->lds, st>>>(g);
        KCHK();
        return APK_OK;
    }
    template <class K, int BOUNDS, class... P>
    int klaunch(hipStream_t st, dim3 grid, dim3 block, size_t lds, P... p) {
        Slot* m = gang_member();
        static const int defer = env_int("APK_GANG_DEFER", 1, 0, 1);
        if (defer && m && m->in_gang && m->d_pinned && st == m->stream && grid.z == 1) {
            m->cmds.emplace_back();
            Cmd& c = m->cmds.back();
            c.grid = grid; c.block = block; c.lds = (uint32_t)lds;
            c.multi = &launch_multi<K, BOUNDS, P...>;
            new (c.blob) Pack<P...>(make_pack_impl(p...));
            return APK_OK;
        }
        poly1_kernel<K, BOUNDS, P...><<<grid, block, lds, st>>>(p...);
        KCHK();
        return APK_OK;
    }
    // launch what the members of a meeting have recorded: instance i of every member in ONE launch when they are the same kernel
    // with the same launch shape (they are: the members run the same prover over the same circuit), else one by one
    int flush_cmds(GangReq* const* reqs, int count, hipStream_t st) {
        Slot* ms[GANG_MAX];
        int nm = 0;
        size_t longest = 0;
        for (int i = 0; i < count; i++) {
            Slot* m = static_cast<Slot*>(reqs[i]->member);
            if (m && !m->cmds.empty()) { ms[nm++] = m; if (m->cmds.size() > longest) longest = m->cmds.size(); }
        }
        int rc = APK_OK;
        for (size_t i = 0; i < longest && rc == APK_OK; i++) {
            bool done[GANG_MAX] = {false, false, false, false};
            for (int a = 0; a < nm && rc == APK_OK; a++) {
                if (done[a] || i >= ms[a]->cmds.size()) continue;
                const Cmd& ca = ms[a]->cmds[i];
                const Cmd* grp[GANG_MAX];
                int ng = 0;
                for (int b = a; b < nm; b++) {
                    if (done[b] || i >= ms[b]->cmds.size()) continue;
                    const Cmd& cb = ms[b]->cmds[i];
                    if (cb.multi != ca.multi || cb.grid.x != ca.grid.x || cb.grid.y != ca.grid.y || cb.block.x != ca.block.x || cb.lds != ca.lds) continue;
                    grp[ng++] = &cb;
                    done[b] = true;
                }
                rc = ca.multi(st, grp, ng);
                if (ng > 1) path(P_GANG_KERNELS);
            }
        }
        for (int a = 0; a < nm; a++) ms[a]->cmds.clear();
        return rc;
    }

    // ---- gangs (gang.h) ----------------------------------------------------------------------------------------------------------
    static int choose_gang(int log_n, int nslots, int max_slots) {
        const int env = env_int("APK_GANG", 0, 0, GANG_MAX);
        if (nslots <= max_slots) return 1;            // every caller gets a stream of its own
        if (env) return env;
        // Same box, 32 / 64 persistent callers with distinct witnesses (tools/sweep_gangs.sh, profiles/r06_gang_sweep.txt), proofs/s
        // alone -> gangs of 2 (32 callers) -> gangs of 4 (64 callers):
        //   BN254 2^13  2 175 -> . -> 3 846     BN254 2^15  1 499 -> 1 763 -> 1 825     BLS12-381 2^14  1 376 -> 1 604 -> 1 746 .. 1 775
        //   BN254 2^16    958 -> 1 033 -> 1 026   BN254 2^17  538 -> 539 (VALU-bound: nothing left for wider launches to fill)
        //   BLS12-381 2^15  852 -> 933 -> 961;  BLS12-381 2^16  501.6 -> 497.4 with pairs (neutral: off);  BN254 2^14  1 997 -> 2 704 -> 3 029
        return log_n <= 15 ? 4 : (log_n == 16 && FPP::N <= 8) ? 2 : 1;
    }
    static Slot*& gang_member() { static thread_local Slot* p = nullptr; return p; }
    bool gangs_allowed() const {
        static const int graphs = env_int("APK_MSM_GRAPH", 0, 0, 1);
        return gang_cap_ > 1 && !hook_ && !wire_hook_ && !sc_.on() && !stats_on_ && !graphs;
    }
    // a proof's slot, alone or as a member of a gang
    struct MemberGuard {
        CurveBackend* b; Slot* s; bool ok = false;
        explicit MemberGuard(CurveBackend* b_) : b(b_) {
            const SlotGate::Ticket t = b->gate_.acquire_member(b->gangs_allowed());
            s = b->slots_[t.slot];
            Slot* lead = b->slots_[t.lead];
            s->lead = lead; s->gang_idx = t.idx; s->in_gang = t.size > 1; s->res_off = 0;
            if (lead == s) s->own_stream = b->stream_pool_[(size_t)t.stream];
            s->stream = b->stream_pool_[(size_t)t.stream];
            if (s->in_gang) { lead->gang.enter(t.gen, t.size); gang_member() = s; b->path(P_GANG_PROOFS); }
        }
        ~MemberGuard() {
            if (s->in_gang) {
                gang_member() = nullptr;
                // an error return may leave this proof's launches in flight on the shared stream: nobody may get this workspace before them
                if (!ok) (void)hipStreamSynchronize(s->stream);
                Slot* lead = s->lead;
                lead->gang.leave(s->gang_idx);
                if (lead == s) lead->gang.wait_empty();       // the others run on this slot's stream and MSM workspace
            }
            s->lead = s; s->in_gang = false; s->res_off = 0;
            s->cmds.clear();
            b->release(s);
        }
    };
    enum { GANG_KIND_MSM = 1, GANG_KIND_NTT = 2, GANG_KIND_FLUSH = 3 };
    struct MsmReq { Slot* member; const MsmTables* T; MsmBatchArgs a; };
    struct NttReq {
        Slot* member; int which; bool inverse; int count;
        const Fr* ins[NTT_MAX_BATCH]; Fr* outs[NTT_MAX_BATCH]; uint32_t in_lens[NTT_MAX_BATCH];
        uint32_t out_len; const Fr *pre, *post, *scale; int sub_log;
        bool same_shape(const NttReq& o) const {
            return which == o.which && inverse == o.inverse && out_len == o.out_len && pre == o.pre && post == o.post && scale == o.scale && sub_log == o.sub_log;
        }
    };
    // The merged launches of one meeting (called once, by the last member to arrive): requests that can share a launch sequence
    // do - MSM batches over the same table, transforms of the same shape - anything else goes by itself, in member order.
    void gang_launch(GangReq* const* reqs, int count) {
        bool done[GANG_MAX] = {false, false, false, false};
        uint32_t res_base = 0;       // MSM sums of this meeting's launch sequences sit one behind the other in the lead's XYZZ area
        // first what the members recorded on their way here (the kernels in front of this merge point), merged
        {
            Slot* any = static_cast<Slot*>(reqs[0]->member);
            const int frc = any ? flush_cmds(reqs, count, any->stream) : APK_OK;
            if (frc != APK_OK) {
                for (int i = 0; i < count; i++) { reqs[i]->rc = frc; reqs[i]->err = apk_last_error(); }
                return;
            }
        }
        for (int i = 0; i < count; i++) {
            if (done[i]) continue;
            if (reqs[i]->kind == GANG_KIND_FLUSH) { reqs[i]->rc = APK_OK; done[i] = true; continue; }
            if (reqs[i]->kind == GANG_KIND_MSM) {
                MsmReq* first = static_cast<MsmReq*>(reqs[i]->args);
                Slot& lead = *first->member->lead;
                MsmBatchArgs c{};
                int grp[GANG_MAX], ng = 0;
                for (int j = i; j < count; j++) {
                    if (done[j] || reqs[j]->kind != GANG_KIND_MSM) continue;
                    MsmReq* m = static_cast<MsmReq*>(reqs[j]->args);
                    if (m->T != first->T || c.batch + m->a.batch > ws_batch_) continue;
                    m->member->res_off = res_base + c.batch;
                    for (uint32_t b = 0; b < m->a.batch; b++) { c.scalars[c.batch] = m->a.scalars[b]; c.len[c.batch] = m->a.len[b]; c.offset[c.batch] = m->a.offset[b]; c.batch++; }
                    grp[ng++] = j;
                    done[j] = true;
                }
                const uint32_t lead_pending = lead.pending_pts;       // (run_msm counts the whole sequence as the workspace owner's)
                lead.res_write_off = res_base;
                const int rc = run_msm(lead, *first->T, c, nullptr);
                lead.res_write_off = 0;
                lead.pending_pts = lead_pending;
                res_base += c.batch;
                if (ng > 1) path(P_GANG_MSM);
                for (int k = 0; k < ng; k++) {
                    MsmReq* m = static_cast<MsmReq*>(reqs[grp[k]]->args);
                    m->member->pending_pts = rc == APK_OK ? m->a.batch : 0;
                    reqs[grp[k]]->rc = rc;
                    if (rc != APK_OK) reqs[grp[k]]->err = apk_last_error();
                }
            } else {
                NttReq* first = static_cast<NttReq*>(reqs[i]->args);
                Slot& lead = *first->member->lead;
                const Fr* ins[NTT_ARGS_MAX]; Fr* outs[NTT_ARGS_MAX]; uint32_t lens[NTT_ARGS_MAX];
                int total = 0, grp[GANG_MAX], ng = 0;
                for (int j = i; j < count; j++) {
                    if (done[j] || reqs[j]->kind != GANG_KIND_NTT) continue;
                    NttReq* m = static_cast<NttReq*>(reqs[j]->args);
                    if (!m->same_shape(*first) || total + m->count > NTT_MAX_BATCH * gang_cap_) continue;
                    for (int t = 0; t < m->count; t++) { ins[total] = m->ins[t]; outs[total] = m->outs[t]; lens[total] = m->in_lens[t]; total++; }
                    grp[ng++] = j;
                    done[j] = true;
                }
                const int rc = run_ntt_batch_now(lead.own_stream, first->which, first->inverse, total, ins, outs, lens, first->out_len, first->pre,
                                                 first->post, first->scale, first->sub_log);
                if (ng > 1) path(P_GANG_NTT);
                for (int k = 0; k < ng; k++) { reqs[grp[k]]->rc = rc; if (rc != APK_OK) reqs[grp[k]]->err = apk_last_error(); }
            }
        }
    }

    Slot* acquire() {
        const SlotGate::Ticket t = gate_.acquire();
        Slot* s = slots_[t.slot];
        s->own_stream = s->stream = stream_pool_[(size_t)t.stream];
        return s;
    }
    void release(Slot* s) {
        // an error return between a side-stream launch and the next sync leaves transforms in flight: whoever gets the STREAM next
        // is ordered behind them
        if (s->side_pending) { s->side_pending = false; if (s->own_stream) (void)hipStreamWaitEvent(s->own_stream, s->ev_side, 0); }
        s->own_stream = s->stream = nullptr;      // (the workspace lookup by stream must only ever find the slot that holds it now)
        gate_.release(s->index);
    }
    struct SlotGuard {
        CurveBackend* b; Slot* s;
        SlotGuard(CurveBackend* b_) : b(b_), s(b_->acquire()) {}
        ~SlotGuard() { b->release(s); }
    };

    int powers(hipStream_t st, Fr* out, uint32_t count, const Fr& w, const Fr& scale) {
        PowersBatch<FRP> pb{};
        pb.out[0] = out; pb.w[0] = w; pb.scale[0] = scale;
        return klaunch<PowersK<FRP>, 256>(st, dim3(cdiv(cdiv(count, 8), 256), 1), 256, 0, pb, count);
    }
    int powers_batch(hipStream_t st, int k, Fr* const* outs, const Fr* ws, uint32_t count) {
        PowersBatch<FRP> pb{};
        for (int i = 0; i < k; i++) { pb.out[i] = outs[i]; pb.w[i] = ws[i]; pb.scale[i] = Fr::one(); }
        return klaunch<PowersK<FRP>, 256>(st, dim3(cdiv(cdiv(count, 8), 256), k), 256, 0, pb, count);
    }

    // LDS of the sort kernels: 32-bit counters, or packed 16-bit pairs from 2^16 buckets (c = 17)
    size_t digits_lds_bytes() const { return NB_ >= MSM_PACKED_NB ? (size_t)NB_ * 2 : (size_t)NB_ * 4; }
    // windows above 17 bits (2^17..2^19 buckets) have no one-level sort: their histogram does not fit the LDS.  They sort in two
    // levels only (partitions of <= 256 buckets), whatever the load.
    bool one_level_ok() const { return NB_ <= 65536u; }
    // slices of the first level: the stage and its 2 P + 1 cursors share the 160 KiB
    uint32_t part_stage_max() const { return msm_part_stage_max(part_cfg_.P ? part_cfg_.P : 4u); }
    int set_sort_lds_limits() {
        if (one_level_ok() && digits_lds_bytes() > 65536) {
            const int dl = (int)digits_lds_bytes();
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&msm_digits_kernel<FRP, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, dl));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&msm_digits_kernel<FRP, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, dl));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&msm_digits_kernel<FRP, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, dl));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&msm_digits_kernel<FRP, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, dl));
        }
        const int part_lds = (int)(MSM_LDS_WORDS - 63u) * 4;     // stage + cursors (msm_part_stage_max)
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&msm_part_kernel<FRP, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, part_lds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&msm_part_kernel<FRP, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, part_lds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&msm_part_kernel<FRP, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, part_lds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&msm_part_kernel<FRP, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, part_lds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&msm_part_sort_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)MSM_PART_TILE * 4));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&msm_part1_kernel<FRP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, part_lds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&msm_part1_kernel<FRP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, part_lds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&msm_part_sort_runs_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)MSM_PART_TILE * 4));
        return APK_OK;
    }

    int choose_window(int requested, int log_size, int slots = 1) {
        c_ = requested;
        if (c_ == 0) {
            c_ = env_int("APK_MSM_WINDOW", 0, 0, 20);
            if (c_ != 0 && c_ < 7) c_ = 7;
        }
        if (c_ == 0 && log_size >= 20) {
            // Round 5, from 2^20 bases: the widest window the two-level sort's packed entry has room for - 19 bits (14 windows
            // instead of 16) up to 2^21 bases, 18 (15 windows) at 2^22 and 2^23 - else 16.  Same box, one MSM at a time / config 5 with four
            // proofs in flight (profiles/r05_msm_size_sweep.json): BN254 2^20 1.73 -> 1.63 ms, 2^21 3.16 -> 2.97, 2^22 6.15 -> 6.01;
            // BLS12-381 2^20 3.32 -> 3.18, 2^21 6.26 -> 5.73 ms and 17.2 -> 18.6 proofs/s (c = 18: 17.8); c = 20 loses again at
            // 2^20 (1.79 / 3.41 ms: 2^19 buckets per MSM in the reduction) and does not fit the entry at 2^21.
            // (with up to 8 192 partitions 19 bits also fit 2^22 and 18 bits 2^23: measured 6.03 against 5.86 ms at 2^22 - 18 stays
            // there - and 12.74 against 13.03 ms for 18 against 16 bits at 2^23)
            for (int cand : {19, 18}) {
                if (cand == 19 && log_size >= 22) continue;
                if (choose_window(cand, log_size, slots) == APK_OK) return APK_OK;
            }
            c_ = 16;
        }
        // measured flat between log2(n)-4 and log2(n)-2 (tools/sweep.sh).  c = 16 (128 KiB LDS histograms, 16 windows instead of
        // 17) pays from 2^21 up on any context, and from 2^17 up on throughput contexts: at 2^17 with 16 slots +1.4 % proofs/s
        // for +1.5 % latency (tools/ab_args.sh, same box, interleaved: 429 against 423 proofs/s; c = 14: 403)
        if (c_ == 0) {
            c_ = log_size - 2; if (c_ < 8) c_ = 8; if (c_ > 15) c_ = 15;
            if (log_size >= 21 || (log_size >= 17 && slots > 2)) c_ = 16;
            // throughput contexts below 2^17, re-measured with the lean tail kernels (round 3, same-box sweeps): fewer windows
            // pay again - BLS12-381 2^14: c = 12 -> 1 176, 13 -> 1 207, 14 -> 1 147 proofs/s; BN254 2^16: 14 -> 800, 15 -> 839,
            // 16 -> 825; BN254 2^15: 13 -> 1 274, 14 -> 1 224, 15 -> 1 274 (and the lower latency)
            else if (slots > 2 && log_size == 14) c_ = 13;
            else if (slots > 2 && (log_size == 15 || log_size == 16)) c_ = 15;
            // (2^16 went to 16 bits with the long accumulate units under load, below: 976 / 972 -> 985 / 977 proofs/s, same box)
            if (slots > 2 && log_size == 16 && FPP::N <= 8) c_ = 16;
            // 2^18 and 2^19: 17 bits (15 windows; the histogram's packed 16-bit counters hold up to 786 432 bases).  Round 5, same
            // box, two interleaved rounds: 2^18 251 / 253 -> 260 / 261 proofs/s, 2^19 125.1 / 125.3 -> 130.0 / 130.5 (bit-heavy
            // witness +2.6 % / +3 %), a lone proof 5.24 -> 5.12 and 9.59 -> 9.24 ms; at 2^17 the two widths tie (528.8 against
            // 528.7 over three rounds) and 16 stays.
            // BLS12-381 (14-limb field: its reduction tail costs 2.3 x as much per bucket) loses 2 % with 17 bits at 2^18 and ties at 2^19.
            if ((log_size == 18 || log_size == 19) && FPP::N <= 8) c_ = 17;
            // 2^17 on a throughput context: the tie broke when the accumulate units under load went to 48 entries (run_msm_body:
            // a bucket of 30 entries is then ONE partial sum and the merge of 65 536 buckets costs next to nothing).  Same box,
            // two rounds, two boxes: 16 bits 516.8 / 516.1 and 529.4 / 528.4, 17 bits with long units 533.2 / 532.1 and
            // 544.5 / 548.7 proofs/s (+3 %); a lone proof 3.21 -> 3.24 ms.  (18 bits: 497; 2^16 with 17 bits: -1.5 %.)
            if (log_size == 17 && slots > 2 && FPP::N <= 8) c_ = 17;
        }
        if (c_ < 7 || c_ > 20) { set_error("msm_window %d out of [7,20]", c_); return APK_ERR_ARG; }
        // c = 17 counts in packed 16-bit halves: a sort slice (at most msm_G_max_ of them) must stay below 2^16 entries
        if (c_ == 17 && (uint64_t)msm_bases_ > (uint64_t)msm_G_max_ * 3072u) { set_error("msm_window 17 supports at most %u bases", msm_G_max_ * 3072u); return APK_ERR_ARG; }
        W_ = (FRP::BITS + 1 + c_ - 1) / c_;
        NB_ = 1u << (c_ - 1);
        {   // spread the BITS+1 bits over W_ windows of width c_ or c_-1
            const int bits = FRP::BITS + 1;
            const int base = bits / W_, extra = bits % W_;
            win_.W = W_;
            int o = 0;
            for (int j = 0; j < W_; j++) {
                win_.off[j] = (uint16_t)o;
                win_.width[j] = (uint8_t)(base + (j < extra ? 1 : 0));
                o += win_.width[j];
            }
            win_.off[W_] = (uint16_t)o;
        }
        if ((uint64_t)msm_bases_ * W_ >= (1ull << 31)) { set_error("bases*windows exceeds 2^31 table entries"); return APK_ERR_ARG; }
        {   // two-level sort (kernels_msm.h MsmPartCfg): index bits as needed; partitions of <= 256 buckets, halved until the mean
            // partition of a full-length MSM holds <= APK_MSM_PART_TARGET entries (what the second level's largest LDS tile takes
            // with its slack; FEWER, larger partitions measured better for a lone proof at c = 15 - 64 partitions of 35 k entries:
            // 3.42 ms, 256 of 9 k: 3.49, 512: 3.62 - and c = 16 at 2^17 keeps its 128 partitions of 16 k either way), and as far
            // as the bits left beside the index allow.  P = 0: the context's MSMs do not take the two-level sort.
            part_cfg_ = MsmPartCfg{};
            uint32_t idx_bits = 1;
            while (((uint64_t)1 << idx_bits) < (uint64_t)msm_bases_ * W_) idx_bits++;
            const uint32_t target = (uint32_t)env_int("APK_MSM_PART_TARGET", MSM_PART_TILE - 2048, 1024, MSM_PART_TILE - 2048);
            const uint64_t per_msm = (uint64_t)msm_bases_ * W_;
            int pb_log = c_ - 1 < 8 ? c_ - 1 : 8;
            if (idx_bits < 31 && pb_log > (int)(31 - idx_bits)) pb_log = 31 - idx_bits;
            while (pb_log > 2 && (NB_ >> pb_log) < MSM_PART_MAX && per_msm / (NB_ >> pb_log) > target) pb_log--;
            const int pb_env = env_int("APK_MSM_PART_PBLOG", 0, 0, 8);
            if (pb_env && pb_env < c_ - 1 && idx_bits + pb_env <= 31 && (NB_ >> pb_env) <= MSM_PART_MAX) pb_log = pb_env;
            if (idx_bits <= 29 && pb_log >= 2 && idx_bits + pb_log <= 31 && (NB_ >> pb_log) >= 4 && (NB_ >> pb_log) <= MSM_PART_MAX &&
                per_msm / (NB_ >> pb_log) <= MSM_PART_TILE - 2048) {
                part_cfg_.idx_bits = idx_bits; part_cfg_.pb_log = (uint32_t)pb_log; part_cfg_.P = NB_ >> pb_log; part_cfg_.run_lanes = 64;
            }
        }
        if (!one_level_ok()) {
            // 2^17..2^19 buckets: the two-level sort or nothing - its packed entry needs index bits + partition bits <= 31 and at
            // most MSM_PART_MAX partitions, and the first level at most MSM_PART_GMAX slices whose entries fit the LDS stage
            if (part_cfg_.P < 4) { set_error("msm_window %d: %u bases x %d windows leave no room for the partition bits of the two-level sort (index bits + partition bits <= 31, <= %u partitions)", c_, msm_bases_, W_, MSM_PART_MAX); return APK_ERR_ARG; }
            if ((uint64_t)cdiv(msm_bases_, MSM_PART_GMAX) * W_ > part_stage_max()) { set_error("msm_window %d: %u bases need more than %u sort slices", c_, msm_bases_, MSM_PART_GMAX); return APK_ERR_ARG; }
        }
        return APK_OK;
    }

    // ---------------------------------------------------------------------------------------------- init
    // MSM-only context: SRS tables + one MSM workspace per slot, no circuit (single sharded MSM, BASELINE config 4)
    int init_msm_only(int device, const void* bases, uint64_t count, int msm_window) override {
        if (!bases || count == 0 || count >= (1ull << 27)) { set_error("msm context: bad base count"); return APK_ERR_ARG; }
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0) { set_error("no HIP device available (%s); libapk has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e)); return APK_ERR_HIP; }
        if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return APK_ERR_ARG; }
        device_ = device;
        HIPCHK(hipSetDevice(device_));
        { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_) == hipSuccess && cus > 0) simds_ = 4u * (uint32_t)cus; }
        msm_only_ = true;
        ws_batch_ = 1;             // (alloc_slot sizes an MSM-only context's workspace for one MSM at a time)
        msm_bases_ = (uint32_t)count;
        n_ = (uint32_t)count;
        int lg = 0;
        while ((1ull << lg) < count) lg++;
        CHK(choose_window(msm_window, lg));
        CHK(set_sort_lds_limits());
        DevBuf srs;
        CHK(srs.alloc(count * sizeof(Aff)));
        HIPCHK(hipMemcpy(srs.p, bases, count * sizeof(Aff), hipMemcpyHostToDevice));
        CHK(build_tables(nullptr, ptr<Aff>(srs), (uint32_t)count, tab_can_));
        HIPCHK(hipDeviceSynchronize());
        Slot* s = new Slot();
        s->index = slots_.size();
        slots_.push_back(s);
        hipStream_t ps = nullptr;
        HIPCHK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
        stream_pool_.push_back(ps);
        gate_.resize(slots_.size());
        CHK(alloc_slot(*s));
        return APK_OK;
    }

    int init(const apk_circuit_desc* d) override {
        if (d->n < 8 || (d->n & (d->n - 1)) || d->n > (1ull << 24)) { set_error("n=%llu must be a power of two in [8, 2^24]", (unsigned long long)d->n); return APK_ERR_ARG; }
        if (d->nb_commitments > APK_MAX_COMMITMENTS) { set_error("at most %d BSB22 commitments", APK_MAX_COMMITMENTS); return APK_ERR_ARG; }
        if (!d->srs_g1 || !d->ql || !d->qr || !d->qm || !d->qo || !d->qk || !d->perm) { set_error("null circuit pointer"); return APK_ERR_ARG; }
        if (d->nb_commitments && !d->srs_g1_lagrange) { set_error("BSB22 commitments need the Lagrange SRS"); return APK_ERR_ARG; }
        if (d->nb_public > d->n) { set_error("nb_public > n"); return APK_ERR_ARG; }
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0) { set_error("no HIP device available (%s); libapk has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e)); return APK_ERR_HIP; }
        if (d->device < 0 || d->device >= ndev) { set_error("device %d out of range (%d devices)", d->device, ndev); return APK_ERR_ARG; }
        device_ = d->device;
        HIPCHK(hipSetDevice(device_));
        { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_) == hipSuccess && cus > 0) simds_ = 4u * (uint32_t)cus; }
        n_ = (uint32_t)d->n;
        n4_ = 4 * n_;
        log_n_ = 0;
        while ((1u << log_n_) < n_) log_n_++;
        nb_public_ = d->nb_public;
        nb_commit_ = d->nb_commitments;
        for (uint32_t k = 0; k < nb_commit_; k++) {
            cci_[k] = d->commitment_constraint_index[k];
            // the prover writes the commitment's hash into row nb_public + cci of Qk: must be a row of the domain on every path
            if ((uint64_t)nb_public_ + cci_[k] >= n_) { set_error("commitment_constraint_index[%u] = %u: row %llu is outside the domain (n = %u)", k, cci_[k], (unsigned long long)nb_public_ + cci_[k], n_); return APK_ERR_ARG; }
        }
        msm_bases_ = n_ + 3;
        CHK(choose_window(d->msm_window, (int)log_n_, d->slots));
        // domain constants on the host (gnark fft.NewDomain [UPSTREAM]; generator = VK Generator,
        // templateLogicSigBN254.go:57; shift = VK CosetShift :68)
        Fr root = root_of_unity();
        const int adicity = FRP::ADICITY;
        omega4_ = root;
        for (int i = 0; i < adicity - (int)log_n_ - 2; i++) omega4_ = Fr::sqr(omega4_);
        omega_ = Fr::sqr(Fr::sqr(omega4_));
        omega_inv_ = Fr::inv(omega_);
        omega4_inv_ = Fr::inv(omega4_);
        shift_ = fr_u64(FRP::COSET_SHIFT);
        shift_inv_ = Fr::inv(shift_);
        n_inv_ = Fr::inv(fr_u64(n_));
        n4_inv_ = Fr::inv(fr_u64(n4_));
        {   // 1/(x^n - 1) on the 4n coset takes four values: x^n = shift^n * i^k, i = omega4^n
            Fr sn = Fr::pow_u64(shift_, n_);
            Fr i4 = Fr::pow_u64(omega4_, n_);
            Fr cur = sn;
            for (int k = 0; k < 4; k++) { zh_inv_[k] = Fr::inv(cur - Fr::one()); cur = cur * i4; }
        }
        hipStream_t st = nullptr;  // default stream during setup
        const size_t fn = (size_t)n_ * sizeof(Fr), f4 = (size_t)n4_ * sizeof(Fr);
        CHK(tw_n_.alloc(fn / 2)); CHK(twi_n_.alloc(fn / 2)); CHK(tw_4n_.alloc(f4 / 2)); CHK(twi_4n_.alloc(f4 / 2));
        const Fr ru = fr_u64(32);   // R'/R = 2^5: x R -> x R'
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_kernel<FRP, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(((size_t)1 << NTT_TILE_LOG) * sizeof(FeU<FRP>))));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_kernel<FRP, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(((size_t)1 << NTT_TILE_LOG) * sizeof(FeU<FRP>))));
        CHK(twu_n_.alloc(fn / 2));
        CHK(powers(st, ptr<Fr>(tw_n_), n_ / 2, omega_, Fr::one()));
        CHK(powers(st, ptr<Fr>(twu_n_), n_ / 2, omega_, ru));
        CHK(powers(st, ptr<Fr>(twi_n_), n_ / 2, omega_inv_, ru));
        CHK(powers(st, ptr<Fr>(tw_4n_), n4_ / 2, omega4_, ru));
        CHK(powers(st, ptr<Fr>(twi_4n_), n4_ / 2, omega4_inv_, ru));
        if (env_int("APK_NTT_TWU", 0, 0, 1)) {     // measured neutral (profiles/r06_ntt_twiddles.txt): off
            struct { DevBuf* in; DevBuf* out; uint32_t count; } tabs[4] = {{&twu_n_, &twu_n_x_, n_ / 2}, {&twi_n_, &twi_n_x_, n_ / 2},
                                                                            {&tw_4n_, &tw_4n_x_, n4_ / 2}, {&twi_4n_, &twi_4n_x_, n4_ / 2}};
            for (auto& t : tabs) {
                CHK(t.out->alloc((size_t)t.count * sizeof(FeU<FRP>)));
                unpack_table_kernel<FRP><<<cdiv(t.count, 256), 256, 0, st>>>(ptr<Fr>(*t.in), ptr<FeU<FRP>>(*t.out), t.count);
                KCHK();
            }
            ntt_twu_ = true;
        }
        CHK(coset_pre_.alloc((size_t)(n_ + 4) * sizeof(Fr)));
        CHK(powers(st, ptr<Fr>(coset_pre_), n_ + 4, shift_, ru));
        CHK(coset_post_inv_.alloc(f4));
        CHK(powers(st, ptr<Fr>(coset_post_inv_), n4_, shift_inv_, n4_inv_ * ru));
        CHK(scales_.alloc(2 * sizeof(Fr)));
        {
            Fr sc[2] = {n_inv_ * ru, n4_inv_ * ru};
            HIPCHK(hipMemcpy(scales_.p, sc, sizeof sc, hipMemcpyHostToDevice));
        }
        CHK(x4_.alloc(f4));
        CHK(powers(st, ptr<Fr>(x4_), n4_, omega4_, shift_));
        // MSM tables
        DevBuf srs;
        CHK(srs.alloc((size_t)(n_ + 3) * sizeof(Aff)));
        HIPCHK(hipMemcpy(srs.p, d->srs_g1, (size_t)(n_ + 3) * sizeof(Aff), hipMemcpyHostToDevice));
        CHK(build_tables(st, ptr<Aff>(srs), n_ + 3, tab_can_));
        wires_lag_mode_ = env_int("APK_WIRES_LAGRANGE", -1, -1, 1);
        {
            const Aff* g = reinterpret_cast<const Aff*>(d->srs_g1);
            for (int k = 0; k < 3; k++) {
                Pt p = Pt::from_affine(g[n_ + k]);
                p.madd(g[k], /*negate=*/true);
                lag_dk_[k] = p.to_affine();
            }
        }
        if (d->srs_g1_lagrange || wires_lag_mode_ == 1) {
            // the extended Lagrange table NOW: the caller's Lagrange SRS (gnark's ProvingKey.KzgLagrange; BSB22 needs it) or a forced
            // Lagrange route - derived from the canonical SRS when not given (kzg.ToLagrangeG1 as an inverse FFT in the exponent)
            CHK(build_lagrange_table(st, ptr<Aff>(srs), d->srs_g1_lagrange));
        } else if (wires_lag_mode_ != 0) {
            // automatic route, nothing given: derived on first use (ensure_lagrange_table), from the SRS kept here
            CHK(srs_keep_.alloc((size_t)(n_ + 3) * sizeof(Aff)));
            HIPCHK(hipMemcpyAsync(srs_keep_.p, srs.p, (size_t)(n_ + 3) * sizeof(Aff), hipMemcpyDeviceToDevice, st));
        }
        HIPCHK(hipDeviceSynchronize());
        srs.release();
        CHK(set_sort_lds_limits());
        // proving slots
        int nslots = d->slots > 0 ? d->slots : 1;
        // 16 concurrently active streams are the optimum at 2^17 (round 6, sleeping waits: 12 -> 542, 16 -> 549, 20 -> 544, 24 -> 518
        // proofs/s; the collapse rounds 2-5 measured beyond 16 - 24: -25 %, 32: -40 % - was mostly the cgroup freezing a process
        // with more spinning proving threads than its 16-CPU quota); callers beyond the cap wait for a slot, which also hides
        // their host-side gaps - or, on small circuits, join a gang
        static const int max_slots = env_int("APK_MAX_SLOTS", 16, 1, 64);
        // Gangs: a context asked for more slots than it may run streams lets up to gang_cap_ proofs share a stream (gang.h).
        // APK_GANG: 1 = never, 2..4 = members per stream; default by size (choose_gang).  The operands of a merged launch must fit
        // the scan's 2^21 buckets and the sort's totals buffers.
        gang_cap_ = choose_gang((int)log_n_, nslots, max_slots);
        while (gang_cap_ > 1 && ((uint64_t)MSM_MAX_BATCH * gang_cap_ * NB_ > (1ull << 21) || MSM_MAX_BATCH * gang_cap_ > MSM_ARGS_MAX ||
                                 (part_cfg_.P && (uint64_t)MSM_MAX_BATCH * gang_cap_ * part_cfg_.P > (uint64_t)MSM_MAX_BATCH * MSM_PART_MAX)))
            gang_cap_--;
        ws_batch_ = (uint32_t)(MSM_MAX_BATCH * gang_cap_);
        const int max_streams = nslots > max_slots ? max_slots : nslots;
        if (nslots > max_slots * gang_cap_) nslots = max_slots * gang_cap_;
        many_slots_ = nslots > 2;
        for (int i = 0; i < max_streams; i++) {
            hipStream_t ps = nullptr;
            HIPCHK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
            stream_pool_.push_back(ps);
        }
        for (int i = 0; i < nslots; i++) {
            Slot* s = new Slot();
            s->index = slots_.size();
            slots_.push_back(s);
            CHK(alloc_slot(*s));
        }
        static const int gang_wait_us = env_int("APK_GANG_WAIT_US", 300, 0, 100000);
        gate_.configure(slots_.size(), (size_t)max_streams, gang_cap_, gang_wait_us);
        gang_launcher_ = [this](GangReq* const* reqs, int count) { gang_launch(reqs, count); };
        slots_[0]->own_stream = slots_[0]->stream = stream_pool_[0];      // (the trace setup runs on slot 0 before anybody can take it)
        const int trc = setup_trace(d);
        slots_[0]->own_stream = slots_[0]->stream = nullptr;
        return trc;
    }

    static Fr root_of_unity() {
        // primitive 2^ADICITY-th root of unity (generated constant, tools/gen_params.py)
        Fr a;
        for (int i = 0; i < Fr::N; i++) a.l[i] = FRP::root(i);
        return Fr::to_mont(a);
    }

    // ---- trace polynomials, permutation polynomials, their coset evaluations and the VK commitments ----------
    int setup_trace(const apk_circuit_desc* d);

    int get_vk(apk_vk* out) override {
        if (msm_only_) { set_error("MSM-only context has no verifying key"); return APK_ERR_STATE; }
        memset(out, 0, sizeof *out);
        uint8_t* dst[8 + APK_MAX_COMMITMENTS] = {out->ql, out->qr, out->qm, out->qo, out->qk, out->s[0], out->s[1], out->s[2], out->qcp[0], out->qcp[1]};
        for (uint32_t i = 0; i < 8 + nb_commit_; i++) memcpy(dst[i], &vk_pts_[i], sizeof(Aff));
        memcpy(out->size_inv, &n_inv_, sizeof(Fr));
        memcpy(out->generator, &omega_, sizeof(Fr));
        memcpy(out->coset_shift, &shift_, sizeof(Fr));
        return APK_OK;
    }

    // ---------------------------------------------------------------------------------------------- primitives
    int msm(int basis, const void* scalars, uint64_t len, bool on_device, void* out) override {
        HIPCHK(hipSetDevice(device_));
        SlotGuard g(this);
        Slot& s = *g.s;
        if (basis) (void)ensure_lagrange_table(s.stream);      // the automatic mode derives the table on first use
        MsmTables& T = basis ? tab_lag_ : tab_can_;
        if (basis ? !lagrange_ready() : !T.built) { set_error("context has no %s SRS", basis ? "Lagrange" : "canonical"); return APK_ERR_STATE; }
        if (len > T.n_bases) { set_error("msm length %llu exceeds SRS size %u", (unsigned long long)len, T.n_bases); return APK_ERR_ARG; }
        const void* dsc = scalars;
        if (!on_device) {
            HIPCHK(hipMemcpyAsync(s.scratch_in.p, scalars, len * sizeof(Fr), hipMemcpyHostToDevice, s.stream));
            dsc = s.scratch_in.p;
        }
        MsmBatchArgs a{};
        a.batch = 1; a.scalars[0] = dsc; a.len[0] = (uint32_t)len; a.offset[0] = 0; a.plain = T.plain ? 1u : 0u;
        CHK(run_msm(s, T, a, reinterpret_cast<Aff*>(s.h_pinned)));
        CHK(sync_results(s));
        memcpy(out, s.h_pinned, sizeof(Aff));
        return APK_OK;
    }

    // `count` partial MSMs in one batch: sum over scalars[b][0 .. len) * SRS[offset + i]; scalars in device memory
    int msm_batch(int basis, uint32_t count, const void* const* d_scalars, const uint64_t* offsets, const uint64_t* lens, void* out) override {
        HIPCHK(hipSetDevice(device_));
        if (count == 0 || count > MSM_MAX_BATCH) { set_error("msm batch of %u (1..%d)", count, MSM_MAX_BATCH); return APK_ERR_ARG; }
        Slot* own = hook_slot();                       // called from inside this thread's commit hook: use the prover's slot
        bool mine = false;
        for (Slot* t : slots_) mine |= (t == own);
        if (!mine) own = nullptr;
        struct MaybeGuard { CurveBackend* b; Slot* s; bool owned; ~MaybeGuard() { if (owned) b->release(s); } };
        MaybeGuard g{this, own ? own : acquire(), own == nullptr};
        Slot& s = *g.s;
        if (basis) (void)ensure_lagrange_table(s.stream);
        MsmTables& T = basis ? tab_lag_ : tab_can_;
        if (basis ? !lagrange_ready() : !T.built) { set_error("context has no %s SRS", basis ? "Lagrange" : "canonical"); return APK_ERR_STATE; }
        MsmBatchArgs a{};
        a.batch = count;
        a.plain = T.plain ? 1u : 0u;
        for (uint32_t b = 0; b < count; b++) {
            hipPointerAttribute_t at{};
            if (d_scalars[b] && (hipPointerGetAttributes(&at, d_scalars[b]) != hipSuccess || at.type != hipMemoryTypeDevice)) {
                (void)hipGetLastError();
                set_error("msm batch: scalars[%u] is not device memory", b);
                return APK_ERR_ARG;
            }
            if (lens[b] > T.n_bases || offsets[b] > T.n_bases - lens[b] || !d_scalars[b]) {   /* no uint64 wrap-around */ set_error("msm batch: range [%llu, +%llu) outside the %u bases", (unsigned long long)offsets[b], (unsigned long long)lens[b], T.n_bases); return APK_ERR_ARG; }
            a.scalars[b] = d_scalars[b]; a.len[b] = (uint32_t)lens[b]; a.offset[b] = (uint32_t)offsets[b];
        }
        CHK(run_msm(s, T, a, reinterpret_cast<Aff*>(s.h_pinned)));
        CHK(wait_stream(s));
        const Pt* gsum = reinterpret_cast<const Pt*>(reinterpret_cast<const uint8_t*>(s.h_pinned) + PIN_XYZZ);
        for (uint32_t b = 0; b < count; b++) { const Aff r = gsum[b].to_affine(); memcpy(reinterpret_cast<uint8_t*>(out) + b * sizeof(Aff), &r, sizeof r); }
        s.pending_pts = 0;
        return APK_OK;
    }
    int set_commit_hook(apk_commit_hook fn, void* user) override { hook_ = fn; hook_user_ = user; return APK_OK; }
    int set_wire_hook(apk_wire_hook fn, void* user) override { wire_hook_ = fn; wire_hook_user_ = user; return APK_OK; }
    int set_subcoset(int k, int G, apk_gather_hook fn, void* user) override {
        if (msm_only_) { set_error("MSM-only context has no quotient to split"); return APK_ERR_STATE; }
        HIPCHK(hipSetDevice(device_));
        if (G <= 1 || !fn) { sc_.G = 1; sc_.glog = 0; sc_.hook = nullptr; sc_.user = nullptr; return APK_OK; }
        if ((G != 2 && G != 4 && G != 8) || k < 0 || k >= G) { set_error("sub-coset split: rank %d of %d (2, 4 or 8 ranks)", k, G); return APK_ERR_ARG; }
        if (!qk_direct_) { set_error("sub-coset split needs Qk completed inside the quotient kernel (at most %d written rows)", QK_INJECT_MAX); return APK_ERR_STATE; }
        if ((n4_ >> (G == 2 ? 1 : G == 4 ? 2 : 3)) < 16u) { set_error("sub-coset split: the domain is too small"); return APK_ERR_STATE; }
        for (hipStream_t ps : stream_pool_) HIPCHK(hipStreamSynchronize(ps));
        const int glog = G == 2 ? 1 : G == 4 ? 2 : 3;
        const uint32_t m = n4_ >> glog;
        const Fr ru = fr_u64(32);   // x R -> x R'
        hipStream_t st = nullptr;
        const Fr wk = Fr::pow_u64(omega4_, (uint64_t)k), wk_inv = Fr::pow_u64(omega4_inv_, (uint64_t)k);
        CHK(sc_.pre[0].alloc((size_t)(n_ + 4) * sizeof(Fr)));
        CHK(powers(st, ptr<Fr>(sc_.pre[0]), n_ + 4, shift_ * wk, ru));
        CHK(sc_.pre[1].alloc((size_t)(n_ + 4) * sizeof(Fr)));
        CHK(powers(st, ptr<Fr>(sc_.pre[1]), n_ + 4, shift_ * Fr::pow_u64(omega4_, (uint64_t)((k + 4) % 8)), ru));
        CHK(sc_.post.alloc((size_t)m * sizeof(Fr)));
        CHK(powers(st, ptr<Fr>(sc_.post), m, wk_inv, ru));
        const Fr rho_inv = Fr::pow_u64(omega4_inv_, (uint64_t)m);
        Fr cur = ru;
        for (int e = 0; e < 8; e++) { sc_.rho_inv[e] = cur; cur = cur * rho_inv; }
        for (Slot* s : slots_) CHK(s->sc_gather.alloc((size_t)n4_ * sizeof(Fr)));
        HIPCHK(hipDeviceSynchronize());
        sc_.k = k; sc_.G = G; sc_.glog = glog; sc_.hook = fn; sc_.user = user;
        return APK_OK;
    }
    int device_ordinal() override { return device_; }
    int msm_window() override { return c_; }
    uint64_t domain_size() override { return msm_only_ ? 0 : n_; }
    // 4n coset evaluations of a canonical polynomial, device memory in and out; COMPLETE when the call returns.  Inside a hook of
    // this context it runs on the prover's own stream (see msm_batch), otherwise on a free slot.
    int coset_ntt_dev(const void* d_in, uint64_t len, void* d_out) override {
        if (msm_only_) { set_error("MSM-only context has no NTT domain"); return APK_ERR_STATE; }
        if (!d_in || !d_out || len == 0 || len > n4_) { set_error("coset ntt: 1..4n coefficients"); return APK_ERR_ARG; }
        HIPCHK(hipSetDevice(device_));
        Slot* own = hook_slot();
        bool mine = false;
        for (Slot* t : slots_) mine |= (t == own);
        if (!mine) own = nullptr;
        struct MaybeGuard { CurveBackend* b; Slot* s; bool owned; ~MaybeGuard() { if (owned) b->release(s); } };
        MaybeGuard g{this, own ? own : acquire(), own == nullptr};
        CHK(coset_ntt_4n(g.s->stream, reinterpret_cast<const Fr*>(d_in), (uint32_t)len, reinterpret_cast<Fr*>(d_out)));
        HIPCHK(hipStreamSynchronize(g.s->stream));
        return APK_OK;
    }
    // device-to-device copy that has COMPLETED when the call returns (hipMemcpy D2D returns early, and the proving streams are
    // non-blocking streams: they do not order themselves behind the null stream)
    int dev_copy(void* dd, const void* ss, size_t bytes) override {
        HIPCHK(hipSetDevice(device_));
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, dd) != hipSuccess || at.type != hipMemoryTypeDevice) { (void)hipGetLastError(); set_error("apk_device_copy: destination is not device memory"); return APK_ERR_ARG; }
        if (hipPointerGetAttributes(&at, ss) != hipSuccess || at.type != hipMemoryTypeDevice) { (void)hipGetLastError(); set_error("apk_device_copy: source is not device memory"); return APK_ERR_ARG; }
        Slot* own = hook_slot();                       // the prover's stream only when the hook belongs to THIS context
        bool mine = false;
        for (Slot* t : slots_) mine |= (t == own);
        hipStream_t st = mine ? own->stream : nullptr;
        HIPCHK(hipMemcpyAsync(dd, ss, bytes, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        return APK_OK;
    }

    int ntt(int which, int inverse, int coset, void* data) override {
        if (msm_only_) { set_error("MSM-only context has no NTT domain"); return APK_ERR_STATE; }
        HIPCHK(hipSetDevice(device_));
        SlotGuard g(this);
        Slot& s = *g.s;
        const uint32_t N = which ? n4_ : n_;
        if (coset && !which) { set_error("coset transforms are provided on the 4n domain"); return APK_ERR_ARG; }
        Fr* din = ptr<Fr>(s.scratch_in);
        Fr* dout = ptr<Fr>(s.quot);
        HIPCHK(hipMemcpyAsync(din, data, (size_t)N * sizeof(Fr), hipMemcpyHostToDevice, s.stream));
        if (!coset) {
            CHK(run_ntt(s.stream, which, inverse != 0, din, dout, N, N, nullptr, nullptr,
                        inverse ? ptr<Fr>(scales_) + (which ? 1 : 0) : nullptr));
        } else if (!inverse) {
            // forward coset needs shift^i for all i < 4n: build it in hcan for this call
            CHK(powers(s.stream, ptr<Fr>(s.hcan), N, shift_, fr_u64(32)));   // R' form, like every NTT table
            CHK(run_ntt(s.stream, 1, false, din, dout, N, N, ptr<Fr>(s.hcan), nullptr, nullptr));
        } else {
            CHK(run_ntt(s.stream, 1, true, din, dout, N, N, nullptr, ptr<Fr>(coset_post_inv_), nullptr));
        }
        HIPCHK(hipMemcpyAsync(data, dout, (size_t)N * sizeof(Fr), hipMemcpyDeviceToHost, s.stream));
        HIPCHK(hipStreamSynchronize(s.stream));
        return APK_OK;
    }

    int dev_alloc(size_t bytes, void** p) override { HIPCHK(hipSetDevice(device_)); HIPCHK(hipMalloc(p, bytes)); return APK_OK; }
    int dev_free(void* p) override { HIPCHK(hipSetDevice(device_)); HIPCHK(hipFree(p)); return APK_OK; }
    int dev_upload(void* dd, const void* s, size_t bytes) override { HIPCHK(hipSetDevice(device_)); HIPCHK(hipMemcpy(dd, s, bytes, hipMemcpyHostToDevice)); return APK_OK; }
    int dev_download(void* dd, const void* s, size_t bytes) override { HIPCHK(hipSetDevice(device_)); HIPCHK(hipMemcpy(dd, s, bytes, hipMemcpyDeviceToHost)); return APK_OK; }
    int stats_enable(int en) override { stats_on_ = en != 0; return APK_OK; }
    int stats_read(apk_stats* out, int reset) override {
        std::lock_guard<std::mutex> g(stats_mu_);
        *out = stats_;
        if (reset) stats_ = apk_stats{};
        return APK_OK;
    }

    // ---------------------------------------------------------------------------------------------- helpers
    int scan_inplace(hipStream_t st, Fr* data, uint32_t count, bool rev, bool mul_op, Fr* tot, Fr* out, int shift) {
        const uint32_t nb = cdiv(count, SCAN_BLOCK);
        if (mul_op) {
            CHK((klaunch<ScanBlockK<FRP, OpMul>, POLY_THREADS>(st, nb, POLY_THREADS, 0, data, count, rev, tot)));
            CHK((klaunch<ScanTotalsK<FRP, OpMul>, POLY_THREADS>(st, 1, POLY_THREADS, 0, tot, nb)));
            CHK((klaunch<ScanApplyK<FRP, OpMul>, POLY_THREADS>(st, cdiv(count, POLY_THREADS), POLY_THREADS, 0, (const Fr*)data, count, rev, (const Fr*)tot, out, shift)));
        } else {
            CHK((klaunch<ScanBlockK<FRP, OpAdd>, POLY_THREADS>(st, nb, POLY_THREADS, 0, data, count, rev, tot)));
            CHK((klaunch<ScanTotalsK<FRP, OpAdd>, POLY_THREADS>(st, 1, POLY_THREADS, 0, tot, nb)));
            CHK((klaunch<ScanApplyK<FRP, OpAdd>, POLY_THREADS>(st, cdiv(count, POLY_THREADS), POLY_THREADS, 0, (const Fr*)data, count, rev, (const Fr*)tot, out, shift)));
        }
        return APK_OK;
    }

    // q = (f - f(z)) / (X - z) for f of `len` coefficients; pw = z^i, pwi = z^-i tables
    int kzg_quotient(Slot& s, const Fr* f, uint32_t len, const Fr* pw, const Fr* pwi, bool z_is_zero, Fr* q) {
        hipStream_t st = s.stream;
        if (z_is_zero) {
            CHK((klaunch<ShiftDownK<FRP>, 256>(st, cdiv(len, 256), 256, 0, f, len, q)));
            return APK_OK;
        }
        Fr* t = ptr<Fr>(s.tmp);
        CHK((klaunch<MulK<FRP>, 256>(st, cdiv(len, 256), 256, 0, t, f, pw, len)));
        CHK(scan_inplace(st, t, len, true, false, ptr<Fr>(s.scan_tot), t, 0));
        CHK((klaunch<DivFinishK<FRP>, POLY_THREADS>(st, cdiv(len, 256), 256, 0, (const Fr*)t, pwi, len, q)));
        return APK_OK;
    }

    int eval_many(Slot& s, const EvalArgs<FRP>& ea, const Fr* pw, Fr* h_out) {
        hipStream_t st = s.stream;
        uint32_t maxlen = 0;
        for (int i = 0; i < ea.count; i++) if (ea.len[i] > maxlen) maxlen = ea.len[i];
        const uint32_t nblocks = cdiv(maxlen, EVAL_BLOCK);
        CHK((klaunch<EvalPartialK<FRP>, POLY_THREADS>(st, dim3(nblocks, ea.count), POLY_THREADS, 0, ea, pw, nblocks, ptr<Fr>(s.eval_partial))));
        // (h_out lies in the slot's pinned buffer: with the zero-copy view the kernel writes the values there itself)
        const bool direct = s.d_pinned && reinterpret_cast<uint8_t*>(h_out) >= reinterpret_cast<uint8_t*>(s.h_pinned) &&
                            reinterpret_cast<uint8_t*>(h_out) + ea.count * sizeof(Fr) <= reinterpret_cast<uint8_t*>(s.h_pinned) + PIN_BYTES;
        Fr* const ev_out = direct ? reinterpret_cast<Fr*>(s.d_pinned + (reinterpret_cast<uint8_t*>(h_out) - reinterpret_cast<uint8_t*>(s.h_pinned))) : ptr<Fr>(s.eval_result);
        CHK((klaunch<EvalFinalK<FRP>, POLY_THREADS>(st, ea.count, POLY_THREADS, 0, (const Fr*)ptr<Fr>(s.eval_partial), nblocks, ev_out)));
        if (!direct) HIPCHK(hipMemcpyAsync(h_out, s.eval_result.p, ea.count * sizeof(Fr), hipMemcpyDeviceToHost, st));
        return APK_OK;
    }

    static void hash_challenge(const char* name, const uint8_t* prev, const std::vector<const uint8_t*>& parts,
                               const std::vector<size_t>& lens, uint8_t out[32]) {
        Sha256 h;
        h.update(name, strlen(name));
        if (prev) h.update(prev, 32);
        for (size_t i = 0; i < parts.size(); i++) h.update(parts[i], lens[i]);
        h.final(out);
    }

    // gnark fr.Hash(msg, "BSB22-Plonk", 1) = expand_msg_xmd(sha256, 48 bytes) mod r, as the verifier
    // recomputes it (templateLogicSigBN254.go:386-397)
    static Fr hash_fr_point(const uint8_t* p, size_t len) {
        static const uint8_t dst_prime[12] = {'B', 'S', 'B', '2', '2', '-', 'P', 'l', 'o', 'n', 'k', 0x0b};
        uint8_t b0[32], b1[32], b2[32], zeros[64] = {0};
        const uint8_t lib[3] = {0x00, 0x30, 0x00};
        Sha256 h;
        h.update(zeros, 64); h.update(p, len); h.update(lib, 3); h.update(dst_prime, 12); h.final(b0);
        uint8_t one = 1, two = 2;
        h.reset(); h.update(b0, 32); h.update(&one, 1); h.update(dst_prime, 12); h.final(b1);
        uint8_t x[32];
        for (int i = 0; i < 32; i++) x[i] = b0[i] ^ b1[i];
        h.reset(); h.update(x, 32); h.update(&two, 1); h.update(dst_prime, 12); h.final(b2);
        uint8_t lo[32] = {0};
        memcpy(lo + 16, b2, 16);
        Fr two128 = Fr::zero();
        two128.l[4] = 1;
        two128 = Fr::to_mont(two128);
        return fr_from_be(b1) * two128 + fr_from_be(lo);
    }

    static void store_pt(uint8_t* slot, const Aff& p) { memset(slot, 0, APK_G1_MAX_BYTES); memcpy(slot, &p, sizeof(Aff)); }

    int prove(const void* L, const void* R, const void* O, bool on_device, const void* pub, const void* blinding,
              const void* const* pi2, apk_proof* out) override;
};

// =====================================================================================================================
template <class FRP, class FPP, int CURVE_ID>
int CurveBackend<FRP, FPP, CURVE_ID>::setup_trace(const apk_circuit_desc* d) {
    Slot& s = *slots_[0];
    hipStream_t st = s.stream;
    const size_t fn = (size_t)n_ * sizeof(Fr), f4 = (size_t)n4_ * sizeof(Fr);
    // L_0 on the coset: (x^n - 1) / (n (x - 1)).  x^n - 1 = 1/zh_inv[i&3]; computed on the host side as a
    // canonical polynomial (1/n) * sum X^i and pushed through the coset NTT - no per-point inversion.
    {
        std::vector<Fr> l0(n_, n_inv_);
        HIPCHK(hipMemcpyAsync(s.tmp.p, l0.data(), fn, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        CHK(l0_4_.alloc(f4));
        CHK(coset_ntt_4n(st, ptr<Fr>(s.tmp), n_, ptr<Fr>(l0_4_)));
    }
    // selector columns: Lagrange -> canonical -> coset
    const void* cols[5] = {d->ql, d->qr, d->qm, d->qo, d->qk};
    DevBuf* can[5] = {&ql_c_, &qr_c_, &qm_c_, &qo_c_, &qk_c_};
    DevBuf* cos[4] = {&eql_, &eqr_, &eqm_, &eqo_};
    for (int i = 0; i < 5; i++) {
        CHK(can[i]->alloc(fn));
        HIPCHK(hipMemcpyAsync(s.wl.p, cols[i], fn, hipMemcpyHostToDevice, st));
        if (i == 4) { CHK(qk_lag_trace_.alloc(fn)); HIPCHK(hipMemcpyAsync(qk_lag_trace_.p, s.wl.p, fn, hipMemcpyDeviceToDevice, st)); }
        CHK(inv_ntt_n(st, ptr<Fr>(s.wl), ptr<Fr>(*can[i])));
        if (i < 4) {
            CHK(cos[i]->alloc(f4));
            CHK(coset_ntt_4n(st, ptr<Fr>(*can[i]), n_, ptr<Fr>(*cos[i])));
            // into the quotient kernel's radix (kernels_poly.h): 32 q, and 1024 q for qm
            scale_kernel<FRP><<<cdiv(n4_, POLY_THREADS), POLY_THREADS, 0, st>>>(ptr<Fr>(*cos[i]), fr_u64(i == 2 ? 1u << 10 : 1u << 5), n4_); KCHK();
        }
        HIPCHK(hipStreamSynchronize(st));
    }
    // rows of Qk the prover writes per proof: public inputs 0..nb_public-1 and the commitment rows.  With few of them the
    // completed column never goes through iNTT + coset NTT again: the quotient kernel adds delta_j * L_row_j(x) instead.
    inj_row_.clear(); inj_trace_val_.clear();
    for (uint32_t i = 0; i < nb_public_; i++) inj_row_.push_back(i);
    for (uint32_t k = 0; k < nb_commit_; k++) inj_row_.push_back(nb_public_ + cci_[k]);
    qk_direct_ = inj_row_.size() <= (size_t)QK_INJECT_MAX && !getenv("APK_QK_NTT");
    if (qk_direct_) {
        CHK(eqk_trace_.alloc(f4));
        CHK(coset_ntt_4n(st, ptr<Fr>(qk_c_), n_, ptr<Fr>(eqk_trace_)));
        for (size_t j = 0; j < inj_row_.size(); j++) {
            const uint32_t row = inj_row_[j];
            if (row >= n_) { set_error("Qk row %u out of range", row); return APK_ERR_ARG; }
            inj_trace_val_.push_back(reinterpret_cast<const Fr*>(d->qk)[row]);
            if (row == 0) continue;   // L_0 is l0_4_
            // L_row(X) = (1/n) sum_i omega^(-row i) X^i
            CHK(inj_tab_[j].alloc(f4));
            CHK(powers(st, ptr<Fr>(s.tmp), n_, Fr::pow_u64(omega_inv_, row), n_inv_));
            CHK(coset_ntt_4n(st, ptr<Fr>(s.tmp), n_, ptr<Fr>(inj_tab_[j])));
        }
        HIPCHK(hipStreamSynchronize(st));
    }
    for (uint32_t k = 0; k < nb_commit_; k++) {
        CHK(qcp_c_[k].alloc(fn)); CHK(eqcp_[k].alloc(f4));
        HIPCHK(hipMemcpyAsync(s.wl.p, d->qcp[k], fn, hipMemcpyHostToDevice, st));
        CHK(inv_ntt_n(st, ptr<Fr>(s.wl), ptr<Fr>(qcp_c_[k])));
        CHK(coset_ntt_4n(st, ptr<Fr>(qcp_c_[k]), n_, ptr<Fr>(eqcp_[k])));
        scale_kernel<FRP><<<cdiv(n4_, POLY_THREADS), POLY_THREADS, 0, st>>>(ptr<Fr>(eqcp_[k]), fr_u64(1u << 5), n4_); KCHK();
        HIPCHK(hipStreamSynchronize(st));
    }
    // permutation polynomials: S_j[i] = u^(p/n) * omega^(p mod n), p = perm[j n + i]   (gnark trace.S)
    {
        std::vector<Fr> wpow(n_);
        {
            std::vector<Fr> half(n_ / 2);
            HIPCHK(hipMemcpy(half.data(), tw_n_.p, fn / 2, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < n_ / 2; i++) { wpow[i] = half[i]; wpow[i + n_ / 2] = Fr::neg(half[i]); }
        }
        Fr us[3] = {Fr::one(), shift_, shift_ * shift_};
        std::vector<Fr> col(n_);
        for (int j = 0; j < 3; j++) {
            for (uint32_t i = 0; i < n_; i++) {
                int64_t p = d->perm[(size_t)j * n_ + i];
                if (p < 0 || p >= (int64_t)3 * n_) { set_error("perm[%zu]=%lld out of range", (size_t)j * n_ + i, (long long)p); return APK_ERR_ARG; }
                uint32_t blk = (uint32_t)(p / n_), pos = (uint32_t)(p % n_);
                col[i] = blk == 0 ? wpow[pos] : us[blk] * wpow[pos];
            }
            CHK(s_lag_[j].alloc(fn)); CHK(s_c_[j].alloc(fn)); CHK(es_[j].alloc(f4));
            HIPCHK(hipMemcpy(s_lag_[j].p, col.data(), fn, hipMemcpyHostToDevice));
            CHK(inv_ntt_n(st, ptr<Fr>(s_lag_[j]), ptr<Fr>(s_c_[j])));
            CHK(coset_ntt_4n(st, ptr<Fr>(s_c_[j]), n_, ptr<Fr>(es_[j])));
            HIPCHK(hipStreamSynchronize(st));
        }
    }
    // VK commitments: the 8+k MSMs of plonk.Setup (setup/setup.go:107,149), canonical basis
    const Fr* polys[8 + APK_MAX_COMMITMENTS] = {ptr<Fr>(ql_c_), ptr<Fr>(qr_c_), ptr<Fr>(qm_c_), ptr<Fr>(qo_c_), ptr<Fr>(qk_c_),
                                                ptr<Fr>(s_c_[0]), ptr<Fr>(s_c_[1]), ptr<Fr>(s_c_[2]), ptr<Fr>(qcp_c_[0]), ptr<Fr>(qcp_c_[1])};
    const uint32_t total = 8 + nb_commit_;
    for (uint32_t base = 0; base < total; base += MSM_MAX_BATCH) {
        MsmBatchArgs a{};
        a.batch = total - base < MSM_MAX_BATCH ? total - base : MSM_MAX_BATCH;
        for (uint32_t b = 0; b < a.batch; b++) { a.scalars[b] = polys[base + b]; a.len[b] = n_; a.offset[b] = 0; }
        CHK(run_msm(s, tab_can_, a, reinterpret_cast<Aff*>(s.h_pinned)));
        CHK(sync_results(s));
        memcpy(&vk_pts_[base], s.h_pinned, a.batch * sizeof(Aff));
    }
    {   // the five of them that enter every proof's [lin] with a fresh coefficient: fixed-base tables on the host (a few ms)
        const Aff fixed[5] = {vk_pts_[0], vk_pts_[1], vk_pts_[2], vk_pts_[3], vk_pts_[7]};
        vk_fixed_.build(fixed, 5);
    }
    return APK_OK;
}

// =====================================================================================================================
template <class FRP, class FPP, int CURVE_ID>
int CurveBackend<FRP, FPP, CURVE_ID>::prove(const void* L, const void* R, const void* O, bool on_device, const void* pub,
                                            const void* blinding, const void* const* pi2, apk_proof* out) {
    if (msm_only_) { set_error("MSM-only context cannot prove"); return APK_ERR_STATE; }
    HIPCHK(hipSetDevice(device_));
    auto t_start = std::chrono::steady_clock::now();
    if (!L || !R || !O || !blinding || !out || (nb_public_ && !pub) || (nb_commit_ && !pi2)) { set_error("null argument to apk_prove"); return APK_ERR_ARG; }
    const size_t fn = (size_t)n_ * sizeof(Fr);
    // host inputs: on their way to the device on a copy stream BEFORE this caller queues for a proving slot (InputSet)
    InputGuard in(this);
    if (!on_device) {
        CHK(in.take());
        const void* src[3] = {L, R, O};
        for (int j = 0; j < 3; j++) HIPCHK(hipMemcpyAsync(in.set->w[j].p, src[j], fn, hipMemcpyHostToDevice, in.set->copy));
        for (uint32_t k = 0; k < nb_commit_; k++) {
            if (!pi2[k]) { set_error("null BSB22 column"); return APK_ERR_ARG; }
            HIPCHK(hipMemcpyAsync(in.set->pi2[k].p, pi2[k], fn, hipMemcpyHostToDevice, in.set->copy));
        }
        HIPCHK(hipEventRecord(in.set->ready, in.set->copy));
    }
    MemberGuard guard(this);
    Slot& s = *guard.s;
    hipStream_t st = s.stream;
    const uint32_t n = n_;
    const Fr* bl = reinterpret_cast<const Fr*>(blinding);
    const Fr* pubv = reinterpret_cast<const Fr*>(pub);
    Aff* hp = reinterpret_cast<Aff*>(s.h_pinned);
    Fr* hfr = reinterpret_cast<Fr*>(reinterpret_cast<uint8_t*>(s.h_pinned) + PIN_FR);
    memset(out, 0, sizeof *out);
    out->curve = CURVE_ID;
    out->nb_commitments = nb_commit_;

    // ---------------- round 1: wire polynomials, BSB22 commitments, completed Qk ----------------------------
    const Fr* dL = reinterpret_cast<const Fr*>(L);
    const Fr* dR = reinterpret_cast<const Fr*>(R);
    const Fr* dO = reinterpret_cast<const Fr*>(O);
    if (!on_device) {
        HIPCHK(hipStreamWaitEvent(st, in.set->ready, 0));
        dL = ptr<Fr>(in.set->w[0]); dR = ptr<Fr>(in.set->w[1]); dO = ptr<Fr>(in.set->w[2]);
        path(P_HOST_INPUTS);
    }
    Aff bsb[APK_MAX_COMMITMENTS];
    Fr cval[APK_MAX_COMMITMENTS];
    uint8_t bsb_bytes[APK_MAX_COMMITMENTS][2 * FPB];
    for (uint32_t k = 0; k < nb_commit_; k++) {
        // kzg.Commit(pi2, Lagrange SRS) then hash_to_field (templateLogicSigBN254.go:386-397)
        // (the committed column is only read: the caller's device buffer, or the input set's copy of the caller's host buffer)
        const Fr* col = on_device ? reinterpret_cast<const Fr*>(pi2[k]) : ptr<Fr>(in.set->pi2[k]);
        MsmBatchArgs a{};
        a.batch = 1; a.scalars[0] = col; a.len[0] = n; a.offset[0] = 0;
        CHK(commit(s, tab_lag_, 1, a, hp));
        CHK(inv_ntt_n(st, col, ptr<Fr>(s.pi2_can[k])));
        {
            const Fr* cin = ptr<Fr>(s.pi2_can[k]); Fr* eout = ptr<Fr>(s.epi2[k]);
            CHK(coset_eval(st, 1, &cin, &n, &eout));
        }
        CHK(sync_results(s));
        bsb[k] = hp[0];
        g1_raw_bytes(bsb[k], bsb_bytes[k]);
        cval[k] = hash_fr_point(bsb_bytes[k], 2 * FPB);
        store_pt(out->bsb22[k], bsb[k]);
    }
    Fr* canon[3] = {ptr<Fr>(s.cl), ptr<Fr>(s.cr), ptr<Fr>(s.co)};
    const Fr* wires[3] = {dL, dR, dO};
    // [L][R][O] over the Lagrange SRS (gnark's way) or over the canonical one: the same group elements, so the choice is about
    // work only.  Over the Lagrange SRS the scalars are the witness values - in real circuits mostly 0, 1 or small, i.e. few
    // non-zero digits on the plain table - over the canonical SRS they are the uniform coefficients of the blinded polynomial.
    // auto: the Lagrange route when the wires' non-zero digits (msm_density_kernel, every 16th proof of the context; the first
    // proof waits for its own count) are below 90 % of what uniform scalars have.  Never with a commit hook installed (the
    // multi-GPU schedules deal canonical batches).
    constexpr uint32_t DENSITY_UNKNOWN = 0xffffffffu;
    volatile uint32_t* h_density = reinterpret_cast<volatile uint32_t*>(reinterpret_cast<uint8_t*>(s.h_pinned) + PIN_DENSITY);
    bool use_lag = false, measuring = false;
    if (lagrange_possible() && !hook_ && wires_lag_mode_ != 0) {
        const uint32_t seq = proof_seq_.fetch_add(1, std::memory_order_relaxed);
        uint32_t pm = wire_density_pm_.load(std::memory_order_relaxed);
        if (wires_lag_mode_ == 1) use_lag = true;
        else {
            measuring = pm == DENSITY_UNKNOWN || (seq & 15u) == 0;
            if (measuring) {
                uint32_t* d_cnt = ptr<uint32_t>(s.tail_flag) + 2;
                HIPCHK(hipMemsetAsync(d_cnt, 0, 4, st));
                MsmBatchArgs da{};
                da.batch = 3;
                for (int j = 0; j < 3; j++) { da.scalars[j] = wires[j]; da.len[j] = n; }
                msm_density_kernel<FRP><<<dim3(cdiv(n, 256) < 512 ? cdiv(n, 256) : 512, 3), 256, 0, st>>>(da, win_, d_cnt); KCHK();
                HIPCHK(hipMemcpyAsync(const_cast<uint32_t*>(h_density), d_cnt, 4, hipMemcpyDeviceToHost, st));
            }
            auto per_mille = [&]() { return (uint32_t)((uint64_t)*h_density * 1000u / ((uint64_t)3 * n * W_)); };
            if (pm == DENSITY_UNKNOWN) {        // the context's first proof: wait for the count
                HIPCHK(hipStreamSynchronize(st));
                pm = per_mille();
                wire_density_pm_.store(pm, std::memory_order_relaxed);
                measuring = false;
            }
            use_lag = pm < 900u;
        }
        // (automatic mode without a caller's Lagrange SRS: the table is derived now, once, on this proof's stream - or not at all)
        if (use_lag && !lagrange_ready()) use_lag = ensure_lagrange_table(st);
    }
    Fr* lagv[3] = {ptr<Fr>(s.wl), ptr<Fr>(s.wr), ptr<Fr>(s.wo)};
    if (use_lag)
        for (int j = 0; j < 3; j++) HIPCHK(hipMemcpyAsync(lagv[j], wires[j], fn, hipMemcpyDeviceToDevice, st));
    // (no zero-fill of the tails: blind_kernel assigns the coefficients n .. n+d-1 and nothing reads beyond them)
    {
        const uint32_t lens[3] = {n, n, n};
        CHK(run_ntt_batch(st, 0, true, 3, wires, canon, lens, n, nullptr, nullptr, ptr<Fr>(scales_)));
    }
    {
        Blind3<FRP> b3{};
        for (int j = 0; j < 3; j++) {
            b3.p[j] = canon[j]; b3.b[j].v[0] = bl[2 * j]; b3.b[j].v[1] = bl[2 * j + 1];
            b3.lag[j] = use_lag ? lagv[j] : nullptr;     // the blinding scalars behind the n witness values: the scalars of D_0, D_1
        }
        CHK((klaunch<Blind3K<FRP>, 256>(st, 3, 64, 0, b3, n, 2)));
    }
    const int fill = tail_fill(s);
    if (fill) path(P_TAIL_FILL);
    {
        MsmBatchArgs a{};
        a.batch = 3;
        for (int j = 0; j < 3; j++) { a.scalars[j] = use_lag ? lagv[j] : canon[j]; a.len[j] = n + 2; a.offset[j] = 0; }
        a.plain = use_lag ? 1u : 0u;
        if (use_lag) path(P_LAGRANGE_WIRES);
        s.mark_acc = fill;
        const int rc = use_lag ? commit(s, tab_lag_, 1, a, hp) : commit(s, tab_can_, 0, a, hp);
        s.mark_acc = 0;
        CHK(rc);
    }
    // completed Qk: public inputs and commitment values written into the Lagrange column, then iNTT
    if (!qk_direct_) {
        HIPCHK(hipMemcpyAsync(s.qk_lag.p, qk_lag_trace_.p, fn, hipMemcpyDeviceToDevice, st));
        if (nb_public_) HIPCHK(hipMemcpyAsync(s.qk_lag.p, pub, (size_t)nb_public_ * sizeof(Fr), hipMemcpyHostToDevice, st));
        for (uint32_t k = 0; k < nb_commit_; k++)
            HIPCHK(hipMemcpyAsync(ptr<Fr>(s.qk_lag) + nb_public_ + cci_[k], &cval[k], sizeof(Fr), hipMemcpyHostToDevice, st));
        CHK(inv_ntt_n(st, ptr<Fr>(s.qk_lag), ptr<Fr>(s.qk_can)));
        CHK(coset_ntt_4n(st, ptr<Fr>(s.qk_can), n, ptr<Fr>(s.eqk)));
    }
    {
        Fr* ev[3] = {ptr<Fr>(s.el), ptr<Fr>(s.er), ptr<Fr>(s.eo)};
        const uint32_t lens[3] = {n + 2, n + 2, n + 2};
        if (wire_hook_ && !sc_.on()) {
            // per-wire transforms dealt to other GPUs (csrc/comm.cpp): the hook returns with the three
            // evaluation vectors complete in this context's memory
            HIPCHK(hipStreamSynchronize(st));
            const void* cin[3] = {canon[0], canon[1], canon[2]};
            void* eout[3] = {ev[0], ev[1], ev[2]};
            hook_slot() = &s;
            const int rc = wire_hook_(wire_hook_user_, 3, cin, lens, eout);
            hook_slot() = nullptr;
            if (rc != APK_OK) { set_error("wire hook failed with %d", rc); return rc == APK_ERR_ARG ? APK_ERR_ARG : APK_ERR_STATE; }
        } else if (fill) {
            CHK(side_begin(s));
            CHK(run_ntt_batch(s.side, 1, false, 3, canon, ev, lens, n4_, ptr<Fr>(coset_pre_), nullptr, nullptr));
            CHK(side_end(s));
        } else {
            CHK(coset_eval(st, 3, canon, lens, ev));
        }
    }
    CHK(sync_results(s));
    if (measuring) wire_density_pm_.store((uint32_t)((uint64_t)*h_density * 1000u / ((uint64_t)3 * n * W_)), std::memory_order_relaxed);
    Aff lro[3] = {hp[0], hp[1], hp[2]};
    for (int j = 0; j < 3; j++) store_pt(out->lro[j], lro[j]);

    double phase_ms[4] = {0, 0, 0, 0}, lincomb_ms = 0;   // R1..R4 of SURVEY.md section 3.3, host wall clock (stats only)
    auto t_phase = t_start;
    auto mark = [&](int r) {
        if (!stats_on_) return;
        const auto now = std::chrono::steady_clock::now();
        phase_ms[r] = std::chrono::duration<double, std::milli>(now - t_phase).count();
        t_phase = now;
    };
    mark(0);

    // ---------------- gamma, beta (templateLogicSigBN254.go:131-133) -----------------------------------------
    uint8_t vk_bytes[8 + APK_MAX_COMMITMENTS][2 * FPB];
    for (uint32_t i = 0; i < 8 + nb_commit_; i++) g1_raw_bytes(vk_pts_[i], vk_bytes[i]);
    std::vector<uint8_t> pub_bytes((size_t)nb_public_ * 32);
    for (uint32_t i = 0; i < nb_public_; i++) fe_to_be<FRP>(pubv[i], pub_bytes.data() + 32 * i);
    uint8_t lro_bytes[3][2 * FPB];
    for (int j = 0; j < 3; j++) g1_raw_bytes(lro[j], lro_bytes[j]);
    uint8_t gamma_raw[32], beta_raw[32], alpha_raw[32], zeta_raw[32];
    {
        // order: S1,S2,S3,Ql,Qr,Qm,Qo,Qk,Qcp..  (vk_pts_ order is Ql,Qr,Qm,Qo,Qk,S1,S2,S3,Qcp..)
        std::vector<const uint8_t*> parts = {vk_bytes[5], vk_bytes[6], vk_bytes[7], vk_bytes[0], vk_bytes[1], vk_bytes[2], vk_bytes[3], vk_bytes[4]};
        std::vector<size_t> lens(8, 2 * FPB);
        for (uint32_t k = 0; k < nb_commit_; k++) { parts.push_back(vk_bytes[8 + k]); lens.push_back(2 * FPB); }
        parts.push_back(pub_bytes.data()); lens.push_back(pub_bytes.size());
        for (int j = 0; j < 3; j++) { parts.push_back(lro_bytes[j]); lens.push_back(2 * FPB); }
        hash_challenge("gamma", nullptr, parts, lens, gamma_raw);
        hash_challenge("beta", gamma_raw, {}, {}, beta_raw);
    }
    const Fr gamma = fr_from_be(gamma_raw), beta = fr_from_be(beta_raw);
    const Fr beta_u = beta * shift_, beta_u2 = beta_u * shift_;

    // ---------------- round 2: grand product Z -------------------------------------------
    {
        const uint32_t nb = cdiv(n, SCAN_BLOCK);
        GpScan<FRP> g{};
        g.data[0] = ptr<Fr>(s.ratio); g.data[1] = ptr<Fr>(s.tmp);
        g.tot[0] = ptr<Fr>(s.scan_tot); g.tot[1] = ptr<Fr>(s.scan_tot) + nb + 1;
        // (the kernel multiplies on unsaturated limbs: the challenges in the product's radix R' = 32 R)
        CHK((klaunch<GpTermsK<FRP>, POLY_THREADS>(st, cdiv(n, POLY_THREADS), POLY_THREADS, 0, dL, dR, dO, (const Fr*)ptr<Fr>(s_lag_[0]),
                                                  (const Fr*)ptr<Fr>(s_lag_[1]), (const Fr*)ptr<Fr>(s_lag_[2]), (const Fr*)ptr<Fr>(tw_n_), n, beta * fr_u64(32), gamma,
                                                  beta_u * fr_u64(32), beta_u2 * fr_u64(32), g.data[0], g.data[1])));
        CHK((klaunch<GpScanBlockK<FRP>, POLY_THREADS>(st, dim3(nb, 2), POLY_THREADS, 0, g, n)));
        Fr* const tot_out = s.d_pinned ? reinterpret_cast<Fr*>(s.d_pinned + PIN_FR) : g.tot[1] + nb;     // hfr[0], from the device
        CHK((klaunch<GpScanTotalsK<FRP>, POLY_THREADS>(st, 2, POLY_THREADS, 0, g, nb, tot_out)));
        if (!s.d_pinned) HIPCHK(hipMemcpyAsync(hfr, g.tot[1] + nb, sizeof(Fr), hipMemcpyDeviceToHost, st));
        CHK(sync_results(s));
        const Fr den_total_inv = Fr::inv(hfr[0]);
        CHK((klaunch<GpFinishK<FRP>, POLY_THREADS>(st, cdiv(n, POLY_THREADS), POLY_THREADS, 0, g, n, den_total_inv, ptr<Fr>(s.zlag))));
    }
    CHK(inv_ntt_n(st, ptr<Fr>(s.zlag), ptr<Fr>(s.cz)));
    {
        Fr4<FRP> b{};
        b.v[0] = bl[6]; b.v[1] = bl[7]; b.v[2] = bl[8];
        CHK((klaunch<BlindK<FRP>, 256>(st, 1, 64, 0, ptr<Fr>(s.cz), n, b, 3)));
        MsmBatchArgs a{};
        a.batch = 1; a.scalars[0] = s.cz.p; a.len[0] = n + 3; a.offset[0] = 0;
        s.mark_acc = fill;
        const int rc = commit(s, tab_can_, 0, a, hp);
        s.mark_acc = 0;
        CHK(rc);
    }
    if (fill) {
        CHK(side_begin(s));
        CHK(coset_ntt_4n(s.side, ptr<Fr>(s.cz), n + 3, ptr<Fr>(s.ez)));
        CHK(side_end(s));
    } else {
        const Fr* cin = ptr<Fr>(s.cz); Fr* eout = ptr<Fr>(s.ez);
        const uint32_t zl = n + 3;
        CHK(coset_eval(st, 1, &cin, &zl, &eout));
        if (sc_.on() && sc_.glog == 3) {   // Z(omega X) lives in class (k + 4) mod 8: evaluated there as well (eqk is free: Qk is completed in the kernel)
            Fr* e2 = ptr<Fr>(s.eqk);
            CHK(coset_eval(st, 1, &cin, &zl, &e2, 1));
        }
    }
    CHK(sync_results(s));
    const Aff zcom = hp[0];
    store_pt(out->z, zcom);
    mark(1);
    uint8_t z_bytes[2 * FPB];
    g1_raw_bytes(zcom, z_bytes);
    {
        std::vector<const uint8_t*> parts; std::vector<size_t> lens;
        for (uint32_t k = 0; k < nb_commit_; k++) { parts.push_back(bsb_bytes[k]); lens.push_back(2 * FPB); }
        parts.push_back(z_bytes); lens.push_back(2 * FPB);
        hash_challenge("alpha", beta_raw, parts, lens, alpha_raw);
    }
    const Fr alpha = fr_from_be(alpha_raw);

    // ---------------- round 3: quotient on the 4n coset ------------------------------------------------------
    {
        QuotientArgs<FRP> q{};
        q.l = ptr<Fr>(s.el); q.r = ptr<Fr>(s.er); q.o = ptr<Fr>(s.eo); q.z = ptr<Fr>(s.ez); q.qk = ptr<Fr>(s.eqk);
        if (qk_direct_) {
            q.qk = ptr<Fr>(eqk_trace_);
            q.nb_inject = (int)inj_row_.size();
            for (size_t j = 0; j < inj_row_.size(); j++) {
                const Fr written = j < nb_public_ ? pubv[j] : cval[j - nb_public_];
                q.inj_delta[j] = written - inj_trace_val_[j];
                q.inj_tab[j] = inj_row_[j] == 0 ? ptr<Fr>(l0_4_) : ptr<Fr>(inj_tab_[j]);
            }
        }
        q.ql = ptr<Fr>(eql_); q.qr = ptr<Fr>(eqr_); q.qm = ptr<Fr>(eqm_); q.qo = ptr<Fr>(eqo_);
        q.s1 = ptr<Fr>(es_[0]); q.s2 = ptr<Fr>(es_[1]); q.s3 = ptr<Fr>(es_[2]);
        q.x = ptr<Fr>(x4_); q.l0 = ptr<Fr>(l0_4_);
        q.nb_commit = (int)nb_commit_;
        for (uint32_t k = 0; k < nb_commit_; k++) { q.qcp[k] = ptr<Fr>(eqcp_[k]); q.pi2[k] = ptr<Fr>(s.epi2[k]); }
        // the kernel multiplies in the radix R' = 32 R: constants carry the factors its header lists
        const Fr c5 = fr_u64(1u << 5), c10 = fr_u64(1u << 10), c20 = fr_u64(1u << 20);
        q.alpha = alpha * c20; q.beta = beta * c5; q.gamma = gamma; q.beta_u = beta_u * c5; q.beta_u2 = beta_u2 * c5;
        q.alpha2 = alpha * alpha * c10;
        for (int k = 0; k < 4; k++) q.zh_inv[k] = zh_inv_[k] * c5;
        for (int j = 0; j < q.nb_inject; j++) q.inj_delta[j] = q.inj_delta[j] * c5;
        q.n4 = n4_;
        if (!sc_.on()) {
            CHK((klaunch<QuotientK<FRP>, POLY_THREADS>(st, cdiv(n4_, POLY_THREADS), POLY_THREADS, 0, q, ptr<Fr>(s.quot))));
            CHK(run_ntt(st, 1, true, ptr<Fr>(s.quot), ptr<Fr>(s.hcan), n4_, n4_, nullptr, ptr<Fr>(coset_post_inv_), nullptr));
        } else {
            // sub-coset split: the quotient on this rank's m points, the local part of the inverse transform, ONE all-gather,
            // the last log2(G) stages on every rank - hcan comes out bit for bit as above
            const uint32_t m = n4_ >> sc_.glog;
            q.n4 = m; q.sub_k = (uint32_t)sc_.k; q.sub_glog = (uint32_t)sc_.glog; q.zs = ptr<Fr>(s.eqk);
            quotient_kernel<FRP><<<cdiv(m, POLY_THREADS), POLY_THREADS, 0, st>>>(q, ptr<Fr>(s.quot)); KCHK();
            Fr* mine = ptr<Fr>(s.sc_gather) + (size_t)sc_.k * m;
            CHK(run_ntt(st, 1, true, ptr<Fr>(s.quot), mine, m, m, nullptr, ptr<Fr>(sc_.post), nullptr, sc_.glog));
            HIPCHK(hipStreamSynchronize(st));
            hook_slot() = &s;
            const int grc = sc_.hook(sc_.user, s.sc_gather.p, (size_t)m * sizeof(Fr));
            hook_slot() = nullptr;
            if (grc != APK_OK) { set_error("sub-coset gather hook failed with %d", grc); return grc == APK_ERR_ARG ? APK_ERR_ARG : APK_ERR_STATE; }
            SubMergeArgs<FRP> ma{};
            for (int e = 0; e < 8; e++) ma.rho_inv[e] = sc_.rho_inv[e];
            ma.m = m; ma.glog = (uint32_t)sc_.glog;
            subcoset_merge_kernel<FRP><<<cdiv(m, POLY_THREADS), POLY_THREADS, 0, st>>>(ma, ptr<Fr>(s.sc_gather), ptr<Fr>(coset_post_inv_), ptr<Fr>(s.hcan)); KCHK();
        }
        // h = h1 + X^(n+2) h2 + X^(2(n+2)) h3   (templateLogicSigBN254.go:79,220-226)
        MsmBatchArgs a{};
        a.batch = 3;
        for (int j = 0; j < 3; j++) { a.scalars[j] = ptr<Fr>(s.hcan) + (size_t)j * (n + 2); a.len[j] = n + 2; a.offset[j] = 0; }
        CHK(commit(s, tab_can_, 0, a, hp));
        // the quotient is a polynomial of degree < 3n+6 iff the witness satisfies the circuit: every coefficient of the tail
        // h[3(n+2) .. 4n) must vanish (OR-reduce on the device, one flag word back)
        s.epoch++;
        const uint32_t tail4 = (n4_ - 3 * (n + 2)) * (uint32_t)(sizeof(Fr) / 16);
        CHK((klaunch<TailNonzeroK<0>, 256>(st, cdiv(tail4, 256 * 8) < 512 ? cdiv(tail4, 256 * 8) : 512, 256, 0,
                                            reinterpret_cast<const uint4*>(ptr<Fr>(s.hcan) + 3 * (size_t)(n + 2)), tail4, s.epoch,
                                            s.d_pinned ? reinterpret_cast<uint32_t*>(s.d_pinned + PIN_TAIL) : ptr<uint32_t>(s.tail_flag))));
        if (!s.d_pinned) HIPCHK(hipMemcpyAsync(reinterpret_cast<uint8_t*>(s.h_pinned) + PIN_TAIL, s.tail_flag.p, 4, hipMemcpyDeviceToHost, st));
    }
    CHK(sync_results(s));
    if (*reinterpret_cast<const volatile uint32_t*>(reinterpret_cast<const uint8_t*>(s.h_pinned) + PIN_TAIL) == s.epoch) {
        set_error("quotient is not a polynomial: the witness does not satisfy the circuit");
        in.ok = true;      // (the stream was just drained: nothing of this proof is still reading its inputs)
        guard.ok = true;
        return APK_ERR_WITNESS;
    }
    Aff hcom[3] = {hp[0], hp[1], hp[2]};
    mark(2);
    uint8_t h_bytes[3][2 * FPB];
    for (int j = 0; j < 3; j++) { store_pt(out->h[j], hcom[j]); g1_raw_bytes(hcom[j], h_bytes[j]); }
    hash_challenge("zeta", alpha_raw, {h_bytes[0], h_bytes[1], h_bytes[2]}, {2 * FPB, 2 * FPB, 2 * FPB}, zeta_raw);
    const Fr zeta = fr_from_be(zeta_raw);

    // ---------------- round 4: evaluations, linearised polynomial, openings ----------------------------------
    Fr zn_m1, zn2, mz;   // zeta^n - 1, zeta^(n+2), 1 - zeta^n: worked out below while the GPU evaluates
    // The context's parked host threads for the [lin] combination while this proof has the context (nearly) to itself - then the
    // GPU idles through it; with many proofs in flight the callers' own threads already keep the host busy.  Since the combination
    // takes GLV halves and fixed-base tables (host_msm.h) one thread does BN254's in 0.10 ms (0.15 before) and waking three more
    // costs what they save; BLS12-381's (0.23 ms) still gains a little from three (same box: tools/ab_lone.sh).
    static const int lc_threads = env_int("APK_HOST_LINCOMB_THREADS", FPP::N <= 8 ? 1 : 3, 1, 8);
    static const bool host_glv = env_int("APK_HOST_GLV", 1, 0, 1) != 0;   // 0: full-length scalars (measurement aid; same bytes)
    HostPool* pool = nullptr;
    const bool host_idle = gate_.busy() <= 2;     // no other proofs' threads competing for the host while the GPU works on this one
    if (lc_threads > 1 && host_idle) {
        std::lock_guard<std::mutex> lk2(mu_);
        if (!lc_pool_) lc_pool_.reset(new HostPool(lc_threads - 1));
        pool = lc_pool_.get();
        path(P_LINCOMB_POOL);
    }
    XYZZ<FPP, Fe64<FPP>> lin_h = XYZZ<FPP, Fe64<FPP>>::inf();
    bool lin_h_done = false;
    const Fr zw = zeta * omega_;
    const bool z0 = zeta.is_zero();
    const Fr zeta_inv = Fr::inv(zeta), zw_inv = zeta_inv * omega_inv_;   // (one host inversion, ~10 us, before the launches below)
    {
        Fr* outs[4] = {ptr<Fr>(s.pw_z), ptr<Fr>(s.pw_zw), ptr<Fr>(s.pw_zi), ptr<Fr>(s.pw_zwi)};
        const Fr ws[4] = {zeta, zw, zeta_inv, zw_inv};
        static const int derive = env_int("APK_POWERS_DERIVE", 1, 0, 1);
        if (z0 || !derive) {
            CHK(powers_batch(st, z0 ? 2 : 4, outs, ws, n + 3));
        } else {
            // zeta^i and zeta^-i by square-and-multiply walks (~6 products per element); (omega zeta)^(+-i) from them and the
            // transform's own table of omega^j: one product per element
            Fr* two[2] = {outs[0], outs[2]};
            const Fr w2[2] = {zeta, zeta_inv};
            CHK(powers_batch(st, 2, two, w2, n + 3));
            DerivePowers<FRP> dp{};
            dp.in[0] = outs[0]; dp.out[0] = outs[1]; dp.inverse[0] = 0;
            dp.in[1] = outs[2]; dp.out[1] = outs[3]; dp.inverse[1] = 1;
            CHK((klaunch<DerivePowersK<FRP>, POLY_THREADS>(st, dim3(cdiv(n + 3, POLY_THREADS), 2), POLY_THREADS, 0, dp, (const Fr*)ptr<Fr>(twu_n_), n, n + 3)));
        }
    }
    std::vector<Fr> lag_w, lag_den;
    Fr ev[EVAL_MAX];
    {
        EvalArgs<FRP> ea{};
        const Fr* fs[5] = {ptr<Fr>(s.cl), ptr<Fr>(s.cr), ptr<Fr>(s.co), ptr<Fr>(s_c_[0]), ptr<Fr>(s_c_[1])};
        const uint32_t ls[5] = {n + 2, n + 2, n + 2, n, n};
        for (int i = 0; i < 5; i++) { ea.f[i] = fs[i]; ea.len[i] = ls[i]; }
        ea.count = 5;
        for (uint32_t k = 0; k < nb_commit_; k++) { ea.f[ea.count] = ptr<Fr>(qcp_c_[k]); ea.len[ea.count] = n; ea.count++; }
        // Z(omega*zeta) rides in the same launch with its own table of powers
        const int iz = ea.count;
        ea.f[iz] = ptr<Fr>(s.cz); ea.len[iz] = n + 3; ea.pw[iz] = ptr<Fr>(s.pw_zw); ea.count++;
        CHK(eval_many(s, ea, ptr<Fr>(s.pw_z), hfr));
        // the opening quotient of Z at omega*zeta does not wait for anything the host derives from the evaluations
        CHK(kzg_quotient(s, ptr<Fr>(s.cz), n + 3, ptr<Fr>(s.pw_zw), ptr<Fr>(s.pw_zwi), z0, ptr<Fr>(s.q2)));
        // [lin] (below) is a combination of commitments the host holds; the coefficients of [H1..3] depend on zeta alone, so a
        // lone proof's host thread works that part out while the GPU evaluates (the rest needs the evaluations).  With other
        // proofs in flight it stays one pass: a second pass repeats the doublings, and the host is what those proofs share.
        zn_m1 = Fr::pow_u64(zeta, n) - Fr::one();
        zn2 = Fr::pow_u64(zeta, n + 2);
        mz = Fr::neg(zn_m1);
        // (so do the Lagrange terms: worked out here, while the GPU evaluates)
        // L_i(zeta) = omega^i (zeta^n - 1) / (n (zeta - omega^i)) for the rows the prover wrote into Qk (public inputs, commitment
        // hashes) and for row 0: one shared inversion
        {
            std::vector<uint32_t> rows = {0};
            for (uint32_t i = 1; i < nb_public_; i++) rows.push_back(i);
            for (uint32_t k = 0; k < nb_commit_; k++) rows.push_back(nb_public_ + cci_[k]);
            Fr wi = Fr::one();
            uint32_t at = 0;
            for (uint32_t row : rows) {
                if (row == at + 1) wi = wi * omega_; else if (row != at) wi = Fr::pow_u64(omega_, row);
                at = row;
                lag_w.push_back(wi);
                lag_den.push_back(zeta - wi);
            }
            std::vector<Fr> pref(lag_den.size() + 1, Fr::one());
            for (size_t i = 0; i < lag_den.size(); i++) pref[i + 1] = pref[i] * lag_den[i];
            Fr inv = Fr::inv(pref.back());
            const Fr scale = zn_m1 * n_inv_;
            for (size_t i = lag_den.size(); i-- > 0;) {
                const Fr di = inv * pref[i];
                inv = inv * lag_den[i];
                lag_w[i] = lag_w[i] * scale * di;        // = L_row(zeta)
            }
        }
        static const int early_h = env_int("APK_LIN_EARLY_H", 1, 0, 1);
        if (host_idle && early_h && (pool || FPP::N <= 8)) {   // (one thread takes longer over BLS12-381's three than the GPU over the evaluations)
            const auto t_lc = std::chrono::steady_clock::now();
            const Aff hpts[3] = {hcom[0], hcom[1], hcom[2]};
            const Fr hks[3] = {mz, mz * zn2, mz * zn2 * zn2};
            lin_h = host_lincomb_sum<FRP, FPP>(hpts, hks, 3, pool, host_glv);
            lin_h_done = true;
            if (stats_on_) lincomb_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_lc).count();
        }
        CHK(sync_results(s));
        const Fr c32 = fr_u64(32);        // eval_partial_kernel multiplies value x value on the product's radix: f(z) / 32 comes back
        for (int i = 0; i < ea.count; i++) ev[i] = hfr[i] * c32;
    }
    const Fr lz = ev[0], rz = ev[1], oz = ev[2], s1z = ev[3], s2z = ev[4];
    const Fr zshift = ev[5 + nb_commit_];
    // coefficients of the linearised polynomial (templateLogicSigBN254.go:195-201,231-254)
    const Fr alpha2 = alpha * alpha;
    const Fr lag0 = lag_w[0];
    // lin(zeta) as the verifier derives it from the quotient identity (templateLogicSigBN254.go:203-218) - the
    // identity holds exactly (the tail check above), so this IS the evaluation of the linearised polynomial
    Fr pi_z = Fr::zero();   // lag_w = [L_0, L_1 .. L_{nbPublic-1}, L_{nbPublic+cci_0} ..](zeta)
    for (uint32_t i = 0; i < nb_public_; i++) pi_z = pi_z + pubv[i] * lag_w[i];
    for (uint32_t k = 0; k < nb_commit_; k++) pi_z = pi_z + cval[k] * lag_w[(nb_public_ ? nb_public_ : 1) + k];
    const Fr lin_z = Fr::neg(pi_z + alpha * zshift * (lz + beta * s1z + gamma) * (rz + beta * s2z + gamma) * (oz + gamma) - alpha2 * lag0);
    const Fr c_s3 = alpha * beta * zshift * (lz + beta * s1z + gamma) * (rz + beta * s2z + gamma);
    const Fr c_z = alpha2 * lag0 - alpha * (lz + beta * zeta + gamma) * (rz + beta_u * zeta + gamma) * (oz + beta_u2 * zeta + gamma);
    // terms of the linearised polynomial: the polynomial (device), its commitment (host), its coefficient
    struct LinTerm { const Fr* poly; uint32_t len; Aff com; Fr coef; };
    std::vector<LinTerm> lin_terms = {
        {ptr<Fr>(ql_c_), n, vk_pts_[0], lz}, {ptr<Fr>(qr_c_), n, vk_pts_[1], rz}, {ptr<Fr>(qm_c_), n, vk_pts_[2], lz * rz},
        {ptr<Fr>(qo_c_), n, vk_pts_[3], oz}, {ptr<Fr>(qk_c_), n, vk_pts_[4], Fr::one()}, {ptr<Fr>(s_c_[2]), n, vk_pts_[7], c_s3},
        {ptr<Fr>(s.cz), n + 3, zcom, c_z}, {ptr<Fr>(s.hcan), n + 2, hcom[0], mz},
        {ptr<Fr>(s.hcan) + (n + 2), n + 2, hcom[1], mz * zn2}, {ptr<Fr>(s.hcan) + 2 * (size_t)(n + 2), n + 2, hcom[2], mz * zn2 * zn2}};
    for (uint32_t k = 0; k < nb_commit_; k++) lin_terms.push_back({ptr<Fr>(s.pi2_can[k]), n, bsb[k], ev[5 + k]});
    // [lin] = sum coef_i * [poly_i]: the group element kzg.Commit(lin) would give, taken from commitments already in hand
    // (host_msm.h) instead of a tenth size-n MSM
    Aff lin_com;
    {
        // terms 7..9 are the [H] part (taken above when the pool was there), term 4 is [Qk] with coefficient one: a plain addition
        Aff lp[HOST_MSM_MAX];
        Fr lk[HOST_MSM_MAX];
        // terms 0..3 and 5 ([Ql][Qr][Qm][Qo][S3]) come out of the context's fixed-base tables: 33 additions each
        static const bool host_fixed = env_int("APK_HOST_FIXED", 1, 0, 1) != 0;   // 0: through the Straus pass (measurement aid; same bytes)
        int cnt = 0;
        for (size_t i = 0; i < lin_terms.size(); i++) {
            if (i == 4 || (lin_h_done && i >= 7 && i <= 9) || (host_fixed && (i <= 3 || i == 5))) continue;
            lp[cnt] = lin_terms[i].com; lk[cnt] = lin_terms[i].coef; cnt++;
        }
        const auto t_lc = std::chrono::steady_clock::now();
        XYZZ<FPP, Fe64<FPP>> sum = host_lincomb_sum<FRP, FPP>(lp, lk, cnt, pool, host_glv);
        if (host_fixed) {
            const Fr fk[5] = {lin_terms[0].coef, lin_terms[1].coef, lin_terms[2].coef, lin_terms[3].coef, lin_terms[5].coef};
            vk_fixed_.template accumulate<FRP>(sum, fk);
        }
        if (lin_h_done) sum.add(lin_h);
        {
            using F64 = Fe64<FPP>;
            sum.madd(Affine<FPP, F64>{F64::from(vk_pts_[4].x), F64::from(vk_pts_[4].y)});
        }
        lin_com = host_xyzz_to_affine<FPP>(sum);
        if (stats_on_) lincomb_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_lc).count();
    }
    memcpy(out->zshift_value, &zshift, sizeof(Fr));
    Fr claimed[6 + APK_MAX_COMMITMENTS] = {lin_z, lz, rz, oz, s1z, s2z};
    for (uint32_t k = 0; k < nb_commit_; k++) claimed[6 + k] = ev[5 + k];
    for (uint32_t i = 0; i < 6 + nb_commit_; i++) memcpy(out->claimed_values[i], &claimed[i], sizeof(Fr));
    // gamma' for the batched opening (templateLogicSigBN254.go:280-286)
    uint8_t gk_raw[32];
    {
        uint8_t zeta_be[32], lin_bytes[2 * FPB], cv_be[6 + APK_MAX_COMMITMENTS][32], zs_be[32];
        fe_to_be<FRP>(zeta, zeta_be);
        g1_raw_bytes(lin_com, lin_bytes);
        for (uint32_t i = 0; i < 6 + nb_commit_; i++) fe_to_be<FRP>(claimed[i], cv_be[i]);
        fe_to_be<FRP>(zshift, zs_be);
        std::vector<const uint8_t*> parts = {zeta_be, lin_bytes, lro_bytes[0], lro_bytes[1], lro_bytes[2], vk_bytes[5], vk_bytes[6]};
        std::vector<size_t> lens = {32, 2 * FPB, 2 * FPB, 2 * FPB, 2 * FPB, 2 * FPB, 2 * FPB};
        for (uint32_t k = 0; k < nb_commit_; k++) { parts.push_back(vk_bytes[8 + k]); lens.push_back(2 * FPB); }
        for (uint32_t i = 0; i < 6 + nb_commit_; i++) { parts.push_back(cv_be[i]); lens.push_back(32); }
        parts.push_back(zs_be); lens.push_back(32);
        hash_challenge("gamma", nullptr, parts, lens, gk_raw);
    }
    const Fr gk = fr_from_be(gk_raw);
    {
        // folded = lin + gk l + gk^2 r + gk^3 o + gk^4 S1 + gk^5 S2 + gk^(6+k) Qcp_k, one fused linear combination over the
        // constituents of lin (the linearised polynomial itself is never materialised)
        LinCombArgs<FRP> lc{};
        int c = 0;
        for (const LinTerm& t : lin_terms) { lc.f[c] = t.poly; lc.len[c] = t.len; lc.coef[c] = t.coef; c++; }
        Fr acc = gk;
        auto push = [&](const Fr* f, uint32_t len) { lc.f[c] = f; lc.len[c] = len; lc.coef[c] = acc; c++; acc = acc * gk; };
        push(ptr<Fr>(s.cl), n + 2); push(ptr<Fr>(s.cr), n + 2); push(ptr<Fr>(s.co), n + 2);
        push(ptr<Fr>(s_c_[0]), n); push(ptr<Fr>(s_c_[1]), n);
        for (uint32_t k = 0; k < nb_commit_; k++) push(ptr<Fr>(qcp_c_[k]), n);
        lc.count = c; lc.out_len = n + 3;
        {   // the kernel multiplies on unsaturated limbs: coefficients in the product's radix R' = 32 R
            const Fr c32 = fr_u64(32);
            for (int k = 0; k < c; k++) lc.coef[k] = lc.coef[k] * c32;
        }
        CHK((klaunch<LincombK<FRP>, POLY_THREADS>(st, cdiv(n + 3, POLY_THREADS), POLY_THREADS, 0, lc, ptr<Fr>(s.folded))));
        CHK(kzg_quotient(s, ptr<Fr>(s.folded), n + 3, ptr<Fr>(s.pw_z), ptr<Fr>(s.pw_zi), z0, ptr<Fr>(s.q1)));
        // both opening proofs in one batch: W_zeta (batched opening) and W_omega*zeta (kzg.Open of Z [UPSTREAM])
        MsmBatchArgs a{};
        a.batch = 2;
        a.scalars[0] = s.q1.p; a.len[0] = n + 2; a.offset[0] = 0;
        a.scalars[1] = s.q2.p; a.len[1] = n + 2; a.offset[1] = 0;
        CHK(commit(s, tab_can_, 0, a, hp));
    }
    CHK(sync_results(s));
    store_pt(out->zshift_h, hp[1]);
    store_pt(out->batched_h, hp[0]);
    memcpy(out->gamma, &gamma, sizeof(Fr)); memcpy(out->beta, &beta, sizeof(Fr)); memcpy(out->alpha, &alpha, sizeof(Fr));
    memcpy(out->zeta, &zeta, sizeof(Fr)); memcpy(out->gamma_kzg, &gk, sizeof(Fr));
    mark(3);
    in.ok = true;
    guard.ok = true;
    path(P_PROOFS);
    if (stats_on_) {
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
        std::lock_guard<std::mutex> g(stats_mu_);
        stats_.prove_ms += ms;
        stats_.proofs += 1;
        for (int r = 0; r < 4; r++) stats_.round_ms[r] += phase_ms[r];
        stats_.host_lincomb_ms += lincomb_ms;
    }
    return APK_OK;
}

// out[i] = scalars[i] * base on `device`; host buffers in, host buffers out
template <class FRP, class FPP>
int g1_mul_batch_impl(int device, const void* base, const void* scalars, uint64_t count, void* out) {
    using Fr = Fe<FRP>;
    using Aff = Affine<FPP>;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) { set_error("no HIP device available; libapk has no CPU fallback"); return APK_ERR_HIP; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range", device); return APK_ERR_ARG; }
    HIPCHK(hipSetDevice(device));
    DevBuf ds, dout;
    CHK(ds.alloc(count * sizeof(Fr)));
    CHK(dout.alloc(count * sizeof(Aff)));
    HIPCHK(hipMemcpy(ds.p, scalars, count * sizeof(Fr), hipMemcpyHostToDevice));
    Aff b;
    memcpy(&b, base, sizeof b);
    g1_mul_batch_kernel<FRP, FPP><<<cdiv(count, 256), 256>>>(b, ptr<Fr>(ds), (uint32_t)count, ptr<Aff>(dout));
    KCHK();
    HIPCHK(hipMemcpy(out, dout.p, count * sizeof(Aff), hipMemcpyDeviceToHost));
    return APK_OK;
}

// ---- setup-time entry points (host buffers in / out) ----------------------------------------------------------------
static inline int pick_device(int device) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) { set_error("no HIP device available; libapk has no CPU fallback"); return APK_ERR_HIP; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range", device); return APK_ERR_ARG; }
    HIPCHK(hipSetDevice(device));
    return APK_OK;
}

template <class FRP, class FPP, int CURVE_ID>
int g1_decompress_impl(int device, const uint8_t* in, uint64_t count, void* out) {
    using Aff = Affine<FPP>;
    CHK(pick_device(device));
    const size_t nb = (size_t)FPP::N * 4;
    DevBuf din, dout, derr;
    CHK(din.alloc(count * nb));
    CHK(dout.alloc(count * sizeof(Aff)));
    CHK(derr.alloc(4));
    HIPCHK(hipMemcpy(din.p, in, count * nb, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(derr.p, 0, 4));
    g1_decompress_kernel<FRP, FPP, CURVE_ID><<<cdiv(count, 256), 256>>>(ptr<uint8_t>(din), (uint32_t)count, ptr<Aff>(dout), ptr<uint32_t>(derr));
    KCHK();
    uint32_t bad = 0;
    HIPCHK(hipMemcpy(&bad, derr.p, 4, hipMemcpyDeviceToHost));
    if (bad) { set_error("%u compressed point(s) are not valid G1 encodings", bad); return APK_ERR_ARG; }
    HIPCHK(hipMemcpy(out, dout.p, count * sizeof(Aff), hipMemcpyDeviceToHost));
    return APK_OK;
}

// kzg.ToLagrangeG1 on device buffers: an inverse FFT in the exponent (kernels_setup.h); complete when it returns
template <class FRP, class FPP>
int g1_to_lagrange_dev(const Affine<FPP>* d_in, uint64_t n, Affine<FPP>* d_out, hipStream_t st) {
    using Fr = Fe<FRP>;
    using Aff = Affine<FPP>;
    using Pt = XYZZ<FPP>;
    if (n < 2 || (n & (n - 1)) || n > (1ull << 24)) { set_error("ToLagrangeG1: size must be a power of two in [2, 2^24]"); return APK_ERR_ARG; }
    int log_n = 0;
    while ((1ull << log_n) < n) log_n++;
    // omega^-1 of the size-n domain, 1/n
    Fr root;
    for (int i = 0; i < Fr::N; i++) root.l[i] = FRP::root(i);
    Fr w = Fr::to_mont(root);
    for (int i = 0; i < FRP::ADICITY - log_n; i++) w = Fr::sqr(w);
    Fr winv = Fr::inv(w);
    Fr nn = Fr::zero();
    nn.l[0] = (uint32_t)n;
    Fr ninv = Fr::inv(Fr::to_mont(nn));
    DevBuf work, twi;
    CHK(work.alloc(n * sizeof(Pt)));
    CHK(twi.alloc((n / 2 + 1) * sizeof(Fr)));
    PowersBatch<FRP> pb{};
    pb.out[0] = ptr<Fr>(twi); pb.w[0] = winv; pb.scale[0] = Fr::one();
    powers_kernel<FRP><<<dim3(cdiv(cdiv(n / 2, 8), 256), 1), 256, 0, st>>>(pb, (uint32_t)(n / 2));
    KCHK();
    lagrange_load_kernel<FPP><<<cdiv(n, 256), 256, 0, st>>>(d_in, (uint32_t)n, log_n, ptr<Pt>(work));
    KCHK();
    for (int t = 0; t < log_n; t++) {
        lagrange_stage_kernel<FRP, FPP><<<cdiv(n / 2, 128), 128, 0, st>>>(ptr<Pt>(work), ptr<Fr>(twi), (uint32_t)n, log_n, t);
        KCHK();
    }
    lagrange_finish_kernel<FRP, FPP><<<cdiv(n, 128), 128, 0, st>>>(ptr<Pt>(work), (uint32_t)n, ninv, d_out);
    KCHK();
    HIPCHK(hipStreamSynchronize(st));     // the work buffers are released on return
    return APK_OK;
}

template <class FRP, class FPP>
int g1_to_lagrange_impl(int device, const void* points, uint64_t n, void* out) {
    using Aff = Affine<FPP>;
    if (n < 2 || (n & (n - 1)) || n > (1ull << 24)) { set_error("ToLagrangeG1: size must be a power of two in [2, 2^24]"); return APK_ERR_ARG; }
    CHK(pick_device(device));
    DevBuf din, dout;
    CHK(din.alloc(n * sizeof(Aff)));
    CHK(dout.alloc(n * sizeof(Aff)));
    HIPCHK(hipMemcpy(din.p, points, n * sizeof(Aff), hipMemcpyHostToDevice));
    CHK((g1_to_lagrange_dev<FRP, FPP>(ptr<Aff>(din), n, ptr<Aff>(dout))));
    HIPCHK(hipMemcpy(out, dout.p, n * sizeof(Aff), hipMemcpyDeviceToHost));
    return APK_OK;
}

}  // namespace apk
