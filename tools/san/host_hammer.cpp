// Sanitizer tier (SURVEY.md section 5 "race detection"): the host-side threading of libapk that has no GPU in it, hammered from
// many threads under -fsanitize=thread (or address).  Built and run by `make -C algoplonk_amd/csrc SAN=thread san-check`
// (tests/test_sanitizers.py runs it in the CPU tier).  What runs here is the library's own code, not a model of it:
//   * SlotGate (slot_gate.h)   - 32 callers on 4 / 16 slots: every slot has one owner at a time, the busy count is never torn
//   * SlotGate + Gang (gang.h) - 48 callers on 32 slots / 8 streams in gangs of up to 2 / 4: never more streams than allowed, every
//     member of a gang sees the same size, every meeting launches exactly once with every member's request, members that leave
//     early (one in five) never strand the others, a lead outlives its followers
//   * HostPool + host_lincomb (host_msm.h) - the [lin] combination on the context's parked threads, 32 callers racing for the pool;
//     every result must be the single-threaded one
// Exit code 0 = no mismatch (the sanitizer reports races on its own and fails the run through TSAN_OPTIONS=halt_on_error=1).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "../../algoplonk_amd/csrc/ff_params.h"
#include "../../algoplonk_amd/csrc/host_msm.h"
#include "../../algoplonk_amd/csrc/slot_gate.h"
#include "../../algoplonk_amd/csrc/gang.h"

using namespace apk;

template <class FRP, class FPP>
static int hammer_lincomb(const char* name, const uint32_t* gx, const uint32_t* gy) {
    using Fr = Fe<FRP>;
    using Fp = Fe<FPP>;
    using Aff = Affine<FPP>;
    using Pt = XYZZ<FPP>;
    // a few distinct points: multiples of the generator
    Aff g;
    for (int i = 0; i < Fp::N; i++) { g.x.l[i] = gx[i]; g.y.l[i] = gy[i]; }
    g.x = Fp::to_mont(g.x); g.y = Fp::to_mont(g.y);
    constexpr int COUNT = 11;
    Aff pts[COUNT];
    Fr ks[COUNT];
    Pt run = Pt::from_affine(g);
    for (int i = 0; i < COUNT; i++) {
        pts[i] = run.to_affine();
        run = Pt::dbl(run); run.madd(g);
        Fr k = Fr::zero();
        for (int w = 0; w < Fr::N; w++) k.l[w] = 0x9e3779b9u * (uint32_t)(i * 8 + w + 1) ^ 0x7f4a7c15u;
        k.l[Fr::N - 1] &= 0x0fffffffu;
        ks[i] = Fr::to_mont(k);
    }
    const Aff want = host_lincomb<FRP, FPP>(pts, ks, COUNT, nullptr);
    // the GLV pass (two half-length columns per point) against the plain full-length one, scalars at the edges included
    int glv_bad = 0;
    {
        Fr edge[COUNT];
        for (int i = 0; i < COUNT; i++) edge[i] = ks[i];
        edge[0] = Fr::zero(); edge[1] = Fr::one(); edge[2] = Fr::neg(Fr::one());
        Fr lam = Fr::zero();
        for (int w = 0; w < 4; w++) { lam.l[2 * w] = (uint32_t)GlvParams<FPP>::lambda[w]; lam.l[2 * w + 1] = (uint32_t)(GlvParams<FPP>::lambda[w] >> 32); }
        edge[3] = Fr::to_mont(lam); edge[4] = Fr::neg(edge[3]);
        Fr p128 = Fr::zero(); p128.l[4] = 1; edge[5] = Fr::to_mont(p128);
        Aff ep[COUNT];
        for (int i = 0; i < COUNT; i++) ep[i] = pts[i];
        ep[6] = Aff::inf(); ep[8] = ep[7];
        for (int c = 1; c <= COUNT; c++) {
            const Aff a = host_xyzz_to_affine<FPP>(host_lincomb_xyzz<FRP, FPP>(ep, edge, c, true));
            const Aff b = host_xyzz_to_affine<FPP>(host_lincomb_xyzz<FRP, FPP>(ep, edge, c, false));
            if (memcmp(&a, &b, sizeof a) != 0) glv_bad++;
        }
        // phi(P) = (beta x, y) = lambda P
        Fp bp;
        for (int w = 0; w < Fp::N; w++) bp.l[w] = GlvParams<FPP>::beta[w];
        const Aff lp = host_lincomb<FRP, FPP>(&pts[2], &edge[3], 1, nullptr);
        if (!(lp.x == Fp::to_mont(bp) * pts[2].x) || !(lp.y == pts[2].y)) glv_bad++;
        // the fixed-base tables (a context's [Ql][Qr][Qm][Qo][S3]) against the Straus pass: an infinity point, a repeated one,
        // the scalars 0 and r - 1 among them
        HostFixedBase<FPP> fb;
        const Aff fp5[5] = {ep[0], ep[1], Aff::inf(), ep[7], ep[7]};
        const Fr fk5[5] = {edge[7], edge[8], edge[9], edge[2], edge[0]};
        fb.build(fp5, 5);
        XYZZ<FPP, Fe64<FPP>> facc = XYZZ<FPP, Fe64<FPP>>::inf();
        fb.template accumulate<FRP>(facc, fk5);
        const Aff fa = host_xyzz_to_affine<FPP>(facc), fw = host_lincomb<FRP, FPP>(fp5, fk5, 5, nullptr);
        if (memcmp(&fa, &fw, sizeof fa) != 0) glv_bad++;
        printf("%s: GLV pass against the plain pass (11 prefixes), the endomorphism, fixed-base tables: %d mismatches\n", name, glv_bad);
    }
    HostPool pool(3);
    std::atomic<int> bad{0}, pooled{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 32; t++)
        th.emplace_back([&] {
            for (int r = 0; r < 6; r++) {
                const Aff got = host_lincomb<FRP, FPP>(pts, ks, COUNT, &pool);
                if (memcmp(&got, &want, sizeof got) != 0) bad++;
                pooled++;
            }
        });
    for (auto& t : th) t.join();
    printf("%s: %d combinations from 32 threads on a 3-worker pool, %d mismatches\n", name, pooled.load(), bad.load());
    return bad.load() + glv_bad;
}

static int hammer_gate(int slots) {
    // (on the heap: std::mutex has no destructor the sanitizer sees, and a later gate at the same STACK address would inherit this one's lock history)
    std::unique_ptr<SlotGate> gate_p(new SlotGate());
    SlotGate& gate = *gate_p;
    gate.resize((size_t)slots);
    std::vector<std::atomic<int>> owner((size_t)slots);
    for (auto& o : owner) o = 0;
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 32; t++)
        th.emplace_back([&, t] {
            for (int r = 0; r < 400; r++) {
                const size_t i = gate.acquire().slot;
                if (owner[i].fetch_add(1) != 0) bad++;          // two owners of one slot
                const int b = gate.busy();
                if (b < 1 || b > slots) bad++;
                if ((r + t) % 7 == 0) std::this_thread::yield();
                owner[i].fetch_sub(1);
                gate.release(i);
            }
        });
    for (auto& t : th) t.join();
    if (gate.busy() != 0) bad++;
    printf("slot gate, %d slots: 32 callers x 400 rounds, %d violations\n", slots, bad.load());
    return bad.load();
}

// the prover's use of the gate and the gang, with counters in place of launches
static int hammer_gangs(int gang_max) {
    constexpr int SLOTS = 32, STREAMS = 8, CALLERS = 48, ROUNDS = 150, MEETINGS = 6;
    std::unique_ptr<SlotGate> gate_p(new SlotGate());
    SlotGate& gate = *gate_p;
    gate.configure(SLOTS, STREAMS, gang_max, 200);
    std::vector<Gang> gangs(SLOTS);
    std::vector<std::atomic<int>> owner(SLOTS);
    for (auto& o : owner) o = 0;
    std::vector<std::atomic<int>> stream_owner(STREAMS);
    for (auto& o : stream_owner) o = 0;
    std::atomic<int> bad{0}, ganged{0}, launches{0}, served{0}, posted{0};
    struct Args { int member; int value; int out; };
    const Gang::Launcher launcher = [&](GangReq* const* reqs, int count) {
        launches++;
        int sum = 0;
        for (int i = 0; i < count; i++) sum += static_cast<Args*>(reqs[i]->args)->value;
        for (int i = 0; i < count; i++) { static_cast<Args*>(reqs[i]->args)->out = sum; reqs[i]->rc = count; served++; }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < CALLERS; t++)
        th.emplace_back([&, t] {
            for (int r = 0; r < ROUNDS; r++) {
                const SlotGate::Ticket tk = gate.acquire_member((r + t) % 9 != 0);       // now and then a caller that must not be ganged
                if (owner[tk.slot].fetch_add(1) != 0) bad++;
                if (gate.streams() > STREAMS || tk.stream < 0 || tk.stream >= STREAMS) bad++;
                if (tk.lead == tk.slot && stream_owner[tk.stream].fetch_add(1) != 0) bad++;      // two gangs on one stream
                if (tk.size < 1 || tk.size > gang_max || tk.idx < 0 || tk.idx >= tk.size || (tk.size == 1 && tk.lead != tk.slot)) bad++;
                if (tk.size > 1) {
                    ganged++;
                    Gang& g = gangs[tk.lead];
                    g.enter(tk.gen, tk.size);
                    const int quit_at = (t * 7 + r) % 5 == 0 ? (t + r) % MEETINGS : MEETINGS;     // one in five leaves early
                    for (int m = 0; m < quit_at; m++) {
                        Args a{tk.idx, 1 << tk.idx, 0};
                        GangReq q;
                        q.kind = 1; q.args = &a;
                        posted++;
                        const int rc = g.meet(tk.idx, q, launcher);
                        if (rc < 1 || rc > tk.size || !(a.out & (1 << tk.idx))) bad++;         // launched with at least this member's request
                    }
                    g.leave(tk.idx);
                    if (tk.lead == tk.slot) g.wait_empty();
                } else std::this_thread::sleep_for(std::chrono::microseconds(100));     // a lone "proof" takes a while too: 48 callers crowd 8 streams
                owner[tk.slot].fetch_sub(1);
                if (tk.lead == tk.slot) stream_owner[tk.stream].fetch_sub(1);
                gate.release(tk.slot);
            }
        });
    for (auto& t : th) t.join();
    if (gate.busy() != 0 || gate.streams() != 0) bad++;
    if (served.load() != posted.load()) bad++;                                                     // every request was served exactly once
    printf("gate + gangs of up to %d: %d callers x %d rounds on %d slots / %d streams, %d ganged proofs, %d requests in %d launches, %d violations\n",
           gang_max, CALLERS, ROUNDS, SLOTS, STREAMS, ganged.load(), posted.load(), launches.load(), bad.load());
    if (ganged.load() == 0) { printf("no gang ever formed\n"); return 1; }
    return bad.load();
}

int main() {
    int bad = 0;
    bad += hammer_gate(4);
    bad += hammer_gate(16);
    bad += hammer_gangs(2);
    bad += hammer_gangs(4);
    // generators: BN254 (1, 2); BLS12-381 G1 generator (canonical little-endian 32-bit words)
    static const uint32_t bn_x[8] = {1, 0, 0, 0, 0, 0, 0, 0}, bn_y[8] = {2, 0, 0, 0, 0, 0, 0, 0};
    static const uint32_t bls_x[12] = {0xdb22c6bb, 0xfb3af00a, 0xf97a1aef, 0x6c55e83f, 0x171bac58, 0xa14e3a3f, 0x9774b905, 0xc3688c4f,
                                       0x4fa9ac0f, 0x2695638c, 0x3197d794, 0x17f1d3a7};
    static const uint32_t bls_y[12] = {0x46c5e7e1, 0x0caa2329, 0xa2888ae4, 0xd03cc744, 0x2c04b3ed, 0x00db18cb, 0xd5d00af6, 0xfcf5e095,
                                       0x741d8ae4, 0xa09e30ed, 0xe3aaa0f1, 0x08b3f481};
    bad += hammer_lincomb<FrBN254, FpBN254>("BN254", bn_x, bn_y);
    bad += hammer_lincomb<FrBLS12381, FpBLS12381>("BLS12-381", bls_x, bls_y);
    printf(bad ? "SAN HAMMER FAILED\n" : "SAN HAMMER OK\n");
    return bad ? 1 : 0;
}
