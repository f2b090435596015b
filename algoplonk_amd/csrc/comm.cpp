// libapk's communicator: the exchange steps of the path behind the C-ABI (include/apk.h "multi-GPU behind the boundary").
//
//   control plane  TCP star through rank 0: rendezvous, headers, status words + 64/96-byte partial sums, barriers
//   data plane     RCCL (librccl dlopen'ed, libapk's own HIP runtime and stream): grouped ncclSend/ncclRecv for the scatter of
//                  scalar slices and for the peer copies of polynomials; or HIP IPC: the sender exports its staging buffer
//                  (hipIpcGetMemHandle), the receiver maps it and PULLS its bytes with one device-to-device copy (over xGMI between
//                  GPUs, on-device when two ranks share a GPU) - used when RCCL cannot be (ranks sharing a device, no librccl,
//                  APK_COMM_RCCL=0); or - CPU tier tests, IPC refused - the same bytes staged through the host and the TCP star
//   schedules      sharded MSM (BASELINE configs[3]); split proof: commitment batches dealt by index range, optional per-wire
//                  coset evaluations dealt by wire
//
// Host only (no kernels): every GPU touch point goes through `apk_compute`, whose built-in table calls the bound context and
// which the CPU tier replaces to run this same code in two processes without a GPU.
#include <arpa/inet.h>
#include <dlfcn.h>
#include <errno.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "backend.h"

struct apk_ctx {   // same layout as in apk_api.cpp
    apk::Backend* be;
    int curve;
};

namespace apk {

// ---- the deal: contiguous shares of a flattened (commitment, index) list ----------------------------------------------------
struct Seg { uint32_t k; uint64_t lo, hi; };
static void my_share(uint64_t total, int rank, int world, uint64_t& lo, uint64_t& hi) {
    const uint64_t base = total / world, rem = total % world;
    lo = (uint64_t)rank * base + ((uint64_t)rank < rem ? rank : rem);
    hi = lo + base + ((uint64_t)rank < rem ? 1 : 0);
}
static std::vector<Seg> deal(const uint32_t* lens, uint32_t count, int rank, int world) {
    uint64_t total = 0;
    for (uint32_t i = 0; i < count; i++) total += lens[i];
    uint64_t lo, hi, base = 0;
    my_share(total, rank, world, lo, hi);
    std::vector<Seg> out;
    for (uint32_t k = 0; k < count; k++) {
        const uint64_t a = lo > base ? lo : base, b = hi < base + lens[k] ? hi : base + lens[k];
        if (a < b) out.push_back({k, a - base, b - base});
        base += lens[k];
    }
    return out;
}

// ---- sockets -------------------------------------------------------------------------------------------------------------------
static int send_all(int fd, const void* buf, size_t n) {
    const uint8_t* p = (const uint8_t*)buf;
    while (n) {
        const ssize_t w = ::send(fd, p, n, MSG_NOSIGNAL);
        if (w < 0) { if (errno == EINTR) continue; set_error("comm: send failed: %s", strerror(errno)); return APK_ERR_STATE; }
        p += w; n -= (size_t)w;
    }
    return APK_OK;
}
static int recv_all(int fd, void* buf, size_t n) {
    uint8_t* p = (uint8_t*)buf;
    while (n) {
        const ssize_t r = ::recv(fd, p, n, 0);
        if (r == 0) { set_error("comm: peer closed the connection"); return APK_ERR_STATE; }
        if (r < 0) {
            if (errno == EINTR) continue;
            set_error("comm: recv failed: %s", errno == EAGAIN || errno == EWOULDBLOCK ? "timed out waiting for a peer" : strerror(errno));
            return APK_ERR_STATE;
        }
        p += r; n -= (size_t)r;
    }
    return APK_OK;
}
// An idle worker waits for the leader's next step as long as it takes (a proving service, host-side witness work): the failure
// timeout (APK_COMM_TIMEOUT_S, SO_RCVTIMEO) applies inside a step only, once its header has started to arrive.
static int wait_readable(int fd) {
    for (;;) {
        struct pollfd pf = {fd, POLLIN, 0};
        const int r = ::poll(&pf, 1, -1);
        if (r > 0) return APK_OK;                       // data, or a hang-up that the following recv reports
        if (r < 0 && errno != EINTR) { set_error("comm: poll failed: %s", strerror(errno)); return APK_ERR_STATE; }
    }
}
// FNV-1a of APK_COMM_TOKEN (0 when unset): a shared secret of one launch, exchanged in the hello so that a stray connection to
// rank 0's port is refused (algoplonk_amd/parallel.py draws it at random and hands it to the ranks through the rendezvous file)
static uint64_t launch_token() {
    const char* t = getenv("APK_COMM_TOKEN");
    uint64_t h = 0;
    if (t && *t) { h = 1469598103934665603ull; for (; *t; t++) { h ^= (uint8_t)*t; h *= 1099511628211ull; } if (!h) h = 1; }
    return h;
}
static void tune(int fd, int timeout_s) {
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    int big = 8 << 20;
    setsockopt(fd, SOL_SOCKET, SO_SNDBUF, &big, sizeof big);
    setsockopt(fd, SOL_SOCKET, SO_RCVBUF, &big, sizeof big);
    struct timeval tv = {timeout_s, 0};
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
}

// ---- RCCL, loaded on demand -----------------------------------------------------------------------------------------------------
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;      // optional: used when a collective timed out
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (h) return true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) { h = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (h) break; }
        if (!h) return false;
#define SYM(f, name) f = reinterpret_cast<decltype(f)>(dlsym(h, name)); if (!f) { dlclose(h); h = nullptr; return false; }
        SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
        SYM(CommCount, "ncclCommCount")
        SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(AllGather, "ncclAllGather") SYM(GroupStart, "ncclGroupStart")
        SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(h, "ncclCommAbort"));
        return true;
    }
};
static Rccl g_rccl;
#define NCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { set_error("comm: RCCL %s: %s", #x, g_rccl.GetErrorString(r_)); return APK_ERR_HIP; } } while (0)
#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("comm: HIP %s: %s", #x, hipGetErrorString(e_)); return APK_ERR_HIP; } } while (0)
#define CHK(x) do { int rc_ = (x); if (rc_ != APK_OK) return rc_; } while (0)

enum : int32_t { OP_COMMIT = 1, OP_WIRES = 2, OP_STOP = 3 };
struct Header { int32_t op, basis; uint32_t count; uint32_t lens[4]; };
struct Hello { int32_t rank; int32_t pad; uint64_t token; };

}  // namespace apk

using namespace apk;

struct apk_comm {
    int rank = 0, world = 1;
    int listen_fd = -1;
    std::vector<int> peer;          // rank 0: fd of every worker (index = rank); workers: peer[0] = fd to rank 0
    int timeout_s = 300;
    apk_ctx* ctx = nullptr;
    apk_compute cp{};               // effective table (built-in entries filled by bind)
    apk_compute user_cp{};          // what set_compute supplied
    bool have_user_cp = false;
    bool device_is_host = false;    // CPU tier: "device" pointers are host pointers
    // RCCL
    bool rccl = false;
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;
    int device = -1;
    // HIP IPC data plane: every rank exports ONE buffer (leader: d_stage, workers: d_wire_out); peers map it and pull
    bool ipc = false;
    uint32_t export_gen = 0;                             // bumped whenever the exported buffer is reallocated
    hipIpcMemHandle_t export_handle{};
    struct Mapping { uint32_t gen = 0; void* p = nullptr; };
    std::vector<Mapping> maps;                           // by peer rank
    std::vector<void*> retired;                          // exported buffers that were outgrown: freed at destroy (peers may map them)
    // grow-only staging
    void* d_stage = nullptr; size_t stage_cap = 0;       // leader: world x chunk; workers: chunk
    void* d_wire_in = nullptr; size_t wire_in_cap = 0;   // workers: a canonical polynomial
    void* d_wire_out = nullptr; size_t wire_out_cap = 0; // workers: its 4n evaluations
    std::vector<uint8_t> h_stage;
    void* d_ag = nullptr; size_t ag_cap = 0;             // RCCL: device staging of the partial-sum all-gather (hipMalloc on `device`)
    void* h_ag = nullptr;                                // ... and its page-locked host mirror (same capacity): no copy of the
                                                         // communicator's stream ever targets caller-owned memory
    bool step_synced = false;       // the current step's status reached every rank of it (serve() keeps serving) - or the transport
                                    // broke mid-step and the stream to the leader is no longer aligned (serve() leaves)
    bool split_on = false;
    bool spmd_on = false;           // replicated prover: this communicator's hooks sit on the bound context (apk_comm_spmd_begin)
    bool subcoset_on = false;       // replicated prover: round 3 on sub-cosets (apk_comm_spmd_begin)
    uint64_t steps = 0;
    std::string why_not_rccl = "not bound";   // apk_comm_transport_reason: what kept the data plane off RCCL ("" when it is RCCL)
    // where a schedule's time goes (apk_comm_phase_ms): this rank's share of the commitment MSMs, the partial sums' exchange, the
    // sub-coset all-gather - host wall clock, summed since the last reset
    double ms_msm = 0, ms_sums = 0, ms_gather = 0;
    uint64_t n_commit_rounds = 0, n_gathers = 0;
    std::mutex step_mu;             // leader: one step at a time (a context with several slots proves concurrently, and every
                                    // proving thread calls the hooks; the workers serve the steps in the order they are announced)

    int fd_of(int r) const { return rank == 0 ? peer[r] : peer[0]; }
};

// ---- built-in compute table: the bound context ------------------------------------------------------------------------------------
static int bi_msm(void* u, int basis, uint32_t count, const void* const* sc, const uint64_t* off, const uint64_t* len, void* out) {
    return apk_msm_g1_batch_device(((apk_comm*)u)->ctx, basis, count, sc, off, len, out);
}
static int bi_coset(void* u, const void* in, uint64_t len, void* out) { return apk_coset_ntt_device(((apk_comm*)u)->ctx, in, len, out); }
static int bi_alloc(void* u, size_t b, void** p) { return apk_device_alloc(((apk_comm*)u)->ctx, b, p); }
static int bi_release(void* u, void* p) {
    apk_comm* c = (apk_comm*)u;
    if (c->ctx && !ctx_alive(c->ctx)) c->ctx = nullptr;
    // (the context these buffers came through is gone - destroyed before its communicator: the allocation itself is plain device
    // memory and is returned to the runtime directly)
    if (!c->ctx) { const hipError_t e = hipFree(p); if (e != hipSuccess) (void)hipGetLastError(); return APK_OK; }
    return apk_device_free(c->ctx, p);
}
static int bi_copy(void* u, void* d, const void* s, size_t b, int kind) {
    apk_ctx* c = ((apk_comm*)u)->ctx;
    return kind == 0 ? apk_device_copy(c, d, s, b) : kind == 1 ? apk_device_upload(c, d, s, b) : apk_device_download(c, d, s, b);
}

static void resolve_compute(apk_comm* c) {
    apk_compute t{};
    t.user = c; t.msm_batch = bi_msm; t.coset_ntt = bi_coset; t.alloc = bi_alloc; t.release = bi_release; t.copy = bi_copy;
    t.g1_bytes = c->ctx ? apk_g1_bytes(c->ctx->curve) : 0;
    t.n = c->ctx ? c->ctx->be->domain_size() : 0;
    c->device_is_host = false;
    if (c->have_user_cp) {
        const apk_compute& u = c->user_cp;
        // a replaced entry runs with the table's own user pointer; mixing built-in and replaced entries is allowed
        if (u.msm_batch) t.msm_batch = u.msm_batch;
        if (u.coset_ntt) t.coset_ntt = u.coset_ntt;
        if (u.alloc) t.alloc = u.alloc;
        if (u.release) t.release = u.release;
        if (u.copy) { t.copy = u.copy; c->device_is_host = true; }
        if (u.g1_bytes) t.g1_bytes = u.g1_bytes;
        if (u.n) t.n = u.n;
    }
    c->cp = t;
}
// the user pointer an entry runs with: replaced entries get the table's, built-in ones the communicator
#define CP_USER(c, field) ((c)->have_user_cp && (c)->user_cp.field ? (c)->user_cp.user : (void*)(c))

// everything that was allocated through the bound context (or the compute table) goes back to it: before the binding changes and
// before the communicator dies - the context must still be alive then (apk_comm_bind(comm, NULL) before apk_ctx_destroy)
static void release_buffers(apk_comm* c) {
    for (auto& mp : c->maps) if (mp.p) { (void)hipIpcCloseMemHandle(mp.p); mp.p = nullptr; mp.gen = 0; }
    if (c->cp.release) {
        void** bufs[3] = {&c->d_stage, &c->d_wire_in, &c->d_wire_out};
        size_t* caps[3] = {&c->stage_cap, &c->wire_in_cap, &c->wire_out_cap};
        for (int i = 0; i < 3; i++) if (*bufs[i]) { (void)c->cp.release(CP_USER(c, release), *bufs[i]); *bufs[i] = nullptr; *caps[i] = 0; }
        for (void* q : c->retired) (void)c->cp.release(CP_USER(c, release), q);
    }
    c->retired.clear();
}

static int ensure(apk_comm* c, void** p, size_t* cap, size_t need) {
    if (*cap >= need && *p) return APK_OK;
    const bool exported = c->ipc && p == (c->rank == 0 ? &c->d_stage : &c->d_wire_out);
    if (*p) {
        if (exported) c->retired.push_back(*p);          // a peer may still have it mapped
        else (void)c->cp.release(CP_USER(c, release), *p);
        *p = nullptr; *cap = 0;
    }
    const size_t want = exported ? need * 2 + 4096 : need + need / 8 + 256;
    CHK(c->cp.alloc(CP_USER(c, alloc), want, p));
    *cap = want;
    if (exported) {
        HCHK(hipSetDevice(c->device));
        HCHK(hipIpcGetMemHandle(&c->export_handle, *p));
        c->export_gen++;
    }
    return APK_OK;
}

// ---- HIP IPC transfers: {generation, handle, offset, bytes} over the control plane, one device-to-device pull, one ack ----
struct IpcMsg { uint32_t gen, pad; uint64_t offset, bytes, cap; hipIpcMemHandle_t h; };   // cap: size of the exported allocation
static int ipc_offer(apk_comm* c, int fd, size_t offset, size_t bytes) {      // the exporter's side
    IpcMsg m{};
    m.gen = c->export_gen; m.offset = offset; m.bytes = bytes; m.h = c->export_handle;
    m.cap = c->rank == 0 ? c->stage_cap : c->wire_out_cap;
    return send_all(fd, &m, sizeof m);
}
static int ipc_wait_ack(int fd) {
    int32_t st = APK_OK;
    CHK(recv_all(fd, &st, 4));
    if (st != APK_OK) { set_error("comm: the peer could not pull from the exported buffer (code %d)", st); return APK_ERR_HIP; }
    return APK_OK;
}
static int ipc_pull(apk_comm* c, int from_rank, int fd, void* d_dst, size_t expect) {   // the importer's side
    IpcMsg m{};
    CHK(recv_all(fd, &m, sizeof m));
    int32_t st = APK_OK;
    hipError_t e = hipSetDevice(c->device);
    apk_comm::Mapping& mp = c->maps[from_rank];
    if (e == hipSuccess && (mp.gen != m.gen || !mp.p)) {
        if (mp.p) (void)hipIpcCloseMemHandle(mp.p);
        mp.p = nullptr;
        e = hipIpcOpenMemHandle(&mp.p, m.h, hipIpcMemLazyEnablePeerAccess);
        mp.gen = m.gen;
    }
    if (e == hipSuccess && m.bytes != expect) { set_error("comm: peer offered %llu bytes, %llu expected", (unsigned long long)m.bytes, (unsigned long long)expect); st = APK_ERR_STATE; }
    if (e == hipSuccess && (m.offset > m.cap || m.bytes > m.cap - m.offset)) { set_error("comm: peer's offer leaves its exported buffer (%llu + %llu of %llu bytes)", (unsigned long long)m.offset, (unsigned long long)m.bytes, (unsigned long long)m.cap); st = APK_ERR_STATE; }
    // hipMemcpy device-to-device returns before the copy is done and the contexts' streams are non-blocking (they do not order
    // themselves behind the null stream): copy on the communicator's own stream and wait for it
    if (e == hipSuccess && st == APK_OK) e = hipMemcpyAsync(d_dst, (const uint8_t*)mp.p + m.offset, m.bytes, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess && st == APK_OK) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("comm: IPC pull failed: %s", hipGetErrorString(e)); st = APK_ERR_HIP; }
    CHK(send_all(fd, &st, 4));
    return st;
}

// ---- control-plane collectives (host memory) ------------------------------------------------------------------------------------
static int ctl_bcast(apk_comm* c, void* buf, size_t n) {
    if (c->world == 1) return APK_OK;
    if (c->rank == 0) { for (int r = 1; r < c->world; r++) CHK(send_all(c->peer[r], buf, n)); return APK_OK; }
    return recv_all(c->peer[0], buf, n);
}
static int ctl_allgather(apk_comm* c, const void* mine, void* all, size_t n) {
    memcpy((uint8_t*)all + (size_t)c->rank * n, mine, n);
    if (c->world == 1) return APK_OK;
    if (c->rank == 0) {
        for (int r = 1; r < c->world; r++) CHK(recv_all(c->peer[r], (uint8_t*)all + (size_t)r * n, n));
        for (int r = 1; r < c->world; r++) CHK(send_all(c->peer[r], all, n * c->world));
        return APK_OK;
    }
    CHK(send_all(c->peer[0], mine, n));
    return recv_all(c->peer[0], all, n * c->world);
}

struct WallMs {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// Whatever this communicator installed on its bound context comes off again - on EVERY rank, for either schedule - before the
// binding changes or the communicator dies: a hook left behind points at freed memory and the next apk_prove on that context
// would call it (an exception between spmd_begin and spmd_end is enough to get there).
static void clear_ctx_hooks(apk_comm* c) {
    // (a host whose finalizers destroyed the context before the communicator: nothing left to take the hooks off)
    if (c->ctx && !ctx_alive(c->ctx)) c->ctx = nullptr;
    if (c->ctx && (c->split_on || c->spmd_on)) {
        (void)apk_ctx_set_commit_hook(c->ctx, nullptr, nullptr);
        (void)apk_ctx_set_wire_hook(c->ctx, nullptr, nullptr);
        (void)apk_ctx_set_subcoset(c->ctx, 0, 1, nullptr, nullptr);
    }
    c->spmd_on = false;
    c->subcoset_on = false;
}

// Wait for the communicator's stream with the control plane's deadline (APK_COMM_TIMEOUT_S).  A collective that a peer never
// joins - it failed locally on the way in - would otherwise hold hipStreamSynchronize forever; here the call fails after the
// timeout like a silent peer on the TCP plane does, and the RCCL plane is given up for the rest of the binding (its stream
// still holds the unfinished collective).
static int stream_wait(apk_comm* c, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; spins++) {
        const hipError_t e = hipStreamQuery(c->stream);
        if (e == hipSuccess) return APK_OK;
        if (e != hipErrorNotReady) { (void)hipGetLastError(); set_error("comm: %s: %s", what, hipGetErrorString(e)); return APK_ERR_HIP; }
        if (spins < 2000) continue;                      // the exchanges this guards take tens of microseconds
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > (double)c->timeout_s) {
            // The stream still holds the unfinished collective and whatever was queued behind it.  The collective is aborted; the
            // stream and the staging buffers it may still write to are ABANDONED (deliberately leaked, never reused or freed) and a
            // fresh stream takes over, so a peer that joins late cannot make anything write into memory with a new owner.
            c->rccl = false;
            if (c->nccl && g_rccl.CommAbort) (void)g_rccl.CommAbort(c->nccl);
            c->nccl = nullptr;
            c->d_ag = nullptr; c->h_ag = nullptr; c->ag_cap = 0;
            c->stream = nullptr;
            if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->stream = nullptr; }
            set_error("comm: %s did not complete within %d s (a peer never joined the collective); RCCL plane abandoned", what, c->timeout_s);
            return APK_ERR_STATE;
        }
        usleep(spins < 20000 ? 20 : 500);
    }
}

// The ONE exchange of a sharded MSM / a dealt commitment batch: `n` bytes (status word + 64/96-byte partial sums) from every
// rank to every rank.  north_star: "RCCL-over-xGMI ... for the final bucket-sum of a single MSM" - ncclAllGather on the
// communicator's stream when RCCL is the data plane (a numeric all-reduce cannot add curve points: the ranks add the gathered
// points themselves, apk_g1_sum); the TCP star otherwise (CPU tier, ranks sharing a GPU, APK_COMM_RCCL=0).
static int sums_allgather(apk_comm* c, const void* mine, void* all, size_t n) {
    if (!c->rccl || c->world == 1) return ctl_allgather(c, mine, all, n);
    HCHK(hipSetDevice(c->device));
    const size_t need = n * ((size_t)c->world + 1);
    if (c->ag_cap < need) {
        // the staging buffer grows on every rank at the same call (need depends on n and the world only): the one local step that
        // can fail before the collective, so the ranks agree on it over the control plane first - nobody enters an all-gather
        // that a peer cannot join
        if (c->d_ag) (void)hipFree(c->d_ag);
        if (c->h_ag) (void)hipHostFree(c->h_ag);
        c->d_ag = nullptr; c->h_ag = nullptr; c->ag_cap = 0;
        int32_t ok = hipMalloc(&c->d_ag, need * 2 + 4096) == hipSuccess && hipHostMalloc(&c->h_ag, need * 2 + 4096, hipHostMallocDefault) == hipSuccess ? 1 : 0;
        if (!ok) { (void)hipGetLastError(); if (c->d_ag) (void)hipFree(c->d_ag); if (c->h_ag) (void)hipHostFree(c->h_ag); c->d_ag = nullptr; c->h_ag = nullptr; }
        std::vector<int32_t> oks(c->world);
        CHK(ctl_allgather(c, &ok, oks.data(), 4));
        for (int r = 0; r < c->world; r++)
            if (!oks[r]) {
                if (c->d_ag) { (void)hipFree(c->d_ag); c->d_ag = nullptr; }
                if (c->h_ag) { (void)hipHostFree(c->h_ag); c->h_ag = nullptr; }
                set_error("comm: rank %d could not allocate the all-gather staging buffer (%zu bytes)", r, need * 2 + 4096);
                return APK_ERR_HIP;
            }
        c->ag_cap = need * 2 + 4096;
    }
    uint8_t* d_all = (uint8_t*)c->d_ag;
    uint8_t* d_mine = d_all + n * (size_t)c->world;
    // through the communicator's own page-locked mirror: should the wait below time out, the copy still queued on the (then
    // abandoned) stream targets memory nobody else will own, not the caller's vector
    uint8_t* h_all = (uint8_t*)c->h_ag;
    uint8_t* h_mine = h_all + n * (size_t)c->world;
    memcpy(h_mine, mine, n);
    HCHK(hipMemcpyAsync(d_mine, h_mine, n, hipMemcpyHostToDevice, c->stream));
    NCHK(g_rccl.AllGather(d_mine, d_all, n, ncclUint8, c->nccl, c->stream));
    HCHK(hipMemcpyAsync(h_all, d_all, n * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
    CHK(stream_wait(c, "the partial sums' ncclAllGather"));
    memcpy(all, h_all, n * (size_t)c->world);
    return APK_OK;
}

// ---- data-plane: scatter of per-rank chunks held by rank 0; peer copies between rank 0 and one worker -------------------------
// d_all (rank 0): world x chunk bytes, rank r's chunk at r * chunk; d_mine: chunk bytes on every rank (rank 0: unused, its share
// is read in place).
static int data_scatter(apk_comm* c, const void* d_all, void* d_mine, size_t chunk) {
    if (c->world == 1 || chunk == 0) return APK_OK;
    if (c->rccl) {
        HCHK(hipSetDevice(c->device));
        NCHK(g_rccl.GroupStart());
        if (c->rank == 0) {
            for (int r = 1; r < c->world; r++) NCHK(g_rccl.Send((const uint8_t*)d_all + (size_t)r * chunk, chunk, ncclUint8, r, c->nccl, c->stream));
        } else {
            NCHK(g_rccl.Recv(d_mine, chunk, ncclUint8, 0, c->nccl, c->stream));
        }
        NCHK(g_rccl.GroupEnd());
        return stream_wait(c, "the scatter's ncclSend / ncclRecv group");
    }
    if (c->ipc) {
        // d_all is the leader's exported staging buffer: every worker pulls its chunk at once, the leader collects the acks
        if (c->rank == 0) {
            for (int r = 1; r < c->world; r++) CHK(ipc_offer(c, c->peer[r], (size_t)r * chunk, chunk));
            int rc = APK_OK;
            for (int r = 1; r < c->world; r++) { const int a = ipc_wait_ack(c->peer[r]); if (a != APK_OK) rc = a; }
            return rc;
        }
        return ipc_pull(c, 0, c->peer[0], d_mine, chunk);
    }
    if (c->h_stage.size() < chunk) c->h_stage.resize(chunk);
    if (c->rank == 0) {
        for (int r = 1; r < c->world; r++) {
            CHK(c->cp.copy(CP_USER(c, copy), c->h_stage.data(), (const uint8_t*)d_all + (size_t)r * chunk, chunk, 2));
            CHK(send_all(c->peer[r], c->h_stage.data(), chunk));
        }
        return APK_OK;
    }
    CHK(recv_all(c->peer[0], c->h_stage.data(), chunk));
    return c->cp.copy(CP_USER(c, copy), d_mine, c->h_stage.data(), chunk, 1);
}
// rank 0 <-> worker w: `to_worker` says which way the bytes flow.  Called on rank 0 and on w only.
static int data_p2p(apk_comm* c, int w, bool to_worker, void* d_buf, size_t bytes) {
    const bool sending = (c->rank == 0) == to_worker;
    if (c->rccl) {
        HCHK(hipSetDevice(c->device));
        const int other = c->rank == 0 ? w : 0;
        NCHK(g_rccl.GroupStart());
        if (sending) NCHK(g_rccl.Send(d_buf, bytes, ncclUint8, other, c->nccl, c->stream));
        else NCHK(g_rccl.Recv(d_buf, bytes, ncclUint8, other, c->nccl, c->stream));
        NCHK(g_rccl.GroupEnd());
        return stream_wait(c, "a peer copy's ncclSend / ncclRecv");
    }
    if (c->ipc) {
        const int fd = c->fd_of(w);
        if (!sending) return ipc_pull(c, c->rank == 0 ? w : 0, fd, d_buf, bytes);
        // the leader's payload is a buffer of the prover: copied into the exported staging buffer first; a worker's payload IS its
        // exported buffer (d_wire_out)
        size_t off = 0;
        if (c->rank == 0) {
            CHK(ensure(c, &c->d_stage, &c->stage_cap, bytes));
            CHK(c->cp.copy(CP_USER(c, copy), c->d_stage, d_buf, bytes, 0));
        } else if (d_buf != c->d_wire_out) {
            set_error("comm: a worker sends from its exported buffer only");
            return APK_ERR_STATE;
        }
        CHK(ipc_offer(c, fd, off, bytes));
        return ipc_wait_ack(fd);
    }
    if (c->h_stage.size() < bytes) c->h_stage.resize(bytes);
    const int fd = c->fd_of(w);
    if (sending) {
        CHK(c->cp.copy(CP_USER(c, copy), c->h_stage.data(), d_buf, bytes, 2));
        return send_all(fd, c->h_stage.data(), bytes);
    }
    CHK(recv_all(fd, c->h_stage.data(), bytes));
    return c->cp.copy(CP_USER(c, copy), d_buf, c->h_stage.data(), bytes, 1);
}

// ---- schedules ---------------------------------------------------------------------------------------------------------------------
// One commitment batch, every rank.  d_scalars only on rank 0.  out_points (count affine points) on rank 0.
// `local` (the replicated prover, apk_comm_spmd_begin): EVERY rank holds the scalar vectors (it ran the same prover on the same
// inputs), so nothing is scattered - each rank commits its index range straight from its own memory - and every rank adds the
// gathered partial sums itself: the same `count` points come back on all ranks, and the transcripts stay in lockstep.
static int commit_round(apk_comm* c, int basis, uint32_t count, const uint32_t* lens, const void* const* d_scalars, void* out_points,
                        bool local = false) {
    const size_t nb = c->cp.g1_bytes;
    uint64_t total = 0;
    c->step_synced = false;
    for (uint32_t i = 0; i < count; i++) total += lens[i];
    const size_t chunk = (size_t)((total + c->world - 1) / c->world) * APK_FR_BYTES;
    // rank 0 reads its own share in place; the workers' slices are packed into the staging buffer and scattered
    if (c->world > 1 && !local) {
        CHK(ensure(c, &c->d_stage, &c->stage_cap, c->rank == 0 ? chunk * c->world : chunk));
        if (c->rank == 0)
            for (int r = 1; r < c->world; r++) {
                size_t at = (size_t)r * chunk;
                for (const Seg& s : deal(lens, count, r, c->world)) {
                    const size_t bytes = (size_t)(s.hi - s.lo) * APK_FR_BYTES;
                    CHK(c->cp.copy(CP_USER(c, copy), (uint8_t*)c->d_stage + at, (const uint8_t*)d_scalars[s.k] + s.lo * APK_FR_BYTES, bytes, 0));
                    at += bytes;
                }
            }
        CHK(data_scatter(c, c->d_stage, c->d_stage, chunk));
    }
    // this rank's partial sums: status word + count points (infinity = all zero where the rank holds no part of a commitment)
    const std::vector<Seg> segs = deal(lens, count, c->rank, c->world);
    const size_t rec = 8 + (size_t)count * nb;
    std::vector<uint8_t> mine(rec, 0), all(rec * c->world, 0);
    int32_t status = APK_OK;
    std::string err;
    if (!segs.empty()) {
        const void* ptrs[4]; uint64_t offs[4], ls[4];
        size_t at = 0;
        for (size_t i = 0; i < segs.size(); i++) {
            const Seg& s = segs[i];
            ptrs[i] = (c->rank == 0 || local) ? (const uint8_t*)d_scalars[s.k] + s.lo * APK_FR_BYTES : (const uint8_t*)c->d_stage + at;
            offs[i] = s.lo; ls[i] = s.hi - s.lo;
            at += (size_t)(s.hi - s.lo) * APK_FR_BYTES;
        }
        std::vector<uint8_t> res(segs.size() * nb);
        const WallMs t_msm;
        status = c->cp.msm_batch(CP_USER(c, msm_batch), basis, (uint32_t)segs.size(), ptrs, offs, ls, res.data());
        c->ms_msm += t_msm.ms();
        if (status == APK_OK) for (size_t i = 0; i < segs.size(); i++) memcpy(mine.data() + 8 + segs[i].k * nb, res.data() + i * nb, nb);
        else err = apk_last_error();
    }
    memcpy(mine.data(), &status, 4);
    {
        const WallMs t_x;
        const int xrc = sums_allgather(c, mine.data(), all.data(), rec);
        c->ms_sums += t_x.ms();
        c->n_commit_rounds++;
        CHK(xrc);
    }
    c->step_synced = true;                     // from here on every rank knows how the step ended
    for (int r = 0; r < c->world; r++) {
        int32_t st;
        memcpy(&st, all.data() + (size_t)r * rec, 4);
        if (st != APK_OK) { set_error("comm: rank %d failed its share of the commitment batch (code %d)%s%s", r, st, r == c->rank ? ": " : "", r == c->rank ? err.c_str() : ""); return st; }
    }
    c->steps++;
    if (c->rank != 0 && !local) return APK_OK;
    std::vector<uint8_t> col((size_t)c->world * nb);
    for (uint32_t k = 0; k < count; k++) {
        for (int r = 0; r < c->world; r++) memcpy(col.data() + (size_t)r * nb, all.data() + (size_t)r * rec + 8 + k * nb, nb);
        CHK(apk_g1_sum(c->ctx ? c->ctx->curve : (nb == 64 ? APK_BN254 : APK_BLS12_381), col.data(), c->world, (uint8_t*)out_points + k * nb));
    }
    return APK_OK;
}

// The 4n-coset evaluations of `count` canonical polynomials, polynomial i on rank i mod world.
// The wires go out in GROUPS of `world` (one per rank): the leader sends a group's dealt polynomials, transforms its own, collects
// the group's results, then turns to the next group - so a worker that owns two wires (world = 2, count = 4: wires 1 and 3) has
// answered the first before the second arrives.  (Sending every dealt wire before collecting any - the first version - crossed
// the worker's status word with the leader's next payload on the IPC plane and deadlocked Send against Send on RCCL.)
static int wires_round(apk_comm* c, uint32_t count, const uint32_t* lens, const void* const* d_can, void* const* d_ev) {
    const size_t ev_bytes = (size_t)4 * c->cp.n * APK_FR_BYTES;
    const uint32_t W = (uint32_t)c->world;
    int32_t status = APK_OK;
    c->step_synced = false;
    if (c->rank == 0) {
        for (uint32_t base = 0; base < count; base += W) {
            const uint32_t end = base + W < count ? base + W : count;
            for (uint32_t i = base; i < end; i++)
                if (i % W) CHK(data_p2p(c, i % W, true, const_cast<void*>(d_can[i]), (size_t)lens[i] * APK_FR_BYTES));
            for (uint32_t i = base; i < end; i++)
                if (i % W == 0 && status == APK_OK) status = c->cp.coset_ntt(CP_USER(c, coset_ntt), d_can[i], lens[i], d_ev[i]);
            for (uint32_t i = base; i < end; i++)
                if (i % W) {
                    int32_t st = APK_OK;
                    CHK(recv_all(c->peer[i % W], &st, 4));
                    if (st != APK_OK) { if (status == APK_OK) { status = st; set_error("comm: rank %u failed the coset evaluation of wire %u (code %d)", i % W, i, st); } continue; }
                    CHK(data_p2p(c, i % W, false, d_ev[i], ev_bytes));
                }
        }
    } else {
        for (uint32_t i = 0; i < count; i++)
            if ((int)(i % W) == c->rank) {
                CHK(ensure(c, &c->d_wire_in, &c->wire_in_cap, (size_t)lens[i] * APK_FR_BYTES));
                CHK(ensure(c, &c->d_wire_out, &c->wire_out_cap, ev_bytes));
                CHK(data_p2p(c, c->rank, true, c->d_wire_in, (size_t)lens[i] * APK_FR_BYTES));
                int32_t st = c->cp.coset_ntt(CP_USER(c, coset_ntt), c->d_wire_in, lens[i], c->d_wire_out);
                CHK(send_all(c->peer[0], &st, 4));
                if (st == APK_OK) CHK(data_p2p(c, c->rank, false, c->d_wire_out, ev_bytes));
                else status = st;
            }
    }
    c->step_synced = true;                     // every transfer of the step completed: a failed transform was reported to the leader
    c->steps++;
    return status;
}

static int hook_commit(void* u, int basis, uint32_t count, const void* const* d_scalars, const uint32_t* lens, void* out_points) {
    return apk_comm_commit((apk_comm*)u, basis, count, d_scalars, lens, out_points);
}
static int hook_commit_local(void* u, int basis, uint32_t count, const void* const* d_scalars, const uint32_t* lens, void* out_points) {
    return apk_comm_commit_local((apk_comm*)u, basis, count, d_scalars, lens, out_points);
}
static int hook_gather(void* u, void* d_all, size_t bytes_per_rank) { return apk_comm_allgather_device((apk_comm*)u, d_all, bytes_per_rank); }
static int hook_wires(void* u, uint32_t count, const void* const* d_can, const uint32_t* lens, void* const* d_ev) {
    return apk_comm_wires((apk_comm*)u, count, d_can, lens, d_ev);
}

extern "C" {

int apk_comm_create(int rank, int world, const char* addr, int port, apk_comm** out) {
    if (!out || world < 1 || rank < 0 || rank >= world || world > 64) { set_error("comm: bad rank / world"); return APK_ERR_ARG; }
    apk_comm* c = new apk_comm();
    c->rank = rank; c->world = world;
    c->timeout_s = env_int("APK_COMM_TIMEOUT_S", 300, 1, 86400);
    if (world == 1) { *out = c; return APK_OK; }
    if (!addr || port <= 0 || port > 65535) { delete c; set_error("comm: address / port"); return APK_ERR_ARG; }
    struct sockaddr_in sa{};
    sa.sin_family = AF_INET; sa.sin_port = htons((uint16_t)port);
    if (inet_pton(AF_INET, addr, &sa.sin_addr) != 1) { delete c; set_error("comm: '%s' is not an IPv4 address", addr); return APK_ERR_ARG; }
    if (rank == 0) {
        c->peer.assign(world, -1);
        c->listen_fd = socket(AF_INET, SOCK_STREAM, 0);
        int one = 1;
        setsockopt(c->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        if (bind(c->listen_fd, (struct sockaddr*)&sa, sizeof sa) != 0 || listen(c->listen_fd, world) != 0) {
            set_error("comm: cannot listen on %s:%d: %s", addr, port, strerror(errno));
            apk_comm_destroy(c);
            return APK_ERR_STATE;
        }
        struct timeval tv = {c->timeout_s, 0};
        setsockopt(c->listen_fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        for (int i = 1; i < world; i++) {
            const int fd = accept(c->listen_fd, nullptr, nullptr);
            if (fd < 0) { set_error("comm: rank 0 waited %d s for %d more rank(s): %s", c->timeout_s, world - i, strerror(errno)); apk_comm_destroy(c); return APK_ERR_STATE; }
            tune(fd, c->timeout_s);
            Hello hi{};
            if (recv_all(fd, &hi, sizeof hi) != APK_OK || hi.rank <= 0 || hi.rank >= world || c->peer[hi.rank] != -1 || hi.token != launch_token()) {
                close(fd); set_error("comm: bad hello from a peer (rank out of range or taken, or another launch's APK_COMM_TOKEN)"); apk_comm_destroy(c); return APK_ERR_STATE;
            }
            c->peer[hi.rank] = fd;
        }
    } else {
        c->peer.assign(1, -1);
        const double deadline = (double)c->timeout_s;
        double waited = 0;
        for (;;) {
            const int fd = socket(AF_INET, SOCK_STREAM, 0);
            if (connect(fd, (struct sockaddr*)&sa, sizeof sa) == 0) { c->peer[0] = fd; break; }
            close(fd);
            if (waited >= deadline) { set_error("comm: rank %d could not reach rank 0 at %s:%d: %s", rank, addr, port, strerror(errno)); apk_comm_destroy(c); return APK_ERR_STATE; }
            usleep(50 * 1000); waited += 0.05;
        }
        tune(c->peer[0], c->timeout_s);
        Hello hi{};
        hi.rank = rank; hi.token = launch_token();
        if (send_all(c->peer[0], &hi, sizeof hi) != APK_OK) { apk_comm_destroy(c); return APK_ERR_STATE; }
    }
    *out = c;
    return APK_OK;
}

void apk_comm_destroy(apk_comm* c) {
    if (!c) return;
    if (c->split_on && c->rank == 0) (void)apk_comm_split_end(c);
    clear_ctx_hooks(c);
    release_buffers(c);
    if (c->d_ag) { (void)hipSetDevice(c->device); (void)hipFree(c->d_ag); }
    if (c->h_ag) (void)hipHostFree(c->h_ag);
    if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (int fd : c->peer) if (fd >= 0) close(fd);
    if (c->listen_fd >= 0) close(c->listen_fd);
    delete c;
}

int apk_comm_rank(const apk_comm* c) { return c ? c->rank : -1; }
int apk_comm_world(const apk_comm* c) { return c ? c->world : 0; }
const char* apk_comm_transport(const apk_comm* c) { return c && c->rccl ? "rccl" : c && c->ipc ? "ipc" : "tcp"; }

int apk_comm_barrier(apk_comm* c) {
    if (!c) { set_error("null communicator"); return APK_ERR_ARG; }
    uint8_t one = 1;
    std::vector<uint8_t> all(c->world);
    return ctl_allgather(c, &one, all.data(), 1);
}

int apk_comm_max_f64(apk_comm* c, double* v) {
    if (!c || !v) { set_error("null argument"); return APK_ERR_ARG; }
    std::vector<double> all(c->world);
    CHK(ctl_allgather(c, v, all.data(), sizeof(double)));
    for (double x : all) if (x > *v) *v = x;
    return APK_OK;
}

int apk_comm_set_compute(apk_comm* c, const apk_compute* t) {
    if (!c) { set_error("null communicator"); return APK_ERR_ARG; }
    c->have_user_cp = t != nullptr;
    if (t) c->user_cp = *t;
    resolve_compute(c);
    return APK_OK;
}

int apk_comm_bind(apk_comm* c, apk_ctx* ctx) {
    if (!c) { set_error("null communicator"); return APK_ERR_ARG; }
    if (c->split_on && c->rank == 0) (void)apk_comm_split_end(c);   // rebinding ends a split proof: the workers leave apk_comm_serve
    clear_ctx_hooks(c);                       // this communicator's hooks must not outlive its binding to the old context
    release_buffers(c);                       // they belong to the previous binding
    c->ipc = false;
    c->ctx = ctx;
    resolve_compute(c);
    if (!ctx && !(c->have_user_cp && c->user_cp.msm_batch)) { c->why_not_rccl = "not bound"; return APK_OK; }     // unbound: call before the context is destroyed
    // data plane: RCCL when every rank owns a different GPU; the ranks agree through the control plane
    c->rccl = false;
    int32_t dev = -1;
    if (ctx && !c->device_is_host) dev = ctx->be->device_ordinal();
    int32_t can = (c->world > 1 && dev >= 0 && env_int("APK_COMM_RCCL", 1, 0, 1) && g_rccl.load()) ? 1 : 0;
    c->why_not_rccl = c->world == 1 ? "single rank" : dev < 0 ? "no device memory behind this rank (host compute table)" :
                      !env_int("APK_COMM_RCCL", 1, 0, 1) ? "APK_COMM_RCCL=0" : !can ? "librccl.so could not be loaded" : "";
    std::vector<int32_t> devs(c->world), cans(c->world);
    CHK(ctl_allgather(c, &dev, devs.data(), 4));
    CHK(ctl_allgather(c, &can, cans.data(), 4));
    bool all_can = c->world > 1;
    for (int r = 0; r < c->world; r++) {
        if (!cans[r] && c->why_not_rccl.empty()) { char b[96]; snprintf(b, sizeof b, "rank %d cannot use RCCL", r); c->why_not_rccl = b; }
        all_can = all_can && cans[r];
        for (int q = 0; q < r; q++)
            if (devs[q] == devs[r]) {   // RCCL refuses two ranks on one device (single node)
                all_can = false;
                if (c->why_not_rccl.empty()) { char b[96]; snprintf(b, sizeof b, "ranks %d and %d share device %d", q, r, devs[r]); c->why_not_rccl = b; }
            }
    }
    if (all_can && !c->nccl) {
        // Nothing here may strand a peer: a rank that fails a step still takes part in the agreement that follows it, and one
        // refusal sends every rank to the next tier (IPC, then the TCP star) instead of failing the bind.
        ncclUniqueId id{};
        int32_t ok = 1;
        if (c->rank == 0 && g_rccl.GetUniqueId(&id) != ncclSuccess) ok = 0;
        CHK(ctl_bcast(c, &ok, 4));
        if (ok) {
            CHK(ctl_bcast(c, &id, sizeof id));
            c->device = dev;
            if (hipSetDevice(dev) != hipSuccess || (!c->stream && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)) { (void)hipGetLastError(); ok = 0; }
            std::vector<int32_t> oks(c->world);
            CHK(ctl_allgather(c, &ok, oks.data(), 4));
            for (int32_t v : oks) ok = ok && v;
            if (ok) {
                const ncclResult_t r = g_rccl.CommInitRank(&c->nccl, c->world, id, c->rank);
                ok = r == ncclSuccess && c->nccl ? 1 : 0;
                if (!ok) set_error("comm: ncclCommInitRank failed on rank %d: %s", c->rank, g_rccl.GetErrorString(r));
                CHK(ctl_allgather(c, &ok, oks.data(), 4));
                for (int32_t v : oks) ok = ok && v;
            }
            if (ok) {
                // first use before first need: the partial-sum exchange itself, on rank numbers
                c->rccl = true;
                std::vector<int32_t> ids(c->world, -1);
                const int32_t me = c->rank;
                int32_t good = sums_allgather(c, &me, ids.data(), 4) == APK_OK ? 1 : 0;
                for (int r = 0; r < c->world && good; r++) good = ids[r] == r;
                c->rccl = false;
                CHK(ctl_allgather(c, &good, oks.data(), 4));
                for (int32_t v : oks) ok = ok && v;
            }
            if (!ok && c->nccl) { (void)g_rccl.CommDestroy(c->nccl); c->nccl = nullptr; }
        }
        all_can = ok != 0;
        if (!ok && c->why_not_rccl.empty()) c->why_not_rccl = "RCCL bring-up failed on some rank (unique id, stream, ncclCommInitRank or the first all-gather)";
    }
    c->rccl = all_can && c->nccl;
    if (c->rccl) c->why_not_rccl.clear();
    // HIP IPC when RCCL is not in use and every rank holds device memory: each rank exports its buffer once and the next rank
    // tries to map it - any refusal (no dmabuf IPC, containers without /dev/kfd sharing ...) and ALL ranks fall back to the TCP star
    c->ipc = false;
    c->maps.assign(c->world, apk_comm::Mapping{});
    if (!c->rccl && c->world > 1) {
        int32_t mine_ok = (dev >= 0 && env_int("APK_COMM_IPC", 1, 0, 1)) ? 1 : 0;
        std::vector<int32_t> oks(c->world);
        CHK(ctl_allgather(c, &mine_ok, oks.data(), 4));
        bool try_ipc = true;
        for (int32_t v : oks) try_ipc = try_ipc && v;
        if (try_ipc) {
            c->device = dev;
            if (!c->stream) { HCHK(hipSetDevice(dev)); HCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); }
            c->ipc = true;                              // ensure() exports what it allocates from here on
            void** slot = c->rank == 0 ? &c->d_stage : &c->d_wire_out;
            size_t* cap = c->rank == 0 ? &c->stage_cap : &c->wire_out_cap;
            int32_t ok = ensure(c, slot, cap, 4096) == APK_OK ? 1 : 0;
            std::vector<hipIpcMemHandle_t> hs(c->world);
            CHK(ctl_allgather(c, &c->export_handle, hs.data(), sizeof(hipIpcMemHandle_t)));
            CHK(ctl_allgather(c, &ok, oks.data(), 4));
            bool all_ok = true;
            for (int32_t v : oks) all_ok = all_ok && v;
            if (all_ok) {
                void* probe = nullptr;
                const int nxt = (c->rank + 1) % c->world;
                if (hipSetDevice(dev) != hipSuccess || hipIpcOpenMemHandle(&probe, hs[nxt], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); ok = 0; }
                else (void)hipIpcCloseMemHandle(probe);
            }
            CHK(ctl_allgather(c, &ok, oks.data(), 4));
            for (int32_t v : oks) all_ok = all_ok && v;
            c->ipc = all_ok;
        }
    }
    return APK_OK;
}

int apk_msm_g1_sharded(apk_comm* c, const void* d_scalars, uint64_t len, void* out) {
    if (!c || !out || !c->cp.msm_batch) { set_error("comm: not bound"); return APK_ERR_ARG; }
    const size_t nb = c->cp.g1_bytes, rec = 8 + nb;
    std::vector<uint8_t> mine(rec, 0), all(rec * c->world, 0);
    int32_t status = APK_OK;
    std::string err;
    if (len) {
        const void* p[1] = {d_scalars}; const uint64_t off[1] = {0}, ls[1] = {len};
        const WallMs t_msm;
        status = c->cp.msm_batch(CP_USER(c, msm_batch), 0, 1, p, off, ls, mine.data() + 8);
        c->ms_msm += t_msm.ms();
        if (status != APK_OK) err = apk_last_error();
    }
    memcpy(mine.data(), &status, 4);
    {
        const WallMs t_x;
        const int xrc = sums_allgather(c, mine.data(), all.data(), rec);      // the ONE exchange step: a 64/96-byte point per rank (ncclAllGather on RCCL)
        c->ms_sums += t_x.ms();
        c->n_commit_rounds++;
        CHK(xrc);
    }
    std::vector<uint8_t> pts((size_t)c->world * nb);
    for (int r = 0; r < c->world; r++) {
        int32_t st;
        memcpy(&st, all.data() + (size_t)r * rec, 4);
        if (st != APK_OK) { set_error("comm: rank %d failed its slice of the sharded MSM (code %d)%s%s", r, st, r == c->rank ? ": " : "", r == c->rank ? err.c_str() : ""); return st; }
        memcpy(pts.data() + (size_t)r * nb, all.data() + (size_t)r * rec + 8, nb);
    }
    c->steps++;
    return apk_g1_sum(c->ctx ? c->ctx->curve : (nb == 64 ? APK_BN254 : APK_BLS12_381), pts.data(), c->world, out);
}

int apk_comm_commit(apk_comm* c, int basis, uint32_t count, const void* const* d_scalars, const uint32_t* lens, void* out_points) {
    if (!c || c->rank != 0 || !count || count > 4 || !d_scalars || !lens || !out_points) { set_error("comm: commit is the leader's call (1..4 commitments)"); return APK_ERR_ARG; }
    std::lock_guard<std::mutex> lk(c->step_mu);
    Header h{};
    h.op = OP_COMMIT; h.basis = basis; h.count = count;
    for (uint32_t i = 0; i < count; i++) h.lens[i] = lens[i];
    CHK(ctl_bcast(c, &h, sizeof h));
    return commit_round(c, basis, count, h.lens, d_scalars, out_points);
}

// The replicated prover's commitment step: called on EVERY rank with that rank's own copy of the scalar vectors.
int apk_comm_commit_local(apk_comm* c, int basis, uint32_t count, const void* const* d_scalars, const uint32_t* lens, void* out_points) {
    if (!c || !count || count > 4 || !d_scalars || !lens || !out_points) { set_error("comm: commit_local takes 1..4 commitments"); return APK_ERR_ARG; }
    std::lock_guard<std::mutex> lk(c->step_mu);
    return commit_round(c, basis, count, lens, d_scalars, out_points, /*local=*/true);
}

int apk_comm_wires(apk_comm* c, uint32_t count, const void* const* d_can, const uint32_t* lens, void* const* d_ev) {
    if (!c || c->rank != 0 || !count || count > 4 || !d_can || !lens || !d_ev) { set_error("comm: wires is the leader's call (1..4 polynomials)"); return APK_ERR_ARG; }
    std::lock_guard<std::mutex> lk(c->step_mu);
    Header h{};
    h.op = OP_WIRES; h.count = count;
    for (uint32_t i = 0; i < count; i++) h.lens[i] = lens[i];
    CHK(ctl_bcast(c, &h, sizeof h));
    return wires_round(c, count, h.lens, d_can, d_ev);
}

int apk_comm_split_begin(apk_comm* c) {
    if (!c || c->rank != 0 || !c->ctx) { set_error("comm: split_begin is the leader's call on a bound circuit context"); return APK_ERR_ARG; }
    CHK(apk_ctx_set_commit_hook(c->ctx, hook_commit, c));
    if (env_int("APK_SPLIT_WIRES", 0, 0, 1) && c->world > 1) {
        if (!c->cp.n) { set_error("comm: the compute table does not state the domain size"); return APK_ERR_STATE; }
        CHK(apk_ctx_set_wire_hook(c->ctx, hook_wires, c));
    }
    c->split_on = true;
    return APK_OK;
}

// In-place all-gather of device memory: rank r's part sits at d_all + r * bytes on every rank when the call returns.
static int allgather_device_impl(apk_comm* c, void* d_all, size_t bytes);
int apk_comm_allgather_device(apk_comm* c, void* d_all, size_t bytes) {
    if (!c || !d_all) { set_error("comm: allgather_device: null argument"); return APK_ERR_ARG; }
    if (c->world == 1 || bytes == 0) return APK_OK;
    const WallMs t;
    const int rc = allgather_device_impl(c, d_all, bytes);
    c->ms_gather += t.ms();
    c->n_gathers++;
    return rc;
}
static int allgather_device_impl(apk_comm* c, void* d_all, size_t bytes) {
    uint8_t* all = (uint8_t*)d_all;
    uint8_t* mine = all + (size_t)c->rank * bytes;
    if (c->rccl) {
        HCHK(hipSetDevice(c->device));
        NCHK(g_rccl.AllGather(mine, all, bytes, ncclUint8, c->nccl, c->stream));     // in place: sendbuff = recvbuff + rank * count
        return stream_wait(c, "the sub-coset ncclAllGather");
    }
    if (c->ipc) {
        // every rank exports its staging buffer (the one allocation a peer can map), parks its part there, and pulls the others'
        void** slot = c->rank == 0 ? &c->d_stage : &c->d_wire_out;
        size_t* cap = c->rank == 0 ? &c->stage_cap : &c->wire_out_cap;
        int32_t ok = ensure(c, slot, cap, bytes) == APK_OK && c->cp.copy(CP_USER(c, copy), *slot, mine, bytes, 0) == APK_OK ? 1 : 0;
        std::vector<IpcMsg> msgs(c->world);
        IpcMsg m{};
        m.gen = c->export_gen; m.offset = 0; m.bytes = ok ? bytes : 0; m.h = c->export_handle; m.cap = *cap;
        CHK(ctl_allgather(c, &m, msgs.data(), sizeof m));
        hipError_t e = hipSetDevice(c->device);
        for (int r = 0; r < c->world && ok; r++) {
            if (r == c->rank) continue;
            if (msgs[r].bytes != bytes || msgs[r].offset > msgs[r].cap || bytes > msgs[r].cap - msgs[r].offset) { ok = 0; break; }
            apk_comm::Mapping& mp = c->maps[r];
            if (e == hipSuccess && (mp.gen != msgs[r].gen || !mp.p)) {
                if (mp.p) (void)hipIpcCloseMemHandle(mp.p);
                mp.p = nullptr;
                e = hipIpcOpenMemHandle(&mp.p, msgs[r].h, hipIpcMemLazyEnablePeerAccess);
                mp.gen = msgs[r].gen;
            }
            if (e == hipSuccess) e = hipMemcpyAsync(all + (size_t)r * bytes, (const uint8_t*)mp.p + msgs[r].offset, bytes, hipMemcpyDeviceToDevice, c->stream);
            if (e != hipSuccess) ok = 0;
        }
        if (ok && hipStreamSynchronize(c->stream) != hipSuccess) ok = 0;
        if (!ok) (void)hipGetLastError();
        // nobody reuses its exported buffer before every peer has pulled from it; and everybody learns how the step went
        std::vector<int32_t> oks(c->world);
        CHK(ctl_allgather(c, &ok, oks.data(), 4));
        for (int r = 0; r < c->world; r++) if (!oks[r]) { set_error("comm: rank %d failed the device all-gather (IPC)", r); return APK_ERR_HIP; }
        return APK_OK;
    }
    // host-staged through the TCP star (CPU tier; IPC off or refused)
    std::vector<uint8_t> h_all((size_t)c->world * bytes);
    CHK(c->cp.copy(CP_USER(c, copy), h_all.data() + (size_t)c->rank * bytes, mine, bytes, 2));
    std::vector<uint8_t> h_mine(h_all.begin() + (size_t)c->rank * bytes, h_all.begin() + (size_t)(c->rank + 1) * bytes);
    CHK(ctl_allgather(c, h_mine.data(), h_all.data(), bytes));
    for (int r = 0; r < c->world; r++)
        if (r != c->rank) CHK(c->cp.copy(CP_USER(c, copy), all + (size_t)r * bytes, h_all.data() + (size_t)r * bytes, bytes, 1));
    return APK_OK;
}

// Replicated prover ("SPMD"): every rank holds the circuit context AND the witness and runs the same apk_prove* call; only the
// commitments are shared out - rank r commits its index range of every batch from its own copy of the polynomials, one
// all-gather of the partial sums (ncclAllGather on the RCCL plane), every rank adds them.  Against the leader / worker split
// (apk_comm_split_begin) nothing is scattered - 604 MB per BLS12-381 2^21 proof stay where they are - and no rank idles through
// the leader's transforms; the price is that those transforms run on every GPU.  One proof at a time per communicator: the
// ranks' batches meet in the order they are issued.
int apk_comm_spmd_begin(apk_comm* c) {
    if (!c || !c->ctx) { set_error("comm: spmd_begin needs a bound circuit context on every rank"); return APK_ERR_ARG; }
    if (c->split_on) { set_error("comm: spmd_begin inside a split proof (call apk_comm_split_end first)"); return APK_ERR_STATE; }
    CHK(apk_ctx_set_commit_hook(c->ctx, hook_commit_local, c));
    (void)apk_ctx_set_wire_hook(c->ctx, nullptr, nullptr);   // a wire hook left by an earlier split has no place in the replicated prover
    c->spmd_on = true;
    // round 3 of the prover on sub-cosets (apk_ctx_set_subcoset): world 2, 4 or 8, unless APK_SPMD_SUBCOSET=0 - and only when
    // EVERY rank's context can take it (the ranks agree first: a rank on its own would wait in an all-gather nobody joins)
    int32_t can = 0;
    if ((c->world == 2 || c->world == 4 || c->world == 8) && env_int("APK_SPMD_SUBCOSET", 1, 0, 1))
        can = apk_ctx_set_subcoset(c->ctx, c->rank, c->world, hook_gather, c) == APK_OK ? 1 : 0;
    std::vector<int32_t> cans(c->world);
    CHK(ctl_allgather(c, &can, cans.data(), 4));
    bool all = true;
    for (int32_t v : cans) all = all && v;
    if (!all) (void)apk_ctx_set_subcoset(c->ctx, 0, 1, nullptr, nullptr);
    c->subcoset_on = all;
    return APK_OK;
}
int apk_comm_spmd_end(apk_comm* c) {
    if (!c) { set_error("null communicator"); return APK_ERR_ARG; }
    if (c->ctx) { (void)apk_ctx_set_commit_hook(c->ctx, nullptr, nullptr); (void)apk_ctx_set_subcoset(c->ctx, 0, 1, nullptr, nullptr); }
    c->subcoset_on = false;
    c->spmd_on = false;
    return APK_OK;
}
int apk_comm_subcoset_active(const apk_comm* c) { return c && c->subcoset_on ? 1 : 0; }

int apk_comm_split_end(apk_comm* c) {
    if (!c || c->rank != 0) { set_error("comm: split_end is the leader's call"); return APK_ERR_ARG; }
    if (c->ctx) { (void)apk_ctx_set_commit_hook(c->ctx, nullptr, nullptr); (void)apk_ctx_set_wire_hook(c->ctx, nullptr, nullptr); }
    c->split_on = false;
    Header h{};
    h.op = OP_STOP;
    return ctl_bcast(c, &h, sizeof h);
}

int apk_comm_serve(apk_comm* c, uint64_t* steps_served) {
    if (!c || c->rank == 0) { set_error("comm: serve is the workers' call"); return APK_ERR_ARG; }
    for (;;) {
        Header h{};
        // between steps the leader may be idle for any length of time: no timeout until the next header starts to arrive
        CHK(wait_readable(c->peer[0]));
        CHK(ctl_bcast(c, &h, sizeof h));
        if (h.op == OP_STOP) break;
        if (h.count == 0 || h.count > 4) { set_error("comm: malformed header"); return APK_ERR_STATE; }
        int rc;
        c->step_synced = false;
        if (h.op == OP_COMMIT) rc = commit_round(c, h.basis, h.count, h.lens, nullptr, nullptr);
        else if (h.op == OP_WIRES) rc = wires_round(c, h.count, h.lens, nullptr, nullptr);
        else { set_error("comm: unknown step %d", h.op); return APK_ERR_STATE; }
        // A step that failed AFTER its status reached every rank (a rank's MSM or transform failed) leaves the streams aligned: the
        // worker stays in the loop for the leader's next call.  A step that broke on the transport (socket, RCCL, IPC) or on a
        // local allocation did not: the next bytes from the leader are not a header, so the worker leaves with the error.
        if (rc != APK_OK && !c->step_synced) return rc;
    }
    if (steps_served) *steps_served = c->steps;
    return APK_OK;
}

const char* apk_comm_transport_reason(const apk_comm* c) { return c ? c->why_not_rccl.c_str() : "null communicator"; }

int apk_comm_phase_ms(apk_comm* c, double* out, int reset) {
    if (!c || !out) { set_error("comm: phase_ms: null argument"); return APK_ERR_ARG; }
    std::lock_guard<std::mutex> lk(c->step_mu);      // the commitment rounds add to these under the same lock
    out[0] = c->ms_msm; out[1] = c->ms_sums; out[2] = c->ms_gather; out[3] = (double)c->n_commit_rounds; out[4] = (double)c->n_gathers;
    if (reset) { c->ms_msm = c->ms_sums = c->ms_gather = 0; c->n_commit_rounds = c->n_gathers = 0; }
    return APK_OK;
}

// One timed pass over the data plane as it came up (collective; after apk_comm_bind on a device context): a ring of ncclSend /
// ncclRecv of `ring_bytes` (every rank sends to its right neighbour and receives from its left one, all links at once - RCCL
// plane only, 0 elsewhere) and an in-place all-gather of `gather_bytes` per rank on whatever plane is active (ncclAllGather,
// IPC pulls or the TCP star).  GB/s per rank = bytes this rank RECEIVED / wall time, the second of two passes (the first warms
// the connections up).  The first multi-GPU run of this library reads its link rate from here (bench.py config.data_plane).
int apk_comm_link_probe(apk_comm* c, size_t ring_bytes, size_t gather_bytes, double* ring_gbps, double* gather_gbps) {
    if (!c || !ring_gbps || !gather_gbps) { set_error("comm: link_probe: null argument"); return APK_ERR_ARG; }
    *ring_gbps = 0; *gather_gbps = 0;
    if (c->world == 1 || !c->cp.alloc || c->device_is_host) return APK_OK;
    const size_t need = (ring_bytes * 2 > gather_bytes * (size_t)c->world ? ring_bytes * 2 : gather_bytes * (size_t)c->world) + 256;
    void* d = nullptr;
    int32_t ok = c->cp.alloc(CP_USER(c, alloc), need, &d) == APK_OK ? 1 : 0;
    std::vector<int32_t> oks(c->world);
    CHK(ctl_allgather(c, &ok, oks.data(), 4));
    for (int32_t v : oks) ok = ok && v;
    int rc = APK_OK;
    if (ok) {
        if (c->rccl && ring_bytes) {
            const int right = (c->rank + 1) % c->world, left = (c->rank + c->world - 1) % c->world;
            for (int pass = 0; pass < 2 && rc == APK_OK; pass++) {
                rc = ctl_allgather(c, &ok, oks.data(), 4);            // start together
                const WallMs t;
                if (rc == APK_OK && hipSetDevice(c->device) != hipSuccess) rc = APK_ERR_HIP;
                if (rc == APK_OK) {
                    ncclResult_t r = g_rccl.GroupStart();
                    if (r == ncclSuccess) r = g_rccl.Send(d, ring_bytes, ncclUint8, right, c->nccl, c->stream);
                    if (r == ncclSuccess) r = g_rccl.Recv((uint8_t*)d + ring_bytes, ring_bytes, ncclUint8, left, c->nccl, c->stream);
                    const ncclResult_t e = g_rccl.GroupEnd();
                    if (r != ncclSuccess || e != ncclSuccess) { set_error("comm: link probe: RCCL ring failed"); rc = APK_ERR_HIP; }
                    else rc = stream_wait(c, "the link probe's ring");
                }
                if (pass == 1 && rc == APK_OK) *ring_gbps = (double)ring_bytes / (t.ms() * 1e6);
            }
        }
        if (gather_bytes && rc == APK_OK)
            for (int pass = 0; pass < 2 && rc == APK_OK; pass++) {
                rc = ctl_allgather(c, &ok, oks.data(), 4);
                const WallMs t;
                if (rc == APK_OK) rc = allgather_device_impl(c, d, gather_bytes);
                if (pass == 1 && rc == APK_OK) *gather_gbps = (double)gather_bytes * (c->world - 1) / (t.ms() * 1e6);
            }
    } else {
        set_error("comm: link probe: a rank could not allocate its buffer");
        rc = APK_ERR_HIP;
    }
    if (d) (void)c->cp.release(CP_USER(c, release), d);
    return rc;
}

int apk_comm_rccl_ranks(const apk_comm* c) {
    if (!c || !c->rccl || !c->nccl) return 0;
    int n = 0;
    if (g_rccl.CommCount(c->nccl, &n) != ncclSuccess) return -1;
    return n;
}

// A world-1 RCCL communicator on `device`, driven through every call the multi-GPU data plane makes (ncclGetUniqueId,
// ncclCommInitRank, grouped ncclSend/ncclRecv to itself on a non-blocking stream, ncclAllGather, ncclCommCount,
// ncclCommDestroy): the one-GPU boxes of this build cannot run two RCCL ranks (RCCL refuses two ranks per device), so this is how
// init, stream use and teardown of that branch execute on hardware at all.  Returns APK_OK and the communicator's size in *ranks.
int apk_comm_rccl_selftest(int device, int* ranks) {
    if (ranks) *ranks = 0;
    if (!g_rccl.load()) { set_error("comm: librccl could not be loaded: %s", dlerror() ? dlerror() : "not found"); return APK_ERR_STATE; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); set_error("no HIP device available; libapk has no CPU fallback"); return APK_ERR_HIP; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return APK_ERR_ARG; }
    HCHK(hipSetDevice(device));
    hipStream_t st = nullptr;
    ncclComm_t comm = nullptr;
    uint8_t *d_a = nullptr, *d_b = nullptr;
    constexpr size_t N = 96 + 8;                // a status word + one BLS12-381 point: the size of the real exchange
    uint8_t h_a[N], h_b[N], h_c[N];
    for (size_t i = 0; i < N; i++) { h_a[i] = (uint8_t)(i * 7 + 3); h_b[i] = 0; h_c[i] = 0; }
    int rc = APK_OK, count = 0;
    auto fail = [&](const char* what, const char* why) { set_error("comm: RCCL self-test: %s: %s", what, why); rc = APK_ERR_HIP; };
#define ST_H(x) if (rc == APK_OK) { hipError_t e_ = (x); if (e_ != hipSuccess) fail(#x, hipGetErrorString(e_)); }
#define ST_N(x) if (rc == APK_OK) { ncclResult_t r_ = (x); if (r_ != ncclSuccess) fail(#x, g_rccl.GetErrorString(r_)); }
    ncclUniqueId id{};
    ST_N(g_rccl.GetUniqueId(&id));
    ST_H(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    ST_N(g_rccl.CommInitRank(&comm, 1, id, 0));
    ST_N(g_rccl.CommCount(comm, &count));
    ST_H(hipMalloc((void**)&d_a, N));
    ST_H(hipMalloc((void**)&d_b, N));
    ST_H(hipMemcpyAsync(d_a, h_a, N, hipMemcpyHostToDevice, st));
    ST_H(hipMemsetAsync(d_b, 0, N, st));
    // the scatter / peer-copy pattern of data_scatter and data_p2p: Send and Recv in ONE group (to itself here)
    ST_N(g_rccl.GroupStart());
    ST_N(g_rccl.Send(d_a, N, ncclUint8, 0, comm, st));
    ST_N(g_rccl.Recv(d_b, N, ncclUint8, 0, comm, st));
    ST_N(g_rccl.GroupEnd());
    ST_H(hipMemcpyAsync(h_b, d_b, N, hipMemcpyDeviceToHost, st));
    ST_H(hipStreamSynchronize(st));
    if (rc == APK_OK && memcmp(h_a, h_b, N) != 0) fail("send to self", "the received bytes differ from the sent ones");
    // the partial-sum exchange of sums_allgather
    ST_H(hipMemsetAsync(d_b, 0, N, st));
    ST_N(g_rccl.AllGather(d_a, d_b, N, ncclUint8, comm, st));
    ST_H(hipMemcpyAsync(h_c, d_b, N, hipMemcpyDeviceToHost, st));
    ST_H(hipStreamSynchronize(st));
    if (rc == APK_OK && memcmp(h_a, h_c, N) != 0) fail("all-gather", "the gathered bytes differ from the contribution");
#undef ST_H
#undef ST_N
    if (comm) (void)g_rccl.CommDestroy(comm);
    if (d_a) (void)hipFree(d_a);
    if (d_b) (void)hipFree(d_b);
    if (st) (void)hipStreamDestroy(st);
    if (rc != APK_OK) (void)hipGetLastError();
    if (ranks) *ranks = count;
    return rc;
}

}  // extern "C"
