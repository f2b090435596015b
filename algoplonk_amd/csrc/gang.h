// Gangs: two to four concurrent proofs of one context that share ONE HIP stream and meet at every MSM batch and NTT batch, so that
// one launch sequence carries all of their operands (DESIGN.md section 4 "cross-proof launch batching").
//
// Why: small circuits are latency-bound under load and the device has about 16 useful hardware queues.  At BLS12-381 2^14 a
// commitment batch is 0.8 M bucket additions for 1 024 SIMDs, its reduction tail a chain of ~36 dependent point operations on a few
// hundred waves; 16 such streams issue ~56 % of what the SIMDs could, and more streams add little (32: +4.5 %).  A gang makes
// every MSM / NTT launch of a stream G times as wide at the same number of launches and streams.
//
// How: the members stay ordinary apk_prove* callers, each running the unchanged prover on its own workspace (Slot).  What they
// share is the LEAD slot's stream, MSM workspace, transform scratch and pinned result area.  Everything a member launches by itself
// (grand product, quotient, evaluations ...) goes straight onto that stream; at a merge point (commit / run_ntt_batch) a member
// posts its arguments and waits; the LAST member to arrive launches ONE sequence for everybody and wakes the others.  Nothing is
// reordered inside a member, and a merged launch sits behind every member's earlier launches, so each proof sees exactly the stream
// order it would have had alone: same kernels, same arithmetic, same bytes.
//
// This header is the GPU-free part (who meets whom, who launches), kept apart like slot_gate.h so that the sanitizer tier
// (tools/san/host_hammer.cpp) runs exactly this code.  A member that leaves early (an error, an unsatisfying witness) resigns with
// leave(): the others stop waiting for it - if they were all waiting already, the leaver launches on their behalf.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>

namespace apk {

constexpr int GANG_MAX = 4;

struct GangReq {
    int kind = 0;        // what kind of merge point this is (the launcher groups compatible requests)
    void* args = nullptr;
    void* member = nullptr;   // the posting member (opaque here: the backend's Slot)
    int rc = 0;          // filled in by the launcher
    std::string err;     // ... with its error message when rc != 0 (the launch ran on another member's thread)
    bool done = false;
};

class Gang {
  public:
    using Launcher = std::function<void(GangReq* const* reqs, int count)>;

    // Every member calls this first, with the gate's ticket (SlotGate::Ticket gen / size): the first one in (re)starts the gang
    // with `n` members, the others find it started.  A member that has entered may meet() at once: the launch waits for all n.
    void enter(uint64_t gen, int n) {
        std::lock_guard<std::mutex> lk(mu_);
        if (started_ && gen_ == gen) return;
        started_ = true; gen_ = gen;
        members_ = n; arrived_ = 0; pending_ = nullptr;
        for (int i = 0; i < GANG_MAX; i++) req_[i] = nullptr;
    }
    int members() {
        std::lock_guard<std::mutex> lk(mu_);
        return members_;
    }
    // A merge point.  `launch` runs ONCE per meeting, on the thread of the last arriver (or of a leaver), with every posted request;
    // it must set rc of each.  Returns this member's rc.
    int meet(int idx, GangReq& r, const Launcher& launch) {
        std::unique_lock<std::mutex> lk(mu_);
        r.done = false;
        req_[idx] = &r;
        pending_ = &launch;
        arrived_++;
        if (arrived_ >= members_) run_locked();
        else cv_.wait(lk, [&] { return r.done; });
        return r.rc;
    }
    // This member will not come to any further merge point.  Returns the number of members left.
    int leave(int idx) {
        std::unique_lock<std::mutex> lk(mu_);
        (void)idx;
        if (members_ > 0) members_--;
        if (members_ > 0 && arrived_ >= members_) run_locked();
        cv_done_.notify_all();
        return members_;
    }
    // the lead's exit: blocks until every other member has left (they run on the lead's stream and workspace)
    void wait_empty() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return members_ <= 0; });
    }

  private:
    void run_locked() {     // mu_ held: everybody who still belongs to the gang is waiting in meet()
        GangReq* list[GANG_MAX];
        int count = 0;
        for (int i = 0; i < GANG_MAX; i++) if (req_[i]) { list[count++] = req_[i]; req_[i] = nullptr; }
        const Launcher* l = pending_;
        pending_ = nullptr;
        arrived_ = 0;
        if (count && l) (*l)(list, count);
        for (int i = 0; i < count; i++) list[i]->done = true;
        cv_.notify_all();
    }
    std::mutex mu_;
    std::condition_variable cv_, cv_done_;
    int members_ = 0, arrived_ = 0;
    bool started_ = false;
    uint64_t gen_ = 0;
    GangReq* req_[GANG_MAX] = {nullptr, nullptr, nullptr, nullptr};
    const Launcher* pending_ = nullptr;
};

}  // namespace apk
