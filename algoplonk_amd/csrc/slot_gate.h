// Which proving slot a caller gets, how many are taken, and who proves together: the piece of the backend's host-side threading that
// has no GPU in it, kept on its own so that the sanitizer tier (tools/san/host_hammer.cpp, `make SAN=thread`) runs exactly this code.
//
// A context has `n` slots (one workspace each) and runs up to `max_streams` HIP streams at a time (16 is the measured optimum at
// 2^17; more cost a few percent: profiles/r06_gang_sweep.txt).  A caller takes a slot; callers beyond the slots wait their turn (SURVEY.md
// section 8b: "safe to call concurrently from several goroutines").  The busy count is what the load-dependent kernel forms are
// chosen from (backend_impl.h run_msm_body / run_ntt_batch / tail_fill).
//
// Gangs (round 6, gang.h): when MORE callers are in the system than the context has streams, callers are paired up - a "lead" that
// owns a stream and up to gang_max - 1 "followers" that prove on the lead's stream in lockstep.  A lead that has to wait for a free
// stream anyway collects followers for free while it waits; one that finds a stream at once waits at most `wait_us` for partners
// (under sustained load they arrive within microseconds: the callers of a finished gang come back together).  With no more callers
// than streams nobody is ever ganged and nobody ever waits: a lone proof's latency is what it was.
#pragma once
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

namespace apk {

class SlotGate {
  public:
    struct Ticket {
        size_t slot = 0;   // the caller's own workspace
        size_t lead = 0;   // the slot whose stream it proves on (== slot when it leads or proves alone)
        int idx = 0;       // member number inside the gang (0 = lead)
        int size = 1;      // members of the gang when it started
        uint64_t gen = 0;  // distinguishes successive gangs of the same lead slot
        int stream = 0;    // which of the context's max_streams streams the gang runs on (a lead's / a lone caller's own)
    };

    void resize(size_t n) { configure(n, n, 1, 0); }
    void configure(size_t n, size_t max_streams, int gang_max, int wait_us) {
        std::lock_guard<std::mutex> lk(mu_);
        busy_.assign(n, 0); holds_stream_.assign(n, 0); count_.assign(n, 0); size_.assign(n, 0); gen_.assign(n, 0); target_.assign(n, 2);
        taken_ = 0; streams_ = 0; waiting_ = 0; outside_ = 0; open_lead_ = NONE;
        max_streams_ = max_streams < 1 ? 1 : max_streams;
        stream_busy_.assign(max_streams_, 0);
        gang_max_ = gang_max < 1 ? 1 : gang_max;
        wait_us_ = wait_us < 0 ? 0 : wait_us;
    }
    size_t size() const { return busy_.size(); }
    int gang_max() const { return gang_max_; }

    // index of a free slot with a stream of its own, taken; blocks while all are busy.  Never ganged (primitives, setup).
    Ticket acquire() { return acquire_member(false); }

    // a proof's way in.  allow_gang = false: as acquire().
    Ticket acquire_member(bool allow_gang) {
        std::unique_lock<std::mutex> lk(mu_);
        const bool gangs = allow_gang && gang_max_ > 1;
        size_t i = NONE;
        for (;;) {
            if (gangs && open_lead_ != NONE) {
                i = free_slot();
                if (i != NONE) {       // ---- follower: join the open gang
                    take(i);
                    const size_t lead = open_lead_;
                    const uint64_t g0 = gen_[lead];
                    Ticket t;
                    t.slot = i; t.lead = lead; t.idx = count_[lead]++;
                    if (count_[lead] >= target_[lead]) open_lead_ = NONE;  // full: closed to further joins, started by its lead
                    cv_.notify_all();
                    cv_.wait(lk, [&] { return gen_[lead] != g0; });        // ... until the lead starts it
                    t.size = size_[lead]; t.gen = gen_[lead]; t.stream = holds_stream_[lead] - 1;
                    return t;
                }
            }
            i = free_slot();
            if (i != NONE) break;
            waiting_++; outside_++;
            cv_.wait(lk);
            waiting_--; outside_--;
        }
        // ---- lead (or alone)
        take(i);
        count_[i] = 1;
        // more callers in the system than streams?  (taken_ counts this caller)
        const bool crowded = gangs && (size_t)(taken_ + waiting_) > max_streams_;
        if (crowded && open_lead_ == NONE) {
            open_lead_ = i;
            // as many members as it takes to put every caller in the system (in flight, or waiting for a slot) on one of the
            // streams: 32 callers on 16 streams prove in pairs even where gangs of four are allowed - sixteen streams of two
            // keep more of the device busy than eight of four
            const size_t in_system = (size_t)(taken_ + outside_);
            int want = (int)((in_system + max_streams_ - 1) / max_streams_);
            target_[i] = want < 2 ? 2 : (want > gang_max_ ? gang_max_ : want);
        }
        // (system_clock: libstdc++ then waits with pthread_cond_timedwait, which ThreadSanitizer intercepts - a steady_clock deadline
        // goes through pthread_cond_clockwait, which gcc 11's libtsan does not know, and every later wait is then misreported;
        // a clock step during these few hundred microseconds only shortens or lengthens one wait for partners)
        const auto deadline = std::chrono::system_clock::now() + std::chrono::microseconds(wait_us_);
        for (;;) {
            const bool have_stream = streams_ < max_streams_;
            const bool collecting = open_lead_ == i;       // still open to joins
            if (have_stream && (!collecting || std::chrono::system_clock::now() >= deadline)) break;
            waiting_++;
            if (have_stream) cv_.wait_until(lk, deadline); else cv_.wait(lk);
            waiting_--;
        }
        if (open_lead_ == i) open_lead_ = NONE;
        streams_++;
        // the lowest free stream: a context only ever touches max_streams streams, whichever slots lead (every further stream is a
        // further hardware queue, and the device time-slices queues beyond what it has: 32 slots leading on 32 streams of their
        // own read 357 against 532 proofs/s, with queues starved for up to 90 ms - profiles/r06_gang_streams.txt)
        int sid = 0;
        while ((size_t)sid + 1 < stream_busy_.size() && stream_busy_[sid]) sid++;
        stream_busy_[sid] = 1;
        holds_stream_[i] = sid + 1;
        size_[i] = count_[i];
        gen_[i]++;
        cv_.notify_all();
        Ticket t;
        t.slot = i; t.lead = i; t.idx = 0; t.size = size_[i]; t.gen = gen_[i]; t.stream = sid;
        return t;
    }
    void release(size_t i) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (busy_[i]) { busy_[i] = 0; taken_--; }
            if (holds_stream_[i]) { stream_busy_[holds_stream_[i] - 1] = 0; holds_stream_[i] = 0; streams_--; }
        }
        cv_.notify_all();
    }
    // slots in use right now (the caller's own included)
    int busy() {
        std::lock_guard<std::mutex> lk(mu_);
        return taken_;
    }
    int streams() {
        std::lock_guard<std::mutex> lk(mu_);
        return (int)streams_;
    }
  private:
    static constexpr size_t NONE = (size_t)-1;
    size_t free_slot() const {
        for (size_t i = 0; i < busy_.size(); i++) if (!busy_[i]) return i;
        return NONE;
    }
    void take(size_t i) { busy_[i] = 1; taken_++; }
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<char> busy_, stream_busy_;
    std::vector<int> holds_stream_;     // per slot: 1 + the stream it holds as a lead / lone caller, 0 = none
    std::vector<int> count_, size_, target_;   // per lead slot: members so far / members at the start / members it closes at
    std::vector<uint64_t> gen_;         // per lead slot: gangs started
    int taken_ = 0, waiting_ = 0, outside_ = 0;   // slots taken; threads waiting in here (with or without a slot); ... without a slot
    size_t streams_ = 0, max_streams_ = 1;
    size_t open_lead_ = NONE;
    int gang_max_ = 1, wait_us_ = 0;
};

}  // namespace apk
