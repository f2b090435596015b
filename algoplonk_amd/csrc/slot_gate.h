// Which proving slot a caller gets, and how many are taken: the one piece of the backend's host-side threading that has no GPU in
// it, kept on its own so that the sanitizer tier (tools/san/host_hammer.cpp, `make SAN=thread`) runs exactly this code.
// A context runs up to `n` proofs concurrently, one slot (HIP stream + workspace) each; callers beyond that wait their turn
// ("safe to call concurrently from several goroutines").  The busy count is what the load-dependent kernel
// forms are chosen from (backend_impl.h run_msm_body / run_ntt_batch / tail_fill).
#pragma once
#include <condition_variable>
#include <mutex>
#include <vector>

namespace apk {

class SlotGate {
  public:
    void resize(size_t n) { std::lock_guard<std::mutex> lk(mu_); busy_.assign(n, 0); }
    size_t size() const { return busy_.size(); }
    // index of a free slot, taken; blocks while all are busy
    size_t acquire() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            for (size_t i = 0; i < busy_.size(); i++)
                if (!busy_[i]) { busy_[i] = 1; taken_++; return i; }
            cv_.wait(lk);
        }
    }
    void release(size_t i) {
        { std::lock_guard<std::mutex> lk(mu_); if (busy_[i]) { busy_[i] = 0; taken_--; } }
        cv_.notify_one();
    }
    // slots in use right now (the caller's own included)
    int busy() {
        std::lock_guard<std::mutex> lk(mu_);
        return taken_;
    }
  private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<char> busy_;
    int taken_ = 0;
};

}  // namespace apk
