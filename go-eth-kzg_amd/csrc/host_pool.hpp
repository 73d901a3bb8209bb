// host_pool.hpp -- a few persistent host threads per context for the sub-millisecond host steps of the verifiers (Horner chains over
// the window sums, the two Miller loops, the gather of cells into pinned staging).  Spawning a std::thread per step costs 25-45 us each
// on the GPU box's host (seven of them per VerifyCellKZGProofBatch call = 9 % of that call); waking a parked worker costs a few.
// The reference gets the same effect from goroutines (verify.go:152-169, internal/multiexp/multiexp.go:20-26).
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace kzg {

class HostPool {
   public:
    explicit HostPool(int n_workers) {
        for (int i = 0; i < n_workers; i++) th_.emplace_back([this] { loop(); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int size() const { return (int)th_.size(); }
    // f(0), ..., f(k-1) on the workers only; returns at once.  *left must be k on entry and reaches 0 when all have finished
    // (the caller keeps f and left alive until then).  With no workers the tasks run here.
    void run_async(int k, const std::function<void(int)>& f, std::atomic<int>* left) {
        if (th_.empty()) {
            for (int i = 0; i < k; i++) { f(i); left->fetch_sub(1, std::memory_order_release); }
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            for (int i = 0; i < k; i++) q_.emplace_back([&f, left, i] { f(i); left->fetch_sub(1, std::memory_order_release); });
        }
        cv_.notify_all();
    }
    HostPool(const HostPool&) = delete;
    HostPool& operator=(const HostPool&) = delete;

    // f(0), ..., f(k-1), each exactly once: f(0) on the calling thread, the others on the workers; returns when all have finished.
    // Safe to call from several threads at once; f must not call parallel() itself.
    void parallel(int k, const std::function<void(int)>& f) {
        if (k <= 1 || th_.empty()) {
            for (int i = 0; i < k; i++) f(i);
            return;
        }
        std::atomic<int> left(k - 1);
        {
            std::lock_guard<std::mutex> g(m_);
            for (int i = 1; i < k; i++) q_.emplace_back([&f, &left, i] { f(i); left.fetch_sub(1, std::memory_order_release); });
        }
        if (k == 2) cv_.notify_one(); else cv_.notify_all();
        f(0);
        while (left.load(std::memory_order_acquire) != 0) std::this_thread::yield();     // the tasks are 50-400 us long
    }

   private:
    void loop() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;            // stop_ and nothing left to do
                job = std::move(q_.front());
                q_.pop_front();
            }
            job();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    bool stop_ = false;
};

}  // namespace kzg
