// host_combine.hpp -- single-item calls that arrive while others are in flight are served as ONE batch call.
//
// The reference's BlobToKZGCommitment, ComputeCellsAndKZGProofs and VerifyBlobKZGProof take ONE blob (prove.go:13-33,
// api_eip7594.go:28-52, verify.go:48-83), and clients call them from many goroutines on one shared Context (verify.go:152-169,
// examples_test.go:11).  On a GPU every such call is a handful of launches that fill a fraction of the chip, a round trip and -- for the
// verifier -- a pairing on the host: K callers get K times that, in a runtime that runs all streams of a process through a few
// hardware queues.  The batch entry points do the same work for K blobs in one set of launches and ONE pairing.  So the single-item entry
// points meet here first (round 5), the way the throughput shape's verifier calls meet in hash rounds (api_verify.hip):
//   * the first arrival that finds no group forming opens one and LEADS it; whoever arrives before it closes joins;
//   * the leader closes its group as soon as fewer than max_running() groups are being processed -- at once for a caller that is alone
//     (a group of one takes the very path the entry point took before: no added latency), after the time one batch takes when the
//     context is busy, which is what makes the groups grow with the load;
//   * the leader processes the whole group through the batch entry point (its own thread, its own lane) and hands every member its
//     result; members sleep on a condition variable meanwhile.
// A group's members are independent calls: a member's input error is its own (the batch paths report per item), and a batch VERDICT
// that is not "all fine" sends every member of the group through the single-item path again (process() decides; see api_verify.hip).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <vector>

namespace kzg {

template <class Item>
class Combiner {
public:
    // groups being processed at a time: one uploads / waits for the device while the other computes.  *Measured* (tools/bench_single_callers.py,
    // 32 / 64 caller threads, calls per second with 2 / 3 / 4 / 6 groups in flight): VerifyBlobKZGProof 14.2 / 12.2 / 7.5 / 9.8 k and
    // 27.0 / 19.3 / 13.5 / 16.9 k, ComputeCellsAndKZGProofs 5.4 / 4.6 / 3.8 / 3.1 k and 7.9 / 7.1 / 5.9 / 2.9 k, BlobToKZGCommitment
    // 26.9 / 28.0 / 30.4 / 26.6 k and 38.0 / 39.7 / 39.9 / 38.3 k -- fewer, larger groups win; every call on its own: 2.9-3.0 k, 0.74-0.79 k
    // and 7.3 k whatever the number of callers.  MKZG_COMBINE_RUNNING overrides (A/B aid).
    static int max_running() {
        static const int v = [] { const char* e = getenv("MKZG_COMBINE_RUNNING"); const int n = e ? atoi(e) : 0; return n > 0 ? n : 2; }();
        return v;
    }
    // process(items): called by the group's leader with every member (size >= 1); must fill each item's result
    template <class F>
    void run(Item* it, F&& process) {
        std::unique_lock<std::mutex> lk(mu_);
        bool leader = false;
        if (!forming_) {
            forming_ = std::make_shared<Group>();
            leader = true;
        }
        std::shared_ptr<Group> g = forming_;
        g->items.push_back(it);
        if (!leader) {
            cv_.wait(lk, [&] { return g->done; });
            return;
        }
        cv_.wait(lk, [&] { return running_ < max_running(); });
        forming_.reset();                      // closed: later arrivals open the next group
        running_++;
        lk.unlock();
        process(g->items);
        lk.lock();
        g->done = true;
        running_--;
        cv_.notify_all();
    }
    // groups of more than one item / items served through them so far (the option "combined_calls" reports the latter)
    long long combined_items() const { return combined_items_.load(std::memory_order_relaxed); }
    void count_combined(size_t n) { combined_items_.fetch_add((long long)n, std::memory_order_relaxed); }

private:
    struct Group {
        std::vector<Item*> items;
        bool done = false;
    };
    std::mutex mu_;
    std::condition_variable cv_;
    std::shared_ptr<Group> forming_;
    int running_ = 0;
    std::atomic<long long> combined_items_{0};
};

}  // namespace kzg
