// host_pool.hpp -- a few persistent host threads per context for the sub-millisecond host steps of the verifiers (Horner chains over
// the window sums, the two Miller loops, the gather of blobs / cells into pinned staging).  Spawning a std::thread per step costs
// 25-45 us each on the GPU box's host (seven of them per VerifyCellKZGProofBatch call = 9 % of that call); waking a parked worker costs
// a few.  The reference gets the same effect from goroutines (verify.go:152-169, internal/multiexp/multiexp.go:20-26).
//
// Sized by the SHARE of the machine a context may use (api.hip pool_workers_for: the cgroup's CPUs divided by the ranks of the job on
// this node or the devices of the context), because every rank of a multi-GPU job builds one.  Nobody burns a CPU while waiting:
//   * in a SMALL pool (up to three workers: a rank's share of a node that runs eight of them) a caller that waits for its tasks first
//     RUNS queued tasks itself (its own or another call's -- all are 50-400 us long): with a single worker the caller is the second
//     pair of hands, with none the pool degenerates to plain loops.  With more workers the caller stays free to put the next copy or
//     launch in flight the moment the chunk it waits for is complete;
//   * with nothing left to run it spins on yield() for SPIN_US, then parks on a futex until the last task signs off.
#pragma once
#include <linux/futex.h>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace kzg {

// tasks still to finish; the last one to sign off wakes whoever parked on the counter
struct Countdown {
    std::atomic<int> v{0};
    Countdown() = default;
    explicit Countdown(int n) : v(n) {}
    void set(int n) { v.store(n, std::memory_order_relaxed); }
    bool done() const { return v.load(std::memory_order_acquire) == 0; }
    void arrive() {
        // the wake follows the decrement: a waiter that saw 0 meanwhile may already have returned and destroyed this object -- the
        // futex call then names a dead address, which the kernel treats as "nobody waits there"; nothing is read or written through it
        if (v.fetch_sub(1, std::memory_order_acq_rel) == 1) (void)syscall(SYS_futex, reinterpret_cast<int*>(&v), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
    }
    void park() {                               // returns when the counter may have changed
        const int x = v.load(std::memory_order_acquire);
        if (x != 0) (void)syscall(SYS_futex, reinterpret_cast<int*>(&v), FUTEX_WAIT_PRIVATE, x, nullptr, nullptr, 0);
    }
};
static_assert(sizeof(std::atomic<int>) == sizeof(int), "futex word");

class HostPool {
   public:
    static constexpr int SPIN_US = 50;          // yield-spin this long before parking (a futex wake-up costs 5-20 us)

    // cpus (may be null / empty = the default): the workers are bound to these CPUs (api.hip device_node_cpus: opt-in, with the measurement
    // that made it opt-in).  Failure to bind is not an error.
    explicit HostPool(int n_workers, const cpu_set_t* cpus = nullptr) {
        for (int i = 0; i < n_workers; i++) th_.emplace_back([this] { loop(); });
        if (cpus && CPU_COUNT(cpus) > 0)
            for (auto& t : th_) (void)pthread_setaffinity_np(t.native_handle(), sizeof(cpu_set_t), cpus);
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
    HostPool(const HostPool&) = delete;
    HostPool& operator=(const HostPool&) = delete;

    // f(0), ..., f(k-1) queued for the workers; returns at once.  left must hold k on entry and reaches 0 when all have finished
    // (the caller keeps f and left alive until then: wait(left) on every exit path -- which is also what runs the tasks of a pool
    // without workers: wait() executes queued tasks itself in pools of up to three workers, front first).
    void run_async(int k, const std::function<void(int)>& f, Countdown* left) {
        if (k <= 0) return;
        {
            std::lock_guard<std::mutex> g(m_);
            for (int i = 0; i < k; i++) q_.emplace_back([&f, left, i] { f(i); left->arrive(); });
        }
        if (k == 1) cv_.notify_one(); else cv_.notify_all();
    }
    // Blocks until left reaches 0.  In a small pool the caller runs queued tasks meanwhile (front first: the tasks of the chunk it
    // waits for were queued before those of later chunks).  always_help: also in a large pool -- for a caller whose wait IS the call's
    // critical path (the lone host-pointer verifier waiting for its challenge hashes: a sixteenth pair of hands instead of a parked thread).
    void wait(Countdown& left, bool always_help = false) {
        const bool help = always_help || th_.size() <= 3;
        while (!left.done()) {
            std::function<void()> job;
            if (help) {
                std::lock_guard<std::mutex> g(m_);
                if (!q_.empty()) {
                    job = std::move(q_.front());
                    q_.pop_front();
                }
            }
            if (job) { job(); continue; }
            // nothing queued: what is left is running on the workers
            const auto t0 = std::chrono::steady_clock::now();
            while (!left.done()) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(SPIN_US)) { left.park(); break; }
                std::this_thread::yield();
            }
        }
    }
    // f(0), ..., f(k-1), each exactly once, on the workers AND the calling thread; returns when all have finished.
    // Safe to call from several threads at once; f must not call parallel() itself.
    void parallel(int k, const std::function<void(int)>& f) {
        if (k <= 1 || th_.empty()) {
            for (int i = 0; i < k; i++) f(i);
            return;
        }
        // the shared state outlives this frame for helpers that start late: they see next >= k and touch nothing else
        struct State { std::atomic<int> next{0}; Countdown left; int k; const std::function<void(int)>* f; };
        auto st = std::make_shared<State>();
        st->k = k;
        st->f = &f;
        st->left.set(k);
        auto drain = [](State& s) {
            for (;;) {
                const int i = s.next.fetch_add(1, std::memory_order_relaxed);
                if (i >= s.k) return;
                (*s.f)(i);                       // i < k: the caller has not returned yet, f is alive
                s.left.arrive();
            }
        };
        const int helpers = std::min(k - 1, (int)th_.size());
        {
            std::lock_guard<std::mutex> g(m_);
            for (int h = 0; h < helpers; h++) q_.emplace_back([st, drain] { drain(*st); });
        }
        if (helpers == 1) cv_.notify_one(); else cv_.notify_all();
        drain(*st);
        const auto t0 = std::chrono::steady_clock::now();
        while (!st->left.done()) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(SPIN_US)) st->left.park();
            else std::this_thread::yield();
        }
    }

   private:
    void loop() {
        (void)pthread_setname_np(pthread_self(), "mkzg-pool");          // visible in /proc/<pid>/task/*/comm, top -H, perf
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
