// tests/host_shim/combiner_stress.cpp -- go-eth-kzg_amd/csrc/host_combine.hpp on its own, under ThreadSanitizer: single-item calls from many
// threads must each be processed exactly once, by a leader that holds at most max_running() groups in flight, a caller that is alone is
// a group of one, and members get the result their leader wrote for THEM.  (The entry points that use it -- BlobToKZGCommitment,
// ComputeCellsAndKZGProofs, VerifyBlobKZGProof -- are driven on the stub device by stub_device_driver.cpp and on the GPU by
// tests/test_gpu_combine.py.)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include "../../go-eth-kzg_amd/csrc/host_combine.hpp"

struct Item { int id = 0; int result = -1; int processed = 0; };

int main() {
    kzg::Combiner<Item> comb;
    std::atomic<int> running{0}, worst{0}, groups{0}, largest{0}, bad{0};
    auto process = [&](std::vector<Item*>& items) {
        const int now = ++running;
        int w = worst.load();
        while (now > w && !worst.compare_exchange_weak(w, now)) {}
        int l = largest.load();
        while ((int)items.size() > l && !largest.compare_exchange_weak(l, (int)items.size())) {}
        groups++;
        std::this_thread::sleep_for(std::chrono::microseconds(300));      // a batch call: long enough for the next group to form behind it
        for (Item* it : items) { it->result = 3 * it->id + 1; it->processed++; }
        --running;
    };
    // 1. a caller that is alone: groups of one
    for (int i = 0; i < 20; i++) {
        Item it;
        it.id = i;
        comb.run(&it, process);
        if (it.result != 3 * i + 1 || it.processed != 1) bad++;
    }
    if (largest.load() != 1 || groups.load() != 20) { fprintf(stderr, "a lone caller was not a group of one (%d groups, largest %d)\n", groups.load(), largest.load()); bad++; }
    // 2. many callers at once
    const int T = 24, N = 200;
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t] {
        for (int i = 0; i < N; i++) {
            Item it;
            it.id = t * N + i;
            comb.run(&it, process);
            if (it.result != 3 * it.id + 1 || it.processed != 1) bad++;
        }
    });
    for (auto& t : th) t.join();
    if (worst.load() > kzg::Combiner<Item>::max_running()) { fprintf(stderr, "%d groups in flight, limit %d\n", worst.load(), kzg::Combiner<Item>::max_running()); bad++; }
    if (largest.load() < 2) { fprintf(stderr, "24 threads never met in a group\n"); bad++; }
    printf("%d groups for %d items, largest %d, at most %d in flight, %d mismatches\n", groups.load(), 20 + T * N, largest.load(), worst.load(), bad.load());
    return bad.load() ? 1 : 0;
}
