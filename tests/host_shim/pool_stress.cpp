// tests/host_shim/pool_stress.cpp -- ThreadSanitizer driver for the product's host pool (go-eth-kzg_amd/csrc/host_pool.hpp): the
// patterns the C ABI uses it in, from several caller threads at once, on pools of 0, 1, 3 and 7 workers.
//   parallel(k, f)                      Horner chains / Miller loops / recovery scatter (api_verify.hip, api_cells.hip, host_pairing.hpp)
//   run_async + wait(chunk) + wait(all) the gather pipelines that run ahead of the caller (api.hip commit_host_piece & co)
// Built and run by tests/test_host_sanitizers.py with -fsanitize=thread; exits 0 when every sum checks out and TSan stays silent.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "../../go-eth-kzg_amd/csrc/host_pool.hpp"
using namespace kzg;

static uint64_t mix(uint64_t x) {          // a few hundred nanoseconds of work that the compiler cannot fold
    for (int i = 0; i < 200; i++) x = x * 6364136223846793005ull + 1442695040888963407ull;
    return x;
}

static int caller(HostPool& pool, int id, int rounds) {
    int bad = 0;
    for (int r = 0; r < rounds; r++) {
        // parallel: k independent results, every index exactly once
        const int k = 1 + (r + id) % 9;
        std::vector<uint64_t> out(k, 0);
        std::vector<int> hits(k, 0);
        pool.parallel(k, [&](int i) { out[i] = mix((uint64_t)(id * 1000003 + r * 101 + i)); hits[i]++; });
        for (int i = 0; i < k; i++) if (hits[i] != 1 || out[i] != mix((uint64_t)(id * 1000003 + r * 101 + i))) bad++;
        // gather pipeline: chunks of TPC tasks queued at once, the caller consumes chunk by chunk
        const int nch = 1 + r % 5, TPC = 1 + (id + r) % 4, per = 64;
        std::vector<uint8_t> src((size_t)nch * TPC * per), dst(src.size(), 0);
        for (size_t i = 0; i < src.size(); i++) src[i] = (uint8_t)(i * 7 + id + r);
        std::vector<Countdown> left(nch);
        for (auto& a : left) a.set(TPC);
        Countdown all_left(nch * TPC);
        const std::function<void(int)> gather = [&](int t) {
            memcpy(dst.data() + (size_t)t * per, src.data() + (size_t)t * per, per);
            left[(size_t)(t / TPC)].arrive();
        };
        pool.run_async(nch * TPC, gather, &all_left);
        for (int c = 0; c < nch; c++) {
            pool.wait(left[c]);
            if (memcmp(dst.data() + (size_t)c * TPC * per, src.data() + (size_t)c * TPC * per, (size_t)TPC * per)) bad++;      // the chunk is complete and visible
        }
        pool.wait(all_left);
        if (dst != src) bad++;
    }
    return bad;
}

int main() {
    int bad = 0;
    for (int workers : {0, 1, 3, 7}) {
        HostPool pool(workers);
        if (pool.size() != workers) bad++;
        std::vector<std::thread> th;
        std::vector<int> res(6, 0);
        for (int t = 0; t < 6; t++) th.emplace_back([&, t] { res[t] = caller(pool, t, 120); });
        for (auto& t : th) t.join();
        for (int v : res) bad += v;
    }
    // a pool torn down with tasks just finished (destructor joins parked workers)
    for (int i = 0; i < 50; i++) {
        HostPool pool(2);
        uint64_t acc[3] = {0, 0, 0};
        pool.parallel(3, [&](int j) { acc[j] = mix((uint64_t)j); });
        if (acc[2] != mix(2)) bad++;
    }
    printf("pool_stress: %d mismatches\n", bad);
    return bad ? 1 : 0;
}
