// tools/ubench_h2d.cpp -- what the host side of a host-pointer VerifyBlobKZGProofBatch(1024) can reach on this box:
//   (1) SHA-NI challenge hash per core, one transcript at a time against two interleaved (host_sha256.hpp), and on T threads;
//   (2) gather of 1024 scattered pageable blobs into pinned staging: memcpy against streaming stores (host_copy.hpp), T threads;
//   (3) H2D of the 134 MB from pinned memory: one stream, two, four; chunk sizes;
//   (4) both at once, chunk by chunk (what the library's pipeline does), with and without the hash on a share of the blobs.
// Build: hipcc -O2 -std=c++17 -o /tmp/ubench_h2d tools/ubench_h2d.cpp -lpthread      (GPU box)
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../go-eth-kzg_amd/csrc/host_copy.hpp"
#include "../go-eth-kzg_amd/csrc/host_sha256.hpp"

using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorName(e_), __LINE__); return 1; } } while (0)

static const size_t B = 131072;

template <class F>
static double run_threads(int T, F f) {          // f(thread index); returns ms
    auto t0 = clk::now();
    std::vector<std::thread> th;
    for (int q = 0; q < T; q++) th.emplace_back([&, q] { f(q); });
    for (auto& t : th) t.join();
    return ms_since(t0);
}

int main(int argc, char** argv) {
    const size_t N = argc > 1 ? (size_t)atoi(argv[1]) : 1024;
    const int TMAX = argc > 2 ? atoi(argv[2]) : 15;
    std::vector<uint8_t*> src(N);
    for (size_t i = 0; i < N; i++) {
        src[i] = (uint8_t*)malloc(B + 4096 * (i % 7)) + 64 * (i % 3);
        for (size_t k = 0; k < B; k += 8) *(uint64_t*)(src[i] + k) = 0x9e3779b97f4a7c15ull * (i * B + k + 1);
    }
    uint8_t cm[48] = {0xc0};
    uint8_t* pinned = nullptr;
    uint8_t* dev = nullptr;
    CK(hipHostMalloc(&pinned, N * B, hipHostMallocDefault));
    CK(hipMalloc(&dev, N * B));
    memset(pinned, 1, N * B);
    std::vector<uint8_t> z(N * 32);
    printf("sha_ni=%d avx2=%d N=%zu\n", (int)kzg::sha256_have_ni(), (int)kzg::copy_have_avx2(), N);

    // (1) hash
    {
        auto t0 = clk::now();
        for (size_t i = 0; i < 64; i++) kzg::host_compute_challenge(&z[i * 32], src[i], cm);
        double one = ms_since(t0) / 64;
        std::vector<uint8_t> z2(64 * 32);
        t0 = clk::now();
        for (size_t i = 0; i < 64; i += 2) kzg::host_compute_challenge_x2(&z2[i * 32], src[i], cm, &z2[i * 32 + 32], src[i + 1], cm);
        double two = ms_since(t0) / 64;
        printf("hash per transcript: one at a time %.1f us, two interleaved %.1f us, same=%d\n", one * 1e3, two * 1e3, (int)!memcmp(z.data(), z2.data(), 64 * 32));
        if (kzg::sha256_have_avx512()) {
            std::vector<uint8_t> z3(64 * 32);
            t0 = clk::now();
            for (size_t i = 0; i < 64; i += 16) {
                const uint8_t* bp[16];
                const uint8_t* cp[16];
                for (int q = 0; q < 16; q++) { bp[q] = src[i + q]; cp[q] = cm; }
                kzg::host_compute_challenge_x16(&z3[i * 32], bp, cp);
            }
            printf("hash per transcript: sixteen AVX-512 lanes %.1f us, same=%d\n", ms_since(t0) / 64 * 1e3, (int)!memcmp(z.data(), z3.data(), 64 * 32));
            for (int T : {8, TMAX, TMAX + 1}) {
                double best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    std::atomic<size_t> next{0};
                    best = std::min(best, run_threads(T, [&](int) {
                        for (;;) {
                            size_t i = next.fetch_add(16);
                            if (i + 16 > N) break;
                            const uint8_t* bp[16];
                            const uint8_t* cp[16];
                            for (int q = 0; q < 16; q++) { bp[q] = src[i + q]; cp[q] = cm; }
                            kzg::host_compute_challenge_x16(&z[i * 32], bp, cp);
                        }
                    }));
                }
                printf("hash %zu transcripts on %2d threads (sixteen lanes): %.3f ms\n", N, T, best);
            }
        }
        for (int T : {4, 8, TMAX}) {
            for (int x2 = 0; x2 < 2; x2++) {
                double best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    std::atomic<size_t> next{0};
                    best = std::min(best, run_threads(T, [&](int) {
                        for (;;) {
                            size_t i = next.fetch_add(2);
                            if (i >= N) break;
                            if (x2) kzg::host_compute_challenge_x2(&z[i * 32], src[i], cm, &z[i * 32 + 32], src[i + 1], cm);
                            else { kzg::host_compute_challenge(&z[i * 32], src[i], cm); kzg::host_compute_challenge(&z[i * 32 + 32], src[i + 1], cm); }
                        }
                    }));
                }
                printf("hash %zu transcripts on %2d threads (%s): %.3f ms\n", N, T, x2 ? "pairs" : "single", best);
            }
        }
    }
    // (2) gather
    for (int nt = 0; nt < 2; nt++)
        for (int T : {4, 8, TMAX}) {
            double best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                std::atomic<size_t> next{0};
                best = std::min(best, run_threads(T, [&](int) {
                    for (;;) {
                        size_t i = next.fetch_add(1);
                        if (i >= N) break;
                        if (nt) kzg::copy_to_staging(pinned + i * B, src[i], B); else memcpy(pinned + i * B, src[i], B);
                    }
                }));
            }
            printf("gather %s threads=%2d  %.3f ms  %.1f GB/s\n", nt ? "stream-stores" : "memcpy       ", T, best, N * B / best / 1e6);
        }
    // (3) H2D
    hipStream_t st[4];
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (size_t chunk : {(size_t)16, (size_t)32, (size_t)64, (size_t)128, (size_t)256}) {
        for (int ns : {1, 2, 4}) {
            double best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipDeviceSynchronize());
                auto t0 = clk::now();
                int k = 0;
                for (size_t lo = 0; lo < N; lo += chunk, k++) {
                    size_t m = std::min(chunk, N - lo);
                    CK(hipMemcpyAsync(dev + lo * B, pinned + lo * B, m * B, hipMemcpyHostToDevice, st[k % ns]));
                }
                for (int q = 0; q < ns; q++) CK(hipStreamSynchronize(st[q]));
                best = std::min(best, ms_since(t0));
            }
            printf("h2d chunk=%3zu blobs streams=%d  %.3f ms  %.1f GB/s\n", chunk, ns, best, N * B / best / 1e6);
        }
    }
    // (4) the pipeline: T gather threads (chunk order), the main thread puts a chunk's copy in flight when its tasks are done;
    //     hash_share percent of the blobs (the LAST ones) are also hashed by the gather threads, in pairs
    for (int nt = 0; nt < 2; nt++)
        for (size_t chunk : {(size_t)32, (size_t)64, (size_t)128})
            for (int share : {0, 50, 100}) {
                const int T = TMAX;
                double best = 1e9, best_gather = 0;
                for (int rep = 0; rep < 4; rep++) {
                    const size_t nch = (N + chunk - 1) / chunk;
                    std::vector<std::atomic<int>> left(nch);
                    const size_t per_task = 8;                   // blobs per task
                    for (size_t k = 0; k < nch; k++) left[k] = (int)((std::min(chunk, N - k * chunk) + per_task - 1) / per_task);
                    std::atomic<size_t> next{0};
                    const size_t first_hashed = N - N * share / 100;
                    CK(hipDeviceSynchronize());
                    auto t0 = clk::now();
                    std::vector<std::thread> th;
                    std::atomic<double> gather_done{0};
                    for (int q = 0; q < T; q++) th.emplace_back([&] {
                        for (;;) {
                            size_t lo = next.fetch_add(per_task);
                            if (lo >= N) break;
                            size_t hi = std::min(N, lo + per_task);
                            for (size_t i = lo; i < hi; i++) { if (nt) kzg::copy_to_staging(pinned + i * B, src[i], B); else memcpy(pinned + i * B, src[i], B); }
                            left[lo / chunk].fetch_sub(1);
                            for (size_t i = lo; i + 1 < hi; i += 2)
                                if (i >= first_hashed) kzg::host_compute_challenge_x2(&z[i * 32], pinned + i * B, cm, &z[i * 32 + 32], pinned + (i + 1) * B, cm);
                        }
                        gather_done = ms_since(t0);
                    });
                    for (size_t k = 0; k < nch; k++) {
                        while (left[k].load() > 0) std::this_thread::yield();
                        size_t lo = k * chunk, m = std::min(chunk, N - lo);
                        CK(hipMemcpyAsync(dev + lo * B, pinned + lo * B, m * B, hipMemcpyHostToDevice, st[0]));
                    }
                    CK(hipStreamSynchronize(st[0]));
                    double up = ms_since(t0);
                    for (auto& t : th) t.join();
                    double all = ms_since(t0);
                    if (all < best) { best = all; best_gather = up; }
                }
                printf("pipeline %s chunk=%3zu hash_share=%3d%% threads=%d: upload done %.3f ms, hash done %.3f ms (%.1f GB/s over the link)\n", nt ? "stream" : "memcpy", chunk, share, T,
                       best_gather, best, N * B / best_gather / 1e6);
            }
    // (5) K callers at once: each its own staging + device buffer + stream, T threads shared by all
    for (int K : {2, 4, 8}) {
        std::vector<uint8_t*> pin(K), dv(K);
        bool ok = true;
        for (int k = 0; k < K; k++) {
            if (hipHostMalloc(&pin[k], N * B, hipHostMallocDefault) != hipSuccess || hipMalloc(&dv[k], N * B) != hipSuccess) { ok = false; break; }
            memset(pin[k], 1, N * B);
        }
        if (!ok) { printf("callers=%d: allocation failed\n", K); break; }
        for (int nt = 0; nt < 2; nt++) {
            const size_t chunk = 64, nch = (N + chunk - 1) / chunk, per_task = 8;
            const int reps = 3;
            std::vector<hipStream_t> ks(K);
            for (auto& s : ks) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            // work queue over (rep, caller, blob-range); every caller thread uploads its own chunks
            std::vector<std::vector<std::atomic<int>>> left(K);
            std::atomic<long> next{0};
            const long tasks_per_call = (long)((N + per_task - 1) / per_task);
            for (int k = 0; k < K; k++) { left[k] = std::vector<std::atomic<int>>(nch * reps); for (size_t c = 0; c < nch * reps; c++) left[k][c] = (int)(chunk / per_task); }
            auto t0 = clk::now();
            std::vector<std::thread> th;
            for (int q = 0; q < TMAX; q++) th.emplace_back([&] {
                for (;;) {
                    long t = next.fetch_add(1);
                    if (t >= tasks_per_call * K * reps) break;
                    // interleave callers: task t -> (rep, range, caller)
                    int k = (int)(t % K);
                    long u = t / K;
                    int rep = (int)(u / tasks_per_call);
                    size_t lo = (size_t)(u % tasks_per_call) * per_task, hi = std::min(N, lo + per_task);
                    for (size_t i = lo; i < hi; i++) { if (nt) kzg::copy_to_staging(pin[k] + i * B, src[i], B); else memcpy(pin[k] + i * B, src[i], B); }
                    left[k][rep * nch + lo / chunk].fetch_sub(1);
                }
            });
            std::vector<std::thread> callers;
            for (int k = 0; k < K; k++) callers.emplace_back([&, k] {
                for (int rep = 0; rep < reps; rep++) {
                    for (size_t c = 0; c < nch; c++) {
                        while (left[k][rep * nch + c].load() > 0) std::this_thread::yield();
                        size_t lo = c * chunk, m = std::min(chunk, N - lo);
                        (void)hipMemcpyAsync(dv[k] + lo * B, pin[k] + lo * B, m * B, hipMemcpyHostToDevice, ks[k]);
                    }
                    (void)hipStreamSynchronize(ks[k]);       // the staging area is reused by the next repetition
                }
            });
            for (auto& t : callers) t.join();
            double all = ms_since(t0);
            for (auto& t : th) t.join();
            printf("callers=%d %s: %d x %zu blobs each in %.3f ms: %.1f GB/s over the link, %.0f blobs/s\n", K, nt ? "stream" : "memcpy", reps, N, all, K * reps * N * B / all / 1e6,
                   K * reps * N / all * 1e3);
            for (auto& s : ks) (void)hipStreamDestroy(s);
        }
        for (int k = 0; k < K; k++) { (void)hipHostFree(pin[k]); (void)hipFree(dv[k]); }
    }
    return 0;
}
