// tools/ubench_gather.cpp -- how fast can host threads gather 256 scattered 128 KiB blobs into one staging buffer on this box?
// (pinned vs pageable destination, thread count).  Build: hipcc -O2 -o /tmp/ubench_gather tools/ubench_gather.cpp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

int main() {
    const size_t B = 131072, N = 256;
    std::vector<char*> src(N);
    for (size_t i = 0; i < N; i++) { src[i] = (char*)malloc(B + 4096 * (i % 7)); memset(src[i], (int)i, B); }
    char* pinned = nullptr;
    if (hipHostMalloc(&pinned, N * B, hipHostMallocDefault) != hipSuccess) return 1;
    char* pinned_nc = nullptr;
    if (hipHostMalloc(&pinned_nc, N * B, hipHostMallocNonCoherent) != hipSuccess) pinned_nc = nullptr;
    char* plain = (char*)malloc(N * B);
    memset(pinned, 0, N * B);
    memset(plain, 0, N * B);
    if (pinned_nc) memset(pinned_nc, 0, N * B);
    struct { const char* name; char* dst; } dsts[] = {{"pinned(default)", pinned}, {"pinned(noncoherent)", pinned_nc}, {"pageable", plain}};
    for (auto& d : dsts) {
        if (!d.dst) continue;
        for (int nthr : {1, 2, 4, 8, 12, 16}) {
            double best = 1e9;
            for (int rep = 0; rep < 7; rep++) {
                auto t0 = std::chrono::steady_clock::now();
                std::vector<std::thread> th;
                for (int q = 0; q < nthr; q++)
                    th.emplace_back([&, q] { for (size_t i = q; i < N; i += nthr) memcpy(d.dst + i * B, src[i], B); });
                for (auto& t : th) t.join();
                best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }
            printf("%-20s threads=%2d  %.3f ms  %.1f GB/s\n", d.name, nthr, best, N * B / best / 1e6);
        }
    }
    return 0;
}
