// tools/ubench_streams.hip -- how many kernels of ONE process does the MI355X run side by side?
//
// The sustained verifier legs keep 3-7 kernels in flight from 12-32 caller threads, each on its own stream.  This
// probe launches, on S non-blocking streams at once, C kernels per stream (stream-ordered) of one small workgroup each that spin for T
// microseconds, and reports the wall time against S x C x T: "concurrency" = how many of them ran at a time.  With the runtime's default
// of four hardware queues (GPU_MAX_HW_QUEUES) the streams share those queues.
//   hipcc --offload-arch=gfx950 -O2 -o ubench_streams tools/ubench_streams.hip && ./ubench_streams [T_us=500]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void __launch_bounds__(64) k_spin(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (sink && threadIdx.x == 0 && ticks < 0) *sink = 1;
}

static double run(int S, int C, long long ticks, bool threads) {
    std::vector<hipStream_t> st(S);
    for (auto& s : st) if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) exit(1);
    for (auto& s : st) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 100, nullptr); }      // warm every stream's queue
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    if (threads) {
        std::vector<std::thread> th;
        for (int i = 0; i < S; i++) th.emplace_back([&, i] {
            for (int c = 0; c < C; c++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], ticks, nullptr);
            (void)hipStreamSynchronize(st[i]);
        });
        for (auto& t : th) t.join();
    } else {
        for (int c = 0; c < C; c++)
            for (int i = 0; i < S; i++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], ticks, nullptr);
        (void)hipDeviceSynchronize();
    }
    const double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    for (auto& s : st) (void)hipStreamDestroy(s);
    return wall;
}

int main(int argc, char** argv) {
    const double T = argc > 1 ? atof(argv[1]) : 500.0;
    const long long ticks = (long long)(T * 100.0);          // wall_clock64: 100 MHz
    printf("kernels of one workgroup spinning %.0f us; S streams x C kernels per stream; concurrency = S*C*T / wall\n", T);
    for (int threads = 0; threads < 2; threads++) {
        printf("-- launched by %s\n", threads ? "one host thread per stream (each waits for its own stream)" : "one host thread, round robin over the streams");
        for (int C : {1, 4}) {
            for (int S : {1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64}) {
                double best = 1e30;
                for (int rep = 0; rep < 3; rep++) best = std::min(best, run(S, C, ticks, threads != 0));
                printf("S %2d  C %d  wall %9.1f us  concurrency %5.2f\n", S, C, best, S * C * T / best);
            }
        }
    }
    return 0;
}
