// tools/ubench_lanes2.hip -- follow-up to ubench_lanes.hip: is it the NUMBER of active lanes that decides the slow mode, or WHICH
// parts of the wavefront are populated?  EXEC = a 64-bit mask given on the command line of the loop below.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(64) k_chain(uint32_t* out, int iters, unsigned long long mask) {
    if (!((mask >> threadIdx.x) & 1ull)) return;
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = a ^ 0x9e3779b9u;
    uint64_t c0 = a, c1 = b, c2 = a + b, c3 = a * 3u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %5, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %4, %3\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");
    }
    out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)(c0 ^ c1 ^ c2 ^ c3) ^ a ^ b;
}

int main() {
    uint32_t* d = nullptr;
    hipMalloc(&d, (size_t)1024 * 64 * 4);
    struct { const char* name; unsigned long long m; } masks[] = {
        {"lane 0", 1ull}, {"lanes 0,32", 1ull | 1ull << 32}, {"lanes 0,16,32,48", 1ull | 1ull << 16 | 1ull << 32 | 1ull << 48},
        {"lanes 0-7", 0xffull}, {"lanes 0-7,32-39", 0xffull | 0xffull << 32}, {"lanes 0-8", 0x1ffull}, {"lanes 0-15", 0xffffull},
        {"lanes 0-16", 0x1ffffull}, {"lanes 0-31", 0xffffffffull}, {"lanes 0-32", 0x1ffffffffull}, {"lanes 32-63", 0xffffffff00000000ull},
        {"even lanes", 0x5555555555555555ull}, {"every 8th", 0x0101010101010101ull}, {"all", ~0ull}};
    for (int blocks : {4, 256, 512, 1024}) {
        printf("%4d waves:", blocks);
        for (auto& mk : masks) {
            const int iters = 3000;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(64), 0, 0, d, 50, mk.m);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(64), 0, 0, d, iters, mk.m);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("  [%s] %.2f", mk.name, ms * 2.4e6 / (iters * 64.0));
            hipEventDestroy(e0);
            hipEventDestroy(e1);
        }
        printf("\n");
    }
    hipFree(d);
    return 0;
}
