// tools/ubench_rowgather.hip -- what does MI355X deliver on RANDOM 128-byte row reads out of a table as large as the commitment MSM's
// direct window table (146 GB), and on a plain stream of the same buffer?
//
// Why (round 6): the one remaining algorithmic lever of the headline kernel is batched-affine addition (6-7 field multiplications per
// table point instead of 9), whose price is HBM traffic: every table row is gathered twice (once for the denominators, once for the
// sums) and the prefix products / intermediate points make a round trip through HBM.  Whether that trade can pay is decided by the
// rate at which the chip gathers random rows -- k_commit_direct_acc today pulls 0.87 TB/s of them and is VALU-bound.
//
//   hipcc --offload-arch=gfx950 -O2 -o ubench_rowgather tools/ubench_rowgather.hip && ./ubench_rowgather [table GiB = 128]
// Prints GB/s for {1, 2, 4, 8} wavefronts per SIMD x {2, 4, 8} independent row reads in flight per lane group, and for a stream.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16(const uint4* p) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
// eight lanes read one 128-byte row (16 bytes each): a wavefront reads eight rows per load instruction, U instructions in flight
template <int U>
__global__ void __launch_bounds__(64) k_gather(const uint4* __restrict__ table, uint64_t nrows, int iters, uint4* __restrict__ sink) {
    const uint64_t wave = blockIdx.x;
    const int lane = threadIdx.x, grp = lane >> 3, sub = lane & 7;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; it++) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t row = mix((wave << 24) ^ ((uint64_t)it << 8) ^ (uint64_t)(u * 8 + grp)) % nrows;
            v[u] = ld16(&table[row * 8 + sub]);
        }
#pragma unroll
        for (int u = 0; u < U; u++) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[blockIdx.x * 64 + lane] = acc;     // never true in practice: keeps the loads alive
}
template <int U>
__global__ void __launch_bounds__(64) k_stream(const uint4* __restrict__ table, uint64_t n16, int iters, uint4* __restrict__ sink) {
    const uint64_t stride = (uint64_t)gridDim.x * 64;
    uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; it++) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { v[u] = ld16(&table[i % n16]); i += stride; }
#pragma unroll
        for (int u = 0; u < U; u++) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int U>
static double run_gather(const uint4* t, uint64_t nrows, int waves, int iters, uint4* sink) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_gather<U>, dim3(waves), dim3(64), 0, 0, t, nrows, iters / 4, sink);        // warm-up (page tables, clocks)
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_gather<U>, dim3(waves), dim3(64), 0, 0, t, nrows, iters, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return (double)waves * iters * U * 8 * 128 / (ms * 1e-3) / 1e9;
}
template <int U>
static double run_stream(const uint4* t, uint64_t n16, int waves, int iters, uint4* sink) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_stream<U>, dim3(waves), dim3(64), 0, 0, t, n16, iters / 4, sink);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_stream<U>, dim3(waves), dim3(64), 0, 0, t, n16, iters, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return (double)waves * iters * U * 64 * 16 / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 128.0;
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30)) & ~127ull;
    uint4* t;
    uint4* sink;
    CK(hipMalloc(&t, bytes));
    CK(hipMalloc(&sink, 8192ull * 64 * 16));
    CK(hipMemset(t, 1, bytes));
    CK(hipDeviceSynchronize());
    const uint64_t nrows = bytes / 128;
    printf("table %.0f GiB = %llu rows of 128 B\n", gib, (unsigned long long)nrows);
    printf("%-28s %10s %10s %10s\n", "random 128-B rows, GB/s", "U=2", "U=4", "U=8");
    for (int occ : {1, 2, 4, 8}) {
        const int waves = 1024 * occ, iters = 4096 / occ;
        printf("%d wavefront(s) per SIMD       %10.0f %10.0f %10.0f\n", occ, run_gather<2>(t, nrows, waves, iters, sink), run_gather<4>(t, nrows, waves, iters, sink),
               run_gather<8>(t, nrows, waves, iters, sink));
    }
    printf("%-28s %10s %10s %10s\n", "stream of the buffer, GB/s", "U=2", "U=4", "U=8");
    for (int occ : {2, 4, 8}) {
        const int waves = 1024 * occ, iters = 4096 / occ;
        printf("%d wavefront(s) per SIMD       %10.0f %10.0f %10.0f\n", occ, run_stream<2>(t, bytes / 16, waves, iters, sink), run_stream<4>(t, bytes / 16, waves, iters, sink),
               run_stream<8>(t, bytes / 16, waves, iters, sink));
    }
    // a table the size of the Infinity Cache, for comparison
    const uint64_t small_rows = (192ull << 20) / 128;
    printf("random rows out of 192 MiB (Infinity Cache resident), 4 wavefronts per SIMD, U=8: %.0f GB/s\n", run_gather<8>(t, small_rows, 4096, 1024, sink));
    CK(hipFree(t));
    CK(hipFree(sink));
    return 0;
}
