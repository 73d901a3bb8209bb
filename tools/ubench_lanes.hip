// tools/ubench_lanes.hip -- does a wavefront with few ACTIVE lanes issue slower on gfx950?  One wavefront per SIMD (grid = CUs x 4
// single-wave workgroups), dependent chains of one instruction kind, EXEC limited to the first `active` lanes by a branch.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_lanes tools/ubench_lanes.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ void __launch_bounds__(64) k_chain(uint32_t* out, int iters, int active) {
    if ((int)threadIdx.x >= active) return;
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = a ^ 0x9e3779b9u;
    uint64_t c0 = a, c1 = b, c2 = a + b, c3 = a * 3u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (KIND == 0) {        // v_mad_u64_u32, four independent accumulators
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %5, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %4, %3\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");
            } else if (KIND == 1) { // v_add_u32 (32-bit integer)
                asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %3\n v_add_u32 %0, %0, %3\n v_add_u32 %1, %1, %2\n" : "+v"(a), "+v"(b) : "v"((uint32_t)c0), "v"((uint32_t)c1));
            } else if (KIND == 2) { // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %3\n v_mul_lo_u32 %0, %0, %3\n v_mul_lo_u32 %1, %1, %2\n" : "+v"(a), "+v"(b) : "v"((uint32_t)c0 | 1u), "v"((uint32_t)c1 | 1u));
            } else {                // v_lshl_add_u64 (64-bit add)
                asm volatile("v_lshl_add_u64 %0, %0, 1, %2\n v_lshl_add_u64 %1, %1, 1, %3\n v_lshl_add_u64 %0, %0, 1, %3\n v_lshl_add_u64 %1, %1, 1, %2\n" : "+v"(c0), "+v"(c1) : "v"(c2), "v"(c3));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)(c0 ^ c1 ^ c2 ^ c3) ^ a ^ b;
}

template <int KIND>
static void run(const char* name, int blocks, uint32_t* d) {
    const int iters = 4000;
    printf("%-14s %4d waves", name, blocks);
    for (int active : {1, 2, 4, 8, 9, 12, 15, 16, 17, 24, 32, 33, 48, 64}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k_chain<KIND>, dim3(blocks), dim3(64), 0, 0, d, 50, active);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_chain<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters, active);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // one wave per SIMD: cycles per instruction at 2.4 GHz = ms * 2.4e6 / (iters * 64 instructions)
        printf(" %2d:%5.2f", active, ms * 2.4e6 / (iters * 64.0));
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
    printf("   (cycles per wave-instruction at a nominal 2.4 GHz; active lanes : value)\n");
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * 4;
    uint32_t* d = nullptr;
    hipMalloc(&d, (size_t)blocks * 64 * 4);
    run<0>("v_mad_u64_u32", blocks, d);
    run<1>("v_add_u32", blocks, d);
    run<2>("v_mul_lo_u32", blocks, d);
    run<3>("v_lshl_add_u64", blocks, d);
    // the same chains with fewer wavefronts on the chip (one per workgroup; the dispatcher spreads them over the CUs)
    for (int nb : {1, 4, 32, 128, 256, 512, 768}) run<0>("v_mad_u64_u32", nb, d);
    for (int nb : {4, 128, 512}) run<1>("v_add_u32", nb, d);
    hipFree(d);
    return 0;
}
