// tools/ubench_fma52.hip -- bounded experiment: would 8 x 48..52-bit limbs on v_fma_f64 beat 14 x 29-bit limbs on
// v_mad_u64_u32 for the 381-bit products of this library?  Measured here: the PRODUCT half of one Montgomery multiplication
// (a * b into column sums; the q * p half costs the same again in both schemes), one wavefront per SIMD, dependent on its own output
// so that nothing is hoisted.
//   mad29 : 14 x 14 = 196 v_mad_u64_u32 into 27 64-bit columns (what mont29.hpp does)
//   fma48 : 8 x 8 = 64 limb products, each as hi = fma_rz(a, b, 2^100 ...) , lo = fma_rz(a, b, -hi) (exact split of the 96-bit
//           product at 2^48, Emmart et al.), the hi parts chained through the FMA addend (exact while the sum stays in one binade),
//           the lo parts need the INDIVIDUAL hi of each product (one subtraction) and their own accumulation: 4 floating-point
//           instructions per limb product = 256 per a * b, before any conversion back to integers.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_fma52 tools/ubench_fma52.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(64) k_mad29(uint32_t* out, int iters) {
    uint32_t a[14], b[14];
    for (int i = 0; i < 14; i++) { a[i] = (threadIdx.x * 2654435761u + i * 40503u) & 0x1fffffffu; b[i] = (a[i] * 2246822519u + 7u) & 0x1fffffffu; }
    for (int it = 0; it < iters; it++) {
        uint64_t t[27];
#pragma unroll
        for (int k = 0; k < 27; k++) t[k] = 0;
#pragma unroll
        for (int i = 0; i < 14; i++)
#pragma unroll
            for (int j = 0; j < 14; j++) t[i + j] += (uint64_t)a[j] * b[i];
        // every column feeds back whole (low and high words), so that no product can be narrowed or dropped
#pragma unroll
        for (int k = 0; k < 27; k++) asm volatile("" : "+v"(t[k]));
#pragma unroll
        for (int i = 0; i < 13; i++) a[i] = ((uint32_t)t[i] ^ (uint32_t)(t[i + 14] >> 32) ^ (uint32_t)(t[i] >> 35) ^ (uint32_t)t[i + 14]) & 0x1fffffffu;
        a[13] = ((uint32_t)t[13] ^ (uint32_t)(t[13] >> 33)) & 0x1fffffffu;
    }
    uint32_t x = 0;
    for (int i = 0; i < 14; i++) x ^= a[i];
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ void __launch_bounds__(64) k_fma48(uint32_t* out, int iters) {
    double a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = (double)((threadIdx.x * 2654435761u + i * 40503u) & 0xffffffu) * 16777216.0 + 12345.0; b[i] = a[i] * 0.5 + 3.0; }
    const double C1 = 0x1p100;
    for (int it = 0; it < iters; it++) {
        double hi[15], lo[15];
#pragma unroll
        for (int k = 0; k < 15; k++) { hi[k] = C1; lo[k] = 0.0; }
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const double h1 = __builtin_fma(a[j], b[i], hi[i + j]);      // hi chain (exact in round-toward-zero while inside the binade)
                const double hp = h1 - hi[i + j];                            // this product's own high part
                const double l = __builtin_fma(a[j], b[i], -hp);             // its low part (exact)
                lo[i + j] += l;
                hi[i + j] = h1;
            }
#pragma unroll
        for (int k = 0; k < 15; k++) asm volatile("" : "+v"(hi[k]), "+v"(lo[k]));
#pragma unroll
        for (int i = 0; i < 7; i++) a[i] = (hi[i] - C1) * 0x1p-52 + lo[i + 8] * 0x1p-4 + (hi[i + 8] - C1) * 0x1p-60 + lo[i] * 0x1p-8 + 1.0;
        a[7] = (hi[7] - C1) * 0x1p-52 + lo[7] * 0x1p-4 + 1.0;
    }
    double x = 0;
    for (int i = 0; i < 8; i++) x += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)(long long)x;
}

template <class K>
static double time_kernel(K k, int blocks, uint32_t* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 10);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms;
}
int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    uint32_t* d = nullptr;
    hipMalloc(&d, (size_t)cus * 8 * 64 * 4);
    const int iters = 20000;
    for (int wps : {1, 2}) {
        const double m1 = time_kernel(k_mad29, cus * 4 * wps, d, iters), m2 = time_kernel(k_fma48, cus * 4 * wps, d, iters);
        printf("%d wavefront(s) per SIMD: a*b product half per wavefront  mad29 (196 v_mad_u64_u32) %.1f ns   fma48 (64 x [fma, sub, fma, add]) %.1f ns   ratio %.2f\n", wps,
               m1 * 1e6 / iters, m2 * 1e6 / iters, m2 / m1);
    }
    hipFree(d);
    return 0;
}
