// tools/ubench_valu.hip -- VALU issue-rate microbenchmark for the integer / fp64 instructions a
// 381-bit Montgomery multiply can be built from on gfx950.  Prints ops/clk/CU per instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void __launch_bounds__(256) bench(uint32_t* out, int iters, uint32_t seed, unsigned long long* clk) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    uint64_t c0 = a0, c1 = a1, c2 = a2, c3 = a3, c4 = a0 + 9, c5 = a1 + 9, c6 = a2 + 9, c7 = a3 + 9;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a0 + 0.5, d5 = a1 + 0.5, d6 = a2 + 0.5, d7 = a3 + 0.5;
    double m = 1.0000001, ad = 0.5;
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {  // v_mad_u64_u32, 8 independent accumulators
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %10, %1\n v_mad_u64_u32 %2, vcc, %9, %10, %2\n v_mad_u64_u32 %3, vcc, %8, %11, %3\n"
                              "v_mad_u64_u32 %4, vcc, %9, %11, %4\n v_mad_u64_u32 %5, vcc, %10, %11, %5\n v_mad_u64_u32 %6, vcc, %8, %8, %6\n v_mad_u64_u32 %7, vcc, %9, %9, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
        } else if (KIND == 1) {  // v_fma_f64
            REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                              "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(m), "v"(ad));)
        } else if (KIND == 2) {  // v_mul_lo_u32
            uint32_t* p = (uint32_t*)&c0;
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                              "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(seed | 1));)
        } else if (KIND == 3) {  // v_mul_hi_u32
            uint32_t* p = (uint32_t*)&c0;
            REP8(asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                              "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(seed | 0xF0000001u));)
        } else if (KIND == 4) {  // v_mad_u32_u24
            uint32_t* p = (uint32_t*)&c0;
            REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %0\n v_mad_u32_u24 %1, %1, %8, %1\n v_mad_u32_u24 %2, %2, %8, %2\n v_mad_u32_u24 %3, %3, %8, %3\n"
                              "v_mad_u32_u24 %4, %4, %8, %4\n v_mad_u32_u24 %5, %5, %8, %5\n v_mad_u32_u24 %6, %6, %8, %6\n v_mad_u32_u24 %7, %7, %8, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(seed | 1));)
        } else if (KIND == 5) {  // v_lshl_add_u64 (64-bit add)
            REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %4\n"
                              "v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 0, %6\n v_lshl_add_u64 %6, %6, 0, %7\n v_lshl_add_u64 %7, %7, 0, %0\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));)
        } else if (KIND == 6) {  // v_add_co_u32 + v_addc_co_u32 pairs
            uint32_t* p = (uint32_t*)&c0;
            REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n v_addc_co_u32 %2, vcc, %2, %6, vcc\n v_addc_co_u32 %3, vcc, %3, %7, vcc\n"
                              "v_add_co_u32 %4, vcc, %4, %0\n v_addc_co_u32 %5, vcc, %5, %1, vcc\n v_addc_co_u32 %6, vcc, %6, %2, vcc\n v_addc_co_u32 %7, vcc, %7, %3, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : : "vcc");)
        } else if (KIND == 7) {  // v_fma_f32 (full-rate reference)
            float* f = (float*)&c0; float mm = 1.0001f;
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                              "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                              : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(mm));)
        } else if (KIND == 8) {  // v_mul_u32_u24 + v_mul_hi_u32_u24
            uint32_t* p = (uint32_t*)&c0;
            REP8(asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_hi_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_hi_u32_u24 %3, %3, %8\n"
                              "v_mul_u32_u24 %4, %4, %8\n v_mul_hi_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_hi_u32_u24 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(seed | 0xF00001u));)
        } else if (KIND == 9) {  // v_mul_f64
            REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                              "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(m));)
        } else if (KIND == 10) {  // v_add_f64
            REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                              "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(ad));)
        }
    }
    uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7) ^ (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
    int iters = 60000;
    int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = one per SIMD
    uint32_t* d; hipMalloc(&d, (size_t)blocks * 256 * 4); unsigned long long* clk; hipMalloc(&clk, 16); unsigned long long hclk[2];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    bench<KIND><<<blocks, 256>>>(d, 10, 1, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    bench<KIND><<<blocks, 256>>>(d, iters, 1, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost); double ghz = (double)hclk[0] / ((double)hclk[1] * 10.0) ;
    double wave_instr = (double)blocks * 4 * iters * 64;    // wave-instructions issued
    double per_simd = wave_instr / (256.0 * 4);            // per SIMD
    double cyc = ms * 1e-3 * ghz * 1e9;                   // at the measured shader clock
    printf("%-28s waves/SIMD=%d  %8.3f ms  clk %.3f GHz  %6.2f cyc/wave-instr/SIMD  %7.2f G wave-instr/s\n", name, waves_per_simd, ms, ghz, cyc / per_simd, wave_instr / ms * 1e-6);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<7>("v_fma_f32", 1); run<0>("v_mad_u64_u32", 1); run<1>("v_fma_f64", 1); run<2>("v_mul_lo_u32", 1); run<3>("v_mul_hi_u32", 1); run<4>("v_mad_u32_u24", 1); run<5>("v_lshl_add_u64", 1); run<6>("v_add_co/addc_co_u32", 1); run<8>("v_mul(_hi)_u32_u24", 1); run<9>("v_mul_f64", 1); run<10>("v_add_f64", 1); }
        if (w == 2) { run<7>("v_fma_f32", 2); run<0>("v_mad_u64_u32", 2); run<1>("v_fma_f64", 2); run<2>("v_mul_lo_u32", 2); run<3>("v_mul_hi_u32", 2); run<4>("v_mad_u32_u24", 2); run<5>("v_lshl_add_u64", 2); run<6>("v_add_co/addc_co_u32", 2); run<8>("v_mul(_hi)_u32_u24", 2); run<9>("v_mul_f64", 2); run<10>("v_add_f64", 2); }
        if (w == 4) { run<7>("v_fma_f32", 4); run<0>("v_mad_u64_u32", 4); run<1>("v_fma_f64", 4); run<2>("v_mul_lo_u32", 4); run<3>("v_mul_hi_u32", 4); run<4>("v_mad_u32_u24", 4); run<5>("v_lshl_add_u64", 4); run<6>("v_add_co/addc_co_u32", 4); run<8>("v_mul(_hi)_u32_u24", 4); run<9>("v_mul_f64", 4); run<10>("v_add_f64", 4); }
    }
    return 0;
}
