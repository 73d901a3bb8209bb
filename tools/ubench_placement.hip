// tools/ubench_placement.hip -- where does the MI355X put the workgroups of small concurrent launches?
//
// Round 5 found that register-lean chain kernels got SLOWER beside other launches: the dispatcher does not look for an idle SIMD, it
// fills the same compute units again.  This probe records (XCC, SE, CU, SIMD) of every single-wavefront workgroup of
//   A  one launch of N workgroups that hold their SIMD's whole register file (v255 + a255 clobbered),
//   B  one launch of N workgroups of few registers,
//   C  two launches of kind B on two streams at the same time,
//   D  as C, the second launch with `pad` leading workgroups that exit at once (does a dispatch keep a placement pointer?),
// each workgroup spinning ~300 us so that all of a launch are resident together, and prints how many (XCC, CU, SIMD) slots hold more than
// one wavefront and how many distinct CUs were used.
//   hipcc --offload-arch=gfx950 -O2 -o ubench_placement tools/ubench_placement.hip && ./ubench_placement [N=129]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <tuple>
#include <vector>

__device__ __forceinline__ uint32_t hw_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v;
}
template <bool WHOLE>
__global__ void __launch_bounds__(64) k_probe(uint32_t* out, int pad, long long spin) {
    if ((int)blockIdx.x < pad) return;
    if (WHOLE) asm volatile("" ::: "v255", "a255");
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[2 * (blockIdx.x - pad)] = hw_id();
        out[2 * (blockIdx.x - pad) + 1] = xcc_id();
    }
}

struct Slot { int xcc, se, cu, simd; };
static Slot decode(uint32_t hw, uint32_t xcc) {
    Slot s;
    s.simd = (hw >> 4) & 3;
    s.cu = (hw >> 8) & 15;
    s.se = (hw >> 13) & 7;          // SH_ID at bit 12 folded below
    s.cu |= ((hw >> 12) & 1) << 4;
    s.xcc = xcc & 15;
    return s;
}
static void report(const char* what, const std::vector<std::vector<uint32_t>>& launches) {
    std::map<std::tuple<int, int, int, int>, int> per_simd;
    std::set<std::tuple<int, int, int>> cus;
    for (const auto& l : launches)
        for (size_t i = 0; i + 1 < l.size(); i += 2) {
            const Slot s = decode(l[i], l[i + 1]);
            per_simd[{s.xcc, s.se, s.cu, s.simd}]++;
            cus.insert({s.xcc, s.se, s.cu});
        }
    int shared = 0, worst = 0;
    for (auto& kv : per_simd) { if (kv.second > 1) shared++; if (kv.second > worst) worst = kv.second; }
    size_t waves = 0;
    for (const auto& l : launches) waves += l.size() / 2;
    printf("%-64s waves %4zu  SIMDs used %4zu  SIMDs with >1 wavefront %4d (worst %d)  CUs used %3zu\n", what, waves, per_simd.size(), shared, worst, cus.size());
    if (getenv("PLACEMENT_DUMP")) {
        for (size_t q = 0; q < launches.size(); q++) {
            printf("  launch %zu:", q);
            for (size_t i = 0; i + 1 < launches[q].size() && i < 2 * 40; i += 2) { const Slot s = decode(launches[q][i], launches[q][i + 1]); printf(" %d.%d.%d.%d", s.xcc, s.se, s.cu, s.simd); }
            printf("\n");
        }
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 129;
    hipStream_t s[2];
    for (auto& q : s) if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) return 1;
    uint32_t* d[2];
    for (auto& p : d) if (hipMalloc(&p, (size_t)N * 8) != hipSuccess) return 1;
    const long long spin = 30000;     // wall_clock64 ticks at 100 MHz: 300 us
    auto fetch = [&](int k) { std::vector<uint32_t> h((size_t)N * 2); (void)hipMemcpy(h.data(), d[k], (size_t)N * 8, hipMemcpyDeviceToHost); return h; };
    hipLaunchKernelGGL((k_probe<true>), dim3(N), dim3(64), 0, s[0], d[0], 0, spin);
    (void)hipDeviceSynchronize();
    report("A  one launch, whole register file per wavefront", {fetch(0)});
    hipLaunchKernelGGL((k_probe<false>), dim3(N), dim3(64), 0, s[0], d[0], 0, spin);
    (void)hipDeviceSynchronize();
    report("B  one launch, few registers", {fetch(0)});
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k_probe<false>), dim3(N), dim3(64), 0, s[0], d[0], 0, spin);
        hipLaunchKernelGGL((k_probe<false>), dim3(N), dim3(64), 0, s[1], d[1], 0, spin);
        (void)hipDeviceSynchronize();
        report("C  two launches on two streams, few registers", {fetch(0), fetch(1)});
    }
    for (int pad : {64, 129, 256, 512, 1024}) {
        hipLaunchKernelGGL((k_probe<false>), dim3(N), dim3(64), 0, s[0], d[0], 0, spin);
        hipLaunchKernelGGL((k_probe<false>), dim3(N + pad), dim3(64), 0, s[1], d[1], pad, spin);
        (void)hipDeviceSynchronize();
        char what[96];
        snprintf(what, sizeof what, "D  as C, second launch behind %d workgroups that exit at once", pad);
        report(what, {fetch(0), fetch(1)});
    }
    hipLaunchKernelGGL((k_probe<true>), dim3(N), dim3(64), 0, s[0], d[0], 0, spin);
    hipLaunchKernelGGL((k_probe<false>), dim3(N), dim3(64), 0, s[1], d[1], 0, spin);
    (void)hipDeviceSynchronize();
    report("E  whole-register launch beside a few-register launch", {fetch(0), fetch(1)});
    return 0;
}
