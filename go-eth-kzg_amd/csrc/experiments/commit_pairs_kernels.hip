// experiments/commit_pairs_kernels.hip -- BlobToKZGCommitment through the direct window table with HALF of the table points added in affine
// coordinates first (round 6).  MEASURED AND REJECTED (profiles/r06_ab_pairs.txt): bit-identical commitments, 8 % SLOWER
// than k_commit_direct_acc (2.82 against 2.60 ms per 256-blob step over four streams on one box).  Not part of the product build; compiled in by
//     tools/build_variant.sh pairs "-DMKZG_AFFINE_PAIRS" "api.hip api_verify.hip api_cells.hip api_multi.hip api_exchange.hip msm_kernels.hip" experiments/commit_pairs_kernels.hip
// (every source that sees ctx.h / DirectSlot: the flag adds members to both),
// which also enables the context option commit_affine_pairs (tools/ab_pairs.py is the A/B).  Reference path: prove.go:13-33 -> internal/kzg/srs.go:56-62 ->
// internal/multiexp/multiexp.go:20-26 (gnark G1Affine.MultiExp, whose bucket accumulation is batch-affine too: the same trick --
// one shared inversion for many independent affine additions -- applied here to pairs of rows of the fixed-base table).
// Integer modular arithmetic only -- no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../g1.hpp"
#include "../msm_kernels.h"
#include "../msm_device.hpp"

namespace kzg {

// ------------------------------------------------------------------------------------------
// k_commit_direct_acc_pairs (round 6, option commit_affine_pairs): the sum of k_commit_direct_acc with HALF of its table points added
// in AFFINE coordinates first.  The two rows of ADJACENT windows (2t, 2t + 1) of one scalar are multiples m 2^(2Ct) L_i and
// m' 2^(C(2t+1)) L_i of the same base with 0 < m, m' <= 2^(C-1): neither equal nor opposite, so their sum is an incomplete affine
// addition, 3 field multiplications given 1 / (x2 - x1) -- and the inversions of all 8 K pairs of a lane are ONE inversion by
// Montgomery's trick (1 + 2 multiplications per pair; the divsteps inversion ~40 multiplications per lane = 0.6 per pair).  One
// wavefront, four loops, each with a small live state (one loop that held the running inverse, the next prefix and the XYZZ accumulator
// together spilled 93 registers):
//     pass 1   pairs forward: the x halves of both rows (plain loads, four pairs in flight in registers), d = x2 - x1, the prefix
//              product to HBM scratch ([pair][q][lane] uint4: coalesced), the running product;
//     inverse  one Fp inversion per lane;
//     pass 2   pairs backward: both full rows by LDS DMA, 1 / d from the prefix and the running inverse, the affine sum -> HBM scratch in the
//              table's row format ([pair][chunk][lane]);
//     pass 3   k_commit_direct_acc's loop over 9 rows per scalar instead of 17: the eight sums (streamed, coalesced) and the odd window out.
// Per scalar: 8 affine additions (~6.6 multiplications each, inversion share included) + 9 mixed additions instead of 17 mixed
// additions: -12.8 % field multiplications.  The price is HBM traffic: every row is gathered twice (1.5 x with the x-only first pass),
// prefix products and sums make one round trip each -- ~6.5 GB per 256-blob launch instead of 2.3 GB, against the 6.3 TB/s the chip
// delivers on random 128-byte rows (tools/ubench_rowgather.hip, profiles/r06_ubench_rowgather.txt).
// Exceptional cases: a zero digit or a row of an identity base leaves the pair's other point (or the identity) as its "sum"; pass 1 and
// pass 2 take the same decision from the same data (the digits and d), so the prefix chain stays consistent; x2 == x1 with both points
// finite cannot happen for a table of multiples of one base and is resolved by the complete XYZZ law all the same (pair_sum_complete).
// ------------------------------------------------------------------------------------------
constexpr int DPAIR_Q = 4;        // uint4 per stored prefix product (14 words + 2 of padding)
constexpr int DPAIR_ROW_Q = 8;    // uint4 per stored sum (the table's 128-byte row: x[14] y[14] inf pad[3])
__device__ __forceinline__ void lds_read_row(G1Aff& p, const uint32_t* buf, int lane) {
    uint32_t rw[32];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const uint4 v = *reinterpret_cast<const uint4*>(buf + q * 256 + lane * 4);
        rw[4 * q] = v.x; rw[4 * q + 1] = v.y; rw[4 * q + 2] = v.z; rw[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int q = 0; q < 14; q++) { p.x.l[q] = rw[q]; p.y.l[q] = rw[14 + q]; }
    p.inf = rw[28];
#pragma unroll
    for (int q = 0; q < 14; q++) asm volatile("" : "+v"(p.x.l[q]), "+v"(p.y.l[q]));          // LDS reads retired before the next DMA into this buffer
    asm volatile("" : "+v"(p.inf));
}
// P + Q for finite affine P, Q with equal x (P = Q or P = -Q): never taken for rows of one base, kept for completeness
__device__ __noinline__ void pair_sum_complete(G1Aff& r, const fp_t& x1, const fp_t& y1, const fp_t& x2, const fp_t& y2) {
    G1X t;
    g1x_from_affine(t, x1, y1);
    g1x_madd(t, x2, y2);
    g1x_to_affine(r, t);
}
template <int C, int INPUT_KIND>
__global__ void __launch_bounds__(DTA, 2) k_commit_direct_acc_pairs(const uint8_t* __restrict__ scalars, size_t scalar_stride_bytes, const uint32_t* __restrict__ direct,
                                                                  uint32_t* __restrict__ parts, int* __restrict__ bad_flags, int K, uint4* __restrict__ prefix) {
    using RC = Recode<C>;
    constexpr bool TIGHT = (C * RC::NWIN < 256);
    constexpr int NPAIR = RC::NWIN / 2;
    constexpr bool SINGLE = (RC::NWIN & 1) != 0;
    constexpr int NJOB = NPAIR + (SINGLE ? 1 : 0);                        // rows per scalar in pass 3: the pair sums, then the odd window
    constexpr int DEPTH = 4;                                              // pairs in flight in pass 1
    __shared__ __attribute__((aligned(16))) uint32_t rowbuf[2 * 8 * 256];  // 16 KB: two table rows per lane
    __shared__ short dig[RC::NWIN * DTA];                                 // the current scalar's signed digits (sign of r - s folded in), [window][lane]
    const int half = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    uint32_t* const bufA = rowbuf;
    uint32_t* const bufB = rowbuf + 8 * 256;
    const int NP = K * NPAIR;                                             // pairs of this lane
    // scratch of this wavefront: NP prefix products [pair][q < 4][lane], then NP sums [pair][q < 8][lane]
    uint4* const wave_base = prefix + ((size_t)b * gridDim.x + half) * ((size_t)NP * (DPAIR_Q + DPAIR_ROW_Q) * 64);
    uint4* const pre_base = wave_base + lane;
    uint4* const sum_base = wave_base + (size_t)NP * DPAIR_Q * 64 + lane;

    auto recode = [&](int k) -> bool {
        const int i = (half * K + k) * DTA + lane;
        uint32_t w[8];
        bool bad = false;
        if (INPUT_KIND == 0) {
            load_scalar_be(w, scalars + (size_t)b * scalar_stride_bytes + (size_t)i * 32);
            bad = !scalar_is_canonical(w);
        } else {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(scalars + (size_t)b * scalar_stride_bytes) + (size_t)i * 12;
            fr_t m, pl;
#pragma unroll
            for (int q = 0; q < 9; q++) m.l[q] = src[q];
            Fr::to_plain(pl, m);
            Fr::to_words32(w, pl);
        }
        bool flip = false;
        if (TIGHT && ((w[7] >> 30) & 1)) {
            flip = true;
            uint64_t br = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint64_t d = (uint64_t)FrParams::MOD32[q] - w[q] - br;
                w[q] = (uint32_t)d;
                br = (d >> 32) & 1;
            }
        }
        RC::for_each_digit(w, [&](int j, int d) { dig[j * DTA + lane] = (short)(flip ? -d : d); });
        return bad;
    };
    // the row of digit sd in window j of scalar slot k; a zero digit fetches row 1 (never used): every lane issues the same loads
    auto row_of = [&](int k, int j, int sd) -> const uint32_t* {
        const int i = (half * K + k) * DTA + lane;
        int m = sd < 0 ? -sd : sd;
        if (m == 0) m = 1;
        return direct + (((size_t)j * 4096 + i) * RC::NBUCKET + (m - 1)) * 32;
    };
    auto dma_row = [&](uint32_t* buf, const uint32_t* row) {
#pragma unroll
        for (int q = 0; q < 8; q++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + 4 * q), (__attribute__((address_space(3))) void*)(buf + q * 256), 16, 0, 0);
    };
#ifdef MKZG_PAIRS_TIMING
    const long long tm0 = wall_clock64();
#endif

    // ---- pass 1: denominators and prefix products, pairs in ascending order, DEPTH pairs' x halves in flight in registers
    bool bad = false;
    fp_t prod;
    Fp::set_one(prod);
    {
        uint4 xa[DEPTH][4], xb[DEPTH][4];
        bool both[DEPTH];
        auto fetch = [&](int slot, int g) __attribute__((always_inline)) {
            const int k = g / NPAIR, t = g - k * NPAIR;
            if (t == 0) bad |= recode(k);
            const int s1 = dig[(2 * t) * DTA + lane], s2 = dig[(2 * t + 1) * DTA + lane];
            both[slot] = s1 != 0 && s2 != 0;
            const uint4* r1 = reinterpret_cast<const uint4*>(row_of(k, 2 * t, s1));
            const uint4* r2 = reinterpret_cast<const uint4*>(row_of(k, 2 * t + 1, s2));
#pragma unroll
            for (int q = 0; q < 4; q++) { xa[slot][q] = r1[q]; xb[slot][q] = r2[q]; }
        };
        static_assert(DEPTH == 4, "the four explicit steps below");
        fetch(0, 0); fetch(1, 1); fetch(2, 2); fetch(3, 3);
#pragma unroll 1
        for (int g0 = 0; g0 < NP; g0 += DEPTH) {
            auto step = [&](auto uc) __attribute__((always_inline)) {                            // slot index as a type: static register indices whatever the unroller decides
                constexpr int u = decltype(uc)::value;
                const int g = g0 + u;
                fp_t x1, x2, d;
                x1.l[0] = xa[u][0].x; x1.l[1] = xa[u][0].y; x1.l[2] = xa[u][0].z; x1.l[3] = xa[u][0].w; x1.l[4] = xa[u][1].x; x1.l[5] = xa[u][1].y; x1.l[6] = xa[u][1].z;
                x1.l[7] = xa[u][1].w; x1.l[8] = xa[u][2].x; x1.l[9] = xa[u][2].y; x1.l[10] = xa[u][2].z; x1.l[11] = xa[u][2].w; x1.l[12] = xa[u][3].x; x1.l[13] = xa[u][3].y;
                x2.l[0] = xb[u][0].x; x2.l[1] = xb[u][0].y; x2.l[2] = xb[u][0].z; x2.l[3] = xb[u][0].w; x2.l[4] = xb[u][1].x; x2.l[5] = xb[u][1].y; x2.l[6] = xb[u][1].z;
                x2.l[7] = xb[u][1].w; x2.l[8] = xb[u][2].x; x2.l[9] = xb[u][2].y; x2.l[10] = xb[u][2].z; x2.l[11] = xb[u][2].w; x2.l[12] = xb[u][3].x; x2.l[13] = xb[u][3].y;
                const bool bth = both[u];
                if (g + DEPTH < NP) fetch(u, g + DEPTH);
                Fp::template sub<0>(d, x2, x1);                   // x2 - x1 + p in (0, 2p)
                if (bth && !Fp::template is_zero_mod<2>(d)) {
                    uint4* dst = pre_base + (size_t)g * DPAIR_Q * 64;
                    dst[0] = make_uint4(prod.l[0], prod.l[1], prod.l[2], prod.l[3]);
                    dst[64] = make_uint4(prod.l[4], prod.l[5], prod.l[6], prod.l[7]);
                    dst[128] = make_uint4(prod.l[8], prod.l[9], prod.l[10], prod.l[11]);
                    dst[192] = make_uint4(prod.l[12], prod.l[13], 0, 0);
                    Fp::opaque(prod);
                    Fp::opaque(d);
                    Fp::mul(prod, prod, d);
                }
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
        }
    }
#ifdef MKZG_PAIRS_TIMING
    const long long tm1 = wall_clock64();
#endif
    fp_t inv;
    {
        fp_t pin = prod, pout;                                    // copies, for the same reason as at pair_sum_complete below
        Fp::inv(pout, pin);
        inv = pout;
    }
#ifdef MKZG_PAIRS_TIMING
    const long long tm2 = wall_clock64();
#endif

    // ---- pass 2: pairs in descending order: the affine sums, left in the scratch as table rows
    {
        uint4 pre_cur[DPAIR_Q], pre_nxt[DPAIR_Q] = {};
        int sd1, sd2;
        auto issue = [&](int g, uint4 (&pre)[DPAIR_Q]) {
            const int k = g / NPAIR, t = g - k * NPAIR;
            if (t == NPAIR - 1) recode(k);
            sd1 = dig[(2 * t) * DTA + lane];
            sd2 = dig[(2 * t + 1) * DTA + lane];
            dma_row(bufA, row_of(k, 2 * t, sd1));
            dma_row(bufB, row_of(k, 2 * t + 1, sd2));
            const uint4* src = pre_base + (size_t)g * DPAIR_Q * 64;
#pragma unroll
            for (int q = 0; q < DPAIR_Q; q++) pre[q] = src[q * 64];
        };
        auto store_sum = [&](int g, const G1Aff& o) {
            uint4* dst = sum_base + (size_t)g * DPAIR_ROW_Q * 64;
            dst[0] = make_uint4(o.x.l[0], o.x.l[1], o.x.l[2], o.x.l[3]);
            dst[64] = make_uint4(o.x.l[4], o.x.l[5], o.x.l[6], o.x.l[7]);
            dst[128] = make_uint4(o.x.l[8], o.x.l[9], o.x.l[10], o.x.l[11]);
            dst[192] = make_uint4(o.x.l[12], o.x.l[13], o.y.l[0], o.y.l[1]);
            dst[256] = make_uint4(o.y.l[2], o.y.l[3], o.y.l[4], o.y.l[5]);
            dst[320] = make_uint4(o.y.l[6], o.y.l[7], o.y.l[8], o.y.l[9]);
            dst[384] = make_uint4(o.y.l[10], o.y.l[11], o.y.l[12], o.y.l[13]);
            dst[448] = make_uint4(o.inf, 0, 0, 0);
        };
        issue(NP - 1, pre_cur);
        G1Aff o;                                                  // the sum of the pair one step back: its stores go out at the START of the next
        o.inf = 1;                                                // step, so that the wait at the top of a step never meets a store just issued
#pragma unroll 1
        for (int g = NP - 1; g >= 0; g--) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            G1Aff p1, p2;
            lds_read_row(p1, bufA, lane);
            lds_read_row(p2, bufB, lane);
            const int c1 = sd1, c2 = sd2;
            if (g > 0) issue(g - 1, pre_nxt);
            if (g < NP - 1) store_sum(g + 1, o);
            const bool v1 = c1 != 0 && !p1.inf, v2 = c2 != 0 && !p2.inf;
            if (c1 < 0) Fp::template neg<0>(p1.y, p1.y);
            if (c2 < 0) Fp::template neg<0>(p2.y, p2.y);
            o.inf = !(v1 || v2);                                  // the pair's "sum": P1 + P2, or the one valid point, or the identity
#pragma unroll
            for (int q = 0; q < 14; q++) { o.x.l[q] = v1 ? p1.x.l[q] : p2.x.l[q]; o.y.l[q] = v1 ? p1.y.l[q] : p2.y.l[q]; }
            if (c1 != 0 && c2 != 0) {
                fp_t d;
                Fp::template sub<0>(d, p2.x, p1.x);
                if (!Fp::template is_zero_mod<2>(d)) {
                    fp_t pre, dinv;
                    pre.l[0] = pre_cur[0].x; pre.l[1] = pre_cur[0].y; pre.l[2] = pre_cur[0].z; pre.l[3] = pre_cur[0].w;
                    pre.l[4] = pre_cur[1].x; pre.l[5] = pre_cur[1].y; pre.l[6] = pre_cur[1].z; pre.l[7] = pre_cur[1].w;
                    pre.l[8] = pre_cur[2].x; pre.l[9] = pre_cur[2].y; pre.l[10] = pre_cur[2].z; pre.l[11] = pre_cur[2].w;
                    pre.l[12] = pre_cur[3].x; pre.l[13] = pre_cur[3].y;
                    Fp::opaque(inv);
                    Fp::opaque(pre);
                    Fp::opaque(d);
                    Fp::mul(dinv, inv, pre);                      // 1 / (x2 - x1)
                    Fp::mul(inv, inv, d);
                    if (v1 && v2) {
                        fp_t lam, tt, x3, y3;
                        Fp::template sub<0>(tt, p2.y, p1.y);      // y2 - y1 + p
                        Fp::mul(lam, tt, dinv);
                        Fp::sqr(x3, lam);
                        Fp::add(tt, p1.x, p2.x);
                        Fp::template sub<1>(x3, x3, tt);          // lam^2 - x1 - x2 + 2p      (< 3.01p)
                        Fp::template sub<2>(tt, p1.x, x3);        // x1 - x3 + 4p
                        Fp::mul(y3, lam, tt);
                        Fp::template sub<0>(y3, y3, p1.y);        // lam (x1 - x3) - y1 + p     (< 2.01p)
                        Fp::reduce_once(o.y, y3);                 // <= 1.01p: the bound g1x_madd's exceptional doubling needs
                        o.x = x3;
                    }
                } else if (v1 && v2) {
                    // copies: the call takes addresses, and an object whose address escapes lives in scratch memory for the whole loop body
                    G1Aff so;
                    fp_t ax = p1.x, ay = p1.y, bx = p2.x, by = p2.y;
                    pair_sum_complete(so, ax, ay, bx, by);
                    o = so;
                }
            }
#pragma unroll
            for (int q = 0; q < DPAIR_Q; q++) pre_cur[q] = pre_nxt[q];
        }
        store_sum(0, o);
    }
#ifdef MKZG_PAIRS_TIMING
    const long long tm3 = wall_clock64();
#endif

    // ---- pass 3: the accumulate loop of k_commit_direct_acc over NJOB rows per scalar: its pair sums (this lane's own stores: visible
    // to its later loads in program order) and the odd window's table row; the row of job j + 1 travels to LDS while job j is added
    G1X acc;
    g1x_set_inf(acc);
    {
        bool cur = false, cneg = false;
#pragma unroll 1
        for (int g = 0; g <= K * NJOB; g++) {
            G1Aff p;
            p.inf = 1;
            if (cur) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_read_row(p, bufA, lane);
            }
            bool nxt = false, nneg = false;
            if (g < K * NJOB) {
                const int k = g / NJOB, s = g - k * NJOB;
                if (SINGLE && s == NPAIR) {
                    recode(k);
                    const int sd = dig[(2 * NPAIR) * DTA + lane];
                    if (sd != 0) {
                        nxt = true;
                        nneg = sd < 0;
                        dma_row(bufA, row_of(k, 2 * NPAIR, sd));
                    }
                } else {
                    nxt = true;
                    const uint4* src = sum_base + (size_t)(k * NPAIR + s) * DPAIR_ROW_Q * 64;
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 64), (__attribute__((address_space(3))) void*)(bufA + q * 256), 16, 0, 0);
                }
            }
            if (!p.inf) {
                if (cneg) Fp::template neg<0>(p.y, p.y);
                g1x_madd(acc, p.x, p.y);
            }
            cur = nxt;
            cneg = nneg;
        }
    }
    store_xyzz(parts + (((size_t)b * gridDim.x + half) * DTA + lane) * 64, acc);
    const int anybad = __syncthreads_or(bad);
    if (lane == 0) bad_flags[b * DBAD + half] = anybad;
#ifdef MKZG_PAIRS_TIMING
    if (lane == 0 && (b % 37) == 0 && half == 3)      // wall_clock64: 100 MHz
        printf("pairs timing blob %d: start %lld, pass1 %lld, inversion %lld, pass2 %lld, pass3 %lld (10 ns units)\n", b, tm0, tm1 - tm0, tm2 - tm1, tm3 - tm2, wall_clock64() - tm3);
#endif
}

size_t direct_prefix_bytes(int max_batch) { return (size_t)max_batch * (4096 / (8 * DTA)) * (8 * (Recode<15>::NWIN / 2)) * (DPAIR_Q + DPAIR_ROW_Q) * 64 * sizeof(uint4); }
void commit_direct_acc_pairs_launch(int input_kind, dim3 grid, hipStream_t s, const uint8_t* d_scalars, size_t stride, const uint32_t* d_direct, const DirectSlot& ws) {
    if (input_kind == 0) hipLaunchKernelGGL((k_commit_direct_acc_pairs<15, 0>), grid, dim3(DTA), 0, s, d_scalars, stride, d_direct, ws.d_parts, ws.d_bad, ws.k, ws.d_prefix);
    else hipLaunchKernelGGL((k_commit_direct_acc_pairs<15, 1>), grid, dim3(DTA), 0, s, d_scalars, stride, d_direct, ws.d_parts, ws.d_bad, ws.k, ws.d_prefix);
}

}  // namespace kzg
