// msm_kernels.hip -- fixed-base multi-scalar multiplication over BLS12-381 G1 for MI355X (gfx950).
//
// Hot path of BlobToKZGCommitment (reference prove.go:13-33 -> internal/kzg/srs.go:56-62 ->
// internal/multiexp/multiexp.go:20-26 -> gnark G1Affine.MultiExp): for every blob b of a batch,
//     C_b = sum_i  s_{b,i} * L_i          (4096 scalars, the bit-reversed Lagrange SRS L)
//
// MI355X-first design (not gnark's per-window goroutine Pippenger):
//   * the base is FIXED for the life of the Context, so the table T[j][i] = 2^(c j) L_i is
//     precomputed once (k_build_table) and all NWIN windows of a scalar fall into ONE set of
//     2^(c-1) signed-digit buckets: no per-window doublings, one bucket reduction per blob;
//   * k_digits_sort: one workgroup per blob streams the 128 KiB of big-endian scalars with
//     16-byte coalesced loads (this IS the HBM read of the path), checks canonicity, recodes
//     signed c-bit digits and counting-sorts the (point, window) pairs by bucket in LDS
//     (ds_add histograms), so that no atomics on 224-byte group elements are ever needed;
//   * k_bucket_accumulate: one lane per (blob, bucket) walks its sorted run and performs mixed
//     XYZZ additions against table rows (128-byte rows, L2 / Infinity-Cache resident);
//   * k_bucket_reduce1 / k_bucket_reduce2: work-efficient weighted sum  sum_k (k+1) B_k  by
//     lane-serial running sums over 16-bucket chunks, then a wavefront-level (64-lane DPP/
//     bpermute) scan + tree per blob, one field inversion and the 48-byte compressed output.
// Integer modular arithmetic only -- no MFMA.  Field representation: mont29.hpp.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "g1.hpp"
#include "g1_quad.hpp"
#include "msm_kernels.h"
#include "g1_decode.hpp"
#include "msm_device.hpp"

namespace kzg {

// out-of-line group operations for the low-arithmetic-intensity kernels (keeps code size sane)
__device__ __noinline__ void g1x_add_ni(G1X& r, const G1X& a, const G1X& b) { g1x_add(r, a, b); }
__device__ __noinline__ void g1x_dbl_ni(G1X& r, const G1X& a) { g1x_dbl(r, a); }

// ------------------------------------------------------------------------------------------
// kernel A: decode scalars, recode digits, counting sort by bucket, build the task list.
// One 1024-thread workgroup per blob (4096 scalars -> 4 per thread).
//   entries[b][slot] = table index (j*4096 + i) | sign << 31, grouped by bucket
//   task_off[b][k]   = id of the first task of bucket k (k = |digit| - 1); a bucket with n entries
//                      is cut into ceil(n / PIECE) tasks, so degenerate inputs (constant
//                      polynomials, the short top window) cannot serialise a lane
//   tasks[b][i]      = {first entry, task id << 8 | length}, sorted by length (descending) so
//                      that the 64 lanes of a wavefront run the same number of additions
//   ntasks[b], status[b] (0 or MKZG_ERR_NON_CANONICAL_SCALAR)
// INPUT_KIND 0: 32-byte big-endian scalars (blob bytes).  1: Fr elements in Montgomery radix-2^29
// form (rows of 12 words, as produced by the on-device quotient kernel).
// ------------------------------------------------------------------------------------------
constexpr int PIECE = 64;
template <int C>
struct TaskGeom {
    using RC = Recode<C>;
    static constexpr int NENT = 4096 * RC::NWIN;
    static constexpr int MAXT = ((RC::NBUCKET + NENT / PIECE + 255) / 256) * 256;
};

template <int C, int INPUT_KIND>
__global__ void __launch_bounds__(1024) k_digits_sort(const uint8_t* __restrict__ scalars, size_t scalar_stride_bytes, uint32_t* __restrict__ entries,
                                                      uint32_t* __restrict__ task_off, uint2* __restrict__ tasks, uint32_t* __restrict__ ntasks,
                                                      int* __restrict__ status) {
    using RC = Recode<C>;
    using TG = TaskGeom<C>;
    constexpr int NB = RC::NBUCKET;
    constexpr int PER_THREAD = NB / 1024 > 0 ? NB / 1024 : 1;
    __shared__ uint32_t hist[NB];
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t len_pos[PIECE + 1];
    const int b = blockIdx.x, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    for (int k = t; k < NB; k += 1024) hist[k] = 0;
    __syncthreads();

    uint32_t sc[4][8];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = t + 1024 * k;
        if (INPUT_KIND == 0) {
            load_scalar_be(sc[k], scalars + (size_t)b * scalar_stride_bytes + (size_t)i * 32);
            bad |= !scalar_is_canonical(sc[k]);
        } else {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(scalars + (size_t)b * scalar_stride_bytes) + (size_t)i * 12;   // 12-word Fr rows
            fr_t m, pl;
#pragma unroll
            for (int q = 0; q < 9; q++) m.l[q] = src[q];
            Fr::to_plain(pl, m);
            Fr::to_words32(sc[k], pl);
        }
    }
    if (__syncthreads_or(bad)) {
        if (t == 0) { status[b] = MKZG_ERR_NON_CANONICAL_SCALAR; ntasks[b] = 0; }
        // leave a well-formed (empty) layout behind so the later kernels are harmless
        for (int k = t; k <= NB; k += 1024) task_off[(size_t)b * (NB + 1) + k] = 0;
        return;
    }
    if (t == 0) status[b] = 0;

    // pass 1: histogram
#pragma unroll
    for (int k = 0; k < 4; k++) {
        RC::for_each_digit(sc[k], [&](int, int d) {
            if (d != 0) atomicAdd(&hist[(d < 0 ? -d : d) - 1], 1u);
        });
    }
    __syncthreads();

    // exclusive scan of hist[NB] (thread t owns PER_THREAD consecutive counters)
    const bool owner = t * PER_THREAD < NB;
    uint32_t cnt[PER_THREAD], first[PER_THREAD];
    uint32_t sum = 0;
    if (owner) {
#pragma unroll
        for (int q = 0; q < PER_THREAD; q++) { cnt[q] = hist[t * PER_THREAD + q]; sum += cnt[q]; }
    }
    auto block_excl_scan = [&](uint32_t v) -> uint32_t {   // exclusive prefix of v over the 1024 threads
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        __syncthreads();
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; w++) base += wave_tot[w];
        return base + incl - v;
    };
    uint32_t excl = block_excl_scan(sum);
    __syncthreads();   // everyone has read hist; overwrite it with the running cursors
    if (owner) {
        uint32_t run = excl;
#pragma unroll
        for (int q = 0; q < PER_THREAD; q++) { first[q] = run; hist[t * PER_THREAD + q] = run; run += cnt[q]; }
    }
    __syncthreads();

    // pass 2: scatter
    uint32_t* ent = entries + (size_t)b * TG::NENT;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = t + 1024 * k;
        RC::for_each_digit(sc[k], [&](int j, int d) {
            if (d != 0) {
                uint32_t slot = atomicAdd(&hist[(d < 0 ? -d : d) - 1], 1u);
                ent[slot] = (uint32_t)(j * 4096 + i) | (d < 0 ? 0x80000000u : 0u);
            }
        });
    }

    // task ids: bucket k owns ceil(cnt/PIECE) consecutive ids
    uint32_t np = 0;
    if (owner) {
#pragma unroll
        for (int q = 0; q < PER_THREAD; q++) np += (cnt[q] + PIECE - 1) / PIECE;
    }
    uint32_t tbase = block_excl_scan(np);
    if (owner) {
        uint32_t run = tbase;
#pragma unroll
        for (int q = 0; q < PER_THREAD; q++) {
            task_off[(size_t)b * (NB + 1) + t * PER_THREAD + q] = run;
            run += (cnt[q] + PIECE - 1) / PIECE;
        }
        if (t * PER_THREAD + PER_THREAD == NB) { task_off[(size_t)b * (NB + 1) + NB] = run; ntasks[b] = run; }
    }
    // counting sort of the tasks by length, longest first
    __syncthreads();   // pass 2 is done with hist; reuse hist[0..PIECE] as the length histogram
    if (t <= PIECE) hist[t] = 0;
    __syncthreads();
    if (owner) {
#pragma unroll
        for (int q = 0; q < PER_THREAD; q++) {
            uint32_t full = cnt[q] / PIECE, rem = cnt[q] % PIECE;
            if (full) atomicAdd(&hist[PIECE], full);
            if (rem) atomicAdd(&hist[rem], 1u);
        }
    }
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (int l = PIECE; l >= 1; l--) { len_pos[l] = run; run += hist[l]; }
    }
    __syncthreads();
    if (owner) {
        uint2* tk = tasks + (size_t)b * TG::MAXT;
        uint32_t id = tbase;
#pragma unroll
        for (int q = 0; q < PER_THREAD; q++) {
            uint32_t start = first[q], left = cnt[q];
            while (left) {
                uint32_t len = left < PIECE ? left : PIECE;
                uint32_t slot = atomicAdd(&len_pos[len], 1u);
                tk[slot] = make_uint2(start, (id << 8) | len);
                start += len; left -= len; id++;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// kernel B: one lane per task: mixed additions of table rows into an XYZZ accumulator.  The
// next table row is fetched while the current addition runs.
// ------------------------------------------------------------------------------------------
// QUAD (calls of up to 4 blobs): a quad of lanes per task, g1_quad.hpp xmadd_quad -- the lanes walk runs of up to 64 entries
template <int C, bool QUAD>
__global__ void __launch_bounds__(256) k_bucket_accumulate(const uint32_t* __restrict__ table, const uint32_t* __restrict__ entries, const uint2* __restrict__ tasks,
                                                           const uint32_t* __restrict__ ntasks, uint32_t* __restrict__ parts) {
    using TG = TaskGeom<C>;
    const QuadRole qr(threadIdx.x);
    const int b = blockIdx.y;
    const uint32_t i = (blockIdx.x * 256 + threadIdx.x) / (QUAD ? 4 : 1);
    if (i >= ntasks[b]) return;
    const uint2 d = tasks[(size_t)b * TG::MAXT + i];
    const uint32_t len = d.y & 0xffu, tid = d.y >> 8;
    const uint32_t* ent = entries + (size_t)b * TG::NENT + d.x;
    G1X acc;
    g1x_set_inf(acc);
    uint32_t v = ent[0];
    G1Aff p;
    load_affine(p, table + (size_t)(v & 0x7fffffffu) * 32);
    for (uint32_t e = 0; e < len; e++) {
        const uint32_t vn = ent[e + 1 < len ? e + 1 : e];
        G1Aff pn;
        load_affine(pn, table + (size_t)(vn & 0x7fffffffu) * 32);
        if (!p.inf) {
            if (v >> 31) Fp::template neg<0>(p.y, p.y);     // p - y  (table rows are canonical)
            if constexpr (QUAD) xmadd_quad(acc, p.x, p.y, qr);
            else g1x_madd(acc, p.x, p.y);
        }
        p = pn;
        v = vn;
    }
    if (QUAD && (threadIdx.x & 3) != 0) return;
    store_xyzz(parts + ((size_t)b * TG::MAXT + tid) * 64, acc);
}

// ------------------------------------------------------------------------------------------
// kernel R1: lane-serial running sums over chunks of 16 buckets (a bucket = its task pieces):
//   S_t = sum_{k in chunk} B_k ,   T_t = sum_{k in chunk} (k - 16 t + 1) B_k
// partials[b][t] = (S_t, T_t)
// ------------------------------------------------------------------------------------------
constexpr int CHUNK = 16;
// QUAD (calls of a few blobs, where somebody waits for this 32- to 48-addition chain): four lanes per chunk, g1_quad.hpp xadd_quad
template <int C, bool QUAD>
__global__ void __launch_bounds__(64, 2) k_bucket_reduce1(const uint32_t* __restrict__ parts, const uint32_t* __restrict__ task_off, const int* __restrict__ status,
                                                       uint32_t* __restrict__ partials) {
    using RC = Recode<C>;
    using TG = TaskGeom<C>;
    constexpr int NCHUNK = RC::NBUCKET / CHUNK;
    static_assert(NCHUNK % 64 == 0, "whole wavefronts of chunks (the quad form needs all four lanes of a quad)");
    const QuadRole qr(threadIdx.x);
    const int b = blockIdx.y;
    const int t = (blockIdx.x * 64 + threadIdx.x) / (QUAD ? 4 : 1);
    if (t >= NCHUNK || status[b] != 0) return;
    const uint32_t* off = task_off + (size_t)b * (RC::NBUCKET + 1) + t * CHUNK;
    const uint32_t* src = parts + (size_t)b * TG::MAXT * 64;
    G1X run, sum, bk;
    g1x_set_inf(run);
    g1x_set_inf(sum);
    for (int q = CHUNK - 1; q >= 0; q--) {
        const uint32_t p0 = off[q], p1 = off[q + 1];
        for (uint32_t p = p0; p < p1; p++) {
            load_xyzz(bk, src + (size_t)p * 64);
            if constexpr (QUAD) xadd_quad(run, bk, qr);
            else g1x_add_ni(run, run, bk);
        }
        if constexpr (QUAD) xadd_quad(sum, run, qr);
        else g1x_add_ni(sum, sum, run);
    }
    if (QUAD && (threadIdx.x & 3) != 0) return;
    uint32_t* dst = partials + ((size_t)b * NCHUNK + t) * 128;
    store_xyzz(dst, run);
    store_xyzz(dst + 64, sum);
}

// wave-wide helpers on whole XYZZ points (56 words through ds_bpermute)
__device__ __forceinline__ void shfl_down_xyzz(G1X& r, const G1X& a, int d) {
#pragma unroll
    for (int i = 0; i < 14; i++) {
        r.x.l[i] = __shfl_down(a.x.l[i], d);
        r.y.l[i] = __shfl_down(a.y.l[i], d);
        r.zz.l[i] = __shfl_down(a.zz.l[i], d);
        r.zzz.l[i] = __shfl_down(a.zzz.l[i], d);
    }
}

__device__ __forceinline__ void bcast0_xyzz(G1X& a) {
#pragma unroll
    for (int i = 0; i < 14; i++) {
        a.x.l[i] = __shfl(a.x.l[i], 0);
        a.y.l[i] = __shfl(a.y.l[i], 0);
        a.zz.l[i] = __shfl(a.zz.l[i], 0);
        a.zzz.l[i] = __shfl(a.zzz.l[i], 0);
    }
}
// ------------------------------------------------------------------------------------------
// kernel R2: one wavefront per blob finishes  sum_k (k+1) B_k = sum_t T_t + 16 sum_t t S_t,
// converts to affine and writes the 48-byte compressed commitment.
// Lane j owns chunks t = M j .. M j + M - 1 (M = chunks per blob / 64).
// ------------------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(64, 2) k_bucket_reduce2(const uint32_t* __restrict__ partials, const int* __restrict__ status, uint8_t* __restrict__ out48) {
    using RC = Recode<C>;
    constexpr int NCHUNK = RC::NBUCKET / CHUNK;
    constexpr int M = NCHUNK / 64;
    static_assert(M >= 1 && (M & (M - 1)) == 0, "bucket count must give a power-of-two number of chunks per lane");
    const int b = blockIdx.x, lane = threadIdx.x;
    if (status[b] != 0) return;
    const uint32_t* src = partials + ((size_t)b * NCHUNK + (size_t)lane * M) * 128;
    // per lane: U = sum_i T_i ; SS = sum_i S_i ; Wt = sum_i i S_i   (i = 0..M-1)
    G1X U, run, wsum, t;
    g1x_set_inf(U);
    g1x_set_inf(run);
    g1x_set_inf(wsum);
    for (int i = M - 1; i >= 0; i--) {
        load_xyzz(t, src + i * 128 + 64);
        g1x_add_ni(U, U, t);
        load_xyzz(t, src + i * 128);
        g1x_add_ni(run, run, t);
        if (i > 0) g1x_add_ni(wsum, wsum, run);
    }
    // X = U + 16 * Wt
    for (int q = 0; q < 4; q++) g1x_dbl_ni(wsum, wsum);
    G1X X;
    g1x_add_ni(X, U, wsum);
    // suffix scan of SS over lanes, then sum over lanes 1..63  ->  sum_j j SS_j
    G1X P = run, o;
    for (int d = 1; d < 64; d <<= 1) {
        shfl_down_xyzz(o, P, d);
        if (lane + d < 64) g1x_add_ni(P, P, o);
    }
    if (lane == 0) g1x_set_inf(P);
    for (int d = 32; d >= 1; d >>= 1) {
        shfl_down_xyzz(o, P, d);
        if (lane < d) g1x_add_ni(P, P, o);
        shfl_down_xyzz(o, X, d);
        if (lane < d) g1x_add_ni(X, X, o);
    }
    {
        // total = X + (16 M) * P, held by lane 0 after the trees: broadcast, then every lane runs the serial tail (doublings, inversion,
        // compression) and lane 0 stores -- a wavefront with one active lane issues slower on gfx950 (fk20_kernels.hip FULL_WAVES)
        bcast0_xyzz(X);
        bcast0_xyzz(P);
        int sh = 4;
        for (int m = M; m > 1; m >>= 1) sh++;
        for (int q = 0; q < sh; q++) g1x_dbl_ni(P, P);
        g1x_add_ni(X, X, P);
        G1Aff a;
        g1x_to_affine(a, X);
        uint32_t w[12];
        g1_compress_words(w, a);
        if (lane == 0) {
            uint32_t* o32 = reinterpret_cast<uint32_t*>(out48 + (size_t)b * 48);
#pragma unroll
            for (int i = 0; i < 12; i++) o32[i] = __builtin_bswap32(w[i]);
        }
    }
}
// The same reduction for calls of a few blobs on QUADS of lanes (g1_quad.hpp): FOUR wavefronts, 64 groups of four lanes with
// NCHUNK / 64 chunks each as above; the ~45 group operations of the chain (loop, suffix scan, trees, doublings) are four or three
// products deep instead of fourteen or nine, and the scan and the trees cross the wavefronts through LDS (one record per wavefront).
template <int C>
__global__ void __launch_bounds__(256) k_bucket_reduce2_quad(const uint32_t* __restrict__ partials, const int* __restrict__ status, uint8_t* __restrict__ out48) {
    using RC = Recode<C>;
    constexpr int NCHUNK = RC::NBUCKET / CHUNK;
    constexpr int M = NCHUNK / 64;
    static_assert(M >= 1 && (M & (M - 1)) == 0, "bucket count must give a power-of-two number of chunks per group");
    __shared__ uint32_t xw[3][4][56];                        // per wavefront: its SS total, its P partial, its X partial
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gl = lane >> 2, g = tid >> 2;
    const QuadRole qr(tid);
    if (status[b] != 0) return;
    auto put = [&](int k, const G1X& v) {
#pragma unroll
        for (int i = 0; i < 14; i++) { xw[k][wave][i] = v.x.l[i]; xw[k][wave][14 + i] = v.y.l[i]; xw[k][wave][28 + i] = v.zz.l[i]; xw[k][wave][42 + i] = v.zzz.l[i]; }
    };
    auto get = [&](G1X& v, int k, int w) {
#pragma unroll
        for (int i = 0; i < 14; i++) { v.x.l[i] = xw[k][w][i]; v.y.l[i] = xw[k][w][14 + i]; v.zz.l[i] = xw[k][w][28 + i]; v.zzz.l[i] = xw[k][w][42 + i]; }
    };
    const uint32_t* src = partials + ((size_t)b * NCHUNK + (size_t)g * M) * 128;
    G1X U, run, wsum, t;
    g1x_set_inf(U);
    g1x_set_inf(run);
    g1x_set_inf(wsum);
#pragma unroll 1
    for (int i = M - 1; i >= 0; i--) {
        load_xyzz(t, src + i * 128 + 64);
        xadd_quad(U, t, qr);
        load_xyzz(t, src + i * 128);
        xadd_quad(run, t, qr);
        if (i > 0) xadd_quad(wsum, run, qr);
    }
#pragma unroll 1
    for (int q = 0; q < 4; q++) xdbl_quad(wsum, qr);
    G1X X = U;
    xadd_quad(X, wsum, qr);                                  // X = U + 16 Wt
    G1X P = run, o;
#pragma unroll 1
    for (int d = 1; d < 16; d <<= 1) {                       // suffix scan of SS over the 16 groups of this wavefront
        shfl_down_xyzz(o, P, 4 * d);
        if (gl + d < 16) xadd_quad(P, o, qr);
    }
    if (lane == 0) put(0, P);                                // group 0 of the wavefront now holds the wavefront's total
    __syncthreads();
#pragma unroll 1
    for (int w = wave + 1; w < 4; w++) {                     // ... plus the totals of the wavefronts after this one
        get(o, 0, w);
        xadd_quad(P, o, qr);
    }
    if (g == 0) g1x_set_inf(P);
#pragma unroll 1
    for (int d = 8; d >= 1; d >>= 1) {                       // sum_g g SS_g and sum_g X_g, first inside the wavefront
        shfl_down_xyzz(o, P, 4 * d);
        if (gl < d) xadd_quad(P, o, qr);
        shfl_down_xyzz(o, X, 4 * d);
        if (gl < d) xadd_quad(X, o, qr);
    }
    if (lane == 0) { put(1, P); put(2, X); }
    __syncthreads();
    if (wave != 0) return;
    get(P, 1, 0);                                            // every lane of wavefront 0 adds the four partials (replicated: full wavefront)
    get(X, 2, 0);
#pragma unroll 1
    for (int w = 1; w < 4; w++) {
        get(o, 1, w);
        xadd_quad(P, o, qr);
        get(o, 2, w);
        xadd_quad(X, o, qr);
    }
    int sh = 4;
    for (int m = M; m > 1; m >>= 1) sh++;
#pragma unroll 1
    for (int q = 0; q < sh; q++) xdbl_quad(P, qr);
    xadd_quad(X, P, qr);
    G1Aff a;
    g1x_to_affine(a, X);
    uint32_t w[12];
    g1_compress_words(w, a);
    if (lane == 0) {
        uint32_t* o32 = reinterpret_cast<uint32_t*>(out48 + (size_t)b * 48);
#pragma unroll
        for (int i = 0; i < 12; i++) o32[i] = __builtin_bswap32(w[i]);
    }
}

// ------------------------------------------------------------------------------------------
// set-up kernels
// ------------------------------------------------------------------------------------------
// zcash-compressed G1 -> affine record (g1_decode.hpp).  No r-torsion test here: trusted-setup points are decoded
// without it (reference trusted_setup.go:104-117); the verifiers' points go through k_g1_validate (verify_kernels.hip).
// Two arrays in one launch: the square root is a ~380-squaring chain per lane (latency-bound).
__global__ void __launch_bounds__(64) k_g1_decompress(const uint8_t* __restrict__ in48_a, int n_a, uint32_t* __restrict__ out_aff_a, int* __restrict__ st_a,
                                                      const uint8_t* __restrict__ in48_b, int n_b, uint32_t* __restrict__ out_aff_b, int* __restrict__ st_b) {
    int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_a + n_b) return;
    const uint8_t* in48 = in48_a;
    uint32_t* out_aff = out_aff_a;
    int* st = st_a;
    if (i >= n_a) { i -= n_a; in48 = in48_b; out_aff = out_aff_b; st = st_b; }
    G1Aff a;
    const int rc = g1_decode_lane(a, in48 + (size_t)i * 48);
    st[i] = rc;
    store_affine(out_aff + (size_t)i * 32, a);
}

// table[j][i] = 2^(C j) * src[perm(i)], canonical affine rows.  perm = bit reversal over 4096 when
// bitrev != 0 (reference api.go:131 commitKeyLagrange.ReversePoints()).
template <int C>
__global__ void __launch_bounds__(64) k_build_table(const uint32_t* __restrict__ src_aff, uint32_t* __restrict__ table, int n, int bitrev_bits) {
    using RC = Recode<C>;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    int s = i;
    if (bitrev_bits) s = (int)(__brev((uint32_t)i) >> (32 - bitrev_bits));
    G1Aff a;
    load_affine(a, src_aff + (size_t)s * 32);
    G1X acc;
    if (a.inf) g1x_set_inf(acc); else g1x_from_affine(acc, a.x, a.y);
    for (int j = 0; j < RC::NWIN; j++) {
        if (j > 0) {
            for (int q = 0; q < C; q++) g1x_dbl_ni(acc, acc);
            g1x_to_affine(a, acc);
            if (!a.inf) g1x_from_affine(acc, a.x, a.y);   // restart from Z = 1 (keeps values tight)
        }
        store_affine(table + ((size_t)j * n + i) * 32, a);
    }
}

// ------------------------------------------------------------------------------------------
// DIRECT-TABLE commitment (context option "commit_direct_bits" = C, 12..15): trade HBM capacity for
// arithmetic.  The table holds every signed-digit multiple of every window base,
//     D[(j*4096 + i)][m-1] = m * 2^(C j) * L_i ,   m = 1 .. 2^(C-1)         (C=13: 43 GB, C=15: 146 GB of the 288 GB)
// so a commitment is 4096 * ceil(255/C) mixed additions into per-lane accumulators and one tree sum per blob:
// no buckets, no sort, no bucket reduction, ONE kernel per batch.
// ------------------------------------------------------------------------------------------
// Table construction in C - 1 launches instead of one 2^(C-1)-step chain per row:
//   k_build_direct_pow2 : D[row][2^k - 1] = 2^k B_row                      (k = 0 .. C-1, one lane per row)
//   k_build_direct_level: D[row][2^k + t - 1] = D[row][t - 1] + 2^k B_row  (t = 1 .. 2^k - 1), level k = 1 .. C-2
// A level's additions are independent AFFINE additions with a common addend; each lane takes eight of them and shares one field
// inversion among them (Montgomery's trick: 1 + 2 multiplications per element), so an entry costs ~6 multiplications + 1/8
// inversion instead of a mixed addition + a full inversion.  t B = +-2^k B is impossible for 0 < t < 2^k < r, so the only
// exceptional case is an identity base (all its multiples are the identity).
template <int C>
__global__ void __launch_bounds__(64) k_build_direct_pow2(const uint32_t* __restrict__ table /* [NWIN][4096] window bases */, uint32_t* __restrict__ direct) {
    using RC = Recode<C>;
    const int row = blockIdx.x * 64 + threadIdx.x;            // row = j*4096 + i
    if (row >= RC::NWIN * 4096) return;
    G1Aff a;
    load_affine(a, table + (size_t)row * 32);
    uint32_t* dst = direct + (size_t)row * RC::NBUCKET * 32;
    G1X acc;
    if (!a.inf) g1x_from_affine(acc, a.x, a.y);
    for (int k = 0; k < C; k++) {
        if (k > 0 && !a.inf) {
            g1x_dbl_ni(acc, acc);
            g1x_to_affine(a, acc);
            g1x_from_affine(acc, a.x, a.y);                   // keep Z = 1: tight values
        }
        store_affine(dst + ((size_t)(1u << k) - 1) * 32, a);
    }
}
template <int C>
__global__ void __launch_bounds__(64, 2) k_build_direct_level(uint32_t* __restrict__ direct, int k) {
    using RC = Recode<C>;
    constexpr int G = 8;
    const int n_t = (1 << k) - 1, per_row = (n_t + G - 1) / G;
    const size_t gid = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (gid >= (size_t)RC::NWIN * 4096 * per_row) return;
    const size_t row = gid / per_row;
    const int t0 = 1 + (int)(gid % per_row) * G;
    const int cnt = n_t - (t0 - 1) < G ? n_t - (t0 - 1) : G;
    uint32_t* base = direct + row * RC::NBUCKET * 32;
    G1Aff P;
    load_affine(P, base + ((size_t)(1u << k) - 1) * 32);
    uint32_t* dst = base + ((size_t)(1u << k) + t0 - 1) * 32;
    if (P.inf) {
        for (int g = 0; g < cnt; g++) store_affine(dst + (size_t)g * 32, P);
        return;
    }
    const uint32_t* src = base + ((size_t)t0 - 1) * 32;
    fp_t pre[G], acc, d;
    Fp::set_one(acc);
#pragma unroll
    for (int g = 0; g < G; g++) {
        if (g < cnt) {
            fp_t xg;
#pragma unroll
            for (int q = 0; q < 14; q++) xg.l[q] = src[(size_t)g * 32 + q];
            Fp::template sub<0>(d, P.x, xg);                  // xP - x_g + p, non-zero mod p
            pre[g] = acc;
            Fp::mul(acc, acc, d);
        }
    }
    fp_t inv;
    Fp::inv(inv, acc);
#pragma unroll 1
    for (int g = G - 1; g >= 0; g--) {       // rolled: one copy of the addition; pre[] then lives in the lane's scratch memory
        if (g < cnt) {
            G1Aff e, o;
            load_affine(e, src + (size_t)g * 32);
            fp_t dinv, lam, t, x3, y3;
            Fp::template sub<0>(d, P.x, e.x);
            Fp::mul(dinv, inv, pre[g]);                       // 1 / (xP - x_g)
            Fp::mul(inv, inv, d);
            Fp::template sub<0>(t, P.y, e.y);
            Fp::mul(lam, t, dinv);                            // (yP - y_g) / (xP - x_g)
            Fp::sqr(x3, lam);
            Fp::add(t, P.x, e.x);
            Fp::template sub<1>(x3, x3, t);                   // lam^2 - xP - x_g + 2p      (< 3.01p)
            Fp::template sub<2>(t, e.x, x3);                  // x_g - x3 + 4p
            Fp::mul(y3, lam, t);
            Fp::template sub<0>(y3, y3, e.y);                 // lam (x_g - x3) - y_g + p
            Fp::canon(o.x, x3);
            Fp::canon(o.y, y3);
            o.inf = 0;
            store_affine(dst + (size_t)g * 32, o);
        }
    }
}

// DTA, DPARTS, DBAD: msm_device.hpp
__device__ __forceinline__ void shfl_xor_xyzz(G1X& r, const G1X& a, int m) {
#pragma unroll
    for (int i = 0; i < 14; i++) {
        r.x.l[i] = __shfl_xor(a.x.l[i], m);
        r.y.l[i] = __shfl_xor(a.y.l[i], m);
        r.zz.l[i] = __shfl_xor(a.zz.l[i], m);
        r.zzz.l[i] = __shfl_xor(a.zzz.l[i], m);
    }
}
// Accumulate: grid (4096 / (K * DTA), n blobs).  Each lane walks its K scalars window by window (the 256-bit scalar is a
// shift register: static register indices only), prefetching the table row of window j+1 while the row of
// window j is added.  K = 8 is the throughput shape (256 blobs fill the chip twice over with 136 additions per lane); calls of
// fewer blobs spread each blob over more lanes (K = 4, 2, 1: 4096 lanes per blob, 17 additions each) and pay for it with the
// log-depth fold of the partial sums below -- the caller of one blob waits for 17 + 12 dependent additions instead of 136 + 14, and
// the 12 run on a quad of lanes each.  When C divides 255 the signed recoding has no spare top window; scalars with bit 254
// set are replaced by r - s (top digit then <= 2^(C-1)) and every digit sign is flipped.
template <int C, int INPUT_KIND>     // INPUT_KIND 0: 32-byte big-endian scalars (blobs); 1: 12-word Montgomery Fr rows (quotient polynomials)
// wsplit 2 (K = 1, calls of up to eight blobs: the caller waits for this chain): a PAIR of lanes per scalar, lane 2q the lower ceil(NWIN / 2)
// windows, lane 2q + 1 the upper ones (after running the recoding's carry through the lower ones): 9 dependent additions instead of 17
// on twice the wavefronts (128 per blob); the two records of a scalar are added by the first fold launch (k_commit_direct_fold16 `pairs`).
__global__ void __launch_bounds__(DTA, 2) k_commit_direct_acc(const uint8_t* __restrict__ scalars, size_t scalar_stride_bytes, const uint32_t* __restrict__ direct,
                                                            uint32_t* __restrict__ parts, int* __restrict__ bad_flags, int K, int wsplit) {
    using RC = Recode<C>;
    constexpr bool TIGHT = (C * RC::NWIN < 256);
    __shared__ __attribute__((aligned(16))) uint32_t rowbuf[(DTA / 64) * 8 * 256];     // 32 KB: one table row per lane
    const int half = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63;
    const int part = wsplit == 2 ? (t & 1) : 0;
    const int NS = wsplit == 2 ? (RC::NWIN + 1) / 2 : RC::NWIN;      // windows of this lane: [j0, min(j0 + NS, NWIN))
    const int j0 = part * NS;
    __shared__ uint32_t scal[8 * DTA];              // this lane's scalar, word-major: the window position is wave-uniform
    uint32_t* wb = rowbuf + (t >> 6) * (8 * 256);
    bool bad = false;
    G1X acc;
    g1x_set_inf(acc);
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        const int i = wsplit == 2 ? half * (DTA / 2) + (t >> 1) : (half * K + k) * DTA + t;
        uint32_t w[8];
        if (INPUT_KIND == 0) {
            load_scalar_be(w, scalars + (size_t)b * scalar_stride_bytes + (size_t)i * 32);
            bad |= !scalar_is_canonical(w);
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
        // software pipeline through LDS: the table row of window j+1 travels HBM -> LDS with global_load_lds_dwordx4 (no
        // VGPR destination) while window j is added; a row is read back from LDS only when its addition starts, so the
        // 28 coordinate registers are live for the first two multiplications only (holding the next row in VGPRs across
        // the addition spills).  Per wavefront: 8 chunks x 64 lanes x 16 B, lane-contiguous as the DMA writes them.
#pragma unroll
        for (int q = 0; q < 8; q++) scal[q * DTA + t] = w[q];
        int carry = 0;
#pragma unroll 1
        for (int jj = 0; jj < j0; jj++) {                // the upper half's carry-in: the recoding of the windows below, nothing added
            const int bit = C * jj, wi = bit >> 5, off = bit & 31;
            const uint32_t lo = scal[wi * DTA + t], hi = scal[(wi < 7 ? wi + 1 : 7) * DTA + t];
            const uint32_t v = (uint32_t)((((uint64_t)(wi < 7 ? hi : 0u) << 32) | lo) >> off);
            carry = ((int)(v & ((1u << C) - 1)) + carry) > RC::NBUCKET ? 1 : 0;
        }
        bool cur = false, cneg = false;
#pragma unroll 1
        for (int sl = 0; sl <= NS; sl++) {
            const int j = j0 + sl;
            G1Aff p;
            p.inf = 1;
            if (cur) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                uint32_t rw[32];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint4 v = *reinterpret_cast<const uint4*>(wb + q * 256 + lane * 4);
                    rw[4 * q] = v.x; rw[4 * q + 1] = v.y; rw[4 * q + 2] = v.z; rw[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int q = 0; q < 14; q++) { p.x.l[q] = rw[q]; p.y.l[q] = rw[14 + q]; }
                p.inf = rw[28];
#pragma unroll
                for (int q = 0; q < 14; q++) asm volatile("" : "+v"(p.x.l[q]), "+v"(p.y.l[q]));   // LDS reads retired before the next DMA
                asm volatile("" : "+v"(p.inf));
            }
            bool nxt = false, nneg = false;
            if (sl < NS && j < RC::NWIN) {
                const int bit = C * j, wi = bit >> 5, off = bit & 31;
                const uint32_t lo = scal[wi * DTA + t], hi = scal[(wi < 7 ? wi + 1 : 7) * DTA + t];
                const uint32_t v = (uint32_t)((((uint64_t)(wi < 7 ? hi : 0u) << 32) | lo) >> off);
                int dn = (int)(v & ((1u << C) - 1)) + carry;
                carry = 0;
                if (dn > RC::NBUCKET) { dn -= (1 << C); carry = 1; }
                if (dn != 0) {
                    const int m = dn < 0 ? -dn : dn;
                    nneg = (dn < 0) != flip;
                    nxt = true;
                    const uint32_t* row = direct + (((size_t)j * 4096 + i) * RC::NBUCKET + (m - 1)) * 32;
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + 4 * q),
                                                         (__attribute__((address_space(3))) void*)(wb + q * 256), 16, 0, 0);
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
    store_xyzz(parts + (((size_t)b * gridDim.x + half) * DTA + t) * 64, acc);
    const int anybad = __syncthreads_or(bad);
    if (t == 0) bad_flags[b * DBAD + half] = anybad;
}
// Fold (K < 8, calls of few blobs: somebody waits for this chain): 16 records per wavefront, a QUAD of lanes per record (g1_quad.hpp
// xadd_quad: an addition four products deep instead of fourteen), xor butterfly over the 16 quads, the sum left in the first record.
// Two launches take a blob's 4096 / K records (stride 1) to 4096 / 16K (stride 16) to 4096 / 256K (stride 256) for the finish kernel.
// pairs (the window-split shape's first launch): a wavefront takes 32 records, every quad adds its two first (a scalar's lower and upper windows)
__global__ void __launch_bounds__(64, 2) k_commit_direct_fold16(uint32_t* __restrict__ parts, int per_blob /* records of a blob, in units of `stride` */, int stride, int pairs) {
    const QuadRole qr(threadIdx.x);
    const size_t base = (size_t)blockIdx.y * per_blob * stride + (size_t)blockIdx.x * (16 << pairs) * stride;
    G1X acc, o;
    load_xyzz(acc, parts + (base + ((size_t)(threadIdx.x >> 2) << pairs) * stride) * 64);
    if (pairs) {
        load_xyzz(o, parts + (base + (((size_t)(threadIdx.x >> 2) << 1) + 1) * stride) * 64);
        xadd_quad(acc, o, qr);
    }
#pragma unroll 1
    for (int m = 32; m >= 4; m >>= 1) {
        shfl_xor_xyzz(o, acc, m);
        xadd_quad(acc, o, qr);
    }
    if (threadIdx.x == 0) store_xyzz(parts + base * 64, acc);
}
// Serial pre-sum for the shapes with more than eight 64-record blocks per blob (129 .. DIRECT_MID_MAX blobs): wavefront w of a blob adds
// blocks w, w + W, ..., w + 7 W lane by lane into block w (its own first input: no other wavefront reads or writes it), W = blocks / 8,
// so that the finish kernel's chain is W + 6 additions instead of 8 W + 6.
__global__ void __launch_bounds__(64, 2) k_commit_direct_sum8(uint32_t* __restrict__ parts, int blob_blocks) {
    const int W = blob_blocks / 8, w = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    uint32_t* base = parts + ((size_t)b * blob_blocks * 64 + lane) * 64;
    G1X acc, o;
    load_xyzz(acc, base + (size_t)w * 64 * 64);
#pragma unroll 1
    for (int j = 1; j < 8; j++) {
        load_xyzz(o, base + (size_t)(w + j * W) * 64 * 64);
        g1x_add(acc, acc, o);
    }
    store_xyzz(base + (size_t)w * 64 * 64, acc);
}
// Finish: one wavefront per blob sums the 512 lane accumulators (8 per lane, then an xor butterfly), converts to
// affine and writes the compressed commitment.  Tiny and latency-bound: it runs beside the next accumulate kernel.
// folded (K < 8): the blob's nblk / 4 records sit 256 apart (two rounds of k_commit_direct_fold16), a quad of lanes per record;
// quads beyond them start from the identity.
// raw (may be null): leave the blob's sum as it is -- 56 limb words X, Y, ZZ, ZZZ at raw + 64 b -- instead of the 48 compressed bytes: the
// host-pointer entry points of a few blobs convert and compress on the HOST (api.hip commit_host_piece): the inversion is an ~80 us chain
// for one wavefront and ~5 us for the calling thread, which waits for this kernel anyway.
__global__ void __launch_bounds__(64, 2) k_commit_direct_fin(const uint32_t* __restrict__ parts, const int* __restrict__ bad_flags, uint8_t* __restrict__ out48,
                                                          int* __restrict__ status, int blob_blocks /* 64-record blocks per blob = accumulate workgroups */,
                                                          int nblk /* blocks to sum (the first ones) */, int folded, uint32_t* __restrict__ raw) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int anybad = 0;
    for (int q = 0; q < blob_blocks; q++) anybad |= bad_flags[b * DBAD + q];
    if (anybad) {
        if (lane == 0) status[b] = MKZG_ERR_NON_CANONICAL_SCALAR;
        return;
    }
    G1X acc, o;
    const size_t base = (size_t)b * blob_blocks * 64;
    if (folded) {                                    // = the stride of the records the folds left: 256, or 512 behind the window-split shape
        const QuadRole qr(lane);
        g1x_set_inf(acc);
        if ((lane >> 2) < nblk / 4) load_xyzz(acc, parts + (base + (size_t)(lane >> 2) * folded) * 64);
#pragma unroll 1
        for (int m = 32; m >= 4; m >>= 1) {
            shfl_xor_xyzz(o, acc, m);
            xadd_quad(acc, o, qr);
        }
    } else {
        load_xyzz(acc, parts + (base + lane) * 64);
#pragma unroll 1
        for (int q = 1; q < nblk; q++) {
            load_xyzz(o, parts + (base + q * 64 + lane) * 64);
            g1x_add(acc, acc, o);
        }
    }
#pragma unroll 1
    for (int m = folded ? 0 : 32; m >= 1; m >>= 1) {
        shfl_xor_xyzz(o, acc, m);
        g1x_add(acc, acc, o);
    }
    if (raw) {
        if (lane == 0) {
            status[b] = 0;
            store_xyzz(raw + (size_t)b * 64, acc);
        }
        return;
    }
    // after the xor butterfly every lane holds the sum: all of them run the inversion and the compression (a wavefront with one
    // active lane issues slower on gfx950, see fk20_kernels.hip FULL_WAVES), lane 0 stores
    G1Aff a;
    g1x_to_affine(a, acc);
    uint32_t w12[12];
    g1_compress_words(w12, a);
    if (lane == 0) {
        status[b] = 0;
        uint32_t* o32 = reinterpret_cast<uint32_t*>(out48 + (size_t)b * 48);
#pragma unroll
        for (int i = 0; i < 12; i++) o32[i] = __builtin_bswap32(w12[i]);
    }
}

// The serial form of the finish (8 blocks of 64 records per blob: 256 blobs at eight scalars per lane, or what k_commit_direct_sum8
// leaves of the one-scalar shape) for the caller who WAITS for it -- one stream, the host-pointer entry points: four wavefronts per blob,
// a QUAD of lanes per record column (g1_quad.hpp: an addition four products deep instead of fourteen), quad q adds records q, q + 64,
// ... serially, then a tree over the 64 quads through LDS, then every lane of the first wavefront runs the inversion and the
// compression on the same value (full wavefronts issue faster, see k_commit_direct_fin) and lane 0 stores.  14 dependent additions
// like the one-wavefront form, each a third as long: measured 0.28 -> 0.18 ms behind the last accumulate wavefront of 256 blobs.  2.1x
// the instructions on four times the wavefronts, so only when nothing else is on the GPU (Lane::others_active): behind another call's
// accumulate kernel the one-wavefront form hides entirely and costs less.
__global__ void __launch_bounds__(256) k_commit_direct_fin_quad(const uint32_t* __restrict__ parts, const int* __restrict__ bad_flags, uint8_t* __restrict__ out48,
                                                               int* __restrict__ status, int blob_blocks, int nblk) {
    __shared__ uint32_t pbuf[64 * 56];
    const int b = blockIdx.x, t = threadIdx.x, q = t >> 2, role = t & 3;
    int anybad = 0;
    for (int k = 0; k < blob_blocks; k++) anybad |= bad_flags[b * DBAD + k];
    if (anybad) {
        if (t == 0) status[b] = MKZG_ERR_NON_CANONICAL_SCALAR;
        return;
    }
    const QuadRole qr(t);
    const size_t base = (size_t)b * blob_blocks * 64;
    G1X acc, o;
    load_xyzz(acc, parts + (base + q) * 64);
#pragma unroll 1
    for (int k = 1; k < nblk; k++) {
        load_xyzz(o, parts + (base + (size_t)k * 64 + q) * 64);
        xadd_quad(acc, o, qr);
    }
#pragma unroll 1
    for (int d = 32; d >= 1; d >>= 1) {
        fp_t mine;
        quad_pick(mine, qr, acc.x, acc.y, acc.zz, acc.zzz);
        if (q < 2 * d) {
#pragma unroll
            for (int i = 0; i < 14; i++) pbuf[q * 56 + role * 14 + i] = mine.l[i];
        }
        __syncthreads();
        const bool take = q < d;
        if (take) {
            const uint32_t* src = pbuf + (q + d) * 56;
#pragma unroll
            for (int i = 0; i < 14; i++) { o.x.l[i] = src[i]; o.y.l[i] = src[14 + i]; o.zz.l[i] = src[28 + i]; o.zzz.l[i] = src[42 + i]; }
        }
        __syncthreads();
        if (take) xadd_quad(acc, o, qr);
    }
    // the sum sits in quad 0: publish it, and let the whole first wavefront finish it
    if (q == 0) {
        fp_t mine;
        quad_pick(mine, qr, acc.x, acc.y, acc.zz, acc.zzz);
#pragma unroll
        for (int i = 0; i < 14; i++) pbuf[role * 14 + i] = mine.l[i];
    }
    __syncthreads();
    if (t >= 64) return;
#pragma unroll
    for (int i = 0; i < 14; i++) { acc.x.l[i] = pbuf[i]; acc.y.l[i] = pbuf[14 + i]; acc.zz.l[i] = pbuf[28 + i]; acc.zzz.l[i] = pbuf[42 + i]; }
    G1Aff a;
    g1x_to_affine(a, acc);
    uint32_t w12[12];
    g1_compress_words(w12, a);
    if (t == 0) {
        status[b] = 0;
        uint32_t* o32 = reinterpret_cast<uint32_t*>(out48 + (size_t)b * 48);
#pragma unroll
        for (int i = 0; i < 12; i++) o32[i] = __builtin_bswap32(w12[i]);
    }
}

// ------------------------------------------------------------------------------------------
// VALU calibration: what this particular GPU sustains on the multiplier's instruction mix -- eight v_mad_u64_u32 (80 % of an Fp
// multiplication's instructions), one v_lshl_add_u64 and one v_and_b32 per ten -- with FOUR wavefronts per SIMD on independent
// accumulators (round 5; two wavefronts of multiply-adds only until then, which the commitment kernel's timed region out-ran: not
// a ceiling).  Boxes of one pool differ by up to 1.5x on this number (power management).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_valu_calibrate(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    uint64_t c0 = a0, c1 = a1, c2 = a2, c3 = a3, c4 = a0 + 9, c5 = a1 + 9, c6 = a2 + 9, c7 = a3 + 9;
    uint64_t s0 = a2 + 17, s1 = a3 + 19;
    uint32_t m0 = ~a1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
            asm volatile("v_mad_u64_u32 %[c0], vcc, %[a0], %[a1], %[c0]\n v_mad_u64_u32 %[c1], vcc, %[a0], %[a2], %[c1]\n v_mad_u64_u32 %[c2], vcc, %[a1], %[a2], %[c2]\n"
                         "v_mad_u64_u32 %[c3], vcc, %[a0], %[a3], %[c3]\n v_mad_u64_u32 %[c4], vcc, %[a1], %[a3], %[c4]\n v_mad_u64_u32 %[c5], vcc, %[a2], %[a3], %[c5]\n"
                         "v_mad_u64_u32 %[c6], vcc, %[a0], %[a0], %[c6]\n v_mad_u64_u32 %[c7], vcc, %[a1], %[a1], %[c7]\n"
                         "v_lshl_add_u64 %[s0], %[s0], 3, %[s1]\n v_and_b32 %[m0], %[m0], %[a0]\n"
                         : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [c4] "+v"(c4), [c5] "+v"(c5), [c6] "+v"(c6), [c7] "+v"(c7), [s0] "+v"(s0), [s1] "+v"(s1), [m0] "+v"(m0)
                         : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7 ^ s0 ^ s1) ^ m0;
}
// returns wave-instructions per second (0 on failure)
double valu_calibrate_device(hipStream_t s) {
    int cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0.0;
    const int blocks = cus * 4, iters = 10000;         // 4 wavefronts per workgroup -> 4 per SIMD
    uint32_t* d = nullptr;
    if (hipMalloc(&d, (size_t)blocks * 256 * sizeof(uint32_t)) != hipSuccess) return 0.0;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_valu_calibrate, dim3(blocks), dim3(256), 0, s, d, 100, 1u);
    (void)hipEventRecord(e0, s);
    hipLaunchKernelGGL(k_valu_calibrate, dim3(blocks), dim3(256), 0, s, d, iters, 1u);
    (void)hipEventRecord(e1, s);
    float ms = 0.f;
    double rate = 0.0;
    if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f)
        rate = (double)blocks * 4 * iters * 80 / (ms * 1e-3);              // 8 x (8 mads + 2) instructions per iteration and wavefront
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d);
    return rate;
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
template <int C>
static void launch_commit(const CommitPlan& pl, const uint8_t* d_scalars, size_t stride, int input_kind, int n, uint8_t* d_out48, int* d_status, hipStream_t s, KernelTimer* kt) {
    using RC = Recode<C>;
    using TG = TaskGeom<C>;
    if (n <= 0) return;
    uint2* tasks = reinterpret_cast<uint2*>(pl.d_tasks);
    if (kt) kt->begin(KT_DIGITS_SORT, s);
    if (input_kind == 0) hipLaunchKernelGGL((k_digits_sort<C, 0>), dim3(n), dim3(1024), 0, s, d_scalars, stride, pl.d_entries, pl.d_task_off, tasks, pl.d_ntasks, d_status);
    else hipLaunchKernelGGL((k_digits_sort<C, 1>), dim3(n), dim3(1024), 0, s, d_scalars, stride, pl.d_entries, pl.d_task_off, tasks, pl.d_ntasks, d_status);
    if (kt) kt->end(KT_DIGITS_SORT, s);
    if (kt) kt->begin(KT_BUCKET_ACC, s);
    if (n <= 4) hipLaunchKernelGGL((k_bucket_accumulate<C, true>), dim3(4 * TG::MAXT / 256, n), dim3(256), 0, s, pl.d_table, pl.d_entries, tasks, pl.d_ntasks, pl.d_parts);
    else hipLaunchKernelGGL((k_bucket_accumulate<C, false>), dim3(TG::MAXT / 256, n), dim3(256), 0, s, pl.d_table, pl.d_entries, tasks, pl.d_ntasks, pl.d_parts);
    if (kt) kt->end(KT_BUCKET_ACC, s);
    constexpr int NCHUNK = RC::NBUCKET / CHUNK;
    if (kt) kt->begin(KT_REDUCE1, s);
    if (n <= 16) hipLaunchKernelGGL((k_bucket_reduce1<C, true>), dim3((4 * NCHUNK + 63) / 64, n), dim3(64), 0, s, pl.d_parts, pl.d_task_off, d_status, pl.d_partials);
    else hipLaunchKernelGGL((k_bucket_reduce1<C, false>), dim3((NCHUNK + 63) / 64, n), dim3(64), 0, s, pl.d_parts, pl.d_task_off, d_status, pl.d_partials);
    if (kt) kt->end(KT_REDUCE1, s);
    if (kt) kt->begin(KT_REDUCE2, s);
    if (n <= 16) hipLaunchKernelGGL((k_bucket_reduce2_quad<C>), dim3(n), dim3(256), 0, s, pl.d_partials, d_status, d_out48);
    else hipLaunchKernelGGL((k_bucket_reduce2<C>), dim3(n), dim3(64), 0, s, pl.d_partials, d_status, d_out48);
    if (kt) kt->end(KT_REDUCE2, s);
}

void commit_batch_device(const CommitPlan& pl, const uint8_t* d_scalars, size_t stride, int input_kind, int n, uint8_t* d_out48, int* d_status, hipStream_t s, KernelTimer* kt) {
    switch (pl.c) {
        case 11: launch_commit<11>(pl, d_scalars, stride, input_kind, n, d_out48, d_status, s, kt); break;
        case 12: launch_commit<12>(pl, d_scalars, stride, input_kind, n, d_out48, d_status, s, kt); break;
        default: launch_commit<13>(pl, d_scalars, stride, input_kind, n, d_out48, d_status, s, kt); break;
    }
}
size_t direct_table_bytes(int c) {
    const int nwin = (255 + c - 1) / c;
    return (size_t)nwin * 4096 * ((size_t)1 << (c - 1)) * 128;
}
template <int C>
static void build_direct_t(const uint32_t* d_table, uint32_t* d_direct, hipStream_t s) {
    using RC = Recode<C>;
    const int rows = RC::NWIN * 4096;
    hipLaunchKernelGGL((k_build_direct_pow2<C>), dim3((rows + 63) / 64), dim3(64), 0, s, d_table, d_direct);
    for (int k = 1; k <= C - 2; k++) {
        const size_t lanes = (size_t)rows * ((((size_t)1 << k) - 1 + 7) / 8);
        hipLaunchKernelGGL((k_build_direct_level<C>), dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, s, d_direct, k);
    }
}
void build_direct_device(int c, const uint32_t* d_table, uint32_t* d_direct, hipStream_t s) {
    switch (c) {
        case 12: build_direct_t<12>(d_table, d_direct, s); break;
        case 13: build_direct_t<13>(d_table, d_direct, s); break;
        case 14: build_direct_t<14>(d_table, d_direct, s); break;
        default: build_direct_t<15>(d_table, d_direct, s); break;
    }
}
void build_table_any_device(int c, const uint32_t* d_src_aff, uint32_t* d_table, int n, int bitrev_bits, hipStream_t s) {
    switch (c) {
        case 14: hipLaunchKernelGGL((k_build_table<14>), dim3((n + 63) / 64), dim3(64), 0, s, d_src_aff, d_table, n, bitrev_bits); break;
        case 15: hipLaunchKernelGGL((k_build_table<15>), dim3((n + 63) / 64), dim3(64), 0, s, d_src_aff, d_table, n, bitrev_bits); break;
        default: build_table_device(c, d_src_aff, d_table, n, bitrev_bits, s); break;
    }
}
size_t direct_parts_bytes(int max_batch) { return (size_t)max_batch * DPARTS * 64 * sizeof(uint32_t); }
size_t direct_bad_bytes(int max_batch) { return (size_t)max_batch * DBAD * sizeof(int); }
DirectSlot direct_slot_offset(const DirectSlot& ws, int off) {
    DirectSlot o = ws;
    o.d_parts = ws.d_parts + (size_t)off * (4096 / ws.k) * ws.wsplit * 64;
    o.d_bad = ws.d_bad + (size_t)off * DBAD;
#ifdef MKZG_AFFINE_PAIRS
    if (ws.d_prefix) o.d_prefix = ws.d_prefix + (size_t)off * (direct_prefix_bytes(1) / sizeof(uint4));
#endif
    return o;
}
// The shape of a call of n blobs through a workspace sized for max_batch.
//   up to 128 blobs : the fewest scalars per lane (most lanes per blob, shortest chain) whose wavefronts still find a free slot each
//                     (two per SIMD), the partial sums folded by butterflies on quads of lanes -- somebody waits for this chain;
//   129 .. 255      : ONE scalar per lane, the 64 blocks of partial sums of a blob added up serially (k_commit_direct_sum8, then the
//                     finish kernel).  Eight scalars per lane is the shape of exactly 256 blobs (2048 wavefronts = two per SIMD): with
//                     fewer blobs some SIMDs hold two of those 136-addition wavefronts and some one, and the launch lasts as long as
//                     the full one (2.78 ms from 129 to 256 blobs).  With 17-addition wavefronts, eight times as many, the dispatcher
//                     keeps every SIMD busy until the end: 1.77 ms at 129 blobs, 2.20 at 192, 2.65 at 240 (tools/sweep_sizes.py; two
//                     or four scalars per lane were measured too: steps at 160 / 176 / 208 blobs where a round of wavefronts fills);
//   from 256 on     : eight scalars per lane.
void direct_pick_shape(int n, int max_batch, DirectSlot* slot) {
    slot->k = 8;
    slot->serial = 1;
    slot->wsplit = 1;
    for (int k = 1; k < 8; k *= 2)      // within the 512 records per blob of max_batch that every lane's buffer starts with
        if ((size_t)n * (4096 / k) <= (size_t)max_batch * 512 && n * (4096 / k) / 64 <= 2048) {
            slot->k = k;
            slot->serial = 0;
            // up to eight blobs: a pair of lanes per scalar (128 wavefronts per blob, one per SIMD at eight blobs): 9 dependent additions instead of 17
            if (k == 1 && n <= 8 && (size_t)n * 8192 <= (size_t)max_batch * 512) slot->wsplit = 2;
            return;
        }
    if (n > 128 && n < 256 && (size_t)n * 4096 <= (size_t)max_batch * DPARTS) slot->k = 1;
    // pieces above 256 blobs (max_batch raised by the caller): eight scalars per lane cost whole rounds of 2048 wavefronts (2.75 ms each),
    // one scalar per lane costs 10.8 us per blob + 0.3 ms -- e.g. 300 blobs: two rounds = 5.5 ms against 3.5 ms
    if (n > 256 && (size_t)n * 4096 <= (size_t)max_batch * DPARTS) {
        const int rounds = (8 * n + 2047) / 2048;
        if (rounds * 2.75 > 0.0108 * n + 0.3) slot->k = 1;
    }
}
CommitPlan commit_plan_offset(const CommitPlan& pl, int off) {
    CommitPlanSizes sz;
    commit_plan_sizes(pl.c, 1, &sz);                 // per-item sizes
    CommitPlan o = pl;
    o.d_entries = pl.d_entries + (size_t)off * (sz.entries_bytes / 4);
    o.d_task_off = pl.d_task_off + (size_t)off * (sz.task_off_bytes / 4);
    o.d_tasks = pl.d_tasks + (size_t)off * (sz.tasks_bytes / 4);
    o.d_ntasks = pl.d_ntasks + (size_t)off * (sz.ntasks_bytes / 4);
    o.d_parts = pl.d_parts + (size_t)off * (sz.parts_bytes / 4);
    o.d_partials = pl.d_partials + (size_t)off * (sz.partials_bytes / 4);
    return o;
}
template <int KIND>
static void launch_direct_acc(int c, dim3 grid, hipStream_t s, const uint8_t* d_scalars, size_t stride, const uint32_t* d_direct, const DirectSlot& ws) {
#ifdef MKZG_AFFINE_PAIRS
    if (c == 15 && ws.k == 8 && ws.d_prefix) {       // rejected experiment (experiments/commit_pairs_kernels.hip), A/B builds only
        commit_direct_acc_pairs_launch(KIND, grid, s, d_scalars, stride, d_direct, ws);
        return;
    }
#endif
    switch (c) {
        case 12: hipLaunchKernelGGL((k_commit_direct_acc<12, KIND>), grid, dim3(DTA), 0, s, d_scalars, stride, d_direct, ws.d_parts, ws.d_bad, ws.k, ws.wsplit); break;
        case 13: hipLaunchKernelGGL((k_commit_direct_acc<13, KIND>), grid, dim3(DTA), 0, s, d_scalars, stride, d_direct, ws.d_parts, ws.d_bad, ws.k, ws.wsplit); break;
        case 14: hipLaunchKernelGGL((k_commit_direct_acc<14, KIND>), grid, dim3(DTA), 0, s, d_scalars, stride, d_direct, ws.d_parts, ws.d_bad, ws.k, ws.wsplit); break;
        default: hipLaunchKernelGGL((k_commit_direct_acc<15, KIND>), grid, dim3(DTA), 0, s, d_scalars, stride, d_direct, ws.d_parts, ws.d_bad, ws.k, ws.wsplit); break;
    }
}
// the two halves of a direct-table commitment: accumulate (VALU-bound, one single-wave workgroup per 512 scalars) and finish
// (one wavefront per blob: sum of the lane accumulators, inversion, compression).  Callers that feed the accumulate kernel in chunks
// on several streams (host-pointer pipeline) run ONE finish over all items afterwards.
void commit_direct_acc_device(int c, const uint32_t* d_direct, const DirectSlot& ws, const uint8_t* d_scalars, size_t stride, int input_kind, int n, hipStream_t s, KernelTimer* kt) {
    if (n <= 0) return;
    if (kt) kt->begin(KT_COMMIT_DIRECT, s);
    const dim3 grid(4096 / (ws.k * DTA) * ws.wsplit, n);
    if (input_kind == 0) launch_direct_acc<0>(c, grid, s, d_scalars, stride, d_direct, ws);
    else launch_direct_acc<1>(c, grid, s, d_scalars, stride, d_direct, ws);
    if (!ws.serial && ws.wsplit == 2) {                   // 8192 records per blob: pairs + 16 -> 256 at stride 32 -> 16 at stride 512
        hipLaunchKernelGGL(k_commit_direct_fold16, dim3(256, n), dim3(64), 0, s, ws.d_parts, 8192, 1, 1);
        hipLaunchKernelGGL(k_commit_direct_fold16, dim3(16, n), dim3(64), 0, s, ws.d_parts, 256, 32, 0);
    } else if (!ws.serial) {
        const int pp = 4096 / ws.k;                       // records per blob: 4096, 2048, 1024 -> / 16 -> / 256
        hipLaunchKernelGGL(k_commit_direct_fold16, dim3(pp / 16, n), dim3(64), 0, s, ws.d_parts, pp, 1, 0);
        hipLaunchKernelGGL(k_commit_direct_fold16, dim3(pp / 256, n), dim3(64), 0, s, ws.d_parts, pp / 16, 16, 0);
    }
    if (kt) kt->end(KT_COMMIT_DIRECT, s);
}
void commit_direct_fin_device(const DirectSlot& ws, int n, uint8_t* d_out48, int* d_status, hipStream_t s, KernelTimer* kt, uint32_t* d_raw) {
    if (n <= 0) return;
    if (kt) kt->begin(KT_COMMIT_DIRECT_FIN, s);
    const int blocks = 4096 / (ws.k * DTA) * ws.wsplit;   // accumulate workgroups per blob: bad-scalar flags, 64-record blocks
    int nblk = blocks / ws.wsplit;                        // (folded shapes: four times the records the folds left)
    if (ws.serial && blocks > 8) {
        hipLaunchKernelGGL(k_commit_direct_sum8, dim3(blocks / 8, n), dim3(64), 0, s, ws.d_parts, blocks);
        nblk = blocks / 8;
    }
    if (ws.serial && ws.fin_quad && !d_raw) hipLaunchKernelGGL(k_commit_direct_fin_quad, dim3(n), dim3(256), 0, s, ws.d_parts, ws.d_bad, d_out48, d_status, blocks, nblk);
    else hipLaunchKernelGGL(k_commit_direct_fin, dim3(n), dim3(64), 0, s, ws.d_parts, ws.d_bad, d_out48, d_status, blocks, nblk, ws.serial ? 0 : 256 * ws.wsplit, d_raw);
    if (kt) kt->end(KT_COMMIT_DIRECT_FIN, s);
}
void commit_direct_device(int c, const uint32_t* d_direct, const DirectSlot& ws, const uint8_t* d_scalars, size_t stride, int input_kind, int n, uint8_t* d_out48, int* d_status,
                          hipStream_t s, KernelTimer* kt, uint32_t* d_raw) {
    commit_direct_acc_device(c, d_direct, ws, d_scalars, stride, input_kind, n, s, kt);
    commit_direct_fin_device(ws, n, d_out48, d_status, s, kt, d_raw);
}
void commit_plan_sizes(int c, int max_batch, CommitPlanSizes* sz) {
    const int nwin = (255 + c - 1) / c, nb = 1 << (c - 1);
    const int nent = 4096 * nwin;
    const int maxt = ((nb + nent / PIECE + 255) / 256) * 256;
    sz->nwin = nwin;
    sz->nbucket = nb;
    sz->max_tasks = maxt;
    sz->table_bytes = (size_t)nwin * 4096 * 128;
    sz->entries_bytes = (size_t)max_batch * nent * 4;
    sz->task_off_bytes = (size_t)max_batch * (nb + 1) * 4;
    sz->tasks_bytes = (size_t)max_batch * maxt * 8;
    sz->ntasks_bytes = (size_t)max_batch * 4;
    sz->parts_bytes = (size_t)max_batch * maxt * 256;
    sz->partials_bytes = (size_t)max_batch * (nb / CHUNK) * 512;
}
void g1_decompress2_device(const uint8_t* d_in48_a, int n_a, uint32_t* d_out_aff_a, int* d_st_a, const uint8_t* d_in48_b, int n_b, uint32_t* d_out_aff_b, int* d_st_b,
                           hipStream_t s) {
    const int n = n_a + n_b;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_g1_decompress, dim3((n + 63) / 64), dim3(64), 0, s, d_in48_a, n_a, d_out_aff_a, d_st_a, d_in48_b, n_b, d_out_aff_b, d_st_b);
}
void g1_decompress_device(const uint8_t* d_in48, int n, uint32_t* d_out_aff, int* d_status, hipStream_t s) {
    g1_decompress2_device(d_in48, n, d_out_aff, d_status, nullptr, 0, nullptr, nullptr, s);
}
void build_table_device(int c, const uint32_t* d_src_aff, uint32_t* d_table, int n, int bitrev_bits, hipStream_t s) {
    switch (c) {
        case 11: hipLaunchKernelGGL((k_build_table<11>), dim3((n + 63) / 64), dim3(64), 0, s, d_src_aff, d_table, n, bitrev_bits); break;
        case 12: hipLaunchKernelGGL((k_build_table<12>), dim3((n + 63) / 64), dim3(64), 0, s, d_src_aff, d_table, n, bitrev_bits); break;
        default: hipLaunchKernelGGL((k_build_table<13>), dim3((n + 63) / 64), dim3(64), 0, s, d_src_aff, d_table, n, bitrev_bits); break;
    }
}

}  // namespace kzg
