// recover_kernels.hip -- block-erasure Reed-Solomon recovery for EIP-7594 on MI355X (gfx950).
//
// Reference: api_eip7594.go:93-142 (cell placement, bit reversal) + internal/erasure_code/
// erasure_code.go:75-164:   Z = vanishing polynomial of the missing cells; D*Z = IFFT(E . FFT(Z));
// P = IFFT_coset( FFT_coset(D*Z) / FFT_coset(Z) ), truncated to 4096 coefficients.
//
// MI355X-first restructuring (same results, fewer passes):
//   * Z(x) = S(x^64) with deg S <= 64, so FFT_8192(Z) and FFT_coset(Z) are periodic with period
//     128: S is evaluated at 128 points and only 128 field inversions are needed, instead of two
//     8192-point NTTs and an 8192-way batch inversion (k_recover_vanishing);
//   * the three remaining NTT-8192s run as one radix-2 pass fused into the load plus two
//     LDS-resident NTT-4096s (one workgroup per half, 147 KB of LDS each).  Decimation in
//     frequency (natural -> bit-reversed) and decimation in time (bit-reversed -> natural)
//     alternate, so no permutation pass exists; coset shifts (7^i), 1/8192 and the bit reversal
//     of the cell placement are folded into loads and stores.
// Integer modular arithmetic only -- no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "mont29.hpp"
#include "recover_kernels.h"

namespace kzg {

constexpr size_t RECOVER_SCRATCH_WORDS = (size_t)(256 + 2 * 8192) * 12;   // per blob: zeval | inv_zc | buf0 | buf1 (Fr rows of 12 words)

constexpr int RNT = 1024, RN = 4096;

__device__ __forceinline__ void r_load_fr12(fr_t& r, const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1], c = q[2];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = c.x;
}
__device__ __forceinline__ void r_store_fr12(uint32_t* p, const fr_t& r) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(r.l[0], r.l[1], r.l[2], r.l[3]);
    q[1] = make_uint4(r.l[4], r.l[5], r.l[6], r.l[7]);
    q[2] = make_uint4(r.l[8], 0, 0, 0);
}
__device__ __forceinline__ void r_mul(fr_t& r, const fr_t& a, const fr_t& b) { FrC::mul(r, a, b); }
__device__ __forceinline__ void r_get(fr_t& r, const uint32_t* sh, int e) {
#pragma unroll
    for (int q = 0; q < 9; q++) r.l[q] = sh[q * RN + e];
}
__device__ __forceinline__ void r_put(uint32_t* sh, int e, const fr_t& r) {
#pragma unroll
    for (int q = 0; q < 9; q++) sh[q * RN + e] = r.l[q];
}
// w_8192^(+-e) for any e (tables hold e < 4096)
__device__ __forceinline__ void r_tw_any(fr_t& r, const uint32_t* __restrict__ tw, uint32_t e) {
    e &= 8191u;
    if (e < 4096u) {
        r_load_fr12(r, tw + (size_t)e * 12);
    } else {
        fr_t t;
        r_load_fr12(t, tw + (size_t)(e - 4096u) * 12);
        FrC::neg(r, t);
    }
}
// LDS-resident NTT-4096 over w_4096 = tw[2 j]; tw = forward or inverse table of w_8192^(+-j)
__device__ __forceinline__ void ntt4096_dif(uint32_t* sh, const uint32_t* __restrict__ tw, int t) {   // natural -> bit-reversed
    for (int h = RN / 2; h >= 1; h >>= 1) {
#pragma unroll
        for (int r = 0; r < RN / 2 / RNT; r++) {
            const int p = t + RNT * r;
            const int k = p & (h - 1);
            const int i0 = ((p - k) << 1) + k, i1 = i0 + h;
            fr_t a, b, w, s, d;
            r_get(a, sh, i0);
            r_get(b, sh, i1);
            FrC::add(s, a, b);
            FrC::sub(d, a, b);
            r_load_fr12(w, tw + (size_t)(k * (RN / h)) * 12);
            r_mul(d, d, w);
            r_put(sh, i0, s);
            r_put(sh, i1, d);
        }
        __syncthreads();
    }
}
__device__ __forceinline__ void ntt4096_dit(uint32_t* sh, const uint32_t* __restrict__ tw, int t) {   // bit-reversed -> natural
    for (int h = 1; h < RN; h <<= 1) {
#pragma unroll
        for (int r = 0; r < RN / 2 / RNT; r++) {
            const int p = t + RNT * r;
            const int k = p & (h - 1);
            const int i0 = ((p - k) << 1) + k, i1 = i0 + h;
            fr_t a, b, w, tt;
            r_get(a, sh, i0);
            r_get(b, sh, i1);
            r_load_fr12(w, tw + (size_t)(k * (RN / h)) * 12);
            r_mul(tt, b, w);
            FrC::add(b, a, tt);
            FrC::sub(a, a, tt);
            r_put(sh, i0, b);
            r_put(sh, i1, a);
        }
        __syncthreads();
    }
}
__device__ __forceinline__ uint32_t brev13(uint32_t x) { return __brev(x) >> 19; }
__device__ __forceinline__ uint32_t brev12(uint32_t x) { return __brev(x) >> 20; }

// pow7[i] = 7^i (i < 8192), ipow7[i] = 7^-i (i < 4096), Montgomery rows   (erasure_code.go:57-60: coset generator 7)
__global__ void __launch_bounds__(256) k_recover_tables(uint32_t* __restrict__ pow7, uint32_t* __restrict__ ipow7) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 8192) return;
    fr_t one, seven, acc, base;
    Fr::set_one(one);
    seven = one;
    for (int q = 0; q < 6; q++) FrC::add(seven, seven, one);
    acc = one;
    base = seven;
    for (uint32_t e = (uint32_t)i; e; e >>= 1) {
        if (e & 1) r_mul(acc, acc, base);
        r_mul(base, base, base);
    }
    Fr::reduce_once(acc, acc);
    r_store_fr12(pow7 + (size_t)i * 12, acc);
    if (i < 4096) {
        fr_t inv;
        Fr::pow_words(inv, acc, FrParams::EXP_INV, FrParams::BITS);
        Fr::reduce_once(inv, inv);
        r_store_fr12(ipow7 + (size_t)i * 12, inv);
    }
}

// vanishing polynomial of the missing blocks and its two 128-periodic evaluation tables
//   zeval[j]  = S(w_128^j)              (= FFT_8192(Z)[j'] for j' = j mod 128)
//   inv_zc[j] = 1 / S(7^64 w_128^j)     (= 1 / FFT_coset(Z)[j'])
// All stage kernels take the blob in blockIdx.y: per blob 192 ints of positions (cell_pos[128] | missing[64]), one count of missing
// cells, RECOVER_SCRATCH_WORDS of scratch, 4096 coefficient rows and one status word.
// VAN_COPIES x 128 threads per blob: the 128 evaluation points once per copy, copies beyond the first shadow it and store nothing.
// Why: 64 blobs x 128 lanes = 8192 active lanes is exactly where gfx950 drops into its sparse-wavefront mode (fk20_kernels.hip
// FULL_WAVES), and that mode outlasts the kernel: the next chip-filling launch (k_fk20_msm, 0.5 ms later) then ran at 3.6 ms instead
// of 2.0 (measured by skipping kernels one at a time, tools/bench_recover.py).
constexpr int VAN_COPIES = 4;
__global__ void __launch_bounds__(128 * VAN_COPIES) k_recover_vanishing(const int* __restrict__ pos_all, const int* __restrict__ n_missing_all, const uint32_t* __restrict__ tw_fwd,
                                                           const uint32_t* __restrict__ pow7, uint32_t* __restrict__ scratch_all) {
    __shared__ uint32_t sz[2][65 * 9];
    const int t = threadIdx.x & 127, blob = blockIdx.y;
    const bool live = threadIdx.x < 128;
    const uint32_t* missing = reinterpret_cast<const uint32_t*>(pos_all + (size_t)blob * 192 + 128);
    const int n_missing = n_missing_all[blob];
    uint32_t* zeval = scratch_all + (size_t)blob * RECOVER_SCRATCH_WORDS;
    uint32_t* inv_zc = zeval + 128 * 12;
    auto get = [&](int buf, int j, fr_t& v) {
#pragma unroll
        for (int q = 0; q < 9; q++) v.l[q] = sz[buf][j * 9 + q];
    };
    auto put = [&](int buf, int j, const fr_t& v) {
#pragma unroll
        for (int q = 0; q < 9; q++) sz[buf][j * 9 + q] = v.l[q];
    };
    fr_t one;
    Fr::set_one(one);
    if (t < 65) {
        fr_t z;
        Fr::set_zero(z);
        put(0, t, t == 0 ? one : z);
    }
    __syncthreads();
    int cur = 0;
    for (int m = 0; m < n_missing; m++) {            // S <- S * (X - x_m), x_m = w_128^(missing[m])   (erasure_code.go:151-164)
        fr_t x;
        r_tw_any(x, tw_fwd, 64u * (missing[m] & 127u));
        if (t <= m + 1 && t < 65) {
            fr_t lo, hi, v;
            Fr::set_zero(hi);
            Fr::set_zero(lo);
            if (t >= 1) get(cur, t - 1, hi);
            if (t <= m) get(cur, t, lo);
            r_mul(lo, lo, x);
            FrC::sub(v, hi, lo);
            if (live) put(cur ^ 1, t, v);
        }
        __syncthreads();
        cur ^= 1;
    }
    // Horner at x = w_128^t and at 7^64 w_128^t
    fr_t x0, c64, x1, a0, a1;
    r_tw_any(x0, tw_fwd, 64u * (uint32_t)t);
    r_load_fr12(c64, pow7 + (size_t)64 * 12);
    r_mul(x1, x0, c64);
    Fr::set_zero(a0);
    Fr::set_zero(a1);
    for (int j = n_missing; j >= 0; j--) {
        fr_t cj;
        get(cur, j, cj);
        r_mul(a0, a0, x0);
        FrC::add(a0, a0, cj);
        r_mul(a1, a1, x1);
        FrC::add(a1, a1, cj);
    }
    if (live) r_store_fr12(zeval + (size_t)t * 12, a0);
    fr_t inv;
    Fr::inv(inv, a1);                       // divsteps (modinv30.hpp): ~10x cheaper than the Fermat exponentiation
    Fr::reduce_once(inv, inv);
    if (live) r_store_fr12(inv_zc + (size_t)t * 12, inv);
}

// stage A: dz_br = DIF-IFFT_8192( E . Zeval ) without the 1/8192, E[j] = ext[brev13(j)] from the cells
//   cell_pos[c] = index of cell c among the supplied cells, or -1 (missing)
//   (positions are indices into the whole staged cell array of the call, so ragged batches need no per-blob base)
__global__ void __launch_bounds__(RNT) k_recover_stage_a(const uint8_t* __restrict__ cells, const int* __restrict__ pos_all, uint32_t* __restrict__ scratch_all,
                                                         const uint32_t* __restrict__ tw_inv, int* __restrict__ status_all) {
    extern __shared__ uint32_t sh[];
    const int half = blockIdx.x, t = threadIdx.x, blob = blockIdx.y;
    const int* cell_pos = pos_all + (size_t)blob * 192;
    const uint32_t* zeval = scratch_all + (size_t)blob * RECOVER_SCRATCH_WORDS;
    uint32_t* dz_br = scratch_all + (size_t)blob * RECOVER_SCRATCH_WORDS + 256 * 12;
    int* status = status_all + blob;
    bool bad = false;
    auto load_e = [&](uint32_t j, fr_t& out) {       // E[j] * Zeval[j mod 128]
        const uint32_t src = brev13(j);              // index into the cell-ordered extended blob
        const int pos = cell_pos[src >> 6];
        if (pos < 0) { Fr::set_zero(out); return; }
        const uint8_t* p = cells + (size_t)pos * 2048 + (size_t)(src & 63) * 32;
        const uint4* q = reinterpret_cast<const uint4*>(p);
        uint4 hi = q[0], lo = q[1];
        uint32_t w[8];
        w[7] = __builtin_bswap32(hi.x); w[6] = __builtin_bswap32(hi.y); w[5] = __builtin_bswap32(hi.z); w[4] = __builtin_bswap32(hi.w);
        w[3] = __builtin_bswap32(lo.x); w[2] = __builtin_bswap32(lo.y); w[1] = __builtin_bswap32(lo.z); w[0] = __builtin_bswap32(lo.w);
        bad |= Fr::words32_geq_mod(w);
        fr_t pl, m, z;
        Fr::from_words32(pl, w);
        Fr::from_plain(m, pl);
        Fr::reduce_once(m, m);
        r_load_fr12(z, zeval + (size_t)(j & 127u) * 12);
        r_mul(out, m, z);
    };
#pragma unroll
    for (int k = 0; k < RN / RNT; k++) {
        const uint32_t i = t + RNT * k;
        fr_t a, b, x;
        load_e(i, a);
        load_e(i + 4096, b);
        if (half == 0) {
            FrC::add(x, a, b);
        } else {
            fr_t w;
            FrC::sub(x, a, b);
            r_load_fr12(w, tw_inv + (size_t)i * 12);
            r_mul(x, x, w);
        }
        r_put(sh, (int)i, x);
    }
    if (__syncthreads_or(bad)) {
        if (t == 0) status[0] = MKZG_ERR_NON_CANONICAL_SCALAR;
        return;
    }
    ntt4096_dif(sh, tw_inv, t);
#pragma unroll
    for (int k = 0; k < RN / RNT; k++) {
        const int i = t + RNT * k;
        fr_t v;
        r_get(v, sh, i);
        r_store_fr12(dz_br + ((size_t)half * RN + i) * 12, v);
    }
}

// stage B1: halves of the DIT forward NTT_8192 of (dz[i] 7^i / 8192): Y_h natural
__global__ void __launch_bounds__(RNT) k_recover_stage_b1(uint32_t* __restrict__ scratch_all, const uint32_t* __restrict__ pow7, const uint32_t* __restrict__ tw_fwd,
                                                          const int* __restrict__ status_all) {
    extern __shared__ uint32_t sh[];
    const int half = blockIdx.x, t = threadIdx.x, blob = blockIdx.y;
    const uint32_t* dz_br = scratch_all + (size_t)blob * RECOVER_SCRATCH_WORDS + 256 * 12;
    uint32_t* y_halves = scratch_all + (size_t)blob * RECOVER_SCRATCH_WORDS + (256 + 8192) * 12;
    const int* status = status_all + blob;
    if (status[0] != 0) return;
    fr_t inv8192;                       // 1/8192 in Montgomery form is R / 2^13 = 2^248: a single set limb (2^16 at weight 2^232)
    Fr::set_zero(inv8192);
    inv8192.l[8] = 0x10000u;
#pragma unroll
    for (int k = 0; k < RN / RNT; k++) {
        const int pl = t + RNT * k;
        const uint32_t p = (uint32_t)half * RN + pl;
        fr_t v, s;
        r_load_fr12(v, dz_br + (size_t)p * 12);
        r_load_fr12(s, pow7 + (size_t)brev13(p) * 12);
        r_mul(v, v, s);
        r_mul(v, v, inv8192);
        r_put(sh, pl, v);
    }
    __syncthreads();
    ntt4096_dit(sh, tw_fwd, t);
#pragma unroll
    for (int k = 0; k < RN / RNT; k++) {
        const int i = t + RNT * k;
        fr_t v;
        r_get(v, sh, i);
        r_store_fr12(y_halves + ((size_t)half * RN + i) * 12, v);
    }
}

// stage C: last DIT stage + division by the coset evaluations of Z:  q natural
__global__ void __launch_bounds__(256) k_recover_stage_c(uint32_t* __restrict__ scratch_all, const uint32_t* __restrict__ tw_fwd, const int* __restrict__ status_all) {
    const int m = blockIdx.x * 256 + threadIdx.x, blob = blockIdx.y;
    const uint32_t* inv_zc = scratch_all + (size_t)blob * RECOVER_SCRATCH_WORDS + 128 * 12;
    const uint32_t* y_halves = scratch_all + (size_t)blob * RECOVER_SCRATCH_WORDS + (256 + 8192) * 12;
    uint32_t* q = scratch_all + (size_t)blob * RECOVER_SCRATCH_WORDS + 256 * 12;
    const int* status = status_all + blob;
    if (m >= RN || status[0] != 0) return;
    fr_t y0, y1, w, tt, a, b, iz;
    r_load_fr12(y0, y_halves + (size_t)m * 12);
    r_load_fr12(y1, y_halves + ((size_t)RN + m) * 12);
    r_load_fr12(w, tw_fwd + (size_t)m * 12);
    r_mul(tt, y1, w);
    FrC::add(a, y0, tt);
    FrC::sub(b, y0, tt);
    r_load_fr12(iz, inv_zc + (size_t)(m & 127) * 12);
    r_mul(a, a, iz);
    r_mul(b, b, iz);
    r_store_fr12(q + (size_t)m * 12, a);
    r_store_fr12(q + ((size_t)RN + m) * 12, b);
}

// stage D: coefficients[i] = IFFT_8192(q)[i] * 7^-i / 8192 for i < 4096 (DIF, natural -> bit-reversed)
__global__ void __launch_bounds__(RNT) k_recover_stage_d(const uint32_t* __restrict__ scratch_all, const uint32_t* __restrict__ tw_inv, const uint32_t* __restrict__ ipow7,
                                                         const int* __restrict__ status_all, uint32_t* __restrict__ coeffs_all) {
    extern __shared__ uint32_t sh[];
    const int half = blockIdx.x, t = threadIdx.x, blob = blockIdx.y;
    const uint32_t* q = scratch_all + (size_t)blob * RECOVER_SCRATCH_WORDS + 256 * 12;
    uint32_t* coeffs = coeffs_all + (size_t)blob * 4096 * 12;
    const int* status = status_all + blob;
    if (status[0] != 0) return;
#pragma unroll
    for (int k = 0; k < RN / RNT; k++) {
        const int i = t + RNT * k;
        fr_t a, b, x;
        r_load_fr12(a, q + (size_t)i * 12);
        r_load_fr12(b, q + ((size_t)RN + i) * 12);
        if (half == 0) {
            FrC::add(x, a, b);
        } else {
            fr_t w;
            FrC::sub(x, a, b);
            r_load_fr12(w, tw_inv + (size_t)i * 12);
            r_mul(x, x, w);
        }
        r_put(sh, i, x);
    }
    __syncthreads();
    ntt4096_dif(sh, tw_inv, t);
    fr_t inv8192;                       // 1/8192 in Montgomery form is R / 2^13 = 2^248: a single set limb (2^16 at weight 2^232)
    Fr::set_zero(inv8192);
    inv8192.l[8] = 0x10000u;
#pragma unroll
    for (int k = 0; k < RN / RNT; k++) {
        const int pl = t + RNT * k;
        const uint32_t i = 2u * brev12((uint32_t)pl) + (uint32_t)half;     // natural index of slot (half, pl)
        if (i < 4096u) {
            fr_t v, s;
            r_get(v, sh, pl);
            r_load_fr12(s, ipow7 + (size_t)i * 12);
            r_mul(v, v, s);
            r_mul(v, v, inv8192);
            r_store_fr12(coeffs + (size_t)i * 12, v);
        }
    }
}

// ------------------------------------------------------------------------------------------
static bool g_rattr = false;
static void set_attrs() {
    if (g_rattr) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_recover_stage_a), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 4096 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_recover_stage_b1), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 4096 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_recover_stage_d), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 4096 * 4);
    g_rattr = true;
}
void recover_tables_device(uint32_t* d_pow7, uint32_t* d_ipow7, hipStream_t s) { hipLaunchKernelGGL(k_recover_tables, dim3(32), dim3(256), 0, s, d_pow7, d_ipow7); }
void recover_coeffs_device(const uint8_t* d_cells, const int* d_pos, const int* d_n_missing, int n_blobs, const uint32_t* d_tw_fwd, const uint32_t* d_tw_inv,
                           const uint32_t* d_pow7, const uint32_t* d_ipow7, uint32_t* d_scratch, uint32_t* d_coeffs, int* d_status, hipStream_t s) {
    if (n_blobs <= 0) return;
    set_attrs();
    // scratch per blob: zeval 128 rows | inv_zc 128 rows | buf0 8192 rows | buf1 8192 rows
    const size_t lds = 9 * 4096 * 4;
    (void)hipMemsetAsync(d_status, 0, (size_t)n_blobs * sizeof(int), s);
    hipLaunchKernelGGL(k_recover_vanishing, dim3(1, n_blobs), dim3(128 * VAN_COPIES), 0, s, d_pos, d_n_missing, d_tw_fwd, d_pow7, d_scratch);
    hipLaunchKernelGGL(k_recover_stage_a, dim3(2, n_blobs), dim3(RNT), lds, s, d_cells, d_pos, d_scratch, d_tw_inv, d_status);
    hipLaunchKernelGGL(k_recover_stage_b1, dim3(2, n_blobs), dim3(RNT), lds, s, d_scratch, d_pow7, d_tw_fwd, d_status);
    hipLaunchKernelGGL(k_recover_stage_c, dim3(16, n_blobs), dim3(256), 0, s, d_scratch, d_tw_fwd, d_status);
    hipLaunchKernelGGL(k_recover_stage_d, dim3(2, n_blobs), dim3(RNT), lds, s, d_scratch, d_tw_inv, d_ipow7, d_status, d_coeffs);
}
size_t recover_scratch_words() { return RECOVER_SCRATCH_WORDS; }

}  // namespace kzg
