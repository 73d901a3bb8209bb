// cells_kernels.hip -- EIP-7594 (PeerDAS) device kernels for MI355X (gfx950): the radix-2 Fr NTT
// and everything built on it.
//
//   k_cells_ntt     ComputeCells (reference api_eip7594.go:12-26,54-58 -> fk20.go:58-74 ->
//                   internal/domain/fft.go:95-144): one workgroup per blob keeps the whole
//                   4096-point polynomial in LDS (147 KB of the CU's 160 KB): inverse NTT
//                   (decimation in time on the blob's bit-reversed order, so no permutation
//                   pass), then the 8192-point extension.  Because the upper half of the padded
//                   input is zero the 8192-NTT splits into the blob itself (cells 0..63 are the
//                   input evaluations) and ONE coset NTT-4096 (shift w_8192) whose decimation-
//                   in-frequency output order is exactly the bit-reversed order the cells use.
//   k_fk20_scalars  the 64 circulant-row NTT-128s of FK20 (reference fk20.go:100-124,
//                   toeplitz.go:17-29,95-107), transposed, scaled by 1/2 and exported as plain
//                   scalars for the fixed-base MSMs.
// Twiddles come from precomputed tables (the reference recomputes them by running products,
// internal/domain/fft.go:118-137).  Integer modular arithmetic only -- no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mont29.hpp"
#include "cells_kernels.h"
#include "pub_ticket.hpp"

namespace kzg {

__device__ __forceinline__ void c_load_fr12(fr_t& r, const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1], c = q[2];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = c.x;
}
__device__ __forceinline__ void c_store_fr12(uint32_t* p, const fr_t& r) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(r.l[0], r.l[1], r.l[2], r.l[3]);
    q[1] = make_uint4(r.l[4], r.l[5], r.l[6], r.l[7]);
    q[2] = make_uint4(r.l[8], 0, 0, 0);
}
__device__ __forceinline__ void c_fr_mul(fr_t& r, const fr_t& a, const fr_t& b) { FrC::mul(r, a, b); }

// LDS element access, structure-of-arrays: word q of element e at sh[q * N + e]
template <int N>
__device__ __forceinline__ void lds_get(fr_t& r, const uint32_t* sh, int e) {
#pragma unroll
    for (int q = 0; q < 9; q++) r.l[q] = sh[q * N + e];
}
template <int N>
__device__ __forceinline__ void lds_put(uint32_t* sh, int e, const fr_t& r) {
#pragma unroll
    for (int q = 0; q < 9; q++) sh[q * N + e] = r.l[q];
}

// ------------------------------------------------------------------------------------------
// twiddle tables: tw_fwd[j] = w_8192^j, tw_inv[j] = w_8192^-j for j < 4096 (Montgomery rows),
// plus the plain-integer forms of w_128^(+-j), j < 64, used as scalars by the G1 FFT.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_build_twiddles(uint32_t* __restrict__ tw_fwd, uint32_t* __restrict__ tw_inv, uint32_t* __restrict__ g1tw_fwd,
                                                        uint32_t* __restrict__ g1tw_inv) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= 4096) return;
    fr_t g;
    Fr::set(g, FR_ROOT_2_32);
    for (int s = 0; s < 32 - 13; s++) c_fr_mul(g, g, g);      // generator of the 8192-domain
    auto pw = [&](uint32_t e, fr_t& out) {
        fr_t acc, base = g;
        Fr::set_one(acc);
        for (; e; e >>= 1) {
            if (e & 1) c_fr_mul(acc, acc, base);
            c_fr_mul(base, base, base);
        }
        Fr::reduce_once(out, acc);
    };
    fr_t f, inv;
    pw((uint32_t)j, f);
    pw((uint32_t)((8192 - j) & 8191), inv);
    c_store_fr12(tw_fwd + (size_t)j * 12, f);
    c_store_fr12(tw_inv + (size_t)j * 12, inv);
    if ((j & 63) == 0) {                                       // w_128^(j/64)
        const int m = j >> 6;
        fr_t pl;
        uint32_t w[8];
        Fr::to_plain(pl, f);
        Fr::to_words32(w, pl);
        for (int q = 0; q < 8; q++) g1tw_fwd[m * 8 + q] = w[q];
        Fr::to_plain(pl, inv);
        Fr::to_words32(w, pl);
        for (int q = 0; q < 8; q++) g1tw_inv[m * 8 + q] = w[q];
    }
}

// ------------------------------------------------------------------------------------------
// k_cells_ntt: one 1024-thread workgroup per blob.
//   in : blob (4096 x 32 B big-endian, evaluations in bit-reversed order)
//   out: coeffs[b] (4096 Montgomery rows, natural order)           -- consumed by FK20
//        cells[b]  (8192 x 32 B big-endian; first half = the blob, second half = coset NTT)
// ------------------------------------------------------------------------------------------
constexpr int NT = 1024;
__device__ __forceinline__ void c_load_be32(uint32_t w[8], const uint8_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 hi = q[0], lo = q[1];
    w[7] = __builtin_bswap32(hi.x); w[6] = __builtin_bswap32(hi.y); w[5] = __builtin_bswap32(hi.z); w[4] = __builtin_bswap32(hi.w);
    w[3] = __builtin_bswap32(lo.x); w[2] = __builtin_bswap32(lo.y); w[1] = __builtin_bswap32(lo.z); w[0] = __builtin_bswap32(lo.w);
}
__device__ __forceinline__ void c_store_be32(uint8_t* p, const uint32_t w[8]) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(__builtin_bswap32(w[7]), __builtin_bswap32(w[6]), __builtin_bswap32(w[5]), __builtin_bswap32(w[4]));
    q[1] = make_uint4(__builtin_bswap32(w[3]), __builtin_bswap32(w[2]), __builtin_bswap32(w[1]), __builtin_bswap32(w[0]));
}

// INPUT 0: blob bytes (evaluation form).  INPUT 1: coefficient rows already on the device (recovery).
template <int INPUT>
__global__ void __launch_bounds__(NT) k_cells_ntt(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ coeffs_in, const uint32_t* __restrict__ tw_fwd,
                                                  const uint32_t* __restrict__ tw_inv, uint32_t* __restrict__ coeffs_out, uint8_t* __restrict__ cells,
                                                  int* __restrict__ status) {
    extern __shared__ uint32_t sh[];   // [9][4096]
    constexpr int N = 4096;
    const int b = blockIdx.x, t = threadIdx.x;
    if (INPUT == 0) {
        const uint8_t* blob = blobs + (size_t)b * 131072;
        bool bad = false;
#pragma unroll
        for (int k = 0; k < N / NT; k++) {
            const int i = t + NT * k;
            uint32_t w[8];
            c_load_be32(w, blob + (size_t)i * 32);
            bad |= Fr::words32_geq_mod(w);
            fr_t pl, m;
            Fr::from_words32(pl, w);
            Fr::from_plain(m, pl);
            Fr::reduce_once(m, m);
            lds_put<N>(sh, i, m);
        }
        if (__syncthreads_or(bad)) {
            if (t == 0) status[b] = MKZG_ERR_NON_CANONICAL_SCALAR;
            return;
        }
        if (t == 0) status[b] = 0;
        // inverse NTT, decimation in time: the blob holds evaluations in bit-reversed order, which
        // is the input order DIT wants; output = coefficients in natural order
        for (int h = 1; h < N; h <<= 1) {
#pragma unroll
            for (int r = 0; r < N / 2 / NT; r++) {
                const int p = t + NT * r;
                const int k = p & (h - 1);
                const int i0 = ((p - k) << 1) + k, i1 = i0 + h;
                fr_t a, bb, w, tt;
                lds_get<N>(a, sh, i0);
                lds_get<N>(bb, sh, i1);
                c_load_fr12(w, tw_inv + (size_t)(k * (N / h)) * 12);
                c_fr_mul(tt, bb, w);
                FrC::add(bb, a, tt);
                FrC::sub(a, a, tt);
                lds_put<N>(sh, i0, bb);
                lds_put<N>(sh, i1, a);
            }
            __syncthreads();
        }
    } else {
        if (status[b] != 0) return;
#pragma unroll
        for (int k = 0; k < N / NT; k++) {
            const int i = t + NT * k;
            fr_t m;
            c_load_fr12(m, coeffs_in + ((size_t)b * N + i) * 12);
            lds_put<N>(sh, i, m);
        }
        __syncthreads();
    }
    // (INPUT 0) scale by 1/4096 and publish the coefficients; then shift to the coset w_8192
    fr_t ninv;
    Fr::set(ninv, FR_INV_4096);
#pragma unroll
    for (int k = 0; k < N / NT; k++) {
        const int i = t + NT * k;
        fr_t c, w;
        lds_get<N>(c, sh, i);
        if (INPUT == 0) {
            c_fr_mul(c, c, ninv);
            if (coeffs_out) c_store_fr12(coeffs_out + ((size_t)b * N + i) * 12, c);
        }
        c_load_fr12(w, tw_fwd + (size_t)i * 12);
        c_fr_mul(c, c, w);
        lds_put<N>(sh, i, c);
    }
    if (INPUT == 0 && cells == nullptr) return;      // coefficients only (the batch verifier evaluates them at the challenge)
    __syncthreads();
    // forward NTT-4096, decimation in frequency: natural in, bit-reversed out
    for (int h = N / 2; h >= 1; h >>= 1) {
#pragma unroll
        for (int r = 0; r < N / 2 / NT; r++) {
            const int p = t + NT * r;
            const int k = p & (h - 1);
            const int i0 = ((p - k) << 1) + k, i1 = i0 + h;
            fr_t a, bb, w, s, d;
            lds_get<N>(a, sh, i0);
            lds_get<N>(bb, sh, i1);
            FrC::add(s, a, bb);
            FrC::sub(d, a, bb);
            c_load_fr12(w, tw_fwd + (size_t)(k * (N / h)) * 12);
            c_fr_mul(d, d, w);
            lds_put<N>(sh, i0, s);
            lds_put<N>(sh, i1, d);
        }
        __syncthreads();
    }
    // serialise: second half of the extended blob; the first half is the blob itself
    uint8_t* out = cells + (size_t)b * 262144;
#pragma unroll
    for (int k = 0; k < N / NT; k++) {
        const int i = t + NT * k;
        fr_t v, pl;
        uint32_t w[8];
        lds_get<N>(v, sh, i);
        Fr::to_plain(pl, v);
        Fr::to_words32(w, pl);
        c_store_be32(out + 131072 + (size_t)i * 32, w);
        if (INPUT == 0) {
            const uint4* src = reinterpret_cast<const uint4*>(blobs + (size_t)b * 131072 + (size_t)i * 32);
            uint4* dst = reinterpret_cast<uint4*>(out + (size_t)i * 32);
            dst[0] = src[0];
            dst[1] = src[1];
        }
    }
}

// first half of the cells from coefficients (recovery path): plain NTT-4096 (no coset), DIF order
__global__ void __launch_bounds__(NT) k_cells_first_half(const uint32_t* __restrict__ coeffs_in, const uint32_t* __restrict__ tw_fwd, uint8_t* __restrict__ cells,
                                                         const int* __restrict__ status) {
    extern __shared__ uint32_t sh[];
    constexpr int N = 4096;
    const int b = blockIdx.x, t = threadIdx.x;
    if (status[b] != 0) return;
#pragma unroll
    for (int k = 0; k < N / NT; k++) {
        const int i = t + NT * k;
        fr_t m;
        c_load_fr12(m, coeffs_in + ((size_t)b * N + i) * 12);
        lds_put<N>(sh, i, m);
    }
    __syncthreads();
    for (int h = N / 2; h >= 1; h >>= 1) {
#pragma unroll
        for (int r = 0; r < N / 2 / NT; r++) {
            const int p = t + NT * r;
            const int k = p & (h - 1);
            const int i0 = ((p - k) << 1) + k, i1 = i0 + h;
            fr_t a, bb, w, s, d;
            lds_get<N>(a, sh, i0);
            lds_get<N>(bb, sh, i1);
            FrC::add(s, a, bb);
            FrC::sub(d, a, bb);
            c_load_fr12(w, tw_fwd + (size_t)(k * (N / h)) * 12);
            c_fr_mul(d, d, w);
            lds_put<N>(sh, i0, s);
            lds_put<N>(sh, i1, d);
        }
        __syncthreads();
    }
    uint8_t* out = cells + (size_t)b * 262144;
#pragma unroll
    for (int k = 0; k < N / NT; k++) {
        const int i = t + NT * k;
        fr_t v, pl;
        uint32_t w[8];
        lds_get<N>(v, sh, i);
        Fr::to_plain(pl, v);
        Fr::to_words32(w, pl);
        c_store_be32(out + (size_t)i * 32, w);
    }
}

// ------------------------------------------------------------------------------------------
// k_fk20_scalars: per blob, the 64 circulant rows of FK20 and their NTT-128s.
//   row i (i < 64):  circ_i[0] = c[4095 - i];  circ_i[64 + j] = c[64 j - i - 1] (j = 1..63); else 0
//   V_i = NTT_128(circ_i) / 2      (the transform stage of fk20_kernels.hip works on u / 2, where u = sum_i V_i T_i as the reference has it)
//   out: scalars[b][k][i] as 8 plain little-endian words   (k < 128 outputs, i < 64 rows)
// One 1024-thread workgroup per blob; all 64 x 128 elements live in LDS (64 transforms side by side).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT) k_fk20_scalars(const uint32_t* __restrict__ coeffs, const uint32_t* __restrict__ tw_fwd, const int* __restrict__ status,
                                                     uint32_t* __restrict__ scalars) {
    extern __shared__ uint32_t sh[];   // [9][4096] : 32 rows at a time; element (row r, pos p) at index r*128 + p
    constexpr int N = 4096;
    const int b = blockIdx.x, t = threadIdx.x;
    if (status[b] != 0) return;
    const uint32_t* c = coeffs + (size_t)b * 4096 * 12;
    fr_t inv128;       // the scale folded into the scalars: 1/2
    {
        // 1/2 = 2048 * (1/4096)
        fr_t x;
        Fr::set(x, FR_INV_4096);
        for (int s = 0; s < 11; s++) FrC::add(x, x, x);
        inv128 = x;
    }
    uint32_t* out = scalars + (size_t)b * 128 * 64 * 8;
    for (int half = 0; half < 2; half++) {
        for (int e = t; e < N; e += NT) {
            const int i = (e >> 7) + 32 * half, p = e & 127;
            fr_t v;
            Fr::set_zero(v);
            if (p == 0) c_load_fr12(v, c + (size_t)(4095 - i) * 12);
            else if (p > 64) c_load_fr12(v, c + (size_t)(64 * (p - 64) - i - 1) * 12);
            lds_put<N>(sh, e, v);
        }
        __syncthreads();
        // 32 independent NTT-128s (DIF), w_128 = w_8192^64
        for (int h = 64; h >= 1; h >>= 1) {
            for (int pr = t; pr < N / 2; pr += NT) {
                const int row = pr >> 6, q = pr & 63;
                const int k = q & (h - 1);
                const int i0 = row * 128 + ((q - k) << 1) + k, i1 = i0 + h;
                fr_t a, bb, w, s, d;
                lds_get<N>(a, sh, i0);
                lds_get<N>(bb, sh, i1);
                FrC::add(s, a, bb);
                FrC::sub(d, a, bb);
                c_load_fr12(w, tw_fwd + (size_t)(k * (64 / h) * 64) * 12);
                c_fr_mul(d, d, w);
                lds_put<N>(sh, i0, s);
                lds_put<N>(sh, i1, d);
            }
            __syncthreads();
        }
        // DIF left the outputs bit-reversed within each row: V_i[k] sits at position brev7(k)
        for (int e = t; e < N; e += NT) {
            const int k = e >> 5, r = e & 31;                  // output index [k][i], i = r + 32*half
            const int pos = (int)(__brev((uint32_t)k) >> 25);
            fr_t v, pl;
            uint32_t w[8];
            lds_get<N>(v, sh, r * 128 + pos);
            c_fr_mul(v, v, inv128);
            Fr::to_plain(pl, v);
            Fr::to_words32(w, pl);
            uint4* dst = reinterpret_cast<uint4*>(out + ((size_t)k * 64 + r + 32 * half) * 8);
            dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
            dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// VerifyCellKZGProofBatch (reference internal/kzg_multi/kzg_verify.go:16-105), device side of the
// scalar work.  w_8192^e for any e in [0, 8192): the tables hold e < 4096, the rest is -w^(e-4096).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void tw_any(fr_t& r, const uint32_t* __restrict__ tw, uint32_t e) {
    e &= 8191u;
    if (e < 4096u) {
        c_load_fr12(r, tw + (size_t)e * 12);
    } else {
        fr_t t;
        c_load_fr12(t, tw + (size_t)(e - 4096u) * 12);
        FrC::neg(r, t);
    }
}
__device__ __forceinline__ void fr_reduce_words(uint32_t v[8]) {   // value < 2^256 -> value mod r
    for (int rep = 0; rep < 2; rep++) {
        if (Fr::words32_geq_mod(v)) {
            uint64_t br = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)v[i] - FrParams::MOD32[i] - br; v[i] = (uint32_t)t; br = (t >> 32) & 1; }
        }
    }
}
__device__ __forceinline__ void fr_store_plain8(uint32_t* dst, const fr_t& m) {
    fr_t pl;
    uint32_t w[8];
    Fr::to_plain(pl, m);
    Fr::to_words32(w, pl);
#pragma unroll
    for (int q = 0; q < 8; q++) dst[q] = w[q];
}
// r^k rows, s1_k = r^k, s4_k = r^k * h_{cell_k}^64   (kzg_verify.go:28, 73-79; srs.go:75-78)
// r^(k + exp_offset) in Montgomery form (canonical): the exponent is the GLOBAL index of the cell (sharded batches: exp_offset = first
// cell of this rank)
__device__ __forceinline__ void cell_rpower(fr_t& rp, const fr_t& r, uint32_t e) {
    fr_t base = r;
    Fr::set_one(rp);
    for (; e; e >>= 1) {
        if (e & 1) c_fr_mul(rp, rp, base);
        c_fr_mul(base, base, base);
    }
    Fr::reduce_once(rp, rp);
}
// ONE launch for everything that needs the indices and r only (round 5: two launches until then, and every launch is time in a
// hardware queue the call shares with other calls):
//   blocks [0, nb_cells)            per cell k: r^k (Montgomery row for the interpolation, plain words for the MSM sum r^k pi_k) and
//                                   r^k h_c^64 (the MSM sum r^k h_k^64 pi_k, kzg_verify.go:68-80)
//   blocks [nb_cells, + n_rows)     weights[row] = sum of r^k over the cells k that belong to commitment `row` (kzg_verify.go:35-41): a
//                                   strided scan of the row indices, the power computed HERE for every match (a row of a block has 128
//                                   cells: half a power per thread) -- so the two halves of the grid do not depend on each other --
//                                   then an LDS tree over the Fr partial sums
__global__ void __launch_bounds__(256) k_cell_rpowers(const RWords r_words8, uint32_t exp_offset, const uint64_t* __restrict__ cell_idx, const uint32_t* __restrict__ row_idx, int n,
                                                      int n_rows, int nb_cells, const uint32_t* __restrict__ tw_fwd, uint32_t* __restrict__ rp_rows, uint32_t* __restrict__ s1,
                                                      uint32_t* __restrict__ s4, uint32_t* __restrict__ weights8) {
    __shared__ uint32_t sh[9 * 256];
    const int t = threadIdx.x;
    uint32_t rw[8];
#pragma unroll
    for (int i = 0; i < 8; i++) rw[i] = r_words8.w[i];
    fr_reduce_words(rw);
    fr_t r, pl;
    Fr::from_words32(pl, rw);
    Fr::from_plain(r, pl);
    Fr::reduce_once(r, r);
    if ((int)blockIdx.x < nb_cells) {
        const int k = blockIdx.x * 256 + t;
        if (k >= n) return;
        fr_t rp, rp64, inv64;
        cell_rpower(rp, r, (uint32_t)k + exp_offset);
        Fr::set(inv64, FR_INV_4096);
        for (int sft = 0; sft < 6; sft++) FrC::add(inv64, inv64, inv64);     // 64 / 4096
        c_fr_mul(rp64, rp, inv64);
        c_store_fr12(rp_rows + (size_t)k * 12, rp64);                 // the interpolation's row: r^k / 64 (the inverse transform's scale rides along)
        fr_store_plain8(s1 + (size_t)k * 8, rp);
        const uint32_t c = (uint32_t)cell_idx[k] & 127u;
        fr_t hp, v;
        tw_any(hp, tw_fwd, 64u * (__brev(c) >> 25));                 // h_c^64 = w_8192^(64 brev7(c))
        c_fr_mul(v, rp, hp);
        fr_store_plain8(s4 + (size_t)k * 8, v);
        return;
    }
    const int row = (int)blockIdx.x - nb_cells;
    fr_t acc;
    Fr::set_zero(acc);
    for (int k = t; k < n; k += 256) {
        if ((int)row_idx[k] == row) {
            fr_t v;
            cell_rpower(v, r, (uint32_t)k + exp_offset);
            FrC::add(acc, acc, v);
        }
    }
#pragma unroll
    for (int q = 0; q < 9; q++) sh[q * 256 + t] = acc.l[q];
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (t < st) {
            fr_t a, b2;
#pragma unroll
            for (int q = 0; q < 9; q++) { a.l[q] = sh[q * 256 + t]; b2.l[q] = sh[q * 256 + t + st]; }
            FrC::add(a, a, b2);
#pragma unroll
            for (int q = 0; q < 9; q++) sh[q * 256 + t] = a.l[q];
        }
        __syncthreads();
    }
    if (t == 0) {
#pragma unroll
        for (int q = 0; q < 9; q++) acc.l[q] = sh[q * 256];
        fr_store_plain8(weights8 + (size_t)row * 8, acc);
    }
}
// per cell: canonical check, coset IFFT-64 on the bit-reversed evaluations, times r^k / 64, summed per workgroup into
// partials[block][64]   (kzg_verify.go:50-66, coset_fft.go:55-70).
// Round 5 shape: TWO cells per pass of a single-wavefront workgroup.  A radix-2 stage of a 64-point transform is 32 butterflies, so one
// cell left half the lanes idle through the six stage multiplications; lanes 0-31 now take cell A's butterflies and lanes 32-63 cell
// B's.  The 32 stage twiddles w_64^-j sit in LDS (they were a global load inside every stage of every cell: a ~1 us stall in a chain of
// six), and the 1/64 of the inverse transform rides in the power rows (k_cell_rpowers stores r^k / 64).  Per pair of cells 12 wavefront
// multiplications instead of 20, six barriers instead of twelve.
__global__ void __launch_bounds__(64) k_cell_interp(const uint8_t* __restrict__ cells, const uint64_t* __restrict__ cell_idx, const uint32_t* __restrict__ rp_rows, int n,
                                                    const uint32_t* __restrict__ tw_inv, uint32_t* __restrict__ partials, int* __restrict__ cell_status, uint32_t* __restrict__ interp8) {
    __shared__ uint32_t sh[2][9 * 64];
    __shared__ uint32_t tws[9 * 32];
    const int lane = threadIdx.x, half = lane >> 5, l = lane & 31;
    fr_t acc;
    Fr::set_zero(acc);
    if (lane < 32) {
        fr_t w;
        c_load_fr12(w, tw_inv + (size_t)(lane * 128) * 12);       // w_64^-lane = w_8192^(-128 lane)
        lds_put<32>(tws, lane, w);
    }
    __syncthreads();
    for (int k0 = blockIdx.x; k0 < n; k0 += 2 * (int)gridDim.x) {
        const int k1 = k0 + (int)gridDim.x;
        const bool has1 = k1 < n;
        bool ok[2];
#pragma unroll
        for (int x = 0; x < 2; x++) {
            const int k = x ? k1 : k0;
            if (x && !has1) { ok[1] = false; break; }
            uint32_t w[8];
            c_load_be32(w, cells + (size_t)k * 2048 + (size_t)lane * 32);
            const bool bad = __ballot(Fr::words32_geq_mod(w)) != 0;      // one wavefront: a vote, no barrier
            ok[x] = !bad;
            if (lane == 0) cell_status[k] = bad ? MKZG_ERR_NON_CANONICAL_SCALAR : 0;
            fr_t pl, m;
            Fr::from_words32(pl, w);
            Fr::from_plain(m, pl);
            Fr::reduce_once(m, m);
            lds_put<64>(sh[x], lane, m);
        }
        __syncthreads();
        // DIT with inverse twiddles: the cell is the bit-reversed natural-order input (kzg_verify.go:52)
        if (half == 0 || has1) {
            uint32_t* my = sh[half];
#pragma unroll 1
            for (int h = 1; h < 64; h <<= 1) {
                const int kk = l & (h - 1);
                const int i0 = ((l - kk) << 1) + kk, i1 = i0 + h;
                fr_t a, bb, tw, tt;
                lds_get<64>(a, my, i0);
                lds_get<64>(bb, my, i1);
                lds_get<32>(tw, tws, kk * (32 / h));
                c_fr_mul(tt, bb, tw);
                FrC::add(bb, a, tt);
                FrC::sub(a, a, tt);
                lds_put<64>(my, i0, bb);                             // one wavefront: its LDS operations execute in program order, the barrier
                lds_put<64>(my, i1, a);                              // below only keeps the compiler from moving them across stages
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 2; x++) {
            if ((x && !has1) || !ok[x]) continue;
            const int k = x ? k1 : k0;
            fr_t v, hs, rp;
            lds_get<64>(v, sh[x], lane);
            const uint32_t c = (uint32_t)cell_idx[k] & 127u;
            tw_any(hs, tw_inv, (uint32_t)lane * (__brev(c) >> 25));  // (h_c^-1)^lane
            c_load_fr12(rp, rp_rows + (size_t)k * 12);               // r^k / 64
            c_fr_mul(v, v, hs);
            c_fr_mul(v, v, rp);
            FrC::add(acc, acc, v);
        }
        __syncthreads();
    }
    // interp[i] = the sum of all workgroups' partial sums, by whoever finishes last -- a launch of its own until round 5.  Two levels of
    // tickets (behind the partial rows: zero when the workspace is allocated, reset by the taker of the last one): the last workgroup of
    // every GROUP of 64 adds its group's rows, the last group to finish adds the <= 32 group rows and writes the 64 plain scalars.
    // Rows travel as agent-scope stores / loads (pub_ticket.hpp), nine words each.
    __shared__ int is_last;
    uint32_t* gpart = partials + (size_t)CELL_INTERP_BLOCKS * 64 * 12;         // [CELL_INTERP_BLOCKS / 64][64] Fr rows
    uint32_t* tick = gpart + (size_t)(CELL_INTERP_BLOCKS / 64) * 64 * 12;      // [groups] + 1
    const int g = (int)blockIdx.x >> 6, n_groups = ((int)gridDim.x + 63) >> 6;
    const int g_size = min(64, (int)gridDim.x - (g << 6));
    auto put_row = [&](uint32_t* row, const fr_t& v) {
#pragma unroll
        for (int q = 0; q < 9; q++) pub_store(row + q, v.l[q]);
    };
    auto get_row = [&](fr_t& v, const uint32_t* row) {
#pragma unroll
        for (int q = 0; q < 9; q++) v.l[q] = pub_load(row + q);
    };
    put_row(partials + ((size_t)blockIdx.x * 64 + lane) * 12, acc);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (lane == 0) {
        const uint32_t got = pub_ticket(&tick[g]);
        is_last = got == (uint32_t)(g_size - 1);
        if (is_last) pub_store(&tick[g], 0u);
    }
    __syncthreads();
    if (!is_last) return;
    Fr::set_zero(acc);
    for (int b = 0; b < g_size; b++) {
        fr_t v;
        get_row(v, partials + ((size_t)((g << 6) + b) * 64 + lane) * 12);
        FrC::add(acc, acc, v);
    }
    put_row(gpart + ((size_t)g * 64 + lane) * 12, acc);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (lane == 0) {
        const uint32_t got = pub_ticket(&tick[CELL_INTERP_BLOCKS / 64]);
        is_last = got == (uint32_t)(n_groups - 1);
        if (is_last) pub_store(&tick[CELL_INTERP_BLOCKS / 64], 0u);
    }
    __syncthreads();
    if (!is_last) return;
    Fr::set_zero(acc);
    for (int b = 0; b < n_groups; b++) {
        fr_t v;
        get_row(v, gpart + ((size_t)b * 64 + lane) * 12);
        FrC::add(acc, acc, v);
    }
    fr_store_plain8(interp8 + (size_t)lane * 8, acc);
}
// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
static bool g_attr_set = false;
static void set_lds_attrs() {
    if (g_attr_set) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cells_ntt<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 4096 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cells_ntt<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 4096 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cells_first_half), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 4096 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fk20_scalars), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 4096 * 4);
    g_attr_set = true;
}
void build_twiddles_device(uint32_t* d_tw_fwd, uint32_t* d_tw_inv, uint32_t* d_g1tw_fwd, uint32_t* d_g1tw_inv, hipStream_t s) {
    hipLaunchKernelGGL(k_build_twiddles, dim3(16), dim3(256), 0, s, d_tw_fwd, d_tw_inv, d_g1tw_fwd, d_g1tw_inv);
}
void cells_from_blobs_device(const uint8_t* d_blobs, int n, const uint32_t* d_tw_fwd, const uint32_t* d_tw_inv, uint32_t* d_coeffs, uint8_t* d_cells, int* d_status,
                             hipStream_t s) {
    set_lds_attrs();
    if (n > 0) hipLaunchKernelGGL((k_cells_ntt<0>), dim3(n), dim3(NT), 9 * 4096 * 4, s, d_blobs, (const uint32_t*)nullptr, d_tw_fwd, d_tw_inv, d_coeffs, d_cells, d_status);
}
void cells_from_coeffs_device(const uint32_t* d_coeffs, int n, const uint32_t* d_tw_fwd, const uint32_t* d_tw_inv, uint8_t* d_cells, int* d_status, hipStream_t s) {
    set_lds_attrs();
    if (n <= 0) return;
    hipLaunchKernelGGL(k_cells_first_half, dim3(n), dim3(NT), 9 * 4096 * 4, s, d_coeffs, d_tw_fwd, d_cells, d_status);
    hipLaunchKernelGGL((k_cells_ntt<1>), dim3(n), dim3(NT), 9 * 4096 * 4, s, (const uint8_t*)nullptr, d_coeffs, d_tw_fwd, d_tw_inv, (uint32_t*)nullptr, d_cells, d_status);
}
void fk20_scalars_device(const uint32_t* d_coeffs, int n, const uint32_t* d_tw_fwd, const int* d_status, uint32_t* d_scalars, hipStream_t s) {
    set_lds_attrs();
    if (n > 0) hipLaunchKernelGGL(k_fk20_scalars, dim3(n), dim3(NT), 9 * 4096 * 4, s, d_coeffs, d_tw_fwd, d_status, d_scalars);
}

void cell_verify_scalars_device(const uint32_t* h_r8, uint32_t exp_offset, const uint64_t* d_cell_idx, const uint32_t* d_row_idx, const uint8_t* d_cells, int n, int n_rows,
                                const uint32_t* d_tw_fwd, const uint32_t* d_tw_inv, uint32_t* d_rp_rows, uint32_t* d_s1, uint32_t* d_s4, uint32_t* d_weights8,
                                uint32_t* d_partials, uint32_t* d_interp8, int* d_cell_status, hipStream_t s, int phase) {
    if (n <= 0) return;
    if (phase != 2) {                          // what needs the indices and r only: the caller issues it ahead of the cells' upload
        RWords r;                              // by value: see rlc_scalars_device
        for (int i = 0; i < 8; i++) r.w[i] = h_r8[i];
        const int nb_cells = (n + 255) / 256;
        hipLaunchKernelGGL(k_cell_rpowers, dim3(nb_cells + n_rows), dim3(256), 0, s, r, exp_offset, d_cell_idx, d_row_idx, n, n_rows, nb_cells, d_tw_fwd, d_rp_rows, d_s1, d_s4, d_weights8);
    }
    if (phase != 1) {                          // what reads the cells
        const int nb = n < CELL_INTERP_BLOCKS ? n : CELL_INTERP_BLOCKS;
        hipLaunchKernelGGL(k_cell_interp, dim3(nb), dim3(64), 0, s, d_cells, d_cell_idx, d_rp_rows, n, d_tw_inv, d_partials, d_cell_status, d_interp8);     // (sums its partials itself)
    }
}

}  // namespace kzg
