// fk20_kernels.hip -- FK20 amortised multi-proofs for EIP-7594 on MI355X (gfx950).
//
// Reference: internal/kzg_multi/fk20/fk20.go:23-124 + toeplitz.go:42-125 (+ the G1 FFT of
// internal/domain/fft.go:23-92).  Per blob:
//     V[k][i]  = NTT_128(circulant row i)[k] / 128                 (cells_kernels.hip)
//     R[k]     = sum_{i<64} V[k][i] * T[k][i]                      128 fixed-base MSMs of 64 points
//     h        = IFFT_128^{G1}(R)[0..63], zero-padded to 128
//     proofs   = bitrev(FFT_128^{G1}(h))
// MI355X-first choices:
//   * the 8192 fixed bases T[k][i] get a DIRECT window table in HBM: D[k][i][w][m] = m 2^(11w) T
//     for the 24 eleven-bit windows w and m = 1..1024 (signed Booth digits) -- 25.8 GB, sized for a
//     288 GB part -- so an MSM-64 is 64 x 24 mixed additions with no bucket reduction at all and
//     no sort; 8 (batches) or 24 (few blobs) lanes share one MSM and combine with a wavefront tree.
//   * the two per-blob G1 transforms (IFFT_128, truncate, FFT_128) are ONE fixed linear map: P = u/2 + c (*) u, two 64 x 64 Toeplitz
//     products with fixed scalars.  They run as a three-level Toom-Cook network in transposed form (gen_fk20_toom.py): additions and
//     doublings of points (k_fk20_pre), 686 INDEPENDENT scalar multiplications by fixed scalars (k_fk20_toom_mul: one wavefront per
//     scalar, one lane per blob, GLV + width-5 NAF, Jacobian accumulator), additions again (k_fk20_post) -- one multiplication deep,
//     where the radix-2 G1 FFTs of the reference (internal/domain/fft.go:49-92) are fourteen deep.
//     The one-wavefront radix-2 FFT (g1_fft128_dif, points in LDS) is kept for the one-off set-up of the table.
// Integer modular arithmetic only -- no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "g1.hpp"
#include "g1_quad.hpp"
#include "fk20_kernels.h"
#include "fk20_toom.h"

namespace kzg {

__device__ __noinline__ void f_add(G1X& r, const G1X& a, const G1X& b) { g1x_add(r, a, b); }
__device__ __noinline__ void f_dbl(G1X& r, const G1X& a) { g1x_dbl(r, a); }
__device__ __noinline__ void f_madd(G1X& acc, const fp_t& x, const fp_t& y) { g1x_madd(acc, x, y); }
__device__ __noinline__ void f_to_affine(G1Aff& r, const G1X& p) { g1x_to_affine(r, p); }

__device__ __forceinline__ void f_load_affine(G1Aff& p, const uint32_t* rec) {
    const uint4* q = reinterpret_cast<const uint4*>(rec);
    uint32_t w[32];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint4 v = q[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < 14; i++) { p.x.l[i] = w[i]; p.y.l[i] = w[14 + i]; }
    p.inf = w[28];
}
__device__ __forceinline__ void f_store_affine(uint32_t* rec, const G1Aff& p) {
    uint32_t w[32];
#pragma unroll
    for (int i = 0; i < 14; i++) { w[i] = p.x.l[i]; w[14 + i] = p.y.l[i]; }
    w[28] = p.inf; w[29] = 0; w[30] = 0; w[31] = 0;
    uint4* q = reinterpret_cast<uint4*>(rec);
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
__device__ __forceinline__ void f_load_xyzz(G1X& p, const uint32_t* rec) {
    const uint4* q = reinterpret_cast<const uint4*>(rec);
    uint32_t w[56];
#pragma unroll
    for (int i = 0; i < 14; i++) {
        uint4 v = q[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < 14; i++) { p.x.l[i] = w[i]; p.y.l[i] = w[14 + i]; p.zz.l[i] = w[28 + i]; p.zzz.l[i] = w[42 + i]; }
}
__device__ __forceinline__ void f_store_xyzz(uint32_t* rec, const G1X& p) {
    uint32_t w[56];
#pragma unroll
    for (int i = 0; i < 14; i++) { w[i] = p.x.l[i]; w[14 + i] = p.y.l[i]; w[28 + i] = p.zz.l[i]; w[42 + i] = p.zzz.l[i]; }
    uint4* q = reinterpret_cast<uint4*>(rec);
#pragma unroll
    for (int i = 0; i < 14; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// r = [k]p for a twiddle k given by its GLV halves kk = (k1[4 words], k2[4 words]), k = k1 + k2 x^2:
//   [k]p = [k1]p - [k2]phi(p),  phi(X, Y, ZZ, ZZZ) = (beta X, Y, ZZ, ZZZ)
// Straus/Shamir with signed 5-bit windows over both halves, one table of 16 multiples of p (kept in the lane's private memory).
// Signed 5-bit Booth digit w of a 128-bit half-scalar k (4 words): value in [-16, 16], sum_w d_w 32^w = k, w = 0..25
__device__ __forceinline__ int glv_digit5(const uint32_t* k, int w) {
    const int bit = 5 * w - 1;
    uint32_t v;
    if (bit < 0) {
        v = (k[0] << 1) & 0x3fu;
    } else {
        const int wi = bit >> 5, off = bit & 31;
        v = wi < 4 ? k[wi] >> off : 0u;
        if (off + 6 > 32 && wi + 1 < 4) v |= k[wi + 1] << (32 - off);
        v &= 0x3fu;
    }
    int d = (int)((v + 1) >> 1);
    if (v >> 5) d -= 32;
    return d;
}
__device__ __noinline__ void f_mul_glv(G1X& r, const G1X& p, const uint32_t* kk) {
    // The accumulator is Jacobian (g1.hpp jdbl: 5 squarings + 2 multiplications per doubling, against 3 + 6 in XYZZ -- the 125
    // doublings are two thirds of this routine); the 16 table entries carry Z^2 and Z^3 so that an addition costs what the XYZZ
    // one does.  Group operations are inlined (one copy each, inside rolled loops) so that the accumulator stays in registers;
    // only the table (1..16 times p) is addressed dynamically and lives in the lane's scratch memory.
    // 26 signed 5-bit windows over both half-scalars: 125 doublings + <= 52 additions + 15 for the table.
    G1JC tbl[16];
    {
        G1J run;
        G1JC pc;
        G1D::jac_from_xyzz(run, p);
        G1D::jac_cache(pc, run);
        tbl[0] = pc;
#pragma unroll 1
        for (int i = 1; i < 16; i++) {
            G1JC e;
            G1D::jadd(run, run, pc);
            G1D::jac_cache(e, run);
            tbl[i] = e;
        }
    }
    fp_t beta;
    Fp::set(beta, FP_BETA);
    G1J acc;
    G1D::set_inf(acc);
    int top = 25;
    while (top >= 0 && glv_digit5(kk, top) == 0 && glv_digit5(kk + 4, top) == 0) top--;
#pragma unroll 1
    for (int w = top; w >= 0; w--) {
        if (w != top) {
#pragma unroll 1
            for (int q = 0; q < 5; q++) G1D::jdbl(acc, acc);
        }
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            const int d = glv_digit5(kk + 4 * half, w);
            if (d) {
                G1JC q = tbl[(d < 0 ? -d : d) - 1];
                // second half: -phi(.) = (beta X, -Y); a negative digit flips the sign once more
                if (half) Fp::mul(q.x, q.x, beta);
                if ((d < 0) != (half != 0)) Fp::template neg<3>(q.y, q.y);   // 8p - Y (Y < 5.01p)
                G1D::jadd(acc, acc, q);
            }
        }
    }
    G1D::jac_to_xyzz(r, acc);
}

// r = [k]p for the WHOLE wavefront (k uniform; lane = blob): the scalar's GLV halves as width-5 NAF strings (fk20_toom.h
// FK20_TOOM_WNAF), so control flow is uniform, an addition happens only at the ~1/6 of the bit positions with a non-zero digit
// (43 additions on average against 52 with per-lane fixed windows) and the table holds the 8 odd multiples instead of 16.
// Same Jacobian accumulator and cached-Z table entries as f_mul_glv.
#ifndef MKZG_FK20_COZ
#define MKZG_FK20_COZ 1                           // round 5: co-Z table + mixed incomplete additions (g1.hpp mul_wnaf_coz); 0 builds the form below
#endif
__device__ __noinline__ void f_mul_wnaf(G1X& r, const G1X& p, const uint32_t* __restrict__ row) {
#if MKZG_FK20_COZ
    fp_t beta_c;
    Fp::set(beta_c, FP_BETA);
    G1D::mul_wnaf_coz(r, p, row, beta_c);
    return;
#endif
    G1JC tbl[8];                                  // (2j + 1) p
    {
        G1J run, two;
        G1JC pc, twoc;
        G1D::jac_from_xyzz(run, p);
        G1D::jac_cache(pc, run);
        tbl[0] = pc;
        G1D::jdbl(two, run);
        G1D::jac_cache(twoc, two);
#pragma unroll 1
        for (int j = 1; j < 8; j++) {
            G1JC c;
            G1D::jadd(run, run, twoc);
            G1D::jac_cache(c, run);
            tbl[j] = c;
        }
    }
    fp_t beta;
    Fp::set(beta, FP_BETA);
    G1J acc;
    G1D::set_inf(acc);
    bool started = false;
#pragma unroll 1
    for (int i = 130; i >= 0; i--) {
        if (started) G1D::jdbl(acc, acc);
        const uint32_t code = (row[i >> 1] >> (16 * (i & 1))) & 0xffffu;
        if (code == 0) continue;
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            const int d = (int)(int8_t)(half ? code >> 8 : code & 0xffu);
            if (d) {
                G1JC q = tbl[((d < 0 ? -d : d) - 1) >> 1];
                // second half: -phi(.) = (beta X, -Y); a negative digit flips the sign once more
                if (half) Fp::mul(q.x, q.x, beta);
                if ((d < 0) != (half != 0)) Fp::template neg<3>(q.y, q.y);   // 8p - Y (Y < 5.01p)
                G1D::jadd(acc, acc, q);
                started = true;
            }
        }
    }
    G1D::jac_to_xyzz(r, acc);
}

// The same product on a QUAD or a PAIR of lanes (g1_quad.hpp) for calls of a few blobs, where the wavefronts of the 686 products have
// lanes to spare and the caller waits for the length of this chain: doublings 3 (4) multiplications deep instead of 7, additions 5 (7)
// instead of 13 (k_fk20_toom_mul 1.76 -> 0.99 ms up to 16 blobs, -> ~1.1 ms up to 32).  All lanes of a group hold the same point, table
// and accumulator.
template <int G>     // lanes per point: 4 (quad) or 2 (pair)
__device__ __noinline__ void f_mul_wnaf_group(G1X& r, const G1X& p, const uint32_t* __restrict__ row, int lane) {
    const QuadRole qr(lane);
    const bool isB = (lane & 1) != 0;
#define gdbl(x) do { if constexpr (G == 4) jdbl_quad(x, qr); else jdbl_pair(x, isB); } while (0)
#define gadd(x, y) do { if constexpr (G == 4) jadd_quad(x, y, qr); else jadd_pair(x, y, isB); } while (0)
    G1JC tbl[8];                                  // (2j + 1) p
    {
        G1J run, two;
        G1JC pc, twoc;
        G1D::jac_from_xyzz(run, p);
        G1D::jac_cache(pc, run);
        tbl[0] = pc;
        two = run;
        gdbl(two);
        G1D::jac_cache(twoc, two);
#pragma unroll 1
        for (int j = 1; j < 8; j++) {
            G1JC c;
            gadd(run, twoc);
            G1D::jac_cache(c, run);
            tbl[j] = c;
        }
    }
    fp_t beta;
    Fp::set(beta, FP_BETA);
    G1J acc;
    G1D::set_inf(acc);
    bool started = false;
#pragma unroll 1
    for (int i = 130; i >= 0; i--) {
        if (started) gdbl(acc);
        const uint32_t code = (row[i >> 1] >> (16 * (i & 1))) & 0xffffu;
        if (code == 0) continue;
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            const int d = (int)(int8_t)(half ? code >> 8 : code & 0xffu);
            if (d) {
                G1JC q = tbl[((d < 0 ? -d : d) - 1) >> 1];
                if (half) Fp::mul(q.x, q.x, beta);
                if ((d < 0) != (half != 0)) Fp::template neg<3>(q.y, q.y);   // 8p - Y (Y < 5.01p)
                gadd(acc, q);
                started = true;
            }
        }
    }
    G1D::jac_to_xyzz(r, acc);
#undef gdbl
#undef gadd
}

// ------------------------------------------------------------------------------------------
// G1 FFT-128 in one wavefront.  LDS: structure of arrays, word q of point e at sh[q*128 + e].
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void lp_get(G1X& p, const uint32_t* sh, int e) {
#pragma unroll
    for (int i = 0; i < 14; i++) {
        p.x.l[i] = sh[i * 128 + e];
        p.y.l[i] = sh[(14 + i) * 128 + e];
        p.zz.l[i] = sh[(28 + i) * 128 + e];
        p.zzz.l[i] = sh[(42 + i) * 128 + e];
    }
}
__device__ __forceinline__ void lp_put(uint32_t* sh, int e, const G1X& p) {
#pragma unroll
    for (int i = 0; i < 14; i++) {
        sh[i * 128 + e] = p.x.l[i];
        sh[(14 + i) * 128 + e] = p.y.l[i];
        sh[(28 + i) * 128 + e] = p.zz.l[i];
        sh[(42 + i) * 128 + e] = p.zzz.l[i];
    }
}
// decimation in frequency: natural order in, bit-reversed order out.  tw: 64 x 8 words, GLV halves of w^j.
__device__ __forceinline__ void g1_fft128_dif(uint32_t* sh, const uint32_t* __restrict__ tw, int lane) {
    for (int h = 64; h >= 1; h >>= 1) {
        const int k = lane & (h - 1);
        const int i0 = ((lane - k) << 1) + k, i1 = i0 + h;
        G1X a, b, s, d, nb;
        lp_get(a, sh, i0);
        lp_get(b, sh, i1);
        f_add(s, a, b);
        g1x_neg(nb, b);
        f_add(d, a, nb);
        const int e = k * (64 / h);
        if (e != 0) {
            uint32_t kw[8];
#pragma unroll
            for (int q = 0; q < 8; q++) kw[q] = tw[e * 8 + q];
            f_mul_glv(d, d, kw);
        }
        lp_put(sh, i0, s);
        lp_put(sh, i1, d);
        __syncthreads();
    }
}
// ------------------------------------------------------------------------------------------
// set-up, step 1: T[k][i] = FFT_128(vec_i)[k], vec_i = (reversed monomial SRS without its first 64
// points, stride 64 from offset i, padded with the identity)        reference fk20.go:28-35,
// toeplitz.go:50-93.  One wavefront per i; output affine rows t_aff[k*64 + i].
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_fk20_setup_fft(const uint32_t* __restrict__ monomial_aff, const uint32_t* __restrict__ g1tw_fwd, uint32_t* __restrict__ t_aff) {
    __shared__ uint32_t sh[56 * 128];
    const int i = blockIdx.x, lane = threadIdx.x;
    for (int j = lane; j < 128; j += 64) {
        G1X p;
        g1x_set_inf(p);
        if (j < 63) {
            G1Aff a;
            f_load_affine(a, monomial_aff + (size_t)(4031 - i - 64 * j) * 32);
            if (!a.inf) g1x_from_affine(p, a.x, a.y);
        }
        lp_put(sh, j, p);
    }
    __syncthreads();
    g1_fft128_dif(sh, &G1TW_GLV_FWD[0][0], lane);
    for (int k = lane; k < 128; k += 64) {
        G1X p;
        lp_get(p, sh, (int)(__brev((uint32_t)k) >> 25));
        G1Aff a;
        f_to_affine(a, p);
        f_store_affine(t_aff + ((size_t)k * 64 + i) * 32, a);
    }
}

// set-up, step 2: D[(k*64+i)][w][m-1] = m * 2^(C w) * T[k][i] for the nwin = ceil(256 / C) windows and m = 1 .. 2^(C-1), affine rows.
// One lane per (point, window).
__global__ void __launch_bounds__(64) k_fk20_setup_table(const uint32_t* __restrict__ t_aff, uint32_t* __restrict__ table, int n_points, int cbits, int nwin) {
    const size_t gid = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (gid >= (size_t)n_points * nwin) return;
    const int pt = (int)(gid / nwin), w = (int)(gid % nwin), nmult = 1 << (cbits - 1);
    G1Aff base;
    f_load_affine(base, t_aff + (size_t)pt * 32);
    uint32_t* dst = table + ((size_t)pt * nwin + w) * nmult * 32;
    if (base.inf) {
        for (int m = 0; m < nmult; m++) f_store_affine(dst + (size_t)m * 32, base);
        return;
    }
    G1X b;
    g1x_from_affine(b, base.x, base.y);
    for (int q = 0; q < cbits * w; q++) f_dbl(b, b);
    f_to_affine(base, b);
    G1X acc;
    g1x_set_inf(acc);
    for (int m = 0; m < nmult; m++) {
        f_madd(acc, base.x, base.y);
        G1Aff a;
        f_to_affine(a, acc);
        f_store_affine(dst + (size_t)m * 32, a);
    }
}

// ------------------------------------------------------------------------------------------
// k_fk20_msm: R[b][k] = sum_i V[b][k][i] * T[k][i] through the direct table of C-bit windows (NWIN = ceil(256 / C) signed Booth
// digits per scalar).  An MSM-64 is shared by LW x LP lanes: lane (wg, pg) adds the rows of windows WPL wg .. WPL wg + WPL - 1 for the
// points 64/LP pg .. 64/LP (pg + 1) - 1, so every lane of every configuration walks the same number of additions:
//   C = 11: <8, 1, 3> batches (8 MSMs per wavefront, 192 additions per lane; one wavefront per SIMD at 64 blobs), <12, 1, 2> up to 39
//           blobs (5 MSMs per wavefront, 128 additions per lane), <24, 1, 1> up to 16 (2 per wavefront, 64 additions), <24, 2, 1> up to
//           8 (one MSM per wavefront, 32 additions)
//   C = 13: <4, 2, 5> batches (160 additions per lane), <10, 2, 2> few blobs (64 additions per lane)
//   C = 12: <2, 4, 11> batches (176 additions per lane), <11, 1, 2> / <22, 1, 1> / <22, 2, 1> for fewer blobs
//   C = 8 : <8, 1, 4> batches (256 additions per lane), <16, 1, 2> up to 39 blobs (128 additions per lane)
// k_fk20_msm_reduce adds the lanes of an MSM.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void f_shfl_xor_xyzz(G1X& r, const G1X& a, int m) {
#pragma unroll
    for (int i = 0; i < 14; i++) {
        r.x.l[i] = __shfl_xor(a.x.l[i], m);
        r.y.l[i] = __shfl_xor(a.y.l[i], m);
        r.zz.l[i] = __shfl_xor(a.zz.l[i], m);
        r.zzz.l[i] = __shfl_xor(a.zzz.l[i], m);
    }
}
// signed Booth digit w of a 256-bit scalar s[8] (little-endian words): bits [C w - 1, C w + C), value in [-2^(C-1), 2^(C-1)]
template <int C>
__device__ __forceinline__ int fk_digit(const uint32_t s[8], int w) {
    const int bit = C * w - 1;
    uint32_t v;
    if (bit < 0) {
        v = (s[0] << 1) & ((2u << C) - 1);
    } else {
        const int wi = bit >> 5, off = bit & 31;
        uint64_t x = wi < 8 ? (uint64_t)s[wi] : 0ull;
        if (wi + 1 < 8) x |= (uint64_t)s[wi + 1] << 32;
        v = (uint32_t)(x >> off) & ((2u << C) - 1);
    }
    int d = (int)((v + 1) >> 1);
    if (v >> C) d -= (1 << C);
    return d;
}
template <int C, int LW, int LP, int WPL>
__global__ void __launch_bounds__(64, 2) k_fk20_msm(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ table, const int* __restrict__ status,
                                                    uint32_t* __restrict__ parts) {
    constexpr int NWIN = (256 + C - 1) / C, NMULT = 1 << (C - 1), LANES = LW * LP, PPL = 64 / LP;      // C NWIN >= 256: the top Booth digit of a scalar < 2^255 is not negative
    static_assert(LW * WPL == NWIN && LP * PPL == 64, "lanes x windows per lane = windows of a scalar; point groups tile the 64 points");
    constexpr int NSTEP = PPL * WPL;
    // Table rows travel HBM -> LDS with global_load_lds_dwordx4 (no VGPR destination) TWO additions ahead of their use, in two row
    // buffers: with one wavefront per SIMD (batch 64) a lane has nothing else to hide a row's latency behind, and over the 86 GB table
    // of 13-bit windows a random row takes longer than one addition (measured: one row ahead, 14.4 us per step against 10.4).
    __shared__ __attribute__((aligned(16))) uint32_t rowbuf[2][8 * 256];
    const int b = blockIdx.y;
    if (status[b] != 0) return;
    const int lane = threadIdx.x, sub = lane % LANES, wg = sub % LW, pg = sub / LW;
    const int k0 = blockIdx.x * (64 / LANES) + lane / LANES;
    const bool active = lane < (64 / LANES) * LANES && k0 < 128;       // e.g. <24, 1, 1> leaves 16 lanes without an MSM; 20-lane groups do not divide 128
    const int k = active ? k0 : 0;
    const uint32_t* sc = scalars + (((size_t)b * 128 + k) * 64 + (size_t)pg * PPL) * 8;      // this lane's first point
    G1X acc;
    g1x_set_inf(acc);
    // step u = (point i, window ww) of this lane; its digit and row address are computed when its DMA is issued, two steps early
    auto load_scalar = [&](uint32_t w8[8], int i) {
        const uint4 a = *reinterpret_cast<const uint4*>(sc + i * 8), c4 = *reinterpret_cast<const uint4*>(sc + i * 8 + 4);
        w8[0] = a.x; w8[1] = a.y; w8[2] = a.z; w8[3] = a.w; w8[4] = c4.x; w8[5] = c4.y; w8[6] = c4.z; w8[7] = c4.w;
    };
    uint32_t cur_s[8], nxt_s[8];
    load_scalar(cur_s, 0);
    load_scalar(nxt_s, PPL > 1 ? 1 : 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int iu = 0, wu = 0;                                    // point / window of the next step to ISSUE
    unsigned pend = 0;                                     // bit q: the row of step (t + q) is in flight / landed; bit 4 + q: it is to be subtracted
    auto issue = [&](int slot, int q) {                    // issue the DMA of step (iu, wu) into row buffer `slot`; q = distance to its use (1 or 2)
        const int wi = WPL * wg + wu, pi = pg * PPL + iu;
        const int d = fk_digit<C>(cur_s, wi);
        if (++wu == WPL) {
            wu = 0;
            iu++;
#pragma unroll
            for (int z = 0; z < 8; z++) cur_s[z] = nxt_s[z];
            if (iu + 1 < PPL) load_scalar(nxt_s, iu + 1);  // two VMEM loads, issued BEFORE this step's row: retired by the wait that retires the row before it
        }
        if (d != 0 && active) {
            const int m = d < 0 ? -d : d;
            pend |= (1u << q) | ((d < 0 ? 1u : 0u) << (4 + q));
            const uint32_t* row = table + (((size_t)(k * 64 + pi) * NWIN + wi) * NMULT + (m - 1)) * 32;
#pragma unroll
            for (int z = 0; z < 8; z++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + 4 * z),
                                                 (__attribute__((address_space(3))) void*)(rowbuf[slot] + z * 256), 16, 0, 0);
        } else {
            // keep the in-flight count per step constant (the wait below counts operations): eight harmless reads of this lane's row slot
#pragma unroll
            for (int z = 0; z < 8; z++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(table + 4 * z),
                                                 (__attribute__((address_space(3))) void*)(rowbuf[slot] + z * 256), 16, 0, 0);
        }
    };
    issue(0, 0);                                           // step 0 -> buffer 0
    if (NSTEP > 1) issue(1, 1);                            // step 1 -> buffer 1
#pragma unroll 1
    for (int t = 0; t < NSTEP; t++) {
        // everything but the eight most recent operations (the row of step t + 1) has landed: the row of step t and any scalar loads
        if (t + 1 < NSTEP) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G1Aff p;
        p.inf = 1;
        const bool have = (pend & 1u) != 0, neg = (pend & 16u) != 0;
        if (have) {
            const uint32_t* rb = rowbuf[t & 1];
            uint32_t rw[32];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint4 v = *reinterpret_cast<const uint4*>(rb + q * 256 + lane * 4);
                rw[4 * q] = v.x; rw[4 * q + 1] = v.y; rw[4 * q + 2] = v.z; rw[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int q = 0; q < 14; q++) { p.x.l[q] = rw[q]; p.y.l[q] = rw[14 + q]; }
            p.inf = rw[28];
#pragma unroll
            for (int q = 0; q < 14; q++) asm volatile("" : "+v"(p.x.l[q]), "+v"(p.y.l[q]));      // LDS reads retired before the buffer is refilled
            asm volatile("" : "+v"(p.inf));
        }
        pend = ((pend >> 1) & 0x7u) | (((pend >> 5) & 0x7u) << 4);
        if (t + 2 < NSTEP) issue(t & 1, 1);                // the buffer just read takes the row of step t + 2 (bit 1 after the shift = step t + 2 seen from t + 1)
        if (!p.inf) {
            if (neg) Fp::template neg<0>(p.y, p.y);
            g1x_madd(acc, p.x, p.y);
        }
    }
    // the lanes' partial sums of an MSM are combined by k_fk20_msm_reduce: any group addition here (inlined or called) makes the
    // compiler keep `acc` in scratch memory, whose loads then queue behind the row DMA on vmcnt
    if (active) f_store_xyzz(parts + (((size_t)b * 128 + k) * LANES + sub) * 64, acc);
}
// LANES partial sums per MSM, padded to PAD (a power of two) lanes with the identity for the shuffle tree
template <int LANES, int PAD>
__global__ void __launch_bounds__(64) k_fk20_msm_reduce(const uint32_t* __restrict__ parts, const int* __restrict__ status, uint32_t* __restrict__ results) {
    const int b = blockIdx.y;
    if (status[b] != 0) return;
    const int lane = threadIdx.x, k = blockIdx.x * (64 / PAD) + lane / PAD, sub = lane % PAD;
    G1X acc, o;
    if (sub < LANES) f_load_xyzz(acc, parts + (((size_t)b * 128 + k) * LANES + sub) * 64);
    else g1x_set_inf(acc);
#pragma unroll 1
    for (int m = PAD / 2; m >= 1; m >>= 1) {
        f_shfl_xor_xyzz(o, acc, m);
        g1x_add(acc, acc, o);
    }
    if (sub == 0) f_store_xyzz(results + ((size_t)b * 128 + k) * 64, acc);
}

// ------------------------------------------------------------------------------------------
// The two G1 transforms of a blob, IFFT_128 -> keep h[0..63] -> FFT_128 (fk20.go:76-90, toeplitz.go:95-125), as the Toom-Cook
// network of gen_fk20_toom.py.  With u' = u / 2 (the 1/2 is folded into the MSM's scalars, cells_kernels.hip k_fk20_scalars):
//     P[2a]     = u'[2a]     + sum_b gE[(a - b) mod 64] u'[2b + 1]        block 0: even outputs from odd inputs
//     P[2a + 1] = u'[2a + 1] + sum_b gO[(a - b) mod 64] u'[2b]            block 1: odd outputs from even inputs
// Each 64 x 64 product: three "pre" levels (an axis of 4 becomes an axis of 7: sums of inputs with coefficients 1, 2, 4, 8 and
// signs), 343 multiplications by fixed scalars, three "post" levels (7 back to 4).  Every level is one launch with one LANE per
// (blob, element) and the evaluation point / output index uniform per workgroup (blockIdx.y), so a wavefront runs one straight-line
// program of at most ten group operations.  Point rows between levels: [entry][blob] (a wavefront's lanes read adjacent rows).
// FULL_WAVES: every wavefront of these kernels runs with all 64 lanes active, lanes beyond the work repeating another lane's and
// storing nothing.  Measured on gfx950 (tools/ubench_lanes.hip, profiles/r02_ubench_lanes.txt): a dependent chain issues one
// instruction per ~5.9 cycles with more than 32 active lanes in the wavefront but per ~7.9 with 32 or fewer, and once the whole chip
// holds fewer than ~8200 active lanes the figure grows with the number of resident wavefronts -- 19.6 cycles at one sparse wavefront
// per SIMD.  A single-blob call ran this stage's 686 one-lane wavefronts at 4.7 ms; with full wavefronts it takes the 1.7 ms of a batch.
// ------------------------------------------------------------------------------------------
// QUAD: calls of up to FK_QUAD_MAX_BATCH blobs run every item on four lanes (g1_quad.hpp xadd_quad / xdbl_quad: an addition four
// products deep instead of fourteen) -- the six network launches are chains of ~10 group operations that a single blob's caller waits for
template <bool QUAD>
__device__ __forceinline__ void toom_run(G1X& acc, const uint8_t* __restrict__ prog, const uint32_t* const* __restrict__ src, const QuadRole& qr) {
    // src[x]: row of operand x (4 or 7 of them)
#pragma unroll 1
    for (int i = 0; i < FK20_PROG_LEN; i++) {
        const int code = prog[i], op = code & 15, x = code >> 4;
        if (op == 7) break;
        if (op == 0) {
            f_load_xyzz(acc, src[x]);
        } else if (op == 3) {
            if constexpr (QUAD) xdbl_quad(acc, qr);
            else g1x_dbl(acc, acc);
        } else if (op == 4) {
            g1x_neg(acc, acc);
        } else {
            G1X t;
            f_load_xyzz(t, src[x]);
            if (op == 2) g1x_neg(t, t);
            if constexpr (QUAD) xadd_quad(acc, t, qr);
            else g1x_add(acc, acc, t);
        }
    }
}
// pre level: out[blk][p][r][q] = sum_j e_r[j] in[blk][p][j][q]   (r = blockIdx.y).  FIRST: in = u' rows [b][128], block blk reads the
// inputs of the other parity, (p, j, q) = (0, j, q) -> element 16 j + q.
template <bool FIRST, bool QUAD>
__global__ void __launch_bounds__(64) k_fk20_pre(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const int* __restrict__ status, int n, int npre, int npost) {
    const int r = blockIdx.y;
    const QuadRole qr(threadIdx.x);
    const size_t glane = ((size_t)blockIdx.x * 64 + threadIdx.x) / (QUAD ? 4 : 1), total = (size_t)n * 2 * npre * npost;
    const bool live = glane < total && (!QUAD || (threadIdx.x & 3) == 0);
    const size_t lane = glane % total;                      // surplus lanes of the last wavefront shadow other lanes (FULL_WAVES below)
    const int b = (int)(lane % n);
    size_t item = lane / n;
    const int q = (int)(item % npost);
    item /= npost;
    const int p = (int)(item % npre), blk = (int)(item / npre);
    if (status[b] != 0) return;
    const uint32_t* src[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (FIRST) src[j] = in + ((size_t)b * 128 + 2 * (j * npost + q) + (1 - blk)) * 64;
        else src[j] = in + (((((size_t)blk * npre + p) * 4 + j) * npost + q) * n + b) * 64;
    }
    G1X acc;
    toom_run<QUAD>(acc, FK20_PRE_PROG[r], src, qr);
    if (live) f_store_xyzz(out + (((((size_t)blk * npre + p) * 7 + r) * npost + q) * n + b) * 64, acc);
}
// post level: out[blk][p][A][q] = sum_r e_r[3 - A] in[blk][p][r][q]   (A = blockIdx.y).  LAST: (p, A, q) = (0, A, q) -> element
// a = 16 A + q of the block's output; the result plus u' goes to rows [b][128] at index 2 a + blk.
template <bool LAST, bool QUAD>
__global__ void __launch_bounds__(64) k_fk20_post(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const uint32_t* __restrict__ u, const int* __restrict__ status, int n,
                                                  int npre, int npost) {
    const int A = blockIdx.y;
    const QuadRole qr(threadIdx.x);
    const size_t glane = ((size_t)blockIdx.x * 64 + threadIdx.x) / (QUAD ? 4 : 1), total = (size_t)n * 2 * npre * npost;
    const bool live = glane < total && (!QUAD || (threadIdx.x & 3) == 0);
    const size_t lane = glane % total;
    const int b = (int)(lane % n);
    size_t item = lane / n;
    const int q = (int)(item % npost);
    item /= npost;
    const int p = (int)(item % npre), blk = (int)(item / npre);
    if (status[b] != 0) return;
    const uint32_t* src[7];
#pragma unroll
    for (int r = 0; r < 7; r++) src[r] = in + (((((size_t)blk * npre + p) * 7 + r) * npost + q) * n + b) * 64;
    G1X acc;
    toom_run<QUAD>(acc, FK20_POST_PROG[A], src, qr);
    if (LAST) {
        const size_t j = (size_t)b * 128 + 2 * (A * npost + q) + blk;
        G1X t;
        f_load_xyzz(t, u + j * 64);
        if constexpr (QUAD) xadd_quad(acc, t, qr);
        else g1x_add(acc, acc, t);
        if (live) f_store_xyzz(out + j * 64, acc);
    } else if (live) {
        f_store_xyzz(out + (((((size_t)blk * npre + p) * 4 + A) * npost + q) * n + b) * 64, acc);
    }
}
// the 686 products: one WAVEFRONT per fixed scalar (blockIdx.x), one lane per blob -- uniform control flow through the digit string
// G lanes per blob: 4 for calls of up to FK_QUAD_MAX_BATCH blobs, 2 up to FK_PAIR_MAX_BATCH, else 1 -- see f_mul_wnaf_group
template <int G>
__global__ void __launch_bounds__(64) k_fk20_toom_mul(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const int* __restrict__ status, int n) {
    const int glane = (blockIdx.y * 64 + threadIdx.x) / G;
    const bool live = glane < n && (threadIdx.x & (G - 1)) == 0;
    const int b = glane < n ? glane : glane % n;       // FULL_WAVES: surplus lanes repeat a blob and store nothing
    if (status[b] != 0) return;
    const size_t row = ((size_t)blockIdx.x * n + b) * 64;
    G1X p;
    f_load_xyzz(p, in + row);
    if (G > 1) f_mul_wnaf_group<G>(p, p, FK20_TOOM_WNAF[blockIdx.x], threadIdx.x);
    else f_mul_wnaf(p, p, FK20_TOOM_WNAF[blockIdx.x]);
    if (live) f_store_xyzz(out + row, p);
}
// The same products in SEGMENTS, for calls of 65 to 192 blobs.  A product is a 1.76 ms chain that saturates its SIMD (the field
// multiplications give one wavefront all the independent work the SIMD can issue), so ceil(n / 64) x 686 wavefronts on 1024 SIMDs take
// 1.76 ms times the LARGEST number of wavefronts any SIMD was dealt: 1372 wavefronts (128 blobs) took 3.5 ms, as 2058 (192) and 2744
// (256) took 5.3.  Cut into K segments of equal cost -- the table in segment 0, the digit positions dealt out by the cost of their
// doubling and additions -- the K x units pieces go out as launches of at most 1024 single-wavefront workgroups (one per SIMD),
// segment-major: piece (unit, s) lies units >= 1024 places after (unit, s - 1), hence in a later launch, and the launch boundary is
// the only synchronisation.  The accumulator and the table of a unit live in global memory between launches (fk20_seg_bytes).
constexpr int FK_SEG_ENTRY_Q = 7;                         // uint4 per table entry and lane: an affine point of the co-Z table (28 words)
constexpr int FK_SEG_ZC_Q = 4;                            // the table's common Z (14 words)
constexpr int FK_SEG_ACC_Q = 11;                          // uint4 per accumulator and lane (42 words)
constexpr int FK_SEG_UNIT_Q = 8 * FK_SEG_ENTRY_Q + FK_SEG_ZC_Q + FK_SEG_ACC_Q;
size_t fk20_seg_bytes(int n) { return (size_t)686 * ((n + 63) / 64) * FK_SEG_UNIT_Q * 64 * sizeof(uint4); }
template <int NQ, int NF>
__device__ __forceinline__ void seg_store_fes(uint4* base, int lane, const fp_t* const* f) {      // NF field elements as NQ uint4, stride 64 lanes
    uint32_t w[4 * NQ];
#pragma unroll
    for (int k = 0; k < NF; k++)
#pragma unroll
        for (int i = 0; i < 14; i++) w[14 * k + i] = f[k]->l[i];
#pragma unroll
    for (int i = 14 * NF; i < 4 * NQ; i++) w[i] = 0;
#pragma unroll
    for (int q = 0; q < NQ; q++) base[(size_t)q * 64 + lane] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}
template <int NQ, int NF>
__device__ __forceinline__ void seg_load_fes(fp_t* const* f, const uint4* base, int lane) {
    uint32_t w[4 * NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const uint4 v = base[(size_t)q * 64 + lane];
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int k = 0; k < NF; k++)
#pragma unroll
        for (int i = 0; i < 14; i++) f[k]->l[i] = w[14 * k + i];
}
__device__ __forceinline__ void seg_store_acc(uint4* unit, int lane, const G1J& a) {
    uint32_t w[44];
#pragma unroll
    for (int i = 0; i < 14; i++) { w[i] = a.x.l[i]; w[14 + i] = a.y.l[i]; w[28 + i] = a.z.l[i]; }
    w[42] = 0; w[43] = 0;
#pragma unroll
    for (int q = 0; q < FK_SEG_ACC_Q; q++) unit[(size_t)(8 * FK_SEG_ENTRY_Q + FK_SEG_ZC_Q + q) * 64 + lane] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}
__device__ __forceinline__ void seg_load_acc(G1J& a, const uint4* unit, int lane) {
    uint32_t w[44];
#pragma unroll
    for (int q = 0; q < FK_SEG_ACC_Q; q++) {
        const uint4 v = unit[(size_t)(8 * FK_SEG_ENTRY_Q + FK_SEG_ZC_Q + q) * 64 + lane];
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < 14; i++) { a.x.l[i] = w[i]; a.y.l[i] = w[14 + i]; a.z.l[i] = w[28 + i]; }
}
// costs in field multiplications (round 5, the co-Z form of g1.hpp mul_wnaf_coz): a doubling 7, a mixed addition 10, the table (one doubling, seven co-Z
// additions, seven rescalings) 90
constexpr int FK_SEG_COST_DBL = 7, FK_SEG_COST_ADD = 10, FK_SEG_COST_TABLE = 90;
__global__ void __launch_bounds__(64) k_fk20_toom_seg(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const int* __restrict__ status, int n, int units, int K,
                                                       int base, uint4* __restrict__ seg) {
    const int idx = base + (int)blockIdx.x;
    const int sg = idx / units, u = idx - sg * units;
    const int t = u % 686, y = u / 686;
    const int lane = threadIdx.x;
    const int glane = y * 64 + lane;
    const bool live = glane < n;
    const int b = live ? glane : glane % n;                // FULL_WAVES: surplus lanes repeat a blob and store nothing
    if (status[b] != 0) return;
    const uint32_t* __restrict__ row = FK20_TOOM_WNAF[t];
    uint4* unit = seg + (size_t)u * FK_SEG_UNIT_Q * 64;
    const size_t prow = ((size_t)t * n + b) * 64;
    // the scalar's total cost (uniform: it depends on the digit string only)
    int total = FK_SEG_COST_TABLE;
    {
        bool st = false;
        for (int i = 130; i >= 0; i--) {
            const uint32_t code = (row[i >> 1] >> (16 * (i & 1))) & 0xffffu;
            if (st) total += FK_SEG_COST_DBL;
            if (code & 0xffu) { total += FK_SEG_COST_ADD; st = true; }
            if (code >> 8) { total += FK_SEG_COST_ADD; st = true; }
        }
    }
    G1J acc;
    if (sg == 0) {
        G1X p;
        f_load_xyzz(p, in + prow);
        fp_t tx[8], ty[8], zc;
        G1D::coz_odd_multiples(tx, ty, zc, p);
#pragma unroll 1
        for (int j = 0; j < 8; j++) {
            const fp_t* e[2] = {&tx[j], &ty[j]};
            seg_store_fes<FK_SEG_ENTRY_Q, 2>(unit + (size_t)j * FK_SEG_ENTRY_Q * 64, lane, e);
        }
        const fp_t* z[1] = {&zc};
        seg_store_fes<FK_SEG_ZC_Q, 1>(unit + (size_t)8 * FK_SEG_ENTRY_Q * 64, lane, z);
        Fp::set_zero(acc.x);
        Fp::set_zero(acc.y);
        Fp::set_zero(acc.z);
    } else {
        seg_load_acc(acc, unit, lane);
    }
    fp_t beta;
    Fp::set(beta, FP_BETA);
    bool started = false;
    int cum = FK_SEG_COST_TABLE;
#pragma unroll 1
    for (int i = 130; i >= 0; i--) {
        const uint32_t code = (row[i >> 1] >> (16 * (i & 1))) & 0xffffu;
        // a position belongs to the segment its MIDDLE falls into (the heaviest segment of any scalar: 1.05 x the mean for K = 4, 1.09 x
        // when dealt by the position's start; tests/test_fk20_segments.py restates the rule for all 686 scalars)
        const int pc = (started ? FK_SEG_COST_DBL : 0) + FK_SEG_COST_ADD * (((code & 0xffu) != 0) + ((code >> 8) != 0));
        const int mine = min(K - 1, (cum + pc / 2) * K / total) - sg;        // < 0: an earlier segment's position, > 0: a later one's
        if (mine > 0) break;
        if (started) cum += FK_SEG_COST_DBL;
        if (started && mine == 0) G1D::jdbl(acc, acc);
        if (code == 0) continue;
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            const int d = (int)(int8_t)(half ? code >> 8 : code & 0xffu);
            if (d) {
                cum += FK_SEG_COST_ADD;
                const bool was_started = started;
                started = true;
                if (mine == 0) {
                    fp_t qx, qy;
                    fp_t* e[2] = {&qx, &qy};
                    seg_load_fes<FK_SEG_ENTRY_Q, 2>(e, unit + (size_t)(((d < 0 ? -d : d) - 1) >> 1) * FK_SEG_ENTRY_Q * 64, lane);
                    // second half: -phi(.) = (beta X, -Y); a negative digit flips the sign once more
                    if (half) Fp::mul(qx, qx, beta);
                    if ((d < 0) != (half != 0)) Fp::template neg<2>(qy, qy);     // 4p - y (y < 3.01p)
                    if (was_started) {
                        G1D::jmadd_inc(acc, acc, qx, qy);
                    } else {                                                      // the chain's first addition: the operand itself
                        acc.x = qx;
                        acc.y = qy;
                        Fp::set_one(acc.z);
                    }
                }
            }
        }
    }
    if (sg == K - 1) {
        G1X r;
        fp_t zc;
        fp_t* z[1] = {&zc};
        seg_load_fes<FK_SEG_ZC_Q, 1>(z, unit + (size_t)8 * FK_SEG_ENTRY_Q * 64, lane);
        Fp::mul(acc.z, acc.z, zc);                                                // back from the table's isomorphic curve: (X', Y', Z' Z)
        G1D::jac_to_xyzz(r, acc);
        if (live) f_store_xyzz(out + prow, r);
    } else {
        seg_store_acc(unit, lane, acc);
    }
}

// proofs[i] = X[brev7(i)] (fk20.go:90), affine, compressed
__global__ void __launch_bounds__(64) k_fk20_finish(const uint32_t* __restrict__ x, const int* __restrict__ status, uint8_t* __restrict__ proofs48) {
    const int b = blockIdx.y, i = blockIdx.x * 64 + threadIdx.x;
    if (status[b] != 0) return;
    G1X p;
    f_load_xyzz(p, x + ((size_t)b * 128 + (int)(__brev((uint32_t)i) >> 25)) * 64);
    G1Aff a;
    f_to_affine(a, p);
    uint32_t w[12];
    g1_compress_words(w, a);
    uint32_t* o32 = reinterpret_cast<uint32_t*>(proofs48 + ((size_t)b * 128 + i) * 48);
#pragma unroll
    for (int q = 0; q < 12; q++) o32[q] = __builtin_bswap32(w[q]);
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
void fk20_setup_device(const uint32_t* d_monomial_aff, const uint32_t* d_g1tw_fwd, uint32_t* d_t_aff, uint32_t* d_table, int cbits, hipStream_t s) {
    hipLaunchKernelGGL(k_fk20_setup_fft, dim3(64), dim3(64), 0, s, d_monomial_aff, d_g1tw_fwd, d_t_aff);
    const int nwin = fk20_nwin(cbits);
    const size_t n = (size_t)8192 * nwin;
    hipLaunchKernelGGL(k_fk20_setup_table, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, d_t_aff, d_table, 8192, cbits, nwin);
}
template <int C, int LW, int LP, int WPL, int PAD>
static void launch_fk_msm(const uint32_t* d_scalars, const uint32_t* d_table, int n, const int* d_status, uint32_t* d_parts, uint32_t* d_results, hipStream_t s) {
    constexpr int LANES = LW * LP, PER = 64 / LANES;       // MSMs per wavefront
    const dim3 grid((128 + PER - 1) / PER, n);
    hipLaunchKernelGGL((k_fk20_msm<C, LW, LP, WPL>), grid, dim3(64), 0, s, d_scalars, d_table, d_status, d_parts);
    hipLaunchKernelGGL((k_fk20_msm_reduce<LANES, PAD>), dim3((128 + 64 / PAD - 1) / (64 / PAD), n), dim3(64), 0, s, d_parts, d_status, d_results);
}
void fk20_msm_device(const uint32_t* d_scalars, const uint32_t* d_table, int cbits, int n, const int* d_status, uint32_t* d_parts, uint32_t* d_results, hipStream_t s) {
    if (n <= 0) return;
    // latency over occupancy: the more lanes an MSM gets the shorter each lane's chain, as long as every wavefront still finds a SIMD
    // of its own (a block carries 6-48 blobs: 48 lanes x 32 additions up to 8 blobs -- the points of a sum split in two halves as well
    // as its windows --, 24 x 64 up to 16, 12 x 128 up to 39, 8 x 192 from there on)
    const bool few = n <= FK_DIRECT_MAX_BATCH;
    if (cbits == 13) {
        if (few) launch_fk_msm<13, 10, 2, 2, 32>(d_scalars, d_table, n, d_status, d_parts, d_results, s);
        else launch_fk_msm<13, 4, 2, 5, 8>(d_scalars, d_table, n, d_status, d_parts, d_results, s);
    } else if (cbits == 12) {
        // 22 windows (47 GB): batches with 2 lanes of 11 windows x 4 point groups of 16 per MSM -- 176 additions per lane against the
        // 192 of the 11-bit table (measured: 1.90 against 2.01 ms per 64 blobs, 18.7 k against 18.0 k blobs/s from eight callers);
        // the few-blob shapes mirror the 11-bit ones with 22 lanes where those have 24
        if (n <= FK_SPLIT_MAX_BATCH) launch_fk_msm<12, 22, 2, 1, 64>(d_scalars, d_table, n, d_status, d_parts, d_results, s);
        else if (few) launch_fk_msm<12, 22, 1, 1, 32>(d_scalars, d_table, n, d_status, d_parts, d_results, s);
        else if (n <= FK_MID_MAX_BATCH) launch_fk_msm<12, 11, 1, 2, 16>(d_scalars, d_table, n, d_status, d_parts, d_results, s);
        else launch_fk_msm<12, 2, 4, 11, 8>(d_scalars, d_table, n, d_status, d_parts, d_results, s);
    } else if (cbits == 8) {
        // the 4.3 GB table the context degrades to when the HBM is short (api_cells.hip ensure_fk20_tables): 32 windows per scalar
        if (n <= FK_MID_MAX_BATCH) launch_fk_msm<8, 16, 1, 2, 16>(d_scalars, d_table, n, d_status, d_parts, d_results, s);
        else launch_fk_msm<8, 8, 1, 4, 8>(d_scalars, d_table, n, d_status, d_parts, d_results, s);
    } else {
        if (n <= FK_SPLIT_MAX_BATCH) launch_fk_msm<11, 24, 2, 1, 64>(d_scalars, d_table, n, d_status, d_parts, d_results, s);
        else if (few) launch_fk_msm<11, 24, 1, 1, 32>(d_scalars, d_table, n, d_status, d_parts, d_results, s);
        else if (n <= FK_MID_MAX_BATCH) launch_fk_msm<11, 12, 1, 2, 16>(d_scalars, d_table, n, d_status, d_parts, d_results, s);
        else launch_fk_msm<11, 8, 1, 3, 8>(d_scalars, d_table, n, d_status, d_parts, d_results, s);
    }
}
// MSM results u' [n][128] XYZZ rows in d_results -> proofs [n][128][48 B].  d_tmp: n x 128 rows (the transformed points before
// compression); d_parts: the two buffers of the network, 2 x 686 x n rows (fk20_parts_rows() per blob).
int fk20_seg_count(int n) {                 // segments per product for a call of n blobs; 0: one launch of whole products
#ifdef FK_SEG_OFF
    return 0;
#endif
    const int g = (n + 63) / 64;
    // 65..128 blobs: 2744 pieces -> 3 launches of 915; 129..192: 8232 -> 9 of 915.  *Measured* (tools/sweep_cells.py, whole call):
    // 96 blobs 8.4 -> 7.5 ms, 128: 8.6 -> 7.7, 160: 11.9 -> 10.9, 192: 12.3 -> 11.0; 256 blobs (10976 pieces, 11 launches) 13.9 -> 13.9:
    // eleven launches each as long as its slowest piece give back what the balance gains, so 193 blobs and more run whole products
    return g == 2 ? 2 : g == 3 ? 4 : 0;
}
void fk20_ffts_device(uint32_t* d_results, uint32_t* d_tmp, int n, const int* d_status, uint8_t* d_proofs48, uint32_t* d_parts, void* d_seg, hipStream_t s) {
    if (n <= 0) return;
    uint32_t* bufA = d_parts;
    uint32_t* bufB = d_parts + (size_t)686 * n * 64;
    const bool quad = n <= FK_QUAD_MAX_BATCH;               // the addition networks on a quad of lanes per item, like the products
    auto grid = [&](int npre, int npost, int ny) { return dim3((unsigned)(((size_t)n * 2 * npre * npost * (quad ? 4 : 1) + 63) / 64), (unsigned)ny); };
    if (quad) {
        hipLaunchKernelGGL((k_fk20_pre<true, true>), grid(1, 16, 7), dim3(64), 0, s, d_results, bufA, d_status, n, 1, 16);
        hipLaunchKernelGGL((k_fk20_pre<false, true>), grid(7, 4, 7), dim3(64), 0, s, bufA, bufB, d_status, n, 7, 4);
        hipLaunchKernelGGL((k_fk20_pre<false, true>), grid(49, 1, 7), dim3(64), 0, s, bufB, bufA, d_status, n, 49, 1);
    } else {
        hipLaunchKernelGGL((k_fk20_pre<true, false>), grid(1, 16, 7), dim3(64), 0, s, d_results, bufA, d_status, n, 1, 16);
        hipLaunchKernelGGL((k_fk20_pre<false, false>), grid(7, 4, 7), dim3(64), 0, s, bufA, bufB, d_status, n, 7, 4);
        hipLaunchKernelGGL((k_fk20_pre<false, false>), grid(49, 1, 7), dim3(64), 0, s, bufB, bufA, d_status, n, 49, 1);
    }
    // up to 16 blobs the 686 wavefronts have room for a quad of lanes per blob, up to 32 for a pair (one wavefront per product either
    // way); above that the extra wavefronts would share SIMDs and the shorter chains buy nothing
    if (n <= FK_QUAD_MAX_BATCH) hipLaunchKernelGGL((k_fk20_toom_mul<4>), dim3(686, (4 * n + 63) / 64), dim3(64), 0, s, bufA, bufB, d_status, n);
    else if (n <= FK_PAIR_MAX_BATCH) hipLaunchKernelGGL((k_fk20_toom_mul<2>), dim3(686, (2 * n + 63) / 64), dim3(64), 0, s, bufA, bufB, d_status, n);
    else if (d_seg && fk20_seg_count(n)) {
        const int K = fk20_seg_count(n), units = 686 * ((n + 63) / 64), total = units * K;
        const int rounds = (total + 1023) / 1024, per = (total + rounds - 1) / rounds;       // per <= 1024 <= units: a piece's predecessor is in an earlier launch
        for (int r = 0; r < rounds; r++)
            hipLaunchKernelGGL(k_fk20_toom_seg, dim3((unsigned)std::min(per, total - r * per)), dim3(64), 0, s, bufA, bufB, d_status, n, units, K, r * per, (uint4*)d_seg);
    } else hipLaunchKernelGGL((k_fk20_toom_mul<1>), dim3(686, (n + 63) / 64), dim3(64), 0, s, bufA, bufB, d_status, n);
    if (quad) {
        hipLaunchKernelGGL((k_fk20_post<false, true>), grid(49, 1, 4), dim3(64), 0, s, bufB, bufA, nullptr, d_status, n, 49, 1);
        hipLaunchKernelGGL((k_fk20_post<false, true>), grid(7, 4, 4), dim3(64), 0, s, bufA, bufB, nullptr, d_status, n, 7, 4);
        hipLaunchKernelGGL((k_fk20_post<true, true>), grid(1, 16, 4), dim3(64), 0, s, bufB, d_tmp, d_results, d_status, n, 1, 16);
    } else {
        hipLaunchKernelGGL((k_fk20_post<false, false>), grid(49, 1, 4), dim3(64), 0, s, bufB, bufA, nullptr, d_status, n, 49, 1);
        hipLaunchKernelGGL((k_fk20_post<false, false>), grid(7, 4, 4), dim3(64), 0, s, bufA, bufB, nullptr, d_status, n, 7, 4);
        hipLaunchKernelGGL((k_fk20_post<true, false>), grid(1, 16, 4), dim3(64), 0, s, bufB, d_tmp, d_results, d_status, n, 1, 16);
    }
    hipLaunchKernelGGL(k_fk20_finish, dim3(2, n), dim3(64), 0, s, d_tmp, d_status, d_proofs48);
}

int fk20_nwin(int cbits) { return (256 + cbits - 1) / cbits; }
size_t fk20_table_bytes(int cbits) { return (size_t)8192 * fk20_nwin(cbits) * ((size_t)1 << (cbits - 1)) * 128; }
size_t fk20_parts_rows() { return (size_t)128 * 24; }      // per blob: the MSM's lane partial sums (at most 128 x 24 rows); the transform stage needs 2 x 686 of them

}  // namespace kzg
