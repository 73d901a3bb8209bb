// msm_device.hpp -- device helpers shared by the fixed-base MSM translation units (msm_kernels.hip, commit_pairs_kernels.hip):
// record layouts in HBM, scalar decoding and the signed-digit recoding of the direct window table / bucket method.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "g1.hpp"

namespace kzg {

// ------------------------------------------------------------------------------------------
// record layouts in HBM
//   affine point : 32 words  = x[14] y[14] inf pad[3]            (128 B, one cache line)
//   XYZZ point   : 64 words  = x[14] y[14] zz[14] zzz[14] pad[8] (256 B)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_affine(G1Aff& p, const uint32_t* rec) {
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
__device__ __forceinline__ void store_affine(uint32_t* rec, const G1Aff& p) {
    uint32_t w[32];
#pragma unroll
    for (int i = 0; i < 14; i++) { w[i] = p.x.l[i]; w[14 + i] = p.y.l[i]; }
    w[28] = p.inf; w[29] = 0; w[30] = 0; w[31] = 0;
    uint4* q = reinterpret_cast<uint4*>(rec);
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
__device__ __forceinline__ void load_xyzz(G1X& p, const uint32_t* rec) {
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
__device__ __forceinline__ void store_xyzz(uint32_t* rec, const G1X& p) {
    uint32_t w[56];
#pragma unroll
    for (int i = 0; i < 14; i++) { w[i] = p.x.l[i]; w[14 + i] = p.y.l[i]; w[28 + i] = p.zz.l[i]; w[42 + i] = p.zzz.l[i]; }
    uint4* q = reinterpret_cast<uint4*>(rec);
#pragma unroll
    for (int i = 0; i < 14; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// ------------------------------------------------------------------------------------------
// scalar decoding + signed-digit recoding
// ------------------------------------------------------------------------------------------
// 32 big-endian bytes at p (16-byte aligned) -> 8 little-endian words
__device__ __forceinline__ void load_scalar_be(uint32_t w[8], const uint8_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 hi = q[0], lo = q[1];
    w[7] = __builtin_bswap32(hi.x); w[6] = __builtin_bswap32(hi.y); w[5] = __builtin_bswap32(hi.z); w[4] = __builtin_bswap32(hi.w);
    w[3] = __builtin_bswap32(lo.x); w[2] = __builtin_bswap32(lo.y); w[1] = __builtin_bswap32(lo.z); w[0] = __builtin_bswap32(lo.w);
}
// canonical check of reference serialization.go:141 / internal/utils/utils.go:45-50: value < r
__device__ __forceinline__ bool scalar_is_canonical(const uint32_t w[8]) { return !Fr::words32_geq_mod(w); }

template <int C>
struct Recode {
    static constexpr int NWIN = (255 + C - 1) / C;
    static constexpr int NBUCKET = 1 << (C - 1);
    // digit j of the signed recoding of w (value < 2^255): d in [-2^(C-1), 2^(C-1)], sum d_j 2^(C j) = w
    template <class F>
    static __device__ __forceinline__ void for_each_digit(const uint32_t w[8], F&& f) {
        int carry = 0;
#pragma unroll
        for (int j = 0; j < NWIN; j++) {
            const int bit = C * j, wi = bit >> 5, off = bit & 31;
            uint32_t v = w[wi] >> off;
            if (off + C > 32 && wi + 1 < 8) v |= w[wi + 1] << (32 - off);
            int d = (int)(v & ((1u << C) - 1)) + carry;
            carry = 0;
            if (d > NBUCKET) { d -= (1 << C); carry = 1; }
            f(j, d);
        }
    }
};


#ifndef MKZG_DTA
#define MKZG_DTA 64
#endif
constexpr int DTA = MKZG_DTA; // threads per accumulate workgroup, 8 points per lane (64: single-wave workgroups, any free wave slot takes one)
constexpr int DPARTS = 4096;  // lane accumulators per blob (64-word XYZZ records) of the one-scalar-per-lane shapes: 1 MB per blob (a lane's buffer grows to it on demand, api.hip ensure_lane_direct)
constexpr int DBAD = 2 * 4096 / DTA;   // stride of a blob's bad-scalar flags (one per accumulate workgroup, at most 2 * 4096 / DTA of them: the window-split shape)
static_assert(DTA == 64, "the partial-sum layout (64 parts per accumulate workgroup, folded by one wavefront) assumes single-wave workgroups");

}  // namespace kzg
