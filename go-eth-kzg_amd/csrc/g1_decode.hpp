// g1_decode.hpp -- one lane decodes one zcash-compressed G1 point (48 bytes, big-endian) into a canonical affine record.
// Checks of gnark's SetBytes (reference serialization.go:108-115): flag bits, x < p, on curve.  The r-torsion test is separate
// (verify_kernels.hip: g1_in_subgroup_lane) because trusted-setup points are decoded without it (reference trusted_setup.go:104-117).
// Shared by k_g1_decompress (msm_kernels.hip) and the fused k_g1_validate (verify_kernels.hip).
#pragma once
#include "g1.hpp"
#include "../../include/mi355kzg.h"

namespace kzg {

// First half of the decoding, everything that does not need a square root: 0 = a finite candidate (x in Montgomery form, the three
// flag bits returned), -1 = the identity encoding, otherwise MKZG_ERR_G1_BAD_ENCODING.
__device__ __forceinline__ int g1_parse_lane(fp_t& x, uint32_t& flags, const uint8_t* __restrict__ row48) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(row48);
    uint32_t w[12];
#pragma unroll
    for (int q = 0; q < 12; q++) w[11 - q] = __builtin_bswap32(p[q]);
    flags = w[11] >> 29;
    Fp::set_zero(x);
    if (!(flags & 4) || flags == 7) return MKZG_ERR_G1_BAD_ENCODING;
    if (flags & 2) {
        uint32_t any = w[11] & 0x1fffffffu;
#pragma unroll
        for (int q = 0; q < 11; q++) any |= w[q];
        return any ? MKZG_ERR_G1_BAD_ENCODING : -1;
    }
    w[11] &= 0x1fffffffu;
    if (Fp::words32_geq_mod(w)) return MKZG_ERR_G1_BAD_ENCODING;
    fp_t xp;
    Fp::from_words32(xp, w);
    Fp::from_plain(x, xp);
    return 0;
}
// rhs = x^3 + 4
__device__ __forceinline__ void g1_curve_rhs(fp_t& rhs, const fp_t& x) {
    fp_t four;
    Fp::sqr(rhs, x);
    Fp::mul(rhs, rhs, x);
    Fp::set(four, FP_FOUR);
    Fp::add(rhs, rhs, four);
}
// returns 0 or the MKZG_ERR_* of the first failed check; `a` is the identity record whenever the result is not a finite point
__device__ __forceinline__ int g1_decode_lane(G1Aff& a, const uint8_t* __restrict__ row48) {
    fp_t x, rhs, y, chk;
    uint32_t flags;
    Fp::set_zero(a.x);
    Fp::set_zero(a.y);
    a.inf = 1;
    const int pr = g1_parse_lane(x, flags, row48);
    if (pr != 0) return pr < 0 ? 0 : pr;
    g1_curve_rhs(rhs, x);
    Fp::pow_words(y, rhs, FpParams::EXP_SQRT, FpParams::BITS);
    Fp::sqr(chk, y);
    if (!Fp::eq_mod(chk, rhs)) return MKZG_ERR_G1_NOT_ON_CURVE;
    fp_t yc, ypl;
    uint32_t yw[12];
    Fp::canon(yc, y);
    Fp::to_plain(ypl, yc);
    Fp::to_words32(yw, ypl);
    bool gt = false, lt = false;
#pragma unroll
    for (int j = 11; j >= 0; j--) {
        if (!gt && !lt) {
            if (yw[j] > FpParams::HALF[j]) gt = true;
            else if (yw[j] < FpParams::HALF[j]) lt = true;
        }
    }
    if (gt != ((flags & 1) != 0)) { Fp::template neg<0>(yc, yc); Fp::reduce_once(yc, yc); }
    Fp::canon(a.x, x);
    a.y = yc;
    a.inf = 0;
    return 0;
}

}  // namespace kzg
