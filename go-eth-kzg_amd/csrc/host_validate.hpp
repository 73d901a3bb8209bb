// host_validate.hpp -- product host code: gnark's G1Affine.SetBytes (reference serialization.go:108-115) for ONE or TWO points on the calling
// thread: flag bits, x < p, the square root, the sign bit, and the r-torsion test phi(P) == -[x^2]P through the same G1T template the
// kernels use (g1.hpp in_subgroup), instantiated over HostFp (6 x 64-bit Montgomery, ADX multiplication: ~26 ns).
//
// Why the host (round 6): the single-proof entry points of the reference's API -- VerifyKZGProof, VerifyBlobKZGProof, ComputeBlobKZGProof
// (verify.go:12-83, prove.go:46-77) -- bring one or two points, and their validation is ONE dependent chain of ~1,400 field
// multiplications per point: 0.64 ms on a quad of GPU lanes (k_g1_validate<0>: half of a 1.3 ms VerifyKZGProof call, all of a 0.77 ms
// ComputeBlobKZGProof call's critical path), ~45 us on one host core.  Batches keep the kernels: many points are what a GPU is for.
// The status codes are the kernels' (g1_decode.hpp g1_decode_lane, verify_kernels.hip g1_validate_block): MKZG_ERR_G1_BAD_ENCODING,
// MKZG_ERR_G1_NOT_ON_CURVE, MKZG_ERR_G1_NOT_IN_SUBGROUP, 0; the identity encoding is valid and yields the identity row.
#pragma once
#include <string.h>
#include "../../include/mi355kzg.h"
#include "host_g1.hpp"

namespace kzg {

// beta: the cube root of unity of constants29.h FP_BETA (gen_constants.py), as a HostFp element
inline const fe& g1h_beta() {
    static const fe b = [] {
        static const uint64_t BETA_PLAIN[6] = {0x2e01fffffffefffeull, 0xde17d813620a0002ull, 0xddb3a93be6f89688ull, 0xba69c6076a0f77eaull, 0x5f19672fdf76ce51ull, 0x0000000000000000ull};
        fe v;
        HostFp::from_plain(v, BETA_PLAIN);
        return v;
    }();
    return b;
}
// One compressed point -> its status and (status 0 or MKZG_ERR_G1_NOT_IN_SUBGROUP) the affine record the kernels consume: 32 words, x[14] y[14] in
// the radix-2^29 Montgomery form of mont29.hpp, canonical; inf, pad[3].  Anything but a finite decoded point leaves the identity record.
inline int g1h_validate_row(const uint8_t in48[48], uint32_t row[32], bool subgroup_check = true) {
    memset(row, 0, 128);
    row[28] = 1;
    const uint8_t flags = in48[0] >> 5;                      // compressed, infinity, sign
    if (!(flags & 4) || flags == 7) return MKZG_ERR_G1_BAD_ENCODING;
    if (flags & 2) {
        uint8_t any = in48[0] & 0x1f;
        for (int i = 1; i < 48; i++) any |= in48[i];
        return any ? MKZG_ERR_G1_BAD_ENCODING : MKZG_OK;
    }
    uint8_t xb[48];
    memcpy(xb, in48, 48);
    xb[0] &= 0x1f;
    fe x, y, rhs, four;
    if (!HostFp::from_be48(x, xb)) return MKZG_ERR_G1_BAD_ENCODING;
    HostFp::sqr(rhs, x);
    HostFp::mul(rhs, rhs, x);
    HostFp::set_one(four);
    HostFp::dbl(four, four);
    HostFp::dbl(four, four);
    HostFp::add(rhs, rhs, four);
    if (!HostFp::sqrt(y, rhs)) return MKZG_ERR_G1_NOT_ON_CURVE;
    if (HostFp::lex_largest(y) != ((flags & 1) != 0)) HostFp::neg<0>(y, y);
    // the record: HostFp -> plain words -> the 29-bit-limb Montgomery form (as HostFp::inv goes to the divsteps code and back)
    const fe* src[2] = {&x, &y};
    for (int c = 0; c < 2; c++) {
        uint64_t pl[6];
        HostFp::to_plain(pl, *src[c]);
        uint32_t w[12];
        for (int i = 0; i < 6; i++) { w[2 * i] = (uint32_t)pl[i]; w[2 * i + 1] = (uint32_t)(pl[i] >> 32); }
        fp_t p29, m29, c29;
        Fp::from_words32(p29, w);
        Fp::from_plain(m29, p29);
        Fp::canon(c29, m29);
        for (int i = 0; i < 14; i++) row[14 * c + i] = c29.l[i];
    }
    row[28] = 0;
    if (subgroup_check && !G1H::in_subgroup(x, y, g1h_beta())) return MKZG_ERR_G1_NOT_IN_SUBGROUP;
    return MKZG_OK;
}

}  // namespace kzg
