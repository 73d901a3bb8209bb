// verify_kernels.hip -- device side of the EIP-4844 proof paths on MI355X (gfx950):
//   VerifyBlobKZGProofBatch  reference verify.go:88-143 + internal/kzg/kzg_verify.go:111-202
//   ComputeBlobKZGProof / ComputeKZGProof   reference prove.go:46-111 + internal/kzg/kzg_prove.go
//
//   k_challenge_sha256   compute_challenge (fiatshamir.go:22-33): SHA-256 of the 131,152-byte transcript of every blob
//                        (inherently serial per blob; blobs are the parallel axis), digest reduced mod r.  Two
//                        producer wavefronts (message schedule) + two consumer wavefronts (rounds, a PAIR of
//                        lanes per blob) per 64 blobs; k_challenge_sha256_jobs: the same for the blobs of several calls at once.
//   k_blob_eval          DeserializeBlob + EvaluateLagrangePolynomial (serialization.go:134-146,
//                        internal/domain/domain.go:193-235): one workgroup per blob, barycentric
//                        formula over the bit-reversed 4096-domain, batch inversion per lane.
//   k_blob_quotient      computeQuotientPoly (kzg_prove.go:53-180), both the off-domain and the
//                        on-domain branch; output feeds the fixed-base MSM of msm_kernels.hip.
//   k_g1_validate        gnark's SetBytes for the verifiers' points (serialization.go:108-115): decoding
//                        (g1_decode.hpp) and the r-torsion test as phi(P) == -[x^2]P (Scott 2021) without the y coordinate, a QUAD of
//                        lanes per point (g1_quad.hpp) or one lane per point for calls that share the chip.
//   k_rlc_scalars        utils.ComputePowers + the scalar folds of kzg_verify.go:141-179.
//   k_vmsm_window        variable-base Pippenger, one workgroup per (MSM, 8-bit window): LDS
//                        counting sort of Booth digits, one lane per bucket, wavefront-level
//                        weighted bucket reduction; the per-window sums go to the host, which
//                        does the 255 serial doublings of the Horner combination.
// Integer modular arithmetic only -- no MFMA.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include <stdlib.h>
#include "g1.hpp"
#include "g1_quad.hpp"
#include "verify_kernels.h"
#include "g1_decode.hpp"
#include "pub_ticket.hpp"

namespace kzg {

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_affine_v(G1Aff& p, const uint32_t* rec) {
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
__device__ __noinline__ void g1x_add_nv(G1X& r, const G1X& a, const G1X& b) { g1x_add(r, a, b); }
__device__ __noinline__ void g1x_dbl_nv(G1X& r, const G1X& a) { g1x_dbl(r, a); }
__device__ __noinline__ void g1x_madd_nv(G1X& acc, const fp_t& x, const fp_t& y) { g1x_madd(acc, x, y); }

// Fr table rows are padded to 12 words (48 B) for 16-byte loads
__device__ __forceinline__ void load_fr12(fr_t& r, const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1], c = q[2];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = c.x;
}
__device__ __forceinline__ void store_fr12(uint32_t* p, const fr_t& r) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(r.l[0], r.l[1], r.l[2], r.l[3]);
    q[1] = make_uint4(r.l[4], r.l[5], r.l[6], r.l[7]);
    q[2] = make_uint4(r.l[8], 0, 0, 0);
}
__device__ __forceinline__ void load_be32_words(uint32_t w[8], const uint8_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 hi = q[0], lo = q[1];
    w[7] = __builtin_bswap32(hi.x); w[6] = __builtin_bswap32(hi.y); w[5] = __builtin_bswap32(hi.z); w[4] = __builtin_bswap32(hi.w);
    w[3] = __builtin_bswap32(lo.x); w[2] = __builtin_bswap32(lo.y); w[1] = __builtin_bswap32(lo.z); w[0] = __builtin_bswap32(lo.w);
}
// canonical words (value < r) -> Montgomery
__device__ __forceinline__ void fr_from_words(fr_t& r, const uint32_t w[8]) {
    fr_t pl;
    Fr::from_words32(pl, w);
    fr_t t;
    Fr::from_plain(t, pl);
    Fr::reduce_once(r, t);
}
__device__ __forceinline__ void fr_mul_n(fr_t& r, const fr_t& a, const fr_t& b) { FrC::mul(r, a, b); }
__device__ __noinline__ void fr_inv_n(fr_t& r, const fr_t& a) {
    fr_t t;
    Fr::inv(t, a);             // divsteps (modinv30.hpp)
    Fr::reduce_once(r, t);
}

// ------------------------------------------------------------------------------------------
// SHA-256 (FIPS 180-4), one lane per message
// ------------------------------------------------------------------------------------------
__device__ __constant__ uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe,
    0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7,
    0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b,
    0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
// gfx950 has a 3-input bitwise op (v_bitop3_b32, truth table as immediate): one instruction for the
// three-way XORs of the sigma functions, for Ch and for Maj
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t sha_ch(uint32_t e, uint32_t f, uint32_t g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xca); }    // e ? f : g
__device__ __forceinline__ uint32_t sha_maj(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xe8); }
__device__ __forceinline__ void load16_be(uint32_t* w, const uint8_t* p) {   // 16 bytes -> 4 big-endian words
    uint4 v = *reinterpret_cast<const uint4*>(p);
    w[0] = __builtin_bswap32(v.x); w[1] = __builtin_bswap32(v.y); w[2] = __builtin_bswap32(v.z); w[3] = __builtin_bswap32(v.w);
}

// z_b = SHA256("FSBLOBVERIFY_V1_" || u128be(4096) || blob_b || commitment_b) mod r   (Montgomery rows)
//
// A transcript is a serial chain of 2050 compressions, so the only parallel axis is the blob: one LANE per blob.  What
// can still be taken off that chain is everything that does not depend on the hash state: loading, byte swapping and the
// message schedule (more than half of the instructions).  Each workgroup is therefore three wavefronts on three SIMDs:
// waves 0 and 1 PRODUCE W[i] + K[i] of the even / odd message blocks of their 64 blobs into a four-deep LDS ring while
// wave 2 CONSUMES the two blocks of the previous step (64 rounds each, nothing else); one barrier per two message blocks.
// 1412 -> ~640 instructions per block on the critical wave.
__device__ __forceinline__ void sha_load_block(uint32_t w[16], int k, const uint8_t* __restrict__ blob, const uint8_t* __restrict__ cm) {
    if (k >= 1 && k < 2048) {                       // blob[64k-32 .. 64k+32)
        const uint8_t* p = blob + 64 * k - 32;
        load16_be(w, p);
        load16_be(w + 4, p + 16);
        load16_be(w + 8, p + 32);
        load16_be(w + 12, p + 48);
    } else if (k == 0) {                            // domain separator, degree, blob[0..32)
        w[0] = 0x4653424c; w[1] = 0x4f425645; w[2] = 0x52494659; w[3] = 0x5f56315f;   // "FSBLOBVERIFY_V1_"
        w[4] = 0; w[5] = 0; w[6] = 0; w[7] = 4096;
        load16_be(w + 8, blob);
        load16_be(w + 12, blob + 16);
    } else if (k == 2048) {                         // blob[131040..131072) || commitment[0..32)
        load16_be(w, blob + 131040);
        load16_be(w + 4, blob + 131056);
        load16_be(w + 8, cm);
        load16_be(w + 12, cm + 16);
    } else {                                        // commitment[32..48) || 0x80 || zeros || bit length (131152 * 8)
        load16_be(w, cm + 32);
        w[4] = 0x80000000u;
#pragma unroll
        for (int i = 5; i < 15; i++) w[i] = 0;
        w[15] = 131152u * 8u;
    }
}
constexpr int SHA_NBLOCKS = 2050;
constexpr int EVT_ROWS = 12;       // what k_blob_eval_tree needs of z alone, 12-word rows: z^(2^0..2^11)
// the rows of one challenge (one lane per blob): eleven squarings
__device__ __forceinline__ void eval_prep_rows(const fr_t& z, uint32_t* __restrict__ rows, bool store) {
    fr_t zl = z;
#pragma unroll 1
    for (int l = 0; l < 12; l++) {
        if (store) store_fr12(rows + l * 12, zl);
        if (l < 11) FrC::mul(zl, zl, zl);
    }
}
// The consumer splits every round over a PAIR of lanes, mirror images within a row of 16 (lane i and lane 15 - i): the "E" lane
// (lanes 0..7 of the row: DPP banks 0, 1) holds (e, f, g, h), the "A" lane (banks 2, 3) holds (a, b, c, d), and both execute the SAME
// instructions: per-lane rotation amounts give S1(e) / S0(a), and Maj(a,b,c) = Ch(a ^ c, b, c) shares the Ch instruction.
// The A lane runs TWO ROUNDS BEHIND the E lane:
//     E, round q:      e' = S1(e) + Ch(e,f,g) + [h + W_q + K_q + d_q]      d_q = a_(q-3): the A lane's current "a" one step ago
//     A, round q - 2:  a' = S0(a) + Maj(a,b,c) + [T1]                      T1 = e_(q-1) - d: the E lane's current "e" one step ago
// so the bracketed addends are ready before the step starts and one three-input add ends the round for both lanes: three
// dependent instructions per round (rotate, xor3, add3) and NINE issued, against 14 issued in one lane before -- this wavefront's
// instruction stream IS the kernel's duration.  The addend of the next step is formed during the current one:
//     v_xad_u32   hw', R2, m_a, wk'      E: h' + W + K (next h is this g)          A: ~c + 1 = -d'   (the A lanes read W + K = 1)
//     DPP add     hw' += R0 of the mirror lane      E: the A lane's a = its d three rounds on     A: the E lane's e
// (with the two-round skew both lanes want the partner's R0, hence ONE row_mirror addition).  The feed-forward at the end of a
// block happens two steps apart for the two lanes (bank-masked additions), with the additive corrections the skew asks for;
// tools/sha_pair_model.py is the lane-level model of this exact instruction sequence, checked against hashlib.
// 32 blobs per consumer wavefront, so a workgroup is two producers + two consumers.
#define SHA_DPP_E " row_mask:0xf bank_mask:0x3\n\t"
#define SHA_DPP_A " row_mask:0xf bank_mask:0xc\n\t"
#define SHA_DPP_ALL " row_mask:0xf bank_mask:0xf\n\t"
#define SHA_ID " quad_perm:[0,1,2,3]"
#ifndef SHA_EXP
#define SHA_EXP 0
#endif
#if SHA_EXP == 1      /* diagnostic (tools/ubench_sha.py): one instruction fewer, wrong digests */
#define SHA_X1(R2, HN, WN) ""
#else
#define SHA_X1(R2, HN, WN) "v_xad_u32 %[" HN "], %[" R2 "], %[ma], %[" WN "]\n\t"
#endif
#if SHA_EXP == 2      /* diagnostic: one independent instruction more */
#define SHA_X2 "v_mov_b32 %[tb], %[tc]\n\t"
#else
#define SHA_X2 ""
#endif
#define SHA_STEP(R0, R1, R2, T, HC, HN, WN)                                              \
    "v_alignbit_b32 %[ta], %[" R0 "], %[" R0 "], %[s1]\n\t"                              \
    "v_alignbit_b32 %[tb], %[" R0 "], %[" R0 "], %[s2]\n\t"                              \
    "v_alignbit_b32 %[tc], %[" R0 "], %[" R0 "], %[s3]\n\t"                              \
    "v_bitop3_b32 %[tx], %[" R0 "], %[" R2 "], %[ma] bitop3:0x78\n\t"                    \
    "v_bitop3_b32 %[ta], %[ta], %[tb], %[tc] bitop3:0x96\n\t"                            \
    "v_bitop3_b32 %[tx], %[tx], %[" R1 "], %[" R2 "] bitop3:0xca\n\t"                    \
    SHA_X1(R2, HN, WN) SHA_X2                                                            \
    "v_add3_u32 %[" T "], %[tx], %[" HC "], %[ta]\n\t"                                   \
    "v_add_u32_dpp %[" HN "], %[" R0 "], %[" HN "] row_mirror" SHA_DPP_ALL
// after step 0 of a block (next addend in hY): E: its d of round 1 is the A lane's a + old H2; A: the E lane's e was fed forward
#define SHA_CORR0                                                                        \
    "v_add_u32_dpp %[hY], %[hs2], %[hY] row_mirror" SHA_DPP_E                            \
    "v_add_u32_dpp %[hY], %[ho], %[hY] row_mirror" SHA_DPP_A   /* ho = -(old H4); a mirrored operand of v_subrev is NOT permuted on gfx950 */
// after step 1 of a block (state now in p2, p3, p0, p1; next addend in hX): the A lanes have finished the previous block
#define SHA_BOUNDARY_A                                                                   \
    "v_add_u32_dpp %[hs2], %[p0], %[hs2]" SHA_ID SHA_DPP_A                               \
    "v_add_u32_dpp %[hs0], %[p2], %[hs0]" SHA_ID SHA_DPP_A                               \
    "v_add_u32_dpp %[hX], %[hs1], %[hX] row_mirror" SHA_DPP_E  /* E: d of round 2 = the A lane's a + old H1 */ \
    "v_subrev_u32_dpp %[hX], %[hs3], %[hX]" SHA_ID SHA_DPP_A   /* A: -d of the new block = -(c + old H3) */ \
    "v_add_u32_dpp %[hs1], %[p3], %[hs1]" SHA_ID SHA_DPP_A                               \
    "v_add_u32_dpp %[hs3], %[p1], %[hs3]" SHA_ID SHA_DPP_A                               \
    "v_mov_b32_dpp %[p2], %[hs0]" SHA_ID SHA_DPP_A                                       \
    "v_mov_b32_dpp %[p0], %[hs2]" SHA_ID SHA_DPP_A                                       \
    "v_mov_b32_dpp %[p3], %[hs1]" SHA_ID SHA_DPP_A                                       \
    "v_mov_b32_dpp %[p1], %[hs3]" SHA_ID SHA_DPP_A
// after the 64th step of a block: the E lanes feed forward (minus their old H4 is kept for SHA_CORR0)
#define SHA_END_BODY                                                                     \
    "v_sub_u32 %[ho], 0, %[hs0]\n\t"                                                     \
    "v_add_u32_dpp %[hs1], %[p1], %[hs1]" SHA_ID SHA_DPP_E                               \
    "v_add_u32_dpp %[hs0], %[p0], %[hs0]" SHA_ID SHA_DPP_E                               \
    "v_add_u32_dpp %[hs2], %[p2], %[hs2]" SHA_ID SHA_DPP_E                               \
    "v_add_u32_dpp %[hs3], %[p3], %[hs3]" SHA_ID SHA_DPP_E                               \
    "v_mov_b32_dpp %[p1], %[hs1]" SHA_ID SHA_DPP_E                                       \
    "v_mov_b32_dpp %[p0], %[hs0]" SHA_ID SHA_DPP_E                                       \
    "v_mov_b32_dpp %[p2], %[hs2]" SHA_ID SHA_DPP_E                                       \
    "v_mov_b32_dpp %[p3], %[hs3]" SHA_ID SHA_DPP_E
// the E lanes' addend of round 0: h + W_0 + K_0 + (the A lane's b + its old H3 = d of the new block)
#define SHA_INIT_E                                                                       \
    "s_nop 1\n\t"                                                                        \
    "v_mov_b32_dpp %[hX], %[p3]" SHA_ID SHA_DPP_E                                        \
    "v_add_u32_dpp %[hX], %[w0], %[hX]" SHA_ID SHA_DPP_E                                 \
    "v_add_u32_dpp %[hX], %[p1], %[hX] row_mirror" SHA_DPP_E                             \
    "v_add_u32_dpp %[hX], %[hs3], %[hX] row_mirror" SHA_DPP_E
#define SHA_STATE_OPS [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [hX] "+v"(hX), [hY] "+v"(hY)
#define SHA_TMP_OPS [ta] "=&v"(ta), [tb] "=&v"(tb), [tc] "=&v"(tc), [tx] "=&v"(tx)
#define SHA_HS_OPS [hs0] "+v"(hs0), [hs1] "+v"(hs1), [hs2] "+v"(hs2), [hs3] "+v"(hs3), [ho] "+v"(ho)
#define SHA_CONST_OPS [ma] "v"(m_a), [s1] "v"(r1), [s2] "v"(r2), [s3] "v"(r3)
__device__ __forceinline__ uint32_t sha_from_e_lane(uint32_t x) {   // A lanes: the value of their E lane; E lanes: 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xC, false);   // row_mirror
}
// the body for one workgroup: blobs blk * 64 .. blk * 64 + 63 of one (blobs, commitments, n, z_out) job
// zrows_out (may be null): the evaluation's rows of every challenge (eval_prep_rows) written by the same lanes right behind the digest --
// in the throughput shape this chain then sits in the ROUND's launch, shared by all its calls, instead of in every call's own stream
__device__ __forceinline__ void sha_challenge_block(const uint8_t* __restrict__ blobs, const uint8_t* __restrict__ commitments48, int n, uint32_t* __restrict__ z_out, int blk,
                                                    uint32_t* __restrict__ zrows_out = nullptr) {
    __shared__ uint4 wk[5][16][64];                 // [message block mod 4][round / 4][blob]: four W + K per 16-byte LDS access; [4] stays zero
    // every verifier call waits for this chain and it needs few issue slots: its wavefronts go first on their SIMDs (beside the other
    // kernels of eight calls in flight a launch took 3.8 ms instead of 2.7 without this)
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    if (role < 2) {
        const int b0 = blk * 64 + lane;
        const int b = b0 < n ? b0 : n - 1;              // surplus lanes shadow the last blob (they must keep the barriers)
        const uint8_t* blob = blobs + (size_t)b * 131072;
        const uint8_t* cm = commitments48 + (size_t)b * 48;
        // producers: wave 0 expands the even message blocks, wave 1 the odd ones (the schedule is ~1.5x the instructions of
        // the rounds, so one producer per consumer would be the bottleneck)
        uint32_t wn[16];                                  // this wave's next message block, in flight while the current one is expanded
        sha_load_block(wn, role, blob, cm);
#pragma unroll 1
        for (int j = 0; j <= SHA_NBLOCKS / 2; j++) {
            const int k = 2 * j + role;
            if (k < SHA_NBLOCKS) {
                uint32_t w[16];
#pragma unroll
                for (int i = 0; i < 16; i++) w[i] = wn[i];
                if (k + 2 < SHA_NBLOCKS) sha_load_block(wn, k + 2, blob, cm);
                uint4 (*dst)[64] = wk[k & 3];
                uint32_t q4[4];
#pragma unroll
                for (int i = 0; i < 64; i++) {
                    if (i >= 16) {
                        const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
                        const uint32_t s0 = xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3);
                        const uint32_t s1 = xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
                        w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
                    }
                    q4[i & 3] = w[i & 15] + SHA_K[i];
                    if ((i & 3) == 3) dst[i >> 2][lane] = make_uint4(q4[0], q4[1], q4[2], q4[3]);
                }
            }
            __syncthreads();
        }
        return;
    }
    // consumers: during step j they run the rounds of blocks 2j-2 and 2j-1 (written in step j-1) while the producers fill 2j, 2j+1
    const bool a_side = ((lane >> 3) & 1) != 0;             // A lane: (a, b, c, d); E lane: (e, f, g, h)
    const int col = 32 * (role - 2) + (lane >> 4) * 8 + (a_side ? 15 - (lane & 15) : (lane & 15));   // this pair's blob within the workgroup
    const int b0 = blk * 64 + col;
    const uint32_t m_a = a_side ? 0xffffffffu : 0u;
    const uint32_t r1 = a_side ? 2 : 6, r2 = a_side ? 13 : 11, r3 = a_side ? 22 : 25;
#pragma unroll
    for (int q = 0; q < 16; q++) wk[4][q][lane] = make_uint4(1, 1, 1, 1);      // the A lanes' "W + K": the +1 of the two's complement -d
    // E lanes start at the initial hash value; the A lanes' first two steps are the (void) last rounds of a block before the first
    // one: all-zero state and addends make them no-ops whose feed-forward yields the initial hash value
    uint32_t hs0 = a_side ? 0x6a09e667u : 0x510e527fu, hs1 = a_side ? 0xbb67ae85u : 0x9b05688cu, hs2 = a_side ? 0x3c6ef372u : 0x1f83d9abu,
             hs3 = a_side ? 0xa54ff53au : 0x5be0cd19u;
    uint32_t p0 = a_side ? 0u : hs0, p1 = a_side ? 0u : hs1, p2 = a_side ? 0u : hs2, p3 = a_side ? 0u : hs3;
    uint32_t ho = a_side ? 0u : 0u - hs0, hX = 0, hY = 0, ta, tb, tc, tx;   // minus the E lane's H4 of one block ago
#pragma unroll 1
    for (int j = 0; j <= SHA_NBLOCKS / 2; j++) {
        if (j >= 1) {
            auto load_wk = [&](uint32_t (&wkr)[64], int k) {            // the block's 64 W + K words (A lane: ones)
                const uint4* src = a_side ? &wk[4][0][0] : &wk[k & 3][0][col];
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const uint4 v = src[q * 64];
                    wkr[4 * q] = v.x; wkr[4 * q + 1] = v.y; wkr[4 * q + 2] = v.z; wkr[4 * q + 3] = v.w;
                }
            };
            auto rounds = [&](const uint32_t (&wkr)[64]) {
                asm volatile(SHA_INIT_E
                             SHA_STEP("p0", "p1", "p2", "p3", "hX", "hY", "w1")
                             SHA_CORR0
                             SHA_STEP("p3", "p0", "p1", "p2", "hY", "hX", "w2")
                             SHA_BOUNDARY_A
                             : SHA_STATE_OPS, SHA_HS_OPS, SHA_TMP_OPS
                             : SHA_CONST_OPS, [w0] "v"(wkr[0]), [w1] "v"(wkr[1]), [w2] "v"(wkr[2]));
                asm volatile(SHA_STEP("p2", "p3", "p0", "p1", "hX", "hY", "w3")
                             SHA_STEP("p1", "p2", "p3", "p0", "hY", "hX", "w4")
                             : SHA_STATE_OPS, SHA_TMP_OPS
                             : SHA_CONST_OPS, [w3] "v"(wkr[3]), [w4] "v"(wkr[4]));
#pragma unroll
                for (int g = 1; g < 16; g++) {
                    asm volatile(SHA_STEP("p0", "p1", "p2", "p3", "hX", "hY", "w1")
                                 SHA_STEP("p3", "p0", "p1", "p2", "hY", "hX", "w2")
                                 SHA_STEP("p2", "p3", "p0", "p1", "hX", "hY", "w3")
                                 SHA_STEP("p1", "p2", "p3", "p0", "hY", "hX", "w4")
                                 : SHA_STATE_OPS, SHA_TMP_OPS
                                 : SHA_CONST_OPS, [w1] "v"(wkr[4 * g + 1]), [w2] "v"(wkr[4 * g + 2]), [w3] "v"(wkr[4 * g + 3]), [w4] "v"(wkr[(4 * g + 4) & 63]));
                }
                asm volatile(SHA_END_BODY : SHA_STATE_OPS, SHA_HS_OPS :);
            };
            // both blocks' operands are read up front: one LDS wait per step instead of two (measured with tools/ubench_sha.py: 2.82 ->
            // 2.71 ms per 1024 transcripts beside the call's other kernels, 2.63 -> 2.60 alone on the chip)
            {
                uint32_t wa[64], wb[64];
                load_wk(wa, 2 * j - 2);
                load_wk(wb, 2 * j - 1);
                rounds(wa);
                rounds(wb);
            }
        }
        __syncthreads();
    }
    // the A lanes' last two rounds and feed-forward (the E lanes' registers are scratch by now; their hash words are in hs0..3)
    asm volatile(SHA_STEP("p0", "p1", "p2", "p3", "hX", "hY", "w1")
                 SHA_CORR0
                 SHA_STEP("p3", "p0", "p1", "p2", "hY", "hX", "w1")
                 SHA_BOUNDARY_A
                 : SHA_STATE_OPS, SHA_HS_OPS, SHA_TMP_OPS
                 : SHA_CONST_OPS, [w1] "v"(m_a & 1u));
    // the A lane holds h[0..3], its E lane h[4..7]: every lane fetches its partner's four words, so BOTH lanes of a pair hold the digest --
    // the A lane stores; the E lane runs the same instructions as a shadow when the evaluation's rows follow (a wavefront with 32 or fewer
    // active lanes issues a dependent chain at 7.9 instead of 5.9 cycles per instruction)
    const uint32_t m0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hs0, 0x140, 0xf, 0xf, false), m1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hs1, 0x140, 0xf, 0xf, false),
                   m2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hs2, 0x140, 0xf, 0xf, false), m3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hs3, 0x140, 0xf, 0xf, false);
    if (b0 >= n || (!a_side && !zrows_out)) return;
    const int b = b0;
    const uint32_t h[8] = {a_side ? hs0 : m0, a_side ? hs1 : m1, a_side ? hs2 : m2, a_side ? hs3 : m3, a_side ? m0 : hs0, a_side ? m1 : hs1, a_side ? m2 : hs2, a_side ? m3 : hs3};
    // digest (big-endian h[0..7]) as little-endian words, reduced mod r (2^256 < 3r: at most two subtractions)
    uint32_t v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = h[7 - i];
    for (int rep = 0; rep < 2; rep++) {
        if (Fr::words32_geq_mod(v)) {
            uint64_t br = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                uint64_t t = (uint64_t)v[i] - FrParams::MOD32[i] - br;
                v[i] = (uint32_t)t;
                br = (t >> 32) & 1;
            }
        }
    }
    fr_t z;
    fr_from_words(z, v);
    if (a_side) store_fr12(z_out + (size_t)b * 12, z);
    if (zrows_out) eval_prep_rows(z, zrows_out + (size_t)b * EVT_ROWS * 12, a_side);
}
#ifndef MKZG_SHA_RESERVE
#define MKZG_SHA_RESERVE_STMT
#else
#define MKZG_SHA_RESERVE_STMT asm volatile("" ::: MKZG_SHA_RESERVE);      // A/B aid (round 5): who may share a SIMD with the hash wavefronts, e.g. -DMKZG_SHA_RESERVE='"v175"'
#endif
__global__ void __launch_bounds__(256) k_challenge_sha256(const uint8_t* __restrict__ blobs, const uint8_t* __restrict__ commitments48, int n, uint32_t* __restrict__ z_out,
                                                          uint32_t* __restrict__ zrows_out) {
    MKZG_SHA_RESERVE_STMT
    sha_challenge_block(blobs, commitments48, n, z_out, (int)blockIdx.x, zrows_out);
}
// the challenges of SEVERAL calls in one launch (api_verify.hip, the hash rounds of the throughput shape): the chain is 2.7 ms long
// whatever the number of blobs, and a launch holds its hardware queue for that long -- calls that arrive together share one
__global__ void __launch_bounds__(256) k_challenge_sha256_jobs(ShaJobs jobs) {
    MKZG_SHA_RESERVE_STMT
    int j = 0;
#pragma unroll 1
    while (j + 1 < jobs.count && (int)blockIdx.x >= jobs.first_block[j + 1]) j++;
    sha_challenge_block(jobs.blobs[j], jobs.commitments[j], jobs.n[j], jobs.z_out[j], (int)blockIdx.x - jobs.first_block[j], jobs.zrows_out[j]);
}

// ------------------------------------------------------------------------------------------
// barycentric evaluation / quotient: one 256-thread workgroup per blob, 16 scalars per lane
// ------------------------------------------------------------------------------------------
constexpr int EV_T = 256;          // k_rlc_scalars workgroup
constexpr int EVB_T = 128, EVB_PER = 32;   // k_blob_eval: one wave per blob, 64 scalars per lane -> one inversion per 64 elements

// block-wide sum of one Fr per thread (result valid in thread 0)
template <int BT>
__device__ __forceinline__ void block_sum_fr(fr_t& v, uint32_t* sh /* BT*9 words */) {
    const int t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 9; q++) sh[q * BT + t] = v.l[q];
    __syncthreads();
    for (int s = BT / 2; s >= 1; s >>= 1) {
        if (t < s) {
            fr_t a, b2;
#pragma unroll
            for (int q = 0; q < 9; q++) { a.l[q] = sh[q * BT + t]; b2.l[q] = sh[q * BT + t + s]; }
            FrC::add(a, a, b2);
#pragma unroll
            for (int q = 0; q < 9; q++) sh[q * BT + t] = a.l[q];
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 9; q++) v.l[q] = sh[q * BT];
    __syncthreads();
}

// y_b = f_b(z_b), EvaluateLagrangePolynomial of domain.go:193-235, as the NUMERATOR TREE of the barycentric formula -- no inversion, no
// special case for z in the domain.  With the blob in bit-reversed order the aligned block of 2^L positions j 2^L .. (j + 1) 2^L - 1 is a
// coset g <zeta> of the 2^L-th roots of unity with g^(2^L) = roots_brp[j], so the product of its (z - w_p) is
//     N(L, j) = z^(2^L) - roots_brp[j]                   (and roots_brp[2j + 1] = -roots_brp[2j]).
// The barycentric sum over one block, cleared of its denominators,
//     S(L, j) = sum_{p in block} f_p w_p prod_{q in block, q != p} (z - w_q),
// obeys  S(L, j) = S(L-1, 2j) N(L-1, 2j+1) + S(L-1, 2j+1) N(L-1, 2j) = S_left (Z + c) + S_right (Z - c)   with Z = z^(2^(L-1)), c = roots_brp[2j]:
// TWO products under ONE Montgomery reduction per node (Fr::mul2), and for a pair of leaves  S(1, j) = w [ z (f_0 - f_1) + w (f_0 + f_1) ],
// w = roots_brp[2j]: three products, two reductions.  The root gives the value directly,
//     f(z) = (z^4096 - 1) / 4096 * sum_p f_p w_p / (z - w_p) = S(12, 0) / 4096
// -- a polynomial identity, so z = w_k needs no branch (every term but the k-th vanishes; domain.go:204-207 is the same value).
// 2.5 products and 1.5 reductions per element = 337 multiply-adds, against 3.5 full multiplications = 604 for the inverses run down the
// same tree (round 3: 1 / N(L-1, 2j) = 1 / N(L, j) * N(L-1, 2j+1), then f_p / (z - w_p) per leaf) and an inversion per blob ahead of it.
// The f_p stay PLAIN integers (no conversion: a plain value times a Montgomery one is plain), every S is plain and loose (< 1.1 r), the
// canonical check of the blob's words rides along.  A lane owns the 2^LOG leaves of one level-LOG node (depth first, rolled: every
// level's code exists once), the levels above go through LDS with half of the lanes leaving at each.
// What depends on z alone comes from the hash launch (or a small kernel ahead): the powers z^(2^L), L = 0..11, rows of 12 words.
#ifndef MKZG_EVT_LOG
#define MKZG_EVT_LOG 5
#endif
constexpr int EVT_LOG = MKZG_EVT_LOG;          // leaves per lane = 2^EVT_LOG, lanes per blob = 4096 >> EVT_LOG
__global__ void __launch_bounds__(64) k_blob_eval_prep(const uint32_t* __restrict__ z_in, int n, uint32_t* __restrict__ zrows) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n) return;
    fr_t z;
    load_fr12(z, z_in + (size_t)b * 12);
    eval_prep_rows(z, zrows + (size_t)b * EVT_ROWS * 12, true);
}
// The scalars of the random linear combination that belong to ONE blob (kzg_verify.go:141-179: r^i, r^i z_i, r^i y_i with the global index
// i = exp_offset + b), by the lane that has just finished y_b: in the throughput shape they ride in the evaluation kernel's tail instead of
// a kernel of their own in every call's stream (k_rlc_scalars: one workgroup, 0.19 ms of queue time per call under load).  The Fr sum
// sum r^i y_i is left to the host: n x 8 plain words come back with the statuses.
struct RlcOut { uint32_t* s1; uint32_t* s3; uint32_t* ry; uint32_t exp_offset; RWords r; };
// (inlined: as a called routine its fr_t& arguments travel through scratch memory -- 288 bytes per lane of the whole launch and 27 % more
// HBM-side traffic for the kernel, PMC -- at the same rate: 1.94-1.95 against 1.89-1.92 M blobs/s sustained, within the noise of the box)
__device__ __forceinline__ void rlc_one_blob(const RlcOut& o, int b, const fr_t& z, const fr_t& y) {
    uint32_t rw[8];
#pragma unroll
    for (int i = 0; i < 8; i++) rw[i] = o.r.w[i];
    for (int rep = 0; rep < 2; rep++) {
        if (Fr::words32_geq_mod(rw)) {
            uint64_t br = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) { uint64_t v = (uint64_t)rw[i] - FrParams::MOD32[i] - br; rw[i] = (uint32_t)v; br = (v >> 32) & 1; }
        }
    }
    fr_t base, rp, t, pl;
    fr_from_words(base, rw);
    Fr::set_one(rp);
    Fr::reduce_once(rp, rp);
#pragma unroll 1
    for (uint32_t e = o.exp_offset + (uint32_t)b; e; e >>= 1) {
        if (e & 1) fr_mul_n(rp, rp, base);
        fr_mul_n(base, base, base);
    }
    uint32_t w[8];
    Fr::to_plain(pl, rp);
    Fr::to_words32(w, pl);
#pragma unroll
    for (int q = 0; q < 8; q++) o.s1[(size_t)b * 8 + q] = w[q];
    fr_mul_n(t, rp, z);
    Fr::to_plain(pl, t);
    Fr::to_words32(w, pl);
#pragma unroll
    for (int q = 0; q < 8; q++) o.s3[(size_t)b * 8 + q] = w[q];
    fr_mul_n(t, rp, y);
    Fr::to_plain(pl, t);
    Fr::to_words32(w, pl);
#pragma unroll
    for (int q = 0; q < 8; q++) o.ry[(size_t)b * 8 + q] = w[q];
}
template <int LOG> struct EvtCfg { static constexpr int T = 4096 >> LOG; };
struct EvtCtx {
    const uint8_t* blob;
    const uint32_t* roots_brp;
    const uint32_t* rows;           // z^(2^L), uniform over the workgroup
    bool bad;
};
// S(L, node) from its two children: Z = z^(2^(L-1)) (row L - 1), c = roots_brp[2 node]
template <int L>
__device__ __forceinline__ void evt_combine(fr_t& out, fr_t& sl, fr_t& sr, int node, const EvtCtx& cx) {
    fr_t Z, c, zp, zm;
    load_fr12(Z, cx.rows + (L - 1) * 12);
    load_fr12(c, cx.roots_brp + (size_t)(2 * node) * 12);
    Fr::add(zp, Z, c);                                      // < 2 r
    Fr::sub<0>(zm, Z, c);                                   // Z - c + r < 2 r
    Fr::opaque(sl);
    Fr::opaque(sr);
    Fr::mul2(out, sl, zp, sr, zm);                          // < (2 * 1.1 r * 2 r) / R + r < 1.1 r   (r / R = 2^-6.1)
}
template <int L>
__device__ __forceinline__ void evt_subtree(fr_t& out, int node, EvtCtx& cx) {
    if constexpr (L == 1) {
        const int p0 = 2 * node;
        uint32_t w0[8], w1[8];
        load_be32_words(w0, cx.blob + (size_t)p0 * 32);
        load_be32_words(w1, cx.blob + (size_t)p0 * 32 + 32);
        // canonical check: the top word decides for all but 2^-31 of the canonical values
        if (w0[7] >= FrParams::MOD32[7]) cx.bad |= Fr::words32_geq_mod(w0);
        if (w1[7] >= FrParams::MOD32[7]) cx.bad |= Fr::words32_geq_mod(w1);
        fr_t f0, f1, em, ep, w, z, tt;
        Fr::from_words32(f0, w0);                           // plain; a non-canonical word fails the call, whatever it adds here
        Fr::from_words32(f1, w1);
        Fr::sub<0>(em, f0, f1);                             // f_0 - f_1 + r
        Fr::add(ep, f0, f1);
        load_fr12(w, cx.roots_brp + (size_t)p0 * 12);
        load_fr12(z, cx.rows);
        Fr::mul2(tt, em, z, ep, w);                         // z (f_0 - f_1) + w (f_0 + f_1), plain, < 1.1 r
        Fr::mul(out, tt, w);
    } else {
        fr_t sl, sr;
#pragma unroll 1
        for (int c = 0; c < 2; c++) {
            fr_t v;
            evt_subtree<L - 1>(v, 2 * node + c, cx);
            if (c == 0) sl = v;
            else sr = v;
        }
        evt_combine<L>(out, sl, sr, node, cx);
    }
}
template <int LOG>
__device__ __forceinline__ void evt_blob_block(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ zrows, const uint32_t* __restrict__ roots_brp,
                                               uint32_t* __restrict__ y_out, int* __restrict__ status, const RlcOut& rlc, const int b) {
    constexpr int T = 4096 >> LOG;
    __shared__ uint32_t sh[T * 9];
    __shared__ int sh_bad;
    const int t = threadIdx.x;
    if (t == 0) sh_bad = 0;
    EvtCtx cx;
    cx.blob = blobs + (size_t)b * 131072;
    cx.roots_brp = roots_brp;
    cx.rows = zrows + (size_t)b * EVT_ROWS * 12;
    cx.bad = false;
    fr_t s;
    evt_subtree<LOG>(s, t, cx);
    __syncthreads();                                        // sh_bad = 0 is out
    if (cx.bad) sh_bad = 1;
    // the levels above the lanes: after the step of level L the lanes whose low L - LOG bits are zero hold S(L, t >> (L - LOG))
#pragma unroll
    for (int q = 0; q < 9; q++) sh[q * T + t] = s.l[q];
    __syncthreads();
#pragma unroll 1
    for (int L = LOG + 1; L <= 12; L++) {
        const int step = 1 << (L - 1 - LOG);
        if ((t & (2 * step - 1)) == 0) {
            fr_t sr, Z, c, zp, zm;
#pragma unroll
            for (int q = 0; q < 9; q++) sr.l[q] = sh[q * T + t + step];
            load_fr12(Z, cx.rows + (L - 1) * 12);
            load_fr12(c, roots_brp + (size_t)(2 * (t >> (L - LOG))) * 12);
            Fr::add(zp, Z, c);
            Fr::sub<0>(zm, Z, c);
            Fr::opaque(s);
            Fr::mul2(s, s, zp, sr, zm);
#pragma unroll
            for (int q = 0; q < 9; q++) sh[q * T + t] = s.l[q];
        }
        __syncthreads();
    }
    if (t == 0) {
        if (sh_bad) {
            status[b] = MKZG_ERR_NON_CANONICAL_SCALAR;
        } else {
            fr_t c, r2, y, z;
            Fr::set(c, FR_INV_4096);
            Fr::set(r2, FrParams::R2);
            Fr::mul(c, c, r2);                              // R^2 / 4096: plain S -> Montgomery S / 4096
            FrC::mul(y, s, c);
            store_fr12(y_out + (size_t)b * 12, y);
            status[b] = 0;
            load_fr12(z, cx.rows);
            if (rlc.s1) rlc_one_blob(rlc, b, z, y);
        }
    }
}
template <int LOG>
__global__ void __launch_bounds__(4096 >> LOG) k_blob_eval_tree(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ zrows, const uint32_t* __restrict__ roots_brp,
                                                               uint32_t* __restrict__ y_out, int* __restrict__ status, const RlcOut rlc) {
    evt_blob_block<LOG>(blobs, zrows, roots_brp, y_out, status, rlc, (int)blockIdx.x);
}
// the evaluations of SEVERAL calls in one launch, behind their hash round on the round's stream (api_verify.hip, deep rounds): a context
// runs one kernel at a time per hardware queue, so a call's own launch waits its turn behind other calls' kernels
// and then fills the chip alone -- a round's worth of blobs in one launch does neither
template <int LOG>
__global__ void __launch_bounds__(4096 >> LOG) k_blob_eval_tree_jobs(const EvalJobs jobs, const uint32_t* __restrict__ roots_brp) {
    int j = 0;
#pragma unroll 1
    while (j + 1 < jobs.count && (int)blockIdx.x >= jobs.first_block[j + 1]) j++;
    RlcOut rlc;
    rlc.s1 = jobs.s1[j];
    rlc.s3 = jobs.s3[j];
    rlc.ry = jobs.ry[j];
    rlc.exp_offset = jobs.exp_offset[j];
#pragma unroll
    for (int i = 0; i < 8; i++) rlc.r.w[i] = jobs.r[j][i];
    evt_blob_block<LOG>(jobs.blobs[j], jobs.zrows[j], roots_brp, jobs.y_out[j], jobs.status[j], rlc, (int)blockIdx.x - jobs.first_block[j]);
}

// MODE 0: y_b = f_b(z_b)                                  -> y_out   (the per-lane Montgomery trick; kept for MODE 1 and as the A/B of the tree)
// MODE 1: q_b = (f_b - y_b)/(X - z_b) in evaluation form   -> quot_out (4096 rows of 12 words), also y_out
// z comes from z_in (rows of 12 words); if y_given != nullptr MODE 1 trusts it (not used).
template <int MODE>
__global__ void __launch_bounds__(EVB_T) k_blob_eval(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ z_in, const uint32_t* __restrict__ roots_brp,
                                                    const uint32_t* __restrict__ inv_roots_brp, uint32_t* __restrict__ y_out, uint32_t* __restrict__ quot_out,
                                                    int* __restrict__ status) {
    __shared__ uint32_t sh[EVB_T * 9];
    __shared__ int sh_found;
    __shared__ int sh_bad;
    const int b = blockIdx.x, t = threadIdx.x;
    const uint8_t* blob = blobs + (size_t)b * 131072;
    if (t == 0) { sh_found = -1; sh_bad = 0; }
    __syncthreads();
    fr_t z;
    load_fr12(z, z_in + (size_t)b * 12);

    // pass 1: canonical check, d_i = z - w_i, per-lane prefix products (zeros skipped)
    fr_t pre[EVB_PER];
    fr_t acc;
    Fr::set_one(acc);
    bool bad = false;
    int found = -1;
#pragma unroll 4
    for (int k = 0; k < EVB_PER; k++) {
        const int i = t + EVB_T * k;
        uint32_t w[8];
        load_be32_words(w, blob + (size_t)i * 32);
        bad |= Fr::words32_geq_mod(w);
        fr_t wi, d;
        load_fr12(wi, roots_brp + (size_t)i * 12);
        FrC::sub(d, z, wi);
        if (Fr::limbs_all_zero(d)) found = i;
        else FrC::mul(acc, acc, d);
        pre[k] = acc;
    }
    if (bad) sh_bad = 1;
    if (found >= 0) sh_found = found;
    __syncthreads();
    if (sh_bad) {
        if (t == 0) status[b] = MKZG_ERR_NON_CANONICAL_SCALAR;
        return;
    }
    const int fidx = sh_found;
    if (t == 0) status[b] = 0;

    if (MODE == 0 && fidx >= 0) {
        // z is a domain point: the value is the corresponding evaluation (domain.go:204-207)
        if (t == 0) {
            uint32_t w[8];
            load_be32_words(w, blob + (size_t)fidx * 32);
            fr_t f;
            fr_from_words(f, w);
            store_fr12(y_out + (size_t)b * 12, f);
        }
        return;
    }

    fr_t inv;
    fr_inv_n(inv, acc);   // 1 / prod d_i over this lane's non-zero d_i

    if (MODE == 0 || fidx < 0) {
        // sum_i f_i w_i / (z - w_i) = z sum_i f_i / (z - w_i) - sum_i f_i   (w_i / (z - w_i) = z / (z - w_i) - 1): no product with the
        // roots, and the blob's words never have to be brought into Montgomery form -- the Montgomery product of the PLAIN f_i with
        // the Montgomery form of 1 / (z - w_i) is the plain f_i / (z - w_i).  Four multiplications per element (three of them the
        // batch inversion) instead of six; both sums are plain residues and meet Montgomery form once per lane.
        fr_t s1, s2;
        Fr::set_zero(s1);
        Fr::set_zero(s2);
#pragma unroll 4
        for (int k = EVB_PER - 1; k >= 0; k--) {
            const int i = t + EVB_T * k;
            fr_t wi, d, di, f, term;
            load_fr12(wi, roots_brp + (size_t)i * 12);
            FrC::sub(d, z, wi);
            if (k > 0) FrC::mul(di, inv, pre[k - 1]); else di = inv;
            FrC::mul(inv, inv, d);
            uint32_t w[8];
            load_be32_words(w, blob + (size_t)i * 32);
            Fr::from_words32(f, w);                  // plain, canonical (checked in pass 1)
            FrC::mul(term, f, di);                   // plain f_i / (z - w_i)
            FrC::add(s1, s1, term);
            FrC::add(s2, s2, f);
            if (MODE == 1) pre[k] = di;   // keep 1/(z - w_i) for the quotient
        }
        fr_t part, s1m, s2m;
        Fr::from_plain(s1m, s1);
        Fr::reduce_once(s1m, s1m);
        Fr::from_plain(s2m, s2);
        Fr::reduce_once(s2m, s2m);
        FrC::mul(part, s1m, z);
        FrC::sub(part, part, s2m);
        block_sum_fr<EVB_T>(part, sh);
        // y = (z^4096 - 1)/4096 * sum
        fr_t y;
        {
            fr_t zp = z, one, c;
#pragma unroll 1
            for (int s = 0; s < 12; s++) FrC::mul(zp, zp, zp);
            Fr::set_one(one);
            FrC::sub(zp, zp, one);
            Fr::set(c, FR_INV_4096);
            FrC::mul(zp, zp, c);
            FrC::mul(y, zp, part);
        }
        if (t == 0) store_fr12(y_out + (size_t)b * 12, y);
        if (MODE == 1) {
            // q_i = (f_i - y) / (w_i - z) = (y - f_i) * 1/(z - w_i)          (kzg_prove.go:81-112)
#pragma unroll 4
            for (int k = 0; k < EVB_PER; k++) {
                const int i = t + EVB_T * k;
                uint32_t w[8];
                load_be32_words(w, blob + (size_t)i * 32);
                fr_t f, q;
                fr_from_words(f, w);
                FrC::sub(f, y, f);
                FrC::mul(q, f, pre[k]);
                store_fr12(quot_out + ((size_t)b * 4096 + i) * 12, q);
            }
        }
        return;
    }

    // MODE 1 and z = w_m is in the domain (kzg_prove.go:118-180): y = f_m,
    //   q_j = (f_j - y)/(w_j - w_m) (j != m),   q_m = sum_{j != m} -q_j w_j / w_m
    {
        fr_t y;
        {
            uint32_t w[8];
            load_be32_words(w, blob + (size_t)fidx * 32);
            fr_from_words(y, w);
        }
        if (t == 0) store_fr12(y_out + (size_t)b * 12, y);
        fr_t part;
        Fr::set_zero(part);
#pragma unroll 4
        for (int k = EVB_PER - 1; k >= 0; k--) {
            const int i = t + EVB_T * k;
            fr_t wi, d, di, f, q, term;
            load_fr12(wi, roots_brp + (size_t)i * 12);
            FrC::sub(d, z, wi);
            const bool self = (i == fidx);
            if (!self) {
                // pre[] skipped the zero, so the chain is consistent: previous prefix is pre[k-1]
                if (k > 0) FrC::mul(di, inv, pre[k - 1]); else di = inv;
                FrC::mul(inv, inv, d);
                uint32_t w[8];
                load_be32_words(w, blob + (size_t)i * 32);
                fr_from_words(f, w);
                FrC::sub(f, y, f);               // y - f_i
                FrC::mul(q, f, di);              // (y - f_i)/(z - w_i) = (f_i - y)/(w_i - z)
                store_fr12(quot_out + ((size_t)b * 4096 + i) * 12, q);
                FrC::mul(term, q, wi);
                FrC::sub(part, part, term);      // accumulate -q_j w_j
            }
        }
        block_sum_fr<EVB_T>(part, sh);
        if (t == 0) {
            fr_t iw, qm;
            load_fr12(iw, inv_roots_brp + (size_t)fidx * 12);
            FrC::mul(qm, part, iw);
            store_fr12(quot_out + ((size_t)b * 4096 + fidx) * 12, qm);
        }
    }
}

// y_b = f_b(z_b) from the COEFFICIENTS of f_b (rows of 12 words, Montgomery form): f(z) = sum_t z^t g_t(z^256) with
// g_t(w) = sum_{k<16} c[t + 256 k] w^k -- one lane per t (coalesced rows), Horner in w, then z^t and a workgroup sum.
// ~45 field multiplications per lane against ~160 for the barycentric formula, and no inversion: the verifier takes the
// coefficients from an inverse NTT that runs beside the SHA-256 challenge (it does not depend on z), so that only this
// kernel is left on the critical path after the challenge.  The value is that of EvaluateLagrangePolynomial
// (internal/domain/domain.go:193-235) for every z, domain points included.
__global__ void __launch_bounds__(256) k_poly_eval(const uint32_t* __restrict__ coeffs, const uint32_t* __restrict__ z_in, const int* __restrict__ status,
                                                   uint32_t* __restrict__ y_out) {
    __shared__ uint32_t sh[256 * 9];
    const int b = blockIdx.x, t = threadIdx.x;
    if (status[b] != 0) return;
    fr_t z, w, acc, c;
    load_fr12(z, z_in + (size_t)b * 12);
    w = z;
#pragma unroll 1
    for (int i = 0; i < 8; i++) FrC::mul(w, w, w);                     // z^256
    const uint32_t* col = coeffs + ((size_t)b * 4096 + t) * 12;
    load_fr12(acc, col + (size_t)15 * 256 * 12);
#pragma unroll 1
    for (int k = 14; k >= 0; k--) {
        FrC::mul(acc, acc, w);
        load_fr12(c, col + (size_t)k * 256 * 12);
        FrC::add(acc, acc, c);
    }
    // times z^t
    fr_t base = z, p;
    Fr::set_one(p);
    Fr::reduce_once(p, p);
#pragma unroll 1
    for (int e = t; e; e >>= 1) {
        if (e & 1) FrC::mul(p, p, base);
        FrC::mul(base, base, base);
    }
    FrC::mul(acc, acc, p);
    block_sum_fr<256>(acc, sh);
    if (t == 0) store_fr12(y_out + (size_t)b * 12, acc);
}

// ------------------------------------------------------------------------------------------
// G1 subgroup membership: phi(P) == -[x^2]P, one lane per affine row; st[i] |= NOT_IN_SUBGROUP
// ------------------------------------------------------------------------------------------
// Decoding and subgroup test of the verifiers' points in ONE launch, two point arrays (commitments and proofs) at once.  Both steps
// are per-point latency chains -- a ~380-squaring square root (0.43 ms) and the 126 doublings of phi(P) == -[x^2]P -- and the second
// does not need the first's result: the y-free form of the test (g1.hpp) runs on x and x^3 + 4 alone.  So one wavefront decodes
// (gnark SetBytes, serialization.go:108-115) and writes the affine rows while another runs the r-torsion test; a point that fails to
// decode reports that error, whatever the test computed from its x.
//
// The test's chain is shortened by giving every point a QUAD of lanes (g1_quad.hpp): the a = 0 doubling is 7 field multiplications
// in three dependent levels, so three lanes of the quad take one product of a level each (operands picked by role with v_cndmask,
// results passed with quad_perm DPP moves, ~84 of each per doubling against ~1400 instructions of products) and the chain is 3
// multiplications deep per doubling instead of 7: 1.02 -> 0.68 ms for the launch, on every verifier's critical path.  All four lanes hold the same (X, Y, Z) before and after; the ten additions of
// the chain run replicated.  A block is 16 points: with 2048 points (a 1024-blob batch) that is 128 blocks of two wavefronts,
// one wavefront per SIMD.
// mode 0: both wavefronts, one status per point.  mode 1 (64 threads, 64 points per block) / mode 2 (64 threads, 16 points per
// block): only the decoding / only the r-torsion test, as two launches on two streams -- for the cell verifier, whose MSMs can
// then start on the decoded rows while the test is still running; mode 2 writes MKZG_ERR_G1_NOT_IN_SUBGROUP or 0 to ITS OWN status
// arrays (the caller gives the decoding's status precedence).  mode 3: as mode 0 with ONE lane per point in the test and 64 points
// per block -- the quad form spends 2.1x the instructions for its shorter chain, which a context that has several calls in flight
// (the chip is shared, nobody waits for one chain) does not want (api_verify.hip throughput_shape).
// Every wavefront runs with all 64 lanes active (FULL_WAVES, see fk20_kernels.hip: sparse wavefronts issue up to 3x slower on gfx950,
// and the DPP moves read every lane of a quad): lanes beyond the points shadow another point and store nothing.
// G1D::in_subgroup_yfree with the doublings on a lane quad; every lane of the quad returns the verdict
__device__ __forceinline__ bool in_subgroup_yfree_quad(const fp_t& px, const fp_t& rhs, const fp_t& beta, int role) {
    fp_t rhs2;
    G1J Q;
    G1D::JacYC base;
    G1D::yfree_start(Q, base, rhs2, px, rhs);
    const QuadRole qr(role);
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll 1
        for (int bit = 62; bit >= 0; bit--) {
            jdbl_quad(Q, qr);
            if ((BLS_X_ABS >> bit) & 1) G1D::jadd_y(Q, Q, base, rhs, rhs2);
        }
        if (pass == 0) G1D::jy_cache(base, Q, rhs);
    }
    return G1D::yfree_verdict(Q, px, rhs, rhs2, beta);
}
static const int VAL_PB = 16;                               // points per block of the r-torsion test (a quad of lanes each)
// the body for one workgroup: block `blk` of one (points a, points b) job
__device__ __forceinline__ void g1_validate_block(const uint8_t* __restrict__ in48_a, int n_a, uint32_t* __restrict__ aff_a, int* __restrict__ st_a,
                                                  const uint8_t* __restrict__ in48_b, int n_b, uint32_t* __restrict__ aff_b, int* __restrict__ st_b, int mode, int blk) {
    __shared__ int sub_ok[64];                              // 1 in the subgroup, 0 not, 2 no verdict (identity or undecodable x)
    // modes: 0 decoding + quad test (128 threads), 1 decoding only, 2 quad test only, 3 decoding + one-lane test (128 threads), 4 one-lane
    // test only (64 threads: half the instructions of the quad test and a quarter of its wavefronts, a chain 1.7x as long -- for calls long
    // enough to hide it: the quad test of 8192 proofs holds 516 wavefronts, each with a SIMD's whole register file, and every other
    // kernel's multi-wavefront workgroups wait for a compute unit with four free SIMDs)
    const bool wide = mode == 3 || mode == 4, both = mode == 0 || mode == 3;
    const int lane = threadIdx.x & 63, role = both ? (int)(threadIdx.x >> 6) : (mode == 1 ? 0 : 1);
    const int total = n_a + n_b;
    // which point: the test gives a quad per point (16 per wavefront) unless wide; the decoding a lane per point -- 64 per block on
    // its own (mode 1) or beside the wide test (mode 3), the block's 16 (copied four times over the wavefront) beside the quad test
    const int per_block = (mode == 1 || wide) ? 64 : VAL_PB;
    const int slot = wide ? lane : (role == 1 ? lane >> 2 : (mode == 1 ? lane : lane & (VAL_PB - 1)));
    int i = blk * per_block + slot;
    const bool live = i < total && (wide ? true : (role == 1 ? (lane & 3) == 0 : (mode == 1 || lane < VAL_PB)));
    if (i >= total) i %= total;
    const uint8_t* in48 = in48_a;
    uint32_t* aff = aff_a;
    int* st = st_a;
    if (i >= n_a) { i -= n_a; in48 = in48_b; aff = aff_b; st = st_b; }
    int v = 2, rc = 0;
    G1Aff p;
    p.inf = 1;
    if (role == 1) {
        fp_t x, rhs, beta;
        uint32_t flags;
        if (g1_parse_lane(x, flags, in48 + (size_t)i * 48) == 0) {        // uniform over the quad: its lanes read the same point
            g1_curve_rhs(rhs, x);
            Fp::set(beta, FP_BETA);
            if (wide) v = G1D::in_subgroup_yfree(x, rhs, beta) ? 1 : 0;
            else v = in_subgroup_yfree_quad(x, rhs, beta, lane & 3) ? 1 : 0;
        }
    } else {
        rc = g1_decode_lane(p, in48 + (size_t)i * 48);
        if (live) {
            uint32_t w[32];
#pragma unroll
            for (int q = 0; q < 14; q++) { w[q] = p.x.l[q]; w[14 + q] = p.y.l[q]; }
            w[28] = p.inf; w[29] = 0; w[30] = 0; w[31] = 0;
            uint4* o = reinterpret_cast<uint4*>(aff + (size_t)i * 32);
#pragma unroll
            for (int q = 0; q < 8; q++) o[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
        }
    }
    if (mode == 1) {                                        // decoding only (64 threads, all role 0)
        if (live) st[i] = rc;
        return;
    }
    if (mode == 2 || mode == 4) {                           // r-torsion test only (64 threads, all role 1), its own status arrays
        if (live) st[i] = v == 0 ? MKZG_ERR_G1_NOT_IN_SUBGROUP : 0;
        return;
    }
    // modes 0, 3: both wavefronts meet at ONE barrier; the decoding wavefront then joins the two verdicts
    if (role == 1 && (wide || (lane & 3) == 0)) sub_ok[slot] = v;
    __syncthreads();
    if (role == 0 && live) {
        if (rc == 0 && !p.inf && sub_ok[slot] == 0) rc = MKZG_ERR_G1_NOT_IN_SUBGROUP;
        st[i] = rc;
    }
}
// ---- modes 3 and 4 since round 5: the register-lean chain (g1.hpp in_subgroup_yfree_lean)
// What the chain's ten additions read lives in LDS, one column of 4 x 14 words per lane (conflict-free 32-bit accesses); the loads
// are volatile so that the compiler cannot carry the values through the doubling loops in registers -- which is the whole point:
// mode 3 / 4 held 487 / 474 registers (256 + AGPRs as spill space), one wavefront per SIMD with nothing beside it, for chains that
// issue at ~60 % of a SIMD's rate.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
struct LdsStash {
    uint32_t* col;                                          // &mem[lane]; word (slot * 14 + limb) at col[(slot * 14 + limb) * 64]
    __device__ __forceinline__ void put(int slot, const fp_t& v) {
#pragma unroll
        for (int q = 0; q < 14; q++) col[(slot * 14 + q) * 64] = v.l[q];
    }
    __device__ __forceinline__ void get(fp_t& v, int slot) const {
        const volatile lds_u32* c = (const volatile lds_u32*)col;     // (the address space spelled out: through the struct it decays to flat loads)
#pragma unroll
        for (int q = 0; q < 14; q++) v.l[q] = c[(slot * 14 + q) * 64];
    }
};
// MODE 3: wavefront 0 decodes 64 points, writes their affine rows, leaves (return code, identity flag) in LDS and EXITS -- it used
// to idle at a barrier for two thirds of the launch holding its SIMD's registers; wavefront 1 runs the test on the same 64 points,
// then picks the decoding's verdicts up (long there) and writes the statuses.  MODE 4: the test alone, statuses of its own.
template <int MODE>
__device__ __forceinline__ void g1_validate_wide_block(const uint8_t* __restrict__ in48_a, int n_a, uint32_t* __restrict__ aff_a, int* __restrict__ st_a,
                                                       const uint8_t* __restrict__ in48_b, int n_b, uint32_t* __restrict__ aff_b, int* __restrict__ st_b, int blk) {
    __shared__ uint32_t stash_mem[4 * 14 * 64];
    __shared__ int dec[64];
    constexpr int DEC_READY = 1 << 30;
    const int lane = threadIdx.x & 63, role = MODE == 3 ? (int)(threadIdx.x >> 6) : 1;
    const int total = n_a + n_b;
    int i = blk * 64 + lane;
    const bool live = i < total;
    if (i >= total) i %= total;
    const uint8_t* in48 = in48_a;
    uint32_t* aff = aff_a;
    int* st = st_a;
    if (i >= n_a) { i -= n_a; in48 = in48_b; aff = aff_b; st = st_b; }
    if (MODE == 3) {
        if (role == 0) dec[lane] = 0;
        __syncthreads();
        if (role == 0) {
            G1Aff p;
            p.inf = 1;
            const int rc = g1_decode_lane(p, in48 + (size_t)i * 48);
            if (live) {
                uint32_t w[32];
#pragma unroll
                for (int q = 0; q < 14; q++) { w[q] = p.x.l[q]; w[14 + q] = p.y.l[q]; }
                w[28] = p.inf; w[29] = 0; w[30] = 0; w[31] = 0;
                uint4* o = reinterpret_cast<uint4*>(aff + (size_t)i * 32);
#pragma unroll
                for (int q = 0; q < 8; q++) o[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
            }
            *(volatile int*)&dec[lane] = DEC_READY | (rc & 0xff) | (p.inf ? 0x100 : 0);
            return;
        }
    }
    int v = 2;                                              // 1 in the subgroup, 0 not, 2 no verdict (identity or undecodable x)
    {
        fp_t x, rhs, beta;
        uint32_t flags;
        if (g1_parse_lane(x, flags, in48 + (size_t)i * 48) == 0) {
            g1_curve_rhs(rhs, x);
            Fp::set(beta, FP_BETA);
            LdsStash stash{stash_mem + lane};
            v = G1D::in_subgroup_yfree_lean(x, rhs, beta, stash) ? 1 : 0;
        }
    }
    if (MODE == 4) {
        if (live) st[i] = v == 0 ? MKZG_ERR_G1_NOT_IN_SUBGROUP : 0;
        return;
    }
    int d;
    do { d = *(volatile int*)&dec[lane]; } while (!(d & DEC_READY));
    if (live) {
        int rc = d & 0xff;
        if (rc == 0 && !(d & 0x100) && v == 0) rc = MKZG_ERR_G1_NOT_IN_SUBGROUP;
        st[i] = rc;
    }
}
// (one instantiation per mode: a kernel's registers are the maximum over every path it contains)
template <int MODE>
__global__ void __launch_bounds__(128) k_g1_validate(const uint8_t* __restrict__ in48_a, int n_a, uint32_t* __restrict__ aff_a, int* __restrict__ st_a,
                                                     const uint8_t* __restrict__ in48_b, int n_b, uint32_t* __restrict__ aff_b, int* __restrict__ st_b) {
    g1_validate_block(in48_a, n_a, aff_a, st_a, in48_b, n_b, aff_b, st_b, MODE, (int)blockIdx.x);
}
// Modes 3 and 4 with the register-lean chain.  ONE WAVEFRONT PER SIMD ALL THE SAME (amdgpu_waves_per_eu(1, 1): the compiler pads the
// allocation): *measured* (round 5, tools/ab_r05_3.sh) this GPU's dispatcher does not look for an idle SIMD -- concurrent launches
// start filling the same compute units, and a chain wavefront that CAN share its SIMD with another launch's chain wavefront does:
// at 221 / 226 registers and two wavefronts per SIMD the test alone ran in 0.95 ms (1.07-1.10 with the old chain), but beside the
// cell verifier's decoding launch both took 1.2-1.3 ms (decoding: 0.45 -> 1.16 ms), the lone 8192-cell call went from 1.8 to 2.9 ms
// and 32 callers of the blob verifier from 2.2 to 1.3 M blobs/s.  What the lean chain buys is therefore its instructions (10 products
// per addition instead of 17, no select / copy code in the loop, a loop body that fits the instruction cache) and the decoding
// wavefronts that EXIT instead of idling at a barrier -- not a second chain on the same SIMD.
#ifndef MKZG_VAL_LEAN_ATTR
#define MKZG_VAL_LEAN_ATTR __attribute__((amdgpu_waves_per_eu(1, 1)))
#endif
template <int MODE>
__global__ void __launch_bounds__(128) MKZG_VAL_LEAN_ATTR k_g1_validate_lean(const uint8_t* __restrict__ in48_a, int n_a, uint32_t* __restrict__ aff_a, int* __restrict__ st_a,
                                                                            const uint8_t* __restrict__ in48_b, int n_b, uint32_t* __restrict__ aff_b, int* __restrict__ st_b) {
#ifndef MKZG_VAL_LEAN_SHARED
    // The WHOLE register file of the SIMD (256 + 256), as the old chain's 474-497 registers had in effect: with the allocation the
    // compiler pads to for one wavefront per SIMD (~380) the 108-register evaluation wavefronts of other calls still moved in beside
    // the chain -- 32 callers 1.43 M blobs/s, 1.85 M with this line, tools/ab_r05_4.sh.  -DMKZG_VAL_LEAN_SHARED builds without it.
    asm volatile("" ::: "v255", "a255");
#endif
    g1_validate_wide_block<MODE>(in48_a, n_a, aff_a, st_a, in48_b, n_b, aff_b, st_b, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// RLC scalars (single workgroup): r -> r^i ; s1_i = r^i ; s3_i = r^i z_i ; e = sum r^i y_i
// outputs as plain little-endian 8-word scalars
// ------------------------------------------------------------------------------------------
// mode 0: everything; mode 1: s1 only (does not need z, y: issued before the challenges exist); mode 2: s3 and e only
__global__ void __launch_bounds__(EV_T) k_rlc_scalars(const RWords r_words8, const uint32_t* __restrict__ z_in, const uint32_t* __restrict__ y_in, int n,
                                                      uint32_t exp_offset, uint32_t* __restrict__ s1, uint32_t* __restrict__ s3, uint32_t* __restrict__ e_out, int mode) {
    __shared__ uint32_t sh[EV_T * 9];
    const int t = threadIdx.x;
    uint32_t rw[8];
#pragma unroll
    for (int i = 0; i < 8; i++) rw[i] = r_words8.w[i];
    for (int rep = 0; rep < 2; rep++) {
        if (Fr::words32_geq_mod(rw)) {
            uint64_t br = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) { uint64_t v = (uint64_t)rw[i] - FrParams::MOD32[i] - br; rw[i] = (uint32_t)v; br = (v >> 32) & 1; }
        }
    }
    fr_t r;
    fr_from_words(r, rw);
    fr_t part;
    Fr::set_zero(part);
    for (int i = t; i < n; i += EV_T) {
        // r^i by square-and-multiply on the (small) exponent i
        fr_t rp, base = r;
        Fr::set_one(rp);
        Fr::reduce_once(rp, rp);
        for (uint32_t e = exp_offset + (uint32_t)i; e; e >>= 1) {
            if (e & 1) fr_mul_n(rp, rp, base);
            fr_mul_n(base, base, base);
        }
        fr_t pl;
        uint32_t w[8];
        if (mode != 2) {
            Fr::to_plain(pl, rp);
            Fr::to_words32(w, pl);
#pragma unroll
            for (int q = 0; q < 8; q++) s1[(size_t)i * 8 + q] = w[q];
        }
        if (mode != 1) {
            fr_t z, y, rz, ry;
            load_fr12(z, z_in + (size_t)i * 12);
            load_fr12(y, y_in + (size_t)i * 12);
            fr_mul_n(rz, rp, z);
            fr_mul_n(ry, rp, y);
            FrC::add(part, part, ry);
            Fr::to_plain(pl, rz);
            Fr::to_words32(w, pl);
#pragma unroll
            for (int q = 0; q < 8; q++) s3[(size_t)i * 8 + q] = w[q];
        }
    }
    if (mode == 1) return;
    block_sum_fr<EV_T>(part, sh);
    if (t == 0) {
        fr_t pl;
        uint32_t w[8];
        Fr::to_plain(pl, part);
        Fr::to_words32(w, pl);
#pragma unroll
        for (int q = 0; q < 8; q++) e_out[q] = w[q];
    }
}

// ------------------------------------------------------------------------------------------
// variable-base MSM: one 128-thread workgroup per (window, msm).  8-bit Booth windows, 128 signed
// buckets.  scalars: n x 8 plain words; points: n affine rows.  Output: per-window sum as 4
// canonical plain coordinates (X, Y, ZZ, ZZZ; 12 words each) -> 48 words.
// ------------------------------------------------------------------------------------------
constexpr int VC = 8, VNB = 128, VNWIN = 32, VMAXN = 8192;

__device__ __forceinline__ void shfl_xor_xyzz_v(G1X& r, const G1X& a, int m) {
#pragma unroll
    for (int i = 0; i < 14; i++) {
        r.x.l[i] = __shfl_xor(a.x.l[i], m);
        r.y.l[i] = __shfl_xor(a.y.l[i], m);
        r.zz.l[i] = __shfl_xor(a.zz.l[i], m);
        r.zzz.l[i] = __shfl_xor(a.zzz.l[i], m);
    }
}
__device__ __forceinline__ void shfl_down_xyzz_v(G1X& r, const G1X& a, int d) {
#pragma unroll
    for (int i = 0; i < 14; i++) {
        r.x.l[i] = __shfl_down(a.x.l[i], d);
        r.y.l[i] = __shfl_down(a.y.l[i], d);
        r.zz.l[i] = __shfl_down(a.zz.l[i], d);
        r.zzz.l[i] = __shfl_down(a.zzz.l[i], d);
    }
}

struct VmsmJob {
    const uint32_t* scalars;   // n x 8 words
    const uint32_t* points;    // n affine rows (32 words)
    int n;
};
struct VmsmJobs { VmsmJob j[4]; int k[4]; };    // k: point-range slices per (job, window), 1..VMSM_KMAX

// grid (window, job, slice): a (job, window) pair with many points is cut into up to VMSM_KMAX slices of the point range, each
// with its own 128 buckets -- the bucket walk is a serial chain of mixed additions per lane, so 8192 points in one workgroup
// are 64 additions deep where eight slices are 8 deep; k_vmsm_combine adds the slices' window sums.
// (one wavefront per SIMD by declaration, like k_g1_validate_lean and for the same measured reason: at 255 registers two of these chains
// -- or one of these and a subgroup test -- would share a SIMD whenever two launches meet on a compute unit)
#ifndef MKZG_VMSM_RESERVE
#define MKZG_VMSM_RESERVE "a89"             // 256 + 90 registers: what the kernel held before round 5 (344) -- a challenge-hash wavefront (153) still fits beside it, a second chain does not
#endif
// one (window w, slice z of K) of ONE MSM; parts_m: that MSM's [VNWIN][VMSM_KMAX][64] slice sums (with the tickets), out_m: its [VNWIN][48] window sums
__device__ __forceinline__ void vmsm_window_block(const VmsmJob& job, const int K, uint32_t* __restrict__ parts_m, uint32_t* __restrict__ out_m, const int w, const int z) {
    __shared__ uint32_t ent[VMAXN];
    __shared__ uint32_t hist[VNB + 1];
    __shared__ uint32_t cursor[VNB];
    __shared__ uint32_t xchg[2 * 56];
    __shared__ int last_flag;
    const int t = threadIdx.x;
    if (z >= K) return;
    const int n_all = job.n, lo = (int)((long long)n_all * z / K), hi = (int)((long long)n_all * (z + 1) / K);
    const uint32_t* sc = job.scalars + (size_t)lo * 8;
    const uint32_t* pts = job.points + (size_t)lo * 32;
    const int n = hi - lo;
    hist[t] = 0;
    if (t == 0) hist[VNB] = 0;
    __syncthreads();
    // Booth digit of window w: v = bits [8w-1, 8w+8) ; d = ((v+1)>>1) - (v>>8 ? 256 : 0)
    auto digit = [&](int i) -> int {
        const uint32_t* s = sc + (size_t)i * 8;
        const int bit = VC * w - 1;
        uint32_t v;
        if (bit < 0) {
            v = (s[0] << 1) & 0x1ffu;
        } else {
            const int wi = bit >> 5, off = bit & 31;
            v = s[wi] >> off;
            if (off + VC + 1 > 32 && wi + 1 < 8) v |= s[wi + 1] << (32 - off);
            v &= 0x1ffu;
        }
        int d = (int)((v + 1) >> 1);
        if (v >> VC) d -= (1 << VC);
        return d;
    };
    for (int i = t; i < n; i += VNB) {
        int d = digit(i);
        if (d != 0) atomicAdd(&hist[(d < 0 ? -d : d) - 1], 1u);
    }
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (int k = 0; k < VNB; k++) { uint32_t c = hist[k]; hist[k] = run; cursor[k] = run; run += c; }
        hist[VNB] = run;
    }
    __syncthreads();
    for (int i = t; i < n; i += VNB) {
        int d = digit(i);
        if (d != 0) {
            uint32_t slot = atomicAdd(&cursor[(d < 0 ? -d : d) - 1], 1u);
            ent[slot] = (uint32_t)i | (d < 0 ? 0x80000000u : 0u);
        }
    }
    __syncthreads();
    // bucket t
    G1X acc;
    g1x_set_inf(acc);
    for (uint32_t e = hist[t]; e < hist[t + 1]; e++) {
        const uint32_t v = ent[e];
        G1Aff p;
        load_affine_v(p, pts + (size_t)(v & 0x7fffffffu) * 32);
        if (p.inf) continue;
        if (v >> 31) Fp::template neg<0>(p.y, p.y);
        g1x_madd(acc, p.x, p.y);
    }
    // window sum = sum_k (k+1) B_k = sum over all 128 buckets of the SUFFIX sums  suffix_k = sum_{j >= k} B_j:
    //   rounds 0-5   suffix scan inside each wavefront (64 buckets)
    //   round  6     wavefront 0 adds wavefront 1's total to every one of its suffixes (one parallel addition, through LDS)
    //   rounds 7-12  tree sum of the 64 suffixes of each wavefront
    //   round  13    thread 0 adds wavefront 1's sum: the slice's window sum
    //   rounds 14..  (slices > 1, the LAST workgroup of this (job, window) to get here) thread 0 adds the other slices' sums
    // -- 14 additions deep; the weights never appear, so there is no 64 S_1 = six serial doublings as a (S, Q) formulation has.
    // ONE inlined copy of the addition in a rolled loop, its second operand read from an LDS snapshot where it is used
    // (G1D::add_from) -- until round 5 the operand came by 56 wavefront shuffles into registers and the kernel held 344 of them plus
    // 688 B of scratch, one wavefront per SIMD; the snapshot reuses ent[], dead after the bucket walk.  The slice combination and
    // the export as canonical plain coordinates used to be a launch of their own (k_vmsm_combine): an atomic ticket per (job,
    // window) in the spare words of slice 0's record -- zero when the workspace is allocated, reset by whoever takes the last one.
    struct LdsXyzz {
        const uint32_t* p;
        int stride;
        __device__ __forceinline__ void get(fp_t& v, int c) const {
            const volatile lds_u32* q = (const volatile lds_u32*)p + c * 14 * stride;
#pragma unroll
            for (int i = 0; i < 14; i++) v.l[i] = q[i * stride];
        }
    };
    const int lane = t & 63, wave = t >> 6;
    G1X Pf = acc;
    __syncthreads();                                        // every lane is through with ent[]
    uint32_t* snap = ent;                                   // [56][VNB]
    uint32_t* mine = parts_m + ((size_t)w * VMSM_KMAX + z) * 64;
    uint32_t* ticket = parts_m + ((size_t)w * VMSM_KMAX) * 64 + 56;
    int rounds = 14;
#pragma unroll 1
    for (int round = 0; round < rounds; round++) {
        LdsXyzz src;
        bool take;
        if (round == 6 || round >= 13) {
            uint32_t* x = xchg + (round & 1) * 56;          // round 6 and the odd rounds from 13 on never overlap in time: barriers below
            if (round <= 13) {
                if (wave == 1 && lane == 0) {
#pragma unroll
                    for (int i = 0; i < 14; i++) { x[i] = Pf.x.l[i]; x[14 + i] = Pf.y.l[i]; x[28 + i] = Pf.zz.l[i]; x[42 + i] = Pf.zzz.l[i]; }
                }
            } else if (t == 0) {                            // another slice's sum (published before the ticket we took: pub_ticket.hpp)
                const int zz = round - 14 < z ? round - 14 : round - 13;
                const uint32_t* o = parts_m + ((size_t)w * VMSM_KMAX + zz) * 64;
#pragma unroll
                for (int i = 0; i < 56; i++) x[i] = pub_load(o + i);
            }
            __syncthreads();
            src.p = x;
            src.stride = 1;
            take = round == 6 ? wave == 0 : t == 0;
        } else {
            const int d = round < 6 ? (1 << round) : (32 >> (round - 7));
#pragma unroll
            for (int i = 0; i < 14; i++) {
                snap[i * VNB + t] = Pf.x.l[i];
                snap[(14 + i) * VNB + t] = Pf.y.l[i];
                snap[(28 + i) * VNB + t] = Pf.zz.l[i];
                snap[(42 + i) * VNB + t] = Pf.zzz.l[i];
            }
            src.p = snap + ((t + d) & (VNB - 1));
            src.stride = VNB;
            take = round < 6 ? (lane + d < 64) : (lane < d);
        }
        if (take) G1D::add_from(Pf, src);
        if (round == 13 && K > 1) {                         // publish the slice's sum and take a ticket; the last one carries on
            if (t == 0) {
#pragma unroll
                for (int i = 0; i < 14; i++) { pub_store(mine + i, Pf.x.l[i]); pub_store(mine + 14 + i, Pf.y.l[i]); pub_store(mine + 28 + i, Pf.zz.l[i]); pub_store(mine + 42 + i, Pf.zzz.l[i]); }
                const uint32_t got = pub_ticket(ticket);
                last_flag = got == (uint32_t)(K - 1);
                if (got == (uint32_t)(K - 1)) pub_store(ticket, 0u);
            }
            __syncthreads();
            if (!last_flag) return;
            rounds = 14 + K - 1;
        }
        if (round >= 13) __syncthreads();                   // xchg is rewritten by the next round
    }
    if (t == 0) {                                           // canonical plain coordinates (all-zero = identity) for the host's Horner step
        uint32_t* dst = out_m + (size_t)w * 48;
        const bool inf = g1x_is_inf(Pf);
        fp_t c;
        uint32_t ww[12];
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
            fp_t f;                                         // limb selects, not a reference picked at run time (that would pin Pf to scratch)
#pragma unroll
            for (int i = 0; i < 14; i++) f.l[i] = k == 0 ? Pf.x.l[i] : k == 1 ? Pf.y.l[i] : k == 2 ? Pf.zz.l[i] : Pf.zzz.l[i];
            Fp::to_plain(c, f);
            Fp::to_words32(ww, c);
#pragma unroll
            for (int qq = 0; qq < 12; qq++) dst[12 * k + qq] = inf ? 0u : ww[qq];
        }
    }
}
__global__ void __launch_bounds__(VNB) k_vmsm_window(VmsmJobs jobs, uint32_t* __restrict__ parts /* [msm][VNWIN][VMSM_KMAX][64] XYZZ */, uint32_t* __restrict__ out /* [msm][VNWIN][48] */) {
    asm volatile("" ::: MKZG_VMSM_RESERVE);
    const int m = blockIdx.y;
    vmsm_window_block(jobs.j[m], jobs.k[m], parts + (size_t)m * VNWIN * VMSM_KMAX * 64, out + (size_t)m * VNWIN * 48, (int)blockIdx.x, (int)blockIdx.z);
}
// the three MSMs of the blob verifier (kzg_verify.go:150,225,179: sum r^i Q_i, sum r^i C_i, sum r^i z_i Q_i) of SEVERAL calls in one launch
// (deep rounds, api_verify.hip): grid (window, 3 x call, slice); every call brings its own scratch and output (its lane's verdict block)
__global__ void __launch_bounds__(VNB) k_vmsm_window_calls(const VmsmCalls calls) {
    asm volatile("" ::: MKZG_VMSM_RESERVE);
    const int j = (int)blockIdx.y / 3, m = (int)blockIdx.y % 3;
    VmsmJob job;
    job.scalars = m == 2 ? calls.s3[j] : calls.s1[j];
    job.points = m == 1 ? calls.aff_c[j] : calls.aff_q[j];
    job.n = calls.n[j];
    vmsm_window_block(job, calls.k[j], calls.parts[j] + (size_t)m * VNWIN * VMSM_KMAX * 64, calls.out[j] + (size_t)m * VNWIN * 48, (int)blockIdx.x, (int)blockIdx.z);
}
// The same, a QUAD of lanes per bucket (g1_quad.hpp): for the MSM a lone caller WAITS for -- sum r^i z_i Q_i of the blob batch verifier
// can only start when the challenge hash has finished, and what it costs is the length of its dependent chain: the bucket walk and then
// the 13-addition suffix / tree network, ~14.5 us per addition on one lane.  On a quad an addition is four products deep instead of
// fourteen (xadd_quad / xmadd_quad), the points travel between buckets through LDS instead of wavefront shuffles (128 buckets x 4 lanes
// = 8 wavefronts), and the network is the same suffix scan (7 rounds over 128 buckets) + tree (7 rounds).  2.1x the instructions, a
// third of the latency: used when the launch leaves most of the chip empty anyway (vmsm_windows_device).  Slices of at most VQ_MAXN points.
constexpr int VQT = 4 * VNB, VQ_MAXN = 2048;
__global__ void __launch_bounds__(VQT) k_vmsm_window_quad(VmsmJobs jobs, uint32_t* __restrict__ parts /* [msm][VNWIN][VMSM_KMAX][64] XYZZ */) {
    __shared__ uint32_t ent[VQ_MAXN];
    __shared__ uint32_t hist[VNB + 1];
    __shared__ uint32_t cursor[VNB];
    __shared__ uint32_t pbuf[VNB * 56];
    const int w = blockIdx.x, m = blockIdx.y, t = threadIdx.x, z = blockIdx.z;
    if (z >= jobs.k[m]) return;
    const int n_all = jobs.j[m].n, lo = (int)((long long)n_all * z / jobs.k[m]), hi = (int)((long long)n_all * (z + 1) / jobs.k[m]);
    const uint32_t* sc = jobs.j[m].scalars + (size_t)lo * 8;
    const uint32_t* pts = jobs.j[m].points + (size_t)lo * 32;
    const int n = hi - lo;
    if (t <= VNB) hist[t] = 0;
    __syncthreads();
    auto digit = [&](int i) -> int {                       // Booth digit of window w, as in k_vmsm_window
        const uint32_t* s = sc + (size_t)i * 8;
        const int bit = VC * w - 1;
        uint32_t v;
        if (bit < 0) {
            v = (s[0] << 1) & 0x1ffu;
        } else {
            const int wi = bit >> 5, off = bit & 31;
            v = s[wi] >> off;
            if (off + VC + 1 > 32 && wi + 1 < 8) v |= s[wi + 1] << (32 - off);
            v &= 0x1ffu;
        }
        int d = (int)((v + 1) >> 1);
        if (v >> VC) d -= (1 << VC);
        return d;
    };
    for (int i = t; i < n; i += VQT) {
        const int d = digit(i);
        if (d != 0) atomicAdd(&hist[(d < 0 ? -d : d) - 1], 1u);
    }
    __syncthreads();
    if (t < 64) {                                           // exclusive prefix sum of the 128 counts on one wavefront (two per lane)
        const uint32_t c0 = hist[2 * t], c1 = hist[2 * t + 1];
        uint32_t run = c0 + c1;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const uint32_t up = __shfl_up(run, sft);
            if (t >= sft) run += up;
        }
        const uint32_t before = run - (c0 + c1);
        hist[2 * t] = before;
        cursor[2 * t] = before;
        hist[2 * t + 1] = before + c0;
        cursor[2 * t + 1] = before + c0;
        if (t == 63) hist[VNB] = run;
    }
    __syncthreads();
    for (int i = t; i < n; i += VQT) {
        const int d = digit(i);
        if (d != 0) {
            const uint32_t slot = atomicAdd(&cursor[(d < 0 ? -d : d) - 1], 1u);
            ent[slot] = (uint32_t)i | (d < 0 ? 0x80000000u : 0u);
        }
    }
    __syncthreads();
    const QuadRole qr(t);
    const int b = t >> 2, role = t & 3;
    // (measured: ONE rolled loop with one general addition for both phases spills as much as the two copies below -- a G1X is 56
    // registers and the law needs three of them plus its temporaries -- and its walk is slower: 0.41 ms against 0.32 for this launch)
    G1X Pf, o;
    g1x_set_inf(Pf);
    for (uint32_t e = hist[b]; e < hist[b + 1]; e++) {      // the order within a bucket varies from run to run (atomics): the sum does not
        const uint32_t v = ent[e];
        G1Aff p;
        load_affine_v(p, pts + (size_t)(v & 0x7fffffffu) * 32);
        if (p.inf) continue;
        if (v >> 31) Fp::template neg<0>(p.y, p.y);
        xmadd_quad(Pf, p.x, p.y, qr);
    }
    // window sum = sum of the 128 suffix sums (see k_vmsm_window): 7 rounds of suffix scan, 7 rounds of tree, through LDS
#pragma unroll 1
    for (int round = 0; round < 14; round++) {
        const int d = round < 7 ? (1 << round) : (64 >> (round - 7));
        const bool take = round < 7 ? (b + d < VNB) : (b < d);
        fp_t mine;
        quad_pick(mine, qr, Pf.x, Pf.y, Pf.zz, Pf.zzz);     // each lane of the quad publishes one coordinate
#pragma unroll
        for (int i = 0; i < 14; i++) pbuf[b * 56 + role * 14 + i] = mine.l[i];
        __syncthreads();
        if (take) {
            const uint32_t* src = pbuf + (b + d) * 56;
#pragma unroll
            for (int i = 0; i < 14; i++) { o.x.l[i] = src[i]; o.y.l[i] = src[14 + i]; o.zz.l[i] = src[28 + i]; o.zzz.l[i] = src[42 + i]; }
        }
        __syncthreads();                                    // everybody has read before the next round overwrites
        if (take) xadd_quad(Pf, o, qr);
    }
    if (t == 0) {
        uint32_t* dst = parts + (((size_t)m * VNWIN + w) * VMSM_KMAX + z) * 64;
#pragma unroll
        for (int i = 0; i < 14; i++) { dst[i] = Pf.x.l[i]; dst[14 + i] = Pf.y.l[i]; dst[28 + i] = Pf.zz.l[i]; dst[42 + i] = Pf.zzz.l[i]; }
    }
}
// Jobs of a few pairs (VerifyKZGProof and VerifyBlobKZGProof: ONE pair per job; the blob verifier on the blobs of a block: 6 today):
// the window sum is sum_i d_i P_i with |d_i| <= 128, so a lane per (window, pair) gets d_i P_i by seven doublings and the additions of
// d_i's bits (~90 multiplications deep) and an xor butterfly adds the PN lanes of a window (14 per level) -- 90 + 14 log2(PN) where the
// bucket machinery above spends the 13 additions of its suffix / tree network (~190) on mostly empty buckets.  PN = pairs per job rounded
// up to a power of two, 64 / PN windows per wavefront; output as k_vmsm_combine's.
template <int PN>
__global__ void __launch_bounds__(64) k_vmsm_small(VmsmJobs jobs, uint32_t* __restrict__ out /* [msm][VNWIN][48] */) {
    constexpr int WPW = 64 / PN;                            // windows per wavefront
    const int m = blockIdx.y, lane = threadIdx.x, i = lane % PN;
    const int w0 = blockIdx.x * WPW + lane / PN;
    const bool live = w0 < VNWIN;
    const int w = live ? w0 : w0 % VNWIN;                   // FULL_WAVES: surplus lanes shadow another window and store nothing
    G1X tot;
    g1x_set_inf(tot);
    if (i < jobs.j[m].n) {
        const uint32_t* s = jobs.j[m].scalars + (size_t)i * 8;
        const int bit = VC * w - 1;
        uint32_t v;
        if (bit < 0) {
            v = (s[0] << 1) & 0x1ffu;
        } else {
            const int wi = bit >> 5, off = bit & 31;
            v = s[wi] >> off;
            if (off + VC + 1 > 32 && wi + 1 < 8) v |= s[wi + 1] << (32 - off);
            v &= 0x1ffu;
        }
        int d = (int)((v + 1) >> 1);                       // Booth digit of window w, as in k_vmsm_window
        if (v >> VC) d -= (1 << VC);
        G1Aff p;
        load_affine_v(p, jobs.j[m].points + (size_t)i * 32);
        const int a = d < 0 ? -d : d;
        if (a != 0 && !p.inf) {
            if (d < 0) Fp::template neg<0>(p.y, p.y);
            bool started = false;
#pragma unroll 1
            for (int b = VC; b >= 0; b--) {                // |d| <= 2^VC
                if (started) g1x_dbl(tot, tot);
                if ((a >> b) & 1) {
                    if (started) g1x_madd(tot, p.x, p.y);
                    else g1x_from_affine(tot, p.x, p.y);
                    started = true;
                }
            }
        }
    }
#pragma unroll 1
    for (int sh = PN / 2; sh >= 1; sh >>= 1) {
        G1X o;
        shfl_xor_xyzz_v(o, tot, sh);
        g1x_add(tot, tot, o);
    }
    if (!live || i != 0) return;
    uint32_t* dst = out + ((size_t)m * VNWIN + w) * 48;
    fp_t c;
    uint32_t ww[12];
    const fp_t* srcf[4] = {&tot.x, &tot.y, &tot.zz, &tot.zzz};
    for (int k = 0; k < 4; k++) {
        if (g1x_is_inf(tot)) {
            for (int qq = 0; qq < 12; qq++) dst[12 * k + qq] = 0;
        } else {
            Fp::to_plain(c, *srcf[k]);
            Fp::to_words32(ww, c);
            for (int qq = 0; qq < 12; qq++) dst[12 * k + qq] = ww[qq];
        }
    }
}
// window sum = sum of its slices; exported as canonical plain coordinates (all-zero = identity) for the host's Horner step
__global__ void __launch_bounds__(64) k_vmsm_combine(VmsmJobs jobs, int n_msm, const uint32_t* __restrict__ parts, uint32_t* __restrict__ out /* [msm][VNWIN][48] */) {
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= n_msm * VNWIN) return;
    const int m = g / VNWIN;
    const uint32_t* src = parts + (size_t)g * VMSM_KMAX * 64;
    G1X tot, q;
    g1x_set_inf(tot);
    for (int z = 0; z < jobs.k[m]; z++) {
#pragma unroll
        for (int i = 0; i < 14; i++) { q.x.l[i] = src[z * 64 + i]; q.y.l[i] = src[z * 64 + 14 + i]; q.zz.l[i] = src[z * 64 + 28 + i]; q.zzz.l[i] = src[z * 64 + 42 + i]; }
        g1x_add_nv(tot, tot, q);
    }
    uint32_t* dst = out + (size_t)g * 48;
    fp_t c;
    uint32_t ww[12];
    const fp_t* srcf[4] = {&tot.x, &tot.y, &tot.zz, &tot.zzz};
    for (int k = 0; k < 4; k++) {
        if (g1x_is_inf(tot)) {
            for (int qq = 0; qq < 12; qq++) dst[12 * k + qq] = 0;
        } else {
            Fp::to_plain(c, *srcf[k]);
            Fp::to_words32(ww, c);
            for (int qq = 0; qq < 12; qq++) dst[12 * k + qq] = ww[qq];
        }
    }
}

__global__ void __launch_bounds__(256) k_build_roots(uint32_t* __restrict__ roots_brp, uint32_t* __restrict__ inv_roots_brp, int log_n) {
    const int n = 1 << log_n;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // generator of the order-n subgroup: root_2^32 ^ (2^(32-log_n))
    fr_t g;
    Fr::set(g, FR_ROOT_2_32);
    Fr::reduce_once(g, g);
    for (int s = 0; s < 32 - log_n; s++) fr_mul_n(g, g, g);
    // w^i
    fr_t acc, base = g;
    Fr::set_one(acc);
    Fr::reduce_once(acc, acc);
    for (uint32_t e = (uint32_t)i; e; e >>= 1) {
        if (e & 1) fr_mul_n(acc, acc, base);
        fr_mul_n(base, base, base);
    }
    const int pos = (int)(__brev((uint32_t)i) >> (32 - log_n));
    store_fr12(roots_brp + (size_t)pos * 12, acc);
    // inverse = w^(n-i)
    fr_t inv, b2 = g;
    Fr::set_one(inv);
    Fr::reduce_once(inv, inv);
    for (uint32_t e = (uint32_t)((n - i) & (n - 1)); e; e >>= 1) {
        if (e & 1) fr_mul_n(inv, inv, b2);
        fr_mul_n(b2, b2, b2);
    }
    store_fr12(inv_roots_brp + (size_t)pos * 12, inv);
}

// 32-byte big-endian scalars -> Montgomery rows (canonical check); used for caller-supplied z, y
__global__ void __launch_bounds__(64) k_scalars_to_fr(const uint8_t* __restrict__ in32, int n, uint32_t* __restrict__ out12, int* __restrict__ st) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    uint32_t w[8];
    load_be32_words(w, in32 + (size_t)i * 32);
    if (Fr::words32_geq_mod(w)) { st[i] = MKZG_ERR_NON_CANONICAL_SCALAR; return; }
    st[i] = 0;
    fr_t f;
    fr_from_words(f, w);
    store_fr12(out12 + (size_t)i * 12, f);
}
// Montgomery rows -> 32-byte big-endian scalars
__global__ void __launch_bounds__(64) k_fr_to_scalars(const uint32_t* __restrict__ in12, int n, uint8_t* __restrict__ out32) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    fr_t f, pl;
    load_fr12(f, in12 + (size_t)i * 12);
    Fr::to_plain(pl, f);
    uint32_t w[8];
    Fr::to_words32(w, pl);
    uint32_t* o = reinterpret_cast<uint32_t*>(out32 + (size_t)i * 32);
#pragma unroll
    for (int q = 0; q < 8; q++) o[q] = __builtin_bswap32(w[7 - q]);
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
void challenge_device(const uint8_t* d_blobs, const uint8_t* d_commitments, int n, uint32_t* d_z, hipStream_t s, uint32_t* d_zrows) {
    if (n > 0) hipLaunchKernelGGL(k_challenge_sha256, dim3((n + 63) / 64), dim3(256), 0, s, d_blobs, d_commitments, n, d_z, d_zrows);
}
void challenge_jobs_device(ShaJobs& jobs, hipStream_t s) {
    int blocks = 0;
    for (int j = 0; j < jobs.count; j++) {
        jobs.first_block[j] = blocks;
        blocks += (jobs.n[j] + 63) / 64;
    }
    if (blocks > 0) hipLaunchKernelGGL(k_challenge_sha256_jobs, dim3(blocks), dim3(256), 0, s, jobs);
}
size_t blob_eval_scratch_words() { return (size_t)EVT_ROWS * 12; }
void blob_eval_jobs_device(EvalJobs& jobs, const uint32_t* d_roots, hipStream_t s) {
    int blocks = 0;
    for (int j = 0; j < jobs.count; j++) {
        jobs.first_block[j] = blocks;
        blocks += jobs.n[j];
    }
    if (blocks > 0) hipLaunchKernelGGL((k_blob_eval_tree_jobs<EVT_LOG>), dim3(blocks), dim3(4096 >> EVT_LOG), 0, s, jobs, d_roots);
}
int vmsm_calls_device(VmsmCalls& calls, hipStream_t s) {
    int kmax = 1;
    if (calls.count < 1 || calls.count > DEEP_MAX_JOBS) return -1;
    for (int j = 0; j < calls.count; j++) {
        if (calls.n[j] < 1 || calls.n[j] > VMAXN) return -1;
        calls.k[j] = std::min(VMSM_KMAX, std::max(1, calls.n[j] / 2048));
        kmax = std::max(kmax, calls.k[j]);
    }
    hipLaunchKernelGGL(k_vmsm_window_calls, dim3(VNWIN, 3 * calls.count, kmax), dim3(VNB), 0, s, calls);
    return 0;
}
void blob_eval_device(const uint8_t* d_blobs, const uint32_t* d_z, const uint32_t* d_roots, const uint32_t* d_inv_roots, int n, uint32_t* d_y, int* d_status, uint32_t* d_zrows,
                      hipStream_t s, bool rows_ready, const uint32_t* h_r8, uint32_t exp_offset, uint32_t* d_s1, uint32_t* d_s3, uint32_t* d_ry) {
#ifdef MKZG_EVAL_LANE_TRICK
    if (n > 0) hipLaunchKernelGGL((k_blob_eval<0>), dim3(n), dim3(EVB_T), 0, s, d_blobs, d_z, d_roots, d_inv_roots, d_y, (uint32_t*)nullptr, d_status);
#else
    (void)d_inv_roots;
    if (n > 0) {
        RlcOut rlc;
        rlc.s1 = h_r8 ? d_s1 : nullptr;
        rlc.s3 = d_s3;
        rlc.ry = d_ry;
        rlc.exp_offset = exp_offset;
        for (int i = 0; i < 8; i++) rlc.r.w[i] = h_r8 ? h_r8[i] : 0;
        if (!rows_ready) hipLaunchKernelGGL(k_blob_eval_prep, dim3((n + 63) / 64), dim3(64), 0, s, d_z, n, d_zrows);
        hipLaunchKernelGGL((k_blob_eval_tree<EVT_LOG>), dim3(n), dim3(4096 >> EVT_LOG), 0, s, d_blobs, d_zrows, d_roots, d_y, d_status, rlc);
    }
#endif
}
void blob_quotient_device(const uint8_t* d_blobs, const uint32_t* d_z, const uint32_t* d_roots, const uint32_t* d_inv_roots, int n, uint32_t* d_y, uint32_t* d_quot,
                          int* d_status, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL((k_blob_eval<1>), dim3(n), dim3(EVB_T), 0, s, d_blobs, d_z, d_roots, d_inv_roots, d_y, d_quot, d_status);
}
void poly_eval_device(const uint32_t* d_coeffs, const uint32_t* d_z, int n, uint32_t* d_y, const int* d_status, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_poly_eval, dim3(n), dim3(256), 0, s, d_coeffs, d_z, d_status, d_y);
}
__global__ void __launch_bounds__(G1_ROWS_MAX * 32) k_g1_put_rows(const G1Rows rp, uint32_t* __restrict__ aff_a, int* __restrict__ st_a, uint32_t* __restrict__ aff_b, int* __restrict__ st_b,
                                                                  int* __restrict__ sub_a, int* __restrict__ sub_b) {
    const int pt = threadIdx.x >> 5, w = threadIdx.x & 31;
    if (pt >= rp.na + rp.nb) return;
    const bool b = pt >= rp.na;
    const int i = b ? rp.first_b + pt - rp.na : rp.first_a + pt;
    (b ? aff_b : aff_a)[i * 32 + w] = rp.row[pt][w];
    if (w == 0) {
        int* sub = b ? sub_b : sub_a;
        if (sub) {
            (b ? st_b : st_a)[i] = rp.st[pt];
            sub[i] = rp.sub[pt];
        } else {
            (b ? st_b : st_a)[i] = rp.st[pt] ? rp.st[pt] : rp.sub[pt];
        }
    }
}
void g1_put_rows_device(const G1Rows& rp, uint32_t* d_aff_a, int* d_st_a, uint32_t* d_aff_b, int* d_st_b, hipStream_t s, int* d_sub_a, int* d_sub_b) {
    hipLaunchKernelGGL(k_g1_put_rows, dim3(1), dim3(G1_ROWS_MAX * 32), 0, s, rp, d_aff_a, d_st_a, d_aff_b, d_st_b, d_sub_a, d_sub_b);
}
void g1_validate2_device(const uint8_t* d_in48_a, int n_a, uint32_t* d_aff_a, int* d_st_a, const uint8_t* d_in48_b, int n_b, uint32_t* d_aff_b, int* d_st_b, hipStream_t s,
                         int mode) {
    const int n = n_a + n_b;
    const int per_block = (mode == 1 || mode == 3 || mode == 4) ? 64 : VAL_PB;
    if (n <= 0) return;
    const dim3 grid((n + per_block - 1) / per_block), block((mode == 0 || mode == 3) ? 128 : 64);
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_g1_validate<0>), grid, block, 0, s, d_in48_a, n_a, d_aff_a, d_st_a, d_in48_b, n_b, d_aff_b, d_st_b); break;
        case 1: hipLaunchKernelGGL((k_g1_validate<1>), grid, block, 0, s, d_in48_a, n_a, d_aff_a, d_st_a, d_in48_b, n_b, d_aff_b, d_st_b); break;
        case 2: hipLaunchKernelGGL((k_g1_validate<2>), grid, block, 0, s, d_in48_a, n_a, d_aff_a, d_st_a, d_in48_b, n_b, d_aff_b, d_st_b); break;
        case 3: hipLaunchKernelGGL((k_g1_validate_lean<3>), grid, block, 0, s, d_in48_a, n_a, d_aff_a, d_st_a, d_in48_b, n_b, d_aff_b, d_st_b); break;
        default: hipLaunchKernelGGL((k_g1_validate_lean<4>), grid, block, 0, s, d_in48_a, n_a, d_aff_a, d_st_a, d_in48_b, n_b, d_aff_b, d_st_b); break;
    }
}
void rlc_scalars_device(const uint32_t* h_r8, const uint32_t* d_z, const uint32_t* d_y, int n, uint32_t exp_offset, uint32_t* d_s1, uint32_t* d_s3, uint32_t* d_e,
                        int mode, hipStream_t s) {
    RWords r;                                  // by value in the kernel arguments: a 32-byte upload is a copy KERNEL that reads host memory, and its
    for (int i = 0; i < 8; i++) r.w[i] = h_r8[i];   // system-scope acquire waited for whatever else was running (cell verifier: 0.2 ms per call in the trace)
    hipLaunchKernelGGL(k_rlc_scalars, dim3(1), dim3(EV_T), 0, s, r, d_z, d_y, n, exp_offset, d_s1, d_s3, d_e, mode);
}
int vmsm_windows_device(const uint32_t* const* d_scalars, const uint32_t* const* d_points, const int* n_per_job, int n_msm, uint32_t* d_parts, uint32_t* d_out, hipStream_t s,
                        int points_per_slice, int latency) {
    if (n_msm < 1 || n_msm > 4) return -1;
    VmsmJobs jobs;
    int kmax = 1;
    for (int m = 0; m < 4; m++) {
        const int q = m < n_msm ? m : 0;
        if (n_per_job[q] < 0 || n_per_job[q] > VMAXN) return -1;
        jobs.j[m].scalars = d_scalars[q];
        jobs.j[m].points = d_points[q];
        jobs.j[m].n = n_per_job[q];
        jobs.k[m] = std::min(VMSM_KMAX, std::max(1, n_per_job[q] / points_per_slice));   // 2048 points per slice: 8192-point jobs fill the GPU in one round of workgroups
        if (m < n_msm) kmax = std::max(kmax, jobs.k[m]);
    }
    int nmax = 0;
    for (int m = 0; m < n_msm; m++) nmax = std::max(nmax, n_per_job[m]);
    if (nmax >= 1 && nmax <= 16) {                           // the single-proof verifiers, the batch verifier on the blobs of a block
        const int pn = nmax <= 1 ? 1 : (nmax <= 2 ? 2 : (nmax <= 4 ? 4 : (nmax <= 8 ? 8 : 16)));
        const dim3 grid((VNWIN * pn + 63) / 64, n_msm);
        switch (pn) {
            case 1: hipLaunchKernelGGL((k_vmsm_small<1>), grid, dim3(64), 0, s, jobs, d_out); break;
            case 2: hipLaunchKernelGGL((k_vmsm_small<2>), grid, dim3(64), 0, s, jobs, d_out); break;
            case 4: hipLaunchKernelGGL((k_vmsm_small<4>), grid, dim3(64), 0, s, jobs, d_out); break;
            case 8: hipLaunchKernelGGL((k_vmsm_small<8>), grid, dim3(64), 0, s, jobs, d_out); break;
            default: hipLaunchKernelGGL((k_vmsm_small<16>), grid, dim3(64), 0, s, jobs, d_out); break;
        }
        return 0;
    }
    bool quad = latency != 0 && VNWIN * n_msm * kmax <= 256;                      // somebody waits for this launch and it leaves the chip mostly empty: a quad per bucket
    for (int m = 0; m < n_msm; m++) quad = quad && (n_per_job[m] + jobs.k[m] - 1) / jobs.k[m] <= VQ_MAXN;
    if (quad) hipLaunchKernelGGL(k_vmsm_window_quad, dim3(VNWIN, n_msm, kmax), dim3(VQT), 0, s, jobs, d_parts);
    else { hipLaunchKernelGGL(k_vmsm_window, dim3(VNWIN, n_msm, kmax), dim3(VNB), 0, s, jobs, d_parts, d_out); return 0; }   // combines its slices itself (tickets)
    hipLaunchKernelGGL(k_vmsm_combine, dim3((n_msm * VNWIN + 63) / 64), dim3(64), 0, s, jobs, n_msm, d_parts, d_out);
    return 0;
}
// ------------------------------------------------------------------------------------------
// gnark-crypto in-memory layouts (the finer seam multiexp.MultiExpG1, reference internal/multiexp/multiexp.go:20-26):
//   fr.Element = 4 x u64 little-endian limbs, Montgomery with R = 2^256;  G1Affine = X, Y as fp.Element = 6 x u64 limbs,
//   Montgomery with R = 2^384, identity = (0, 0).
// -> 8-word plain scalars and radix-2^29 affine rows, the inputs of k_vmsm_window.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_gnark_to_vmsm(const uint32_t* __restrict__ scalars_mont /* n x 8 words */, const uint32_t* __restrict__ points_mont /* n x 24 words */, int n,
                                                      uint32_t* __restrict__ out_scalars8, uint32_t* __restrict__ out_aff) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    {   // s = a * 2^-256 mod r: one Montgomery multiplication (radix 2^261) by the plain constant 2^5
        uint32_t w[8];
#pragma unroll
        for (int q = 0; q < 8; q++) w[q] = scalars_mont[(size_t)i * 8 + q];
        fr_t a, k, t;
        Fr::from_words32(a, w);
        Fr::set_zero(k);
        k.l[0] = 32;
        Fr::mul(t, a, k);
        Fr::reduce_once(t, t);
        Fr::to_words32(w, t);
#pragma unroll
        for (int q = 0; q < 8; q++) out_scalars8[(size_t)i * 8 + q] = w[q];
    }
    {   // x 2^384 -> x 2^406: one multiplication by the constant FP_R406_OVER_R384_M
        uint32_t wx[12], wy[12];
        uint32_t any = 0;
#pragma unroll
        for (int q = 0; q < 12; q++) { wx[q] = points_mont[(size_t)i * 24 + q]; wy[q] = points_mont[(size_t)i * 24 + 12 + q]; any |= wx[q] | wy[q]; }
        G1Aff a;
        fp_t c, t;
        Fp::set(c, FP_R406_OVER_R384_M);
        Fp::from_words32(t, wx);
        Fp::mul(t, t, c);
        Fp::canon(a.x, t);
        Fp::from_words32(t, wy);
        Fp::mul(t, t, c);
        Fp::canon(a.y, t);
        a.inf = any == 0 ? 1u : 0u;
        uint32_t row[32];
#pragma unroll
        for (int q = 0; q < 14; q++) { row[q] = a.x.l[q]; row[14 + q] = a.y.l[q]; }
        row[28] = a.inf; row[29] = 0; row[30] = 0; row[31] = 0;
        uint4* dst = reinterpret_cast<uint4*>(out_aff + (size_t)i * 32);
#pragma unroll
        for (int q = 0; q < 8; q++) dst[q] = make_uint4(row[4 * q], row[4 * q + 1], row[4 * q + 2], row[4 * q + 3]);
    }
}
void gnark_to_vmsm_device(const uint32_t* d_scalars_mont, const uint32_t* d_points_mont, int n, uint32_t* d_scalars8, uint32_t* d_aff, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_gnark_to_vmsm, dim3((n + 63) / 64), dim3(64), 0, s, d_scalars_mont, d_points_mont, n, d_scalars8, d_aff);
}
void build_roots_device(uint32_t* d_roots_brp, uint32_t* d_inv_roots_brp, int log_n, hipStream_t s) {
    int n = 1 << log_n;
    hipLaunchKernelGGL(k_build_roots, dim3((n + 255) / 256), dim3(256), 0, s, d_roots_brp, d_inv_roots_brp, log_n);
}
void scalars_to_fr_device(const uint8_t* d_in32, int n, uint32_t* d_out12, int* d_status, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_scalars_to_fr, dim3((n + 63) / 64), dim3(64), 0, s, d_in32, n, d_out12, d_status);
}
void fr_to_scalars_device(const uint32_t* d_in12, int n, uint8_t* d_out32, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_fr_to_scalars, dim3((n + 63) / 64), dim3(64), 0, s, d_in12, n, d_out32);
}

}  // namespace kzg
