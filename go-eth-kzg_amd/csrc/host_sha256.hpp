// host_sha256.hpp -- SHA-256 for the Fiat-Shamir challenge of the host-pointer verifiers (reference fiatshamir.go:22-33).
//
// Why the host has one at all: a blob handed over as a host pointer must cross PCIe (131 KiB, ~2.6 us at 50 GB/s) before the device
// can hash it, and the device's hash of one transcript is a 2050-compression serial chain (2.7 ms whatever the batch size).  A host
// core with SHA extensions hashes the same transcript in ~70 us, so a pool of cores hashes a batch WHILE it uploads and the device
// starts from finished challenges.  Device-resident batches keep the device hash (verify_kernels.hip k_challenge_sha256).
// The compression function comes in two forms: x86 SHA extensions (runtime-detected) and portable C.
#pragma once
#include <stdint.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace kzg {

static const uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74,
    0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d,
    0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e,
    0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5,
    0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static inline uint32_t sha_ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static inline void sha256_blocks_c(uint32_t st[8], const uint8_t* p, size_t nblk) {
    while (nblk--) {
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = sha_ror(w[i - 15], 7) ^ sha_ror(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = sha_ror(w[i - 2], 17) ^ sha_ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = h + (sha_ror(e, 6) ^ sha_ror(e, 11) ^ sha_ror(e, 25)) + ((e & f) ^ (~e & g)) + SHA256_K[i] + w[i];
            const uint32_t t2 = (sha_ror(a, 2) ^ sha_ror(a, 13) ^ sha_ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
        p += 64;
    }
}

#if defined(__x86_64__)
// Four rounds per sha256rnds2 pair; the message schedule of rounds 16.. from sha256msg1 / sha256msg2:
//   W[t..t+3] = msg2(msg1(W[t-16..t-13], W[t-12..t-9]) + (W[t-7..t-4]), W[t-4..t-1])
__attribute__((target("sha,sse4.1,ssse3"))) static inline void sha256_blocks_ni(uint32_t st[8], const uint8_t* p, size_t nblk) {
    const __m128i BSWAP = _mm_set_epi64x(0x0c0d0e0f08090a0bLL, 0x0405060700010203LL);
    __m128i tmp = _mm_loadu_si128((const __m128i*)&st[0]);       // d c b a
    __m128i s1 = _mm_loadu_si128((const __m128i*)&st[4]);        // h g f e
    tmp = _mm_shuffle_epi32(tmp, 0xB1);                          // c d a b
    s1 = _mm_shuffle_epi32(s1, 0x1B);                            // e f g h
    __m128i s0 = _mm_alignr_epi8(tmp, s1, 8);                    // a b e f
    s1 = _mm_blend_epi16(s1, tmp, 0xF0);                         // c d g h
    while (nblk--) {
        const __m128i save0 = s0, save1 = s1;
        __m128i m[4];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            __m128i w;
            if (i < 4) {
                w = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16 * i)), BSWAP);
            } else {
                __m128i t = _mm_sha256msg1_epu32(m[i & 3], m[(i + 1) & 3]);
                t = _mm_add_epi32(t, _mm_alignr_epi8(m[(i + 3) & 3], m[(i + 2) & 3], 4));
                w = _mm_sha256msg2_epu32(t, m[(i + 3) & 3]);
            }
            m[i & 3] = w;
            __m128i msg = _mm_add_epi32(w, _mm_loadu_si128((const __m128i*)&SHA256_K[4 * i]));
            s1 = _mm_sha256rnds2_epu32(s1, s0, msg);
            msg = _mm_shuffle_epi32(msg, 0x0E);
            s0 = _mm_sha256rnds2_epu32(s0, s1, msg);
        }
        s0 = _mm_add_epi32(s0, save0);
        s1 = _mm_add_epi32(s1, save1);
        p += 64;
    }
    tmp = _mm_shuffle_epi32(s0, 0x1B);                           // f e b a
    s1 = _mm_shuffle_epi32(s1, 0xB1);                            // d c h g
    s0 = _mm_blend_epi16(tmp, s1, 0xF0);                         // d c b a
    s1 = _mm_alignr_epi8(s1, tmp, 8);                            // h g f e
    _mm_storeu_si128((__m128i*)&st[0], s0);
    _mm_storeu_si128((__m128i*)&st[4], s1);
}
// Two independent messages interleaved in one instruction stream: sha256rnds2 is a 4-cycle dependent chain per two rounds, so a lone
// message leaves the unit idle more than half of the time; the second message's rounds fill those slots (measured per core in
// tools/ubench_h2d.cpp; the gather threads of the host-pointer verifier hash their transcripts in pairs).
__attribute__((target("sha,sse4.1,ssse3"))) static inline void sha256_blocks_ni_x2(uint32_t sta[8], const uint8_t* pa, uint32_t stb[8], const uint8_t* pb, size_t nblk) {
    const __m128i BSWAP = _mm_set_epi64x(0x0c0d0e0f08090a0bLL, 0x0405060700010203LL);
    __m128i a0, a1, b0, b1;
    {
        __m128i t = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i*)&sta[0]), 0xB1);
        __m128i u = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i*)&sta[4]), 0x1B);
        a0 = _mm_alignr_epi8(t, u, 8);
        a1 = _mm_blend_epi16(u, t, 0xF0);
        t = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i*)&stb[0]), 0xB1);
        u = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i*)&stb[4]), 0x1B);
        b0 = _mm_alignr_epi8(t, u, 8);
        b1 = _mm_blend_epi16(u, t, 0xF0);
    }
    while (nblk--) {
        const __m128i sa0 = a0, sa1 = a1, sb0 = b0, sb1 = b1;
        __m128i ma[4], mb[4];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            __m128i wa, wb;
            if (i < 4) {
                wa = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pa + 16 * i)), BSWAP);
                wb = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pb + 16 * i)), BSWAP);
            } else {
                __m128i t = _mm_sha256msg1_epu32(ma[i & 3], ma[(i + 1) & 3]);
                __m128i u = _mm_sha256msg1_epu32(mb[i & 3], mb[(i + 1) & 3]);
                t = _mm_add_epi32(t, _mm_alignr_epi8(ma[(i + 3) & 3], ma[(i + 2) & 3], 4));
                u = _mm_add_epi32(u, _mm_alignr_epi8(mb[(i + 3) & 3], mb[(i + 2) & 3], 4));
                wa = _mm_sha256msg2_epu32(t, ma[(i + 3) & 3]);
                wb = _mm_sha256msg2_epu32(u, mb[(i + 3) & 3]);
            }
            ma[i & 3] = wa;
            mb[i & 3] = wb;
            const __m128i k = _mm_loadu_si128((const __m128i*)&SHA256_K[4 * i]);
            __m128i ga = _mm_add_epi32(wa, k), gb = _mm_add_epi32(wb, k);
            a1 = _mm_sha256rnds2_epu32(a1, a0, ga);
            b1 = _mm_sha256rnds2_epu32(b1, b0, gb);
            ga = _mm_shuffle_epi32(ga, 0x0E);
            gb = _mm_shuffle_epi32(gb, 0x0E);
            a0 = _mm_sha256rnds2_epu32(a0, a1, ga);
            b0 = _mm_sha256rnds2_epu32(b0, b1, gb);
        }
        a0 = _mm_add_epi32(a0, sa0);
        a1 = _mm_add_epi32(a1, sa1);
        b0 = _mm_add_epi32(b0, sb0);
        b1 = _mm_add_epi32(b1, sb1);
        pa += 64;
        pb += 64;
    }
    {
        __m128i t = _mm_shuffle_epi32(a0, 0x1B), u = _mm_shuffle_epi32(a1, 0xB1);
        _mm_storeu_si128((__m128i*)&sta[0], _mm_blend_epi16(t, u, 0xF0));
        _mm_storeu_si128((__m128i*)&sta[4], _mm_alignr_epi8(u, t, 8));
        t = _mm_shuffle_epi32(b0, 0x1B);
        u = _mm_shuffle_epi32(b1, 0xB1);
        _mm_storeu_si128((__m128i*)&stb[0], _mm_blend_epi16(t, u, 0xF0));
        _mm_storeu_si128((__m128i*)&stb[4], _mm_alignr_epi8(u, t, 8));
    }
}
// SIXTEEN independent messages, one per 32-bit lane of a 512-bit register (AVX-512F/BW: vprord rotations, vpternlogd for the sigmas, Ch
// and Maj): the message blocks are loaded 64 bytes per message, byte-swapped and transposed (16 x 16 words), then every round is ~9 vector
// instructions for 16 messages.  *Measured* (tools/ubench_h2d.cpp): faster per transcript than two interleaved SHA-NI chains on the
// cores that have full-width AVX-512; the host-pointer verifier's hash tasks take it when a call brings enough transcripts to give every
// pool thread sixteen.  p[i] + off: start of message i's next block; all messages advance together.
__attribute__((target("avx512f,avx512bw"))) static inline void sha256_transpose16(__m512i r[16]) {
    __m512i t[16];
    for (int i = 0; i < 16; i += 2) { t[i] = _mm512_unpacklo_epi32(r[i], r[i + 1]); t[i + 1] = _mm512_unpackhi_epi32(r[i], r[i + 1]); }
    for (int i = 0; i < 16; i += 4) {
        r[i] = _mm512_unpacklo_epi64(t[i], t[i + 2]); r[i + 1] = _mm512_unpackhi_epi64(t[i], t[i + 2]);
        r[i + 2] = _mm512_unpacklo_epi64(t[i + 1], t[i + 3]); r[i + 3] = _mm512_unpackhi_epi64(t[i + 1], t[i + 3]);
    }
    for (int i = 0; i < 4; i++) {
        t[i] = _mm512_shuffle_i32x4(r[i], r[i + 4], 0x88); t[i + 4] = _mm512_shuffle_i32x4(r[i], r[i + 4], 0xdd);
        t[i + 8] = _mm512_shuffle_i32x4(r[i + 8], r[i + 12], 0x88); t[i + 12] = _mm512_shuffle_i32x4(r[i + 8], r[i + 12], 0xdd);
    }
    for (int i = 0; i < 4; i++) {
        r[i] = _mm512_shuffle_i32x4(t[i], t[i + 8], 0x88); r[i + 8] = _mm512_shuffle_i32x4(t[i], t[i + 8], 0xdd);
        r[i + 4] = _mm512_shuffle_i32x4(t[i + 4], t[i + 12], 0x88); r[i + 12] = _mm512_shuffle_i32x4(t[i + 4], t[i + 12], 0xdd);
    }
}
__attribute__((target("avx512f,avx512bw"))) static inline void sha256_blocks_avx512_x16(__m512i st[8], const uint8_t* const p[16], size_t off, size_t nblk) {
    const __m512i BSW = _mm512_set4_epi32(0x0c0d0e0f, 0x08090a0b, 0x04050607, 0x00010203);
    __m512i a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (size_t k = 0; k < nblk; k++) {
        __m512i w[16];
        for (int i = 0; i < 16; i++) w[i] = _mm512_shuffle_epi8(_mm512_loadu_si512((const void*)(p[i] + off + 64 * k)), BSW);
        sha256_transpose16(w);                    // w[j], lane i = word j of message i's block
        const __m512i sa = a, sb = b, sc = c, sd = d, se = e, sf = f, sg = g, sh = h;
#pragma GCC unroll 64
        for (int i = 0; i < 64; i++) {
            if (i >= 16) {
                const __m512i w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
                const __m512i s0 = _mm512_ternarylogic_epi32(_mm512_ror_epi32(w15, 7), _mm512_ror_epi32(w15, 18), _mm512_srli_epi32(w15, 3), 0x96);
                const __m512i s1 = _mm512_ternarylogic_epi32(_mm512_ror_epi32(w2, 17), _mm512_ror_epi32(w2, 19), _mm512_srli_epi32(w2, 10), 0x96);
                w[i & 15] = _mm512_add_epi32(_mm512_add_epi32(w[i & 15], s0), _mm512_add_epi32(w[(i - 7) & 15], s1));
            }
            const __m512i S1 = _mm512_ternarylogic_epi32(_mm512_ror_epi32(e, 6), _mm512_ror_epi32(e, 11), _mm512_ror_epi32(e, 25), 0x96);
            const __m512i ch = _mm512_ternarylogic_epi32(e, f, g, 0xca);
            const __m512i t1 = _mm512_add_epi32(_mm512_add_epi32(h, S1), _mm512_add_epi32(_mm512_add_epi32(ch, _mm512_set1_epi32((int)SHA256_K[i])), w[i & 15]));
            const __m512i S0 = _mm512_ternarylogic_epi32(_mm512_ror_epi32(a, 2), _mm512_ror_epi32(a, 13), _mm512_ror_epi32(a, 22), 0x96);
            const __m512i mj = _mm512_ternarylogic_epi32(a, b, c, 0xe8);
            h = g; g = f; f = e; e = _mm512_add_epi32(d, t1); d = c; c = b; b = a; a = _mm512_add_epi32(t1, _mm512_add_epi32(S0, mj));
        }
        a = _mm512_add_epi32(a, sa); b = _mm512_add_epi32(b, sb); c = _mm512_add_epi32(c, sc); d = _mm512_add_epi32(d, sd);
        e = _mm512_add_epi32(e, se); f = _mm512_add_epi32(f, sf); g = _mm512_add_epi32(g, sg); h = _mm512_add_epi32(h, sh);
    }
    st[0] = a; st[1] = b; st[2] = c; st[3] = d; st[4] = e; st[5] = f; st[6] = g; st[7] = h;
}
static inline bool sha256_have_avx512() {
    static const bool have = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw");
    return have;
}
static inline bool sha256_have_ni() {
    static const bool have = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3");
    return have;
}
#else
static inline bool sha256_have_ni() { return false; }
static inline bool sha256_have_avx512() { return false; }
#endif

static inline void sha256_blocks(uint32_t st[8], const uint8_t* p, size_t nblk) {
#if defined(__x86_64__)
    if (sha256_have_ni()) { sha256_blocks_ni(st, p, nblk); return; }
#endif
    sha256_blocks_c(st, p, nblk);
}
// the same number of blocks of two messages
static inline void sha256_blocks_x2(uint32_t sta[8], const uint8_t* pa, uint32_t stb[8], const uint8_t* pb, size_t nblk) {
#if defined(__x86_64__)
    if (sha256_have_ni()) { sha256_blocks_ni_x2(sta, pa, stb, pb, nblk); return; }
#endif
    sha256_blocks_c(sta, pa, nblk);
    sha256_blocks_c(stb, pb, nblk);
}
static inline void sha256_init(uint32_t st[8]) {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(st, iv, sizeof(iv));
}
// one-shot digest (tests)
static inline void sha256_digest(uint8_t out[32], const uint8_t* data, size_t len) {
    uint32_t st[8];
    sha256_init(st);
    sha256_blocks(st, data, len / 64);
    uint8_t tail[128] = {0};
    const size_t rem = len % 64;
    memcpy(tail, data + len - rem, rem);
    tail[rem] = 0x80;
    const size_t tl = rem + 9 <= 64 ? 64 : 128;
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_blocks(st, tail, tl / 64);
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16); out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i]; }
}

// compute_challenge (reference fiatshamir.go:22-33): SHA-256("FSBLOBVERIFY_V1_" || u128_be(4096) || blob || commitment) reduced
// modulo r, written as 32 big-endian bytes (canonical).  The blob is hashed in place: only the first 32 and the last 32 bytes of it
// are copied (they share a compression block with the 32-byte header and with the commitment).
static inline void challenge_head_block(uint8_t blk[64], const uint8_t* blob) {
    memcpy(blk, "FSBLOBVERIFY_V1_", 16);
    memset(blk + 16, 0, 16);
    blk[30] = 0x10;                                        // 4096 = 0x1000, big-endian in the last 8 of the 16 bytes
    memcpy(blk + 32, blob, 32);
}
static inline void challenge_tail_blocks(uint8_t blk[128], const uint8_t* blob, const uint8_t commitment48[48]) {
    memset(blk, 0, 128);
    memcpy(blk, blob + 131072 - 32, 32);
    memcpy(blk + 32, commitment48, 48);
    blk[80] = 0x80;
    const uint64_t bits = (uint64_t)(32 + 131072 + 48) * 8;
    for (int i = 0; i < 8; i++) blk[127 - i] = (uint8_t)(bits >> (8 * i));
}
// digest as a 256-bit big-endian integer, reduced mod r (r > 2^254: at most two subtractions), as 32 big-endian bytes
static inline void challenge_reduce(uint8_t z32[32], const uint32_t st[8]) {
    static const uint32_t RW[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};   // little-endian words of r
    uint32_t w[8];
    for (int i = 0; i < 8; i++) w[i] = st[7 - i];
    for (int pass = 0; pass < 3; pass++) {
        bool ge = true;
        for (int i = 7; i >= 0; i--) {
            if (w[i] != RW[i]) { ge = w[i] > RW[i]; break; }
        }
        if (!ge) break;
        uint64_t br = 0;
        for (int i = 0; i < 8; i++) {
            const uint64_t d = (uint64_t)w[i] - RW[i] - br;
            w[i] = (uint32_t)d;
            br = (d >> 32) & 1;
        }
    }
    for (int i = 0; i < 8; i++) { z32[4 * i] = (uint8_t)(w[7 - i] >> 24); z32[4 * i + 1] = (uint8_t)(w[7 - i] >> 16); z32[4 * i + 2] = (uint8_t)(w[7 - i] >> 8); z32[4 * i + 3] = (uint8_t)w[7 - i]; }
}
static inline void host_compute_challenge(uint8_t z32[32], const uint8_t* blob /* 131072 B */, const uint8_t commitment48[48]) {
    uint32_t st[8];
    sha256_init(st);
    uint8_t blk[128];
    challenge_head_block(blk, blob);
    sha256_blocks(st, blk, 1);
    sha256_blocks(st, blob + 32, (131072 - 64) / 64);      // blob[32 .. 131040)
    challenge_tail_blocks(blk, blob, commitment48);
    sha256_blocks(st, blk, 2);
    challenge_reduce(z32, st);
}
// two transcripts at once (the two hash chains interleaved in one core's instruction stream: sha256_blocks_ni_x2)
static inline void host_compute_challenge_x2(uint8_t z32a[32], const uint8_t* bloba, const uint8_t cma[48], uint8_t z32b[32], const uint8_t* blobb, const uint8_t cmb[48]) {
    uint32_t sa[8], sb[8];
    sha256_init(sa);
    sha256_init(sb);
    uint8_t ka[128], kb[128];
    challenge_head_block(ka, bloba);
    challenge_head_block(kb, blobb);
    sha256_blocks_x2(sa, ka, sb, kb, 1);
    sha256_blocks_x2(sa, bloba + 32, sb, blobb + 32, (131072 - 64) / 64);
    challenge_tail_blocks(ka, bloba, cma);
    challenge_tail_blocks(kb, blobb, cmb);
    sha256_blocks_x2(sa, ka, sb, kb, 2);
    challenge_reduce(z32a, sa);
    challenge_reduce(z32b, sb);
}

#if defined(__x86_64__)
// sixteen transcripts at once (sha256_blocks_avx512_x16); the caller has checked sha256_have_avx512()
__attribute__((target("avx512f,avx512bw"))) static inline void host_compute_challenge_x16(uint8_t* z32x16 /* 16 x 32 B */, const uint8_t* const blobs[16], const uint8_t* const cms[16]) {
    uint32_t iv[8];
    sha256_init(iv);
    __m512i st[8];
    for (int q = 0; q < 8; q++) st[q] = _mm512_set1_epi32((int)iv[q]);
    alignas(64) uint8_t tmp[16][128];
    const uint8_t* p[16];
    for (int i = 0; i < 16; i++) { challenge_head_block(tmp[i], blobs[i]); p[i] = tmp[i]; }
    sha256_blocks_avx512_x16(st, p, 0, 1);
    for (int i = 0; i < 16; i++) p[i] = blobs[i] + 32;
    sha256_blocks_avx512_x16(st, p, 0, (131072 - 64) / 64);
    for (int i = 0; i < 16; i++) { challenge_tail_blocks(tmp[i], blobs[i], cms[i]); p[i] = tmp[i]; }
    sha256_blocks_avx512_x16(st, p, 0, 2);
    alignas(64) uint32_t o[8][16];
    for (int q = 0; q < 8; q++) _mm512_store_si512((void*)o[q], st[q]);
    for (int m = 0; m < 16; m++) {
        uint32_t d[8];
        for (int q = 0; q < 8; q++) d[q] = o[q][m];
        challenge_reduce(z32x16 + 32 * m, d);
    }
}
#endif

}  // namespace kzg
