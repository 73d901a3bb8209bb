// tests/host_shim/sha_shim.cpp -- C entry points over the product's host SHA-256 (go-eth-kzg_amd/csrc/host_sha256.hpp) for
// tests/test_host_sha256.py: both compression functions (SHA extensions when the CPU has them, portable C) and the Fiat-Shamir
// challenge of the host-pointer verifiers.
#include "../../go-eth-kzg_amd/csrc/host_sha256.hpp"
#include "../../go-eth-kzg_amd/csrc/host_copy.hpp"

extern "C" {
int shim_have_sha_ni() { return kzg::sha256_have_ni() ? 1 : 0; }
void shim_sha256(const uint8_t* data, size_t len, uint8_t out[32]) { kzg::sha256_digest(out, data, len); }
// portable compression function only
void shim_sha256_c(const uint8_t* data, size_t len, uint8_t out[32]) {
    uint32_t st[8];
    kzg::sha256_init(st);
    kzg::sha256_blocks_c(st, data, len / 64);
    uint8_t tail[128] = {0};
    const size_t rem = len % 64;
    memcpy(tail, data + len - rem, rem);
    tail[rem] = 0x80;
    const size_t tl = rem + 9 <= 64 ? 64 : 128;
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    kzg::sha256_blocks_c(st, tail, tl / 64);
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16); out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i]; }
}
void shim_challenge(const uint8_t* blob, const uint8_t* commitment48, uint8_t z32[32]) { kzg::host_compute_challenge(z32, blob, commitment48); }
// the gather's copy into a staging area (streaming stores when the destination is 32-byte aligned and the size a multiple of 128)
void shim_copy_to_staging(uint8_t* dst, const uint8_t* src, size_t n) { kzg::copy_to_staging(dst, src, n); }
// sixteen transcripts, one per AVX-512 lane (returns 0 when the CPU has no AVX-512: nothing written)
int shim_challenge_x16(const uint8_t* blobs16 /* 16 x 131072 */, const uint8_t* cms16 /* 16 x 48 */, uint8_t* z16 /* 16 x 32 */) {
#if defined(__x86_64__)
    if (!kzg::sha256_have_avx512()) return 0;
    const uint8_t* b[16];
    const uint8_t* c[16];
    for (int i = 0; i < 16; i++) { b[i] = blobs16 + (size_t)i * 131072; c[i] = cms16 + i * 48; }
    kzg::host_compute_challenge_x16(z16, b, c);
    return 1;
#else
    return 0;
#endif
}
// two transcripts interleaved in one instruction stream (the gather threads of the host-pointer verifier hash in pairs)
void shim_challenge_x2(const uint8_t* blob_a, const uint8_t* cm_a, uint8_t za[32], const uint8_t* blob_b, const uint8_t* cm_b, uint8_t zb[32]) {
    kzg::host_compute_challenge_x2(za, blob_a, cm_a, zb, blob_b, cm_b);
}
}
