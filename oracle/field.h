/*
 * oracle/field.h -- Fp (381-bit) and Fr (255-bit) Montgomery arithmetic, 64-bit limbs, plain C.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the CPU oracle that the HIP product path is checked
 * against; it is never linked into, imported by or executed from the product
 * (go-eth-kzg_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it.
 *
 * The reference (crate-crypto/go-eth-kzg) takes this arithmetic from
 * github.com/consensys/gnark-crypto v0.16.0 (go.mod:8), which is NOT present in
 * /root/reference.  What is restated here is the published algorithm (Montgomery CIOS,
 * little-endian 64-bit limbs, R = 2^384 resp. 2^256 -- the in-memory layout of gnark's
 * fp.Element / fr.Element); parity is pinned on the reference's own golden vectors
 * (tests/golden/, see tests/test_oracle_vectors.py).
 */
#ifndef KZGO_FIELD_H
#define KZGO_FIELD_H
#include <stdint.h>
#include <string.h>
#include "constants.h"

typedef unsigned __int128 u128;
typedef uint64_t u64;

typedef struct { u64 l[6]; } fp_t;
typedef struct { u64 l[4]; } fr_t;

/* ---------- generic N-limb helpers (N is a compile-time constant at every call site) ---------- */
static inline u64 kz_adc(u64 a, u64 b, u64 *c) { u128 t = (u128)a + b + *c; *c = (u64)(t >> 64); return (u64)t; }
static inline u64 kz_sbb(u64 a, u64 b, u64 *br) { u128 t = (u128)a - b - *br; *br = (u64)(t >> 64) & 1; return (u64)t; }
static inline u64 kz_mac(u64 acc, u64 a, u64 b, u64 *c) { u128 t = (u128)a * b + acc + *c; *c = (u64)(t >> 64); return (u64)t; }

static inline int bn_geq(int n, const u64 *a, const u64 *b) {
    for (int i = n - 1; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
    return 1;
}
static inline int bn_is_zero(int n, const u64 *a) { u64 t = 0; for (int i = 0; i < n; i++) t |= a[i]; return t == 0; }
static inline int bn_eq(int n, const u64 *a, const u64 *b) { u64 t = 0; for (int i = 0; i < n; i++) t |= a[i] ^ b[i]; return t == 0; }

static inline void mod_add(int n, u64 *o, const u64 *a, const u64 *b, const u64 *m) {
    u64 c = 0, t[6];
    for (int i = 0; i < n; i++) t[i] = kz_adc(a[i], b[i], &c);
    /* moduli have at least one spare top bit, so c == 0 */
    if (bn_geq(n, t, m)) { u64 br = 0; for (int i = 0; i < n; i++) t[i] = kz_sbb(t[i], m[i], &br); }
    for (int i = 0; i < n; i++) o[i] = t[i];
}
static inline void mod_sub(int n, u64 *o, const u64 *a, const u64 *b, const u64 *m) {
    u64 br = 0, t[6];
    for (int i = 0; i < n; i++) t[i] = kz_sbb(a[i], b[i], &br);
    if (br) { u64 c = 0; for (int i = 0; i < n; i++) t[i] = kz_adc(t[i], m[i], &c); }
    for (int i = 0; i < n; i++) o[i] = t[i];
}
/* Montgomery product a*b*R^-1 mod m (coarsely integrated operand scanning, "no-carry" variant:
 * valid because the top bit of both moduli is clear).  Generated per field by macro so that the
 * limb count is a literal and gcc fully unrolls the loops. */
#define KZ_DEFINE_MONT_MUL(NAME, NL, MOD, INV)                                                   \
    static inline void NAME(u64 *o, const u64 *a, const u64 *b) {                                \
        u64 t[NL] = {0};                                                                         \
        _Pragma("GCC unroll 8") for (int i = 0; i < NL; i++) {                                   \
            u64 A = 0, C = 0, bi = b[i];                                                         \
            u128 x = (u128)a[0] * bi + t[0];                                                     \
            A = (u64)(x >> 64);                                                                  \
            u64 q = (u64)x * INV;                                                                \
            u128 y = (u128)q * MOD[0] + (u64)x;                                                  \
            C = (u64)(y >> 64);                                                                  \
            _Pragma("GCC unroll 8") for (int j = 1; j < NL; j++) {                               \
                x = (u128)a[j] * bi + t[j] + A;                                                  \
                A = (u64)(x >> 64);                                                              \
                y = (u128)q * MOD[j] + (u64)x + C;                                               \
                C = (u64)(y >> 64);                                                              \
                t[j - 1] = (u64)y;                                                               \
            }                                                                                    \
            t[NL - 1] = C + A;                                                                   \
        }                                                                                        \
        if (bn_geq(NL, t, MOD)) { u64 br = 0; for (int i = 0; i < NL; i++) t[i] = kz_sbb(t[i], MOD[i], &br); } \
        for (int i = 0; i < NL; i++) o[i] = t[i];                                                \
    }
KZ_DEFINE_MONT_MUL(fp_mont_mul, 6, FP_MOD, FP_INV)
KZ_DEFINE_MONT_MUL(fr_mont_mul, 4, FR_MOD, FR_INV)

/* ---------- Fp ---------- */
static inline void fp_add(fp_t *o, const fp_t *a, const fp_t *b) { mod_add(6, o->l, a->l, b->l, FP_MOD); }
static inline void fp_sub(fp_t *o, const fp_t *a, const fp_t *b) { mod_sub(6, o->l, a->l, b->l, FP_MOD); }
static inline void fp_mul(fp_t *o, const fp_t *a, const fp_t *b) { fp_mont_mul(o->l, a->l, b->l); }
static inline void fp_sqr(fp_t *o, const fp_t *a) { fp_mont_mul(o->l, a->l, a->l); }
static inline int fp_is_zero(const fp_t *a) { return bn_is_zero(6, a->l); }
static inline int fp_eq(const fp_t *a, const fp_t *b) { return bn_eq(6, a->l, b->l); }
static inline void fp_zero(fp_t *o) { memset(o, 0, sizeof *o); }
static inline void fp_one(fp_t *o) { memcpy(o->l, FP_R, 48); }
static inline void fp_neg(fp_t *o, const fp_t *a) { fp_t z; fp_zero(&z); fp_sub(o, &z, a); }
static inline void fp_dbl(fp_t *o, const fp_t *a) { fp_add(o, a, a); }
static inline void fp_from_plain(fp_t *o, const u64 v[6]) { fp_t t, r2; memcpy(t.l, v, 48); memcpy(r2.l, FP_R2, 48); fp_mul(o, &t, &r2); }
static inline void fp_to_plain(u64 v[6], const fp_t *a) { fp_t one = {{1, 0, 0, 0, 0, 0}}, t; fp_mul(&t, a, &one); memcpy(v, t.l, 48); }
static inline void fp_set_u64(fp_t *o, u64 x) { u64 v[6] = {x, 0, 0, 0, 0, 0}; fp_from_plain(o, v); }
static inline void fp_pow(fp_t *o, const fp_t *a, const u64 *e, int nlimbs) {
    fp_t acc; fp_one(&acc);
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        fp_sqr(&acc, &acc);
        if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(&acc, &acc, a);
    }
    *o = acc;
}
static inline void fp_inv(fp_t *o, const fp_t *a) { fp_pow(o, a, FP_PM2, 6); } /* 0 -> 0 */
/* sqrt for p = 3 mod 4; returns 1 if a is a square */
static inline int fp_sqrt(fp_t *o, const fp_t *a) {
    fp_t s, c; fp_pow(&s, a, FP_SQRT_EXP, 6); fp_sqr(&c, &s);
    if (!fp_eq(&c, a)) return 0;
    *o = s; return 1;
}
/* 48 bytes big-endian -> Fp; returns 0 if value >= p */
static inline int fp_from_be48(fp_t *o, const uint8_t b[48]) {
    u64 v[6];
    for (int i = 0; i < 6; i++) { u64 w = 0; for (int j = 0; j < 8; j++) w = (w << 8) | b[(5 - i) * 8 + j]; v[i] = w; }
    if (bn_geq(6, v, FP_MOD)) return 0;
    fp_from_plain(o, v); return 1;
}
static inline void fp_to_be48(uint8_t b[48], const fp_t *a) {
    u64 v[6]; fp_to_plain(v, a);
    for (int i = 0; i < 6; i++) for (int j = 0; j < 8; j++) b[(5 - i) * 8 + j] = (uint8_t)(v[i] >> (56 - 8 * j));
}
/* "lexicographically largest": a > (p-1)/2 as integers */
static inline int fp_lex_largest(const fp_t *a) {
    u64 v[6]; fp_to_plain(v, a);
    return !bn_geq(6, FP_HALF, v); /* v > half */
}

/* ---------- Fr ---------- */
static inline void fr_add(fr_t *o, const fr_t *a, const fr_t *b) { mod_add(4, o->l, a->l, b->l, FR_MOD); }
static inline void fr_sub(fr_t *o, const fr_t *a, const fr_t *b) { mod_sub(4, o->l, a->l, b->l, FR_MOD); }
static inline void fr_mul(fr_t *o, const fr_t *a, const fr_t *b) { fr_mont_mul(o->l, a->l, b->l); }
static inline void fr_sqr(fr_t *o, const fr_t *a) { fr_mont_mul(o->l, a->l, a->l); }
static inline int fr_is_zero(const fr_t *a) { return bn_is_zero(4, a->l); }
static inline int fr_eq(const fr_t *a, const fr_t *b) { return bn_eq(4, a->l, b->l); }
static inline void fr_zero(fr_t *o) { memset(o, 0, sizeof *o); }
static inline void fr_one(fr_t *o) { memcpy(o->l, FR_R, 32); }
static inline void fr_neg(fr_t *o, const fr_t *a) { fr_t z; fr_zero(&z); fr_sub(o, &z, a); }
static inline void fr_from_plain(fr_t *o, const u64 v[4]) { fr_t t, r2; memcpy(t.l, v, 32); memcpy(r2.l, FR_R2, 32); fr_mul(o, &t, &r2); }
static inline void fr_to_plain(u64 v[4], const fr_t *a) { fr_t one = {{1, 0, 0, 0}}, t; fr_mul(&t, a, &one); memcpy(v, t.l, 32); }
static inline void fr_set_u64(fr_t *o, u64 x) { u64 v[4] = {x, 0, 0, 0}; fr_from_plain(o, v); }
static inline void fr_pow(fr_t *o, const fr_t *a, const u64 *e, int nlimbs) {
    fr_t acc; fr_one(&acc);
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        fr_sqr(&acc, &acc);
        if ((e[i >> 6] >> (i & 63)) & 1) fr_mul(&acc, &acc, a);
    }
    *o = acc;
}
static inline void fr_pow_u64(fr_t *o, const fr_t *a, u64 e) { fr_pow(o, a, &e, 1); }
static inline void fr_inv(fr_t *o, const fr_t *a) { fr_pow(o, a, FR_RM2, 4); } /* 0 -> 0, as gnark's Inverse */
static inline void be32_to_limbs(u64 v[4], const uint8_t b[32]) {
    for (int i = 0; i < 4; i++) { u64 w = 0; for (int j = 0; j < 8; j++) w = (w << 8) | b[(3 - i) * 8 + j]; v[i] = w; }
}
/* canonical decode (reference serialization.go:141, internal/utils/utils.go:45-50): 0 if >= r */
static inline int fr_from_be32_canonical(fr_t *o, const uint8_t b[32]) {
    u64 v[4]; be32_to_limbs(v, b);
    if (bn_geq(4, v, FR_MOD)) return 0;
    fr_from_plain(o, v); return 1;
}
/* reducing decode (reference fiatshamir.go:31, gnark fr.SetBytes): value mod r */
static inline void fr_from_be32_reduce(fr_t *o, const uint8_t b[32]) {
    u64 v[4]; be32_to_limbs(v, b);
    /* 2^256 / r < 3, so at most two subtractions */
    while (bn_geq(4, v, FR_MOD)) { u64 br = 0; for (int i = 0; i < 4; i++) v[i] = kz_sbb(v[i], FR_MOD[i], &br); }
    fr_from_plain(o, v);
}
static inline void fr_to_be32(uint8_t b[32], const fr_t *a) {
    u64 v[4]; fr_to_plain(v, a);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) b[(3 - i) * 8 + j] = (uint8_t)(v[i] >> (56 - 8 * j));
}
/* Montgomery batch inversion; zeros are skipped and stay zero (gnark fr.BatchInvert semantics,
 * relied on by reference internal/kzg/kzg_prove.go:92-95). out may not alias in. */
static inline void fr_batch_inv(fr_t *out, const fr_t *in, size_t n) {
    fr_t acc; fr_one(&acc);
    for (size_t i = 0; i < n; i++) {
        if (fr_is_zero(&in[i])) { fr_zero(&out[i]); continue; }
        out[i] = acc; fr_mul(&acc, &acc, &in[i]);
    }
    fr_inv(&acc, &acc);
    for (size_t i = n; i-- > 0;) {
        if (fr_is_zero(&in[i])) continue;
        fr_t t; fr_mul(&t, &acc, &out[i]); fr_mul(&acc, &acc, &in[i]); out[i] = t;
    }
}

#endif
