/*
 * oracle/curve.h -- BLS12-381 G1 (y^2 = x^3 + 4 over Fp), Fp2, G2 (y^2 = x^3 + 4(1+u) over Fp2),
 * zcash-format point (de)compression.  Plain C, TEST INFRASTRUCTURE ONLY (see field.h header).
 *
 * Reference call sites this stands in for (gnark-crypto v0.16.0, not in /root/reference):
 *   G1Affine.SetBytes / Bytes           serialization.go:98-115, trusted_setup.go:113-116
 *   G1Jac add/double/ScalarMultiplication  internal/kzg/kzg_verify.go:56-83,169; internal/domain/fft.go:39,80
 *   G2Affine decode / G2Jac scalar-mul   trusted_setup.go:130-133; internal/kzg/kzg_verify.go:52-66
 * Identity encoding in affine form is (0,0) (reference internal/multiexp/multiexp_test.go:78-96).
 */
#ifndef KZGO_CURVE_H
#define KZGO_CURVE_H
#include "field.h"

typedef struct { fp_t x, y; } g1a_t;          /* affine, (0,0) = infinity */
typedef struct { fp_t x, y, z; } g1j_t;       /* Jacobian, z = 0 = infinity */

static inline int g1a_is_inf(const g1a_t *p) { return fp_is_zero(&p->x) && fp_is_zero(&p->y); }
static inline int g1j_is_inf(const g1j_t *p) { return fp_is_zero(&p->z); }
static inline void g1j_set_inf(g1j_t *p) { fp_one(&p->x); fp_one(&p->y); fp_zero(&p->z); }
static inline void g1a_set_inf(g1a_t *p) { fp_zero(&p->x); fp_zero(&p->y); }
static inline void g1j_from_affine(g1j_t *o, const g1a_t *a) {
    if (g1a_is_inf(a)) { g1j_set_inf(o); return; }
    o->x = a->x; o->y = a->y; fp_one(&o->z);
}
static inline void g1j_to_affine(g1a_t *o, const g1j_t *p) {
    if (g1j_is_inf(p)) { g1a_set_inf(o); return; }
    fp_t zi, zi2, zi3; fp_inv(&zi, &p->z); fp_sqr(&zi2, &zi); fp_mul(&zi3, &zi2, &zi);
    fp_mul(&o->x, &p->x, &zi2); fp_mul(&o->y, &p->y, &zi3);
}
static inline void g1j_neg(g1j_t *o, const g1j_t *p) { o->x = p->x; o->z = p->z; fp_neg(&o->y, &p->y); }
static inline void g1a_neg(g1a_t *o, const g1a_t *p) { o->x = p->x; fp_neg(&o->y, &p->y); }

/* doubling, a = 0 (dbl-2009-l) */
static inline void g1j_dbl(g1j_t *o, const g1j_t *p) {
    if (g1j_is_inf(p)) { *o = *p; return; }
    fp_t A, B, C, D, E, F, t, x3, y3, z3;
    fp_sqr(&A, &p->x); fp_sqr(&B, &p->y); fp_sqr(&C, &B);
    fp_add(&t, &p->x, &B); fp_sqr(&t, &t); fp_sub(&t, &t, &A); fp_sub(&t, &t, &C); fp_dbl(&D, &t);
    fp_dbl(&E, &A); fp_add(&E, &E, &A);
    fp_sqr(&F, &E);
    fp_dbl(&t, &D); fp_sub(&x3, &F, &t);
    fp_mul(&z3, &p->y, &p->z); fp_dbl(&z3, &z3);
    fp_sub(&t, &D, &x3); fp_mul(&y3, &E, &t);
    fp_dbl(&t, &C); fp_dbl(&t, &t); fp_dbl(&t, &t); fp_sub(&y3, &y3, &t);
    o->x = x3; o->y = y3; o->z = z3;
}
/* general addition (add-2007-bl) with all exceptional cases */
static inline void g1j_add(g1j_t *o, const g1j_t *p, const g1j_t *q) {
    if (g1j_is_inf(p)) { *o = *q; return; }
    if (g1j_is_inf(q)) { *o = *p; return; }
    fp_t z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;
    fp_sqr(&z1z1, &p->z); fp_sqr(&z2z2, &q->z);
    fp_mul(&u1, &p->x, &z2z2); fp_mul(&u2, &q->x, &z1z1);
    fp_mul(&s1, &p->y, &q->z); fp_mul(&s1, &s1, &z2z2);
    fp_mul(&s2, &q->y, &p->z); fp_mul(&s2, &s2, &z1z1);
    if (fp_eq(&u1, &u2)) {
        if (fp_eq(&s1, &s2)) { g1j_dbl(o, p); return; }
        g1j_set_inf(o); return;
    }
    fp_sub(&h, &u2, &u1);
    fp_dbl(&i, &h); fp_sqr(&i, &i);
    fp_mul(&j, &h, &i);
    fp_sub(&rr, &s2, &s1); fp_dbl(&rr, &rr);
    fp_mul(&v, &u1, &i);
    fp_sqr(&x3, &rr); fp_sub(&x3, &x3, &j); fp_sub(&x3, &x3, &v); fp_sub(&x3, &x3, &v);
    fp_sub(&t, &v, &x3); fp_mul(&y3, &rr, &t); fp_mul(&t, &s1, &j); fp_dbl(&t, &t); fp_sub(&y3, &y3, &t);
    fp_add(&z3, &p->z, &q->z); fp_sqr(&z3, &z3); fp_sub(&z3, &z3, &z1z1); fp_sub(&z3, &z3, &z2z2); fp_mul(&z3, &z3, &h);
    o->x = x3; o->y = y3; o->z = z3;
}
/* mixed addition Jacobian + affine (madd-2007-bl), exceptional cases routed to the general code */
static inline void g1j_add_affine(g1j_t *o, const g1j_t *p, const g1a_t *q) {
    if (g1a_is_inf(q)) { *o = *p; return; }
    if (g1j_is_inf(p)) { g1j_from_affine(o, q); return; }
    fp_t z1z1, u2, s2, h, hh, i, j, rr, v, t, x3, y3, z3;
    fp_sqr(&z1z1, &p->z);
    fp_mul(&u2, &q->x, &z1z1);
    fp_mul(&s2, &q->y, &p->z); fp_mul(&s2, &s2, &z1z1);
    if (fp_eq(&u2, &p->x)) { g1j_t qj; g1j_from_affine(&qj, q); g1j_add(o, p, &qj); return; }
    fp_sub(&h, &u2, &p->x);
    fp_sqr(&hh, &h);
    fp_dbl(&i, &hh); fp_dbl(&i, &i);
    fp_mul(&j, &h, &i);
    fp_sub(&rr, &s2, &p->y); fp_dbl(&rr, &rr);
    fp_mul(&v, &p->x, &i);
    fp_sqr(&x3, &rr); fp_sub(&x3, &x3, &j); fp_sub(&x3, &x3, &v); fp_sub(&x3, &x3, &v);
    fp_sub(&t, &v, &x3); fp_mul(&y3, &rr, &t); fp_mul(&t, &p->y, &j); fp_dbl(&t, &t); fp_sub(&y3, &y3, &t);
    fp_add(&z3, &p->z, &h); fp_sqr(&z3, &z3); fp_sub(&z3, &z3, &z1z1); fp_sub(&z3, &z3, &hh);
    o->x = x3; o->y = y3; o->z = z3;
}
static inline void g1j_sub(g1j_t *o, const g1j_t *p, const g1j_t *q) { g1j_t n; g1j_neg(&n, q); g1j_add(o, p, &n); }
/* scalar as plain little-endian limbs */
static inline void g1j_mul_limbs(g1j_t *o, const g1j_t *p, const u64 *k, int nlimbs) {
    g1j_t acc; g1j_set_inf(&acc);
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        g1j_dbl(&acc, &acc);
        if ((k[i >> 6] >> (i & 63)) & 1) g1j_add(&acc, &acc, p);
    }
    *o = acc;
}
static inline void g1j_mul_fr(g1j_t *o, const g1j_t *p, const fr_t *k) { u64 v[4]; fr_to_plain(v, k); g1j_mul_limbs(o, p, v, 4); }
static inline int g1j_eq(const g1j_t *p, const g1j_t *q) {
    if (g1j_is_inf(p) || g1j_is_inf(q)) return g1j_is_inf(p) && g1j_is_inf(q);
    fp_t z1z1, z2z2, a, b;
    fp_sqr(&z1z1, &p->z); fp_sqr(&z2z2, &q->z);
    fp_mul(&a, &p->x, &z2z2); fp_mul(&b, &q->x, &z1z1);
    if (!fp_eq(&a, &b)) return 0;
    fp_mul(&a, &p->y, &q->z); fp_mul(&a, &a, &z2z2);
    fp_mul(&b, &q->y, &p->z); fp_mul(&b, &b, &z1z1);
    return fp_eq(&a, &b);
}
static inline int g1a_on_curve(const g1a_t *p) {
    if (g1a_is_inf(p)) return 1;
    fp_t l, r, four; fp_sqr(&l, &p->y); fp_sqr(&r, &p->x); fp_mul(&r, &r, &p->x); fp_set_u64(&four, 4); fp_add(&r, &r, &four);
    return fp_eq(&l, &r);
}
/* r-torsion check, the definitional way: [r]P == infinity */
static inline int g1a_in_subgroup(const g1a_t *p) {
    g1j_t j, t; g1j_from_affine(&j, p); g1j_mul_limbs(&t, &j, FR_MOD, 4); return g1j_is_inf(&t);
}

/* zcash compressed encoding: byte0 bit7 = compressed, bit6 = infinity, bit5 = y is the larger root */
static inline void g1a_compress(uint8_t out[48], const g1a_t *p) {
    if (g1a_is_inf(p)) { memset(out, 0, 48); out[0] = 0xc0; return; }
    fp_to_be48(out, &p->x);
    out[0] |= 0x80;
    if (fp_lex_largest(&p->y)) out[0] |= 0x20;
}
enum { G1DEC_OK = 0, G1DEC_BAD_ENCODING = 1, G1DEC_NOT_ON_CURVE = 2, G1DEC_NOT_IN_SUBGROUP = 3 };
static inline int g1a_decompress(g1a_t *o, const uint8_t in[48], int subgroup_check) {
    uint8_t flags = in[0] & 0xe0;
    if (!(flags & 0x80)) return G1DEC_BAD_ENCODING;       /* 48 bytes can only hold a compressed point */
    if (flags == 0xe0) return G1DEC_BAD_ENCODING;         /* infinity + sort flag */
    if (flags & 0x40) {
        if (in[0] & 0x1f) return G1DEC_BAD_ENCODING;
        for (int i = 1; i < 48; i++) if (in[i]) return G1DEC_BAD_ENCODING;
        g1a_set_inf(o); return G1DEC_OK;
    }
    uint8_t xb[48]; memcpy(xb, in, 48); xb[0] &= 0x1f;
    fp_t x, y, rhs, four;
    if (!fp_from_be48(&x, xb)) return G1DEC_BAD_ENCODING;
    fp_sqr(&rhs, &x); fp_mul(&rhs, &rhs, &x); fp_set_u64(&four, 4); fp_add(&rhs, &rhs, &four);
    if (!fp_sqrt(&y, &rhs)) return G1DEC_NOT_ON_CURVE;
    if (fp_lex_largest(&y) != ((flags & 0x20) != 0)) fp_neg(&y, &y);
    o->x = x; o->y = y;
    if (subgroup_check && !g1a_in_subgroup(o)) return G1DEC_NOT_IN_SUBGROUP;
    return G1DEC_OK;
}

/* ---------- Fp2 = Fp[u]/(u^2+1) ---------- */
typedef struct { fp_t c0, c1; } fp2_t;
static inline void fp2_add(fp2_t *o, const fp2_t *a, const fp2_t *b) { fp_add(&o->c0, &a->c0, &b->c0); fp_add(&o->c1, &a->c1, &b->c1); }
static inline void fp2_sub(fp2_t *o, const fp2_t *a, const fp2_t *b) { fp_sub(&o->c0, &a->c0, &b->c0); fp_sub(&o->c1, &a->c1, &b->c1); }
static inline void fp2_neg(fp2_t *o, const fp2_t *a) { fp_neg(&o->c0, &a->c0); fp_neg(&o->c1, &a->c1); }
static inline void fp2_dbl(fp2_t *o, const fp2_t *a) { fp2_add(o, a, a); }
static inline void fp2_mul(fp2_t *o, const fp2_t *a, const fp2_t *b) {
    fp_t t0, t1, t2, t3;
    fp_mul(&t0, &a->c0, &b->c0); fp_mul(&t1, &a->c1, &b->c1);
    fp_mul(&t2, &a->c0, &b->c1); fp_mul(&t3, &a->c1, &b->c0);
    fp_sub(&o->c0, &t0, &t1); fp_add(&o->c1, &t2, &t3);
}
static inline void fp2_sqr(fp2_t *o, const fp2_t *a) { fp2_mul(o, a, a); }
static inline void fp2_mul_fp(fp2_t *o, const fp2_t *a, const fp_t *b) { fp_mul(&o->c0, &a->c0, b); fp_mul(&o->c1, &a->c1, b); }
static inline void fp2_conj(fp2_t *o, const fp2_t *a) { o->c0 = a->c0; fp_neg(&o->c1, &a->c1); }
static inline int fp2_is_zero(const fp2_t *a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static inline int fp2_eq(const fp2_t *a, const fp2_t *b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static inline void fp2_zero(fp2_t *o) { fp_zero(&o->c0); fp_zero(&o->c1); }
static inline void fp2_one(fp2_t *o) { fp_one(&o->c0); fp_zero(&o->c1); }
static inline void fp2_inv(fp2_t *o, const fp2_t *a) {
    fp_t n, t; fp_sqr(&n, &a->c0); fp_sqr(&t, &a->c1); fp_add(&n, &n, &t); fp_inv(&n, &n);
    fp_mul(&o->c0, &a->c0, &n); fp_mul(&t, &a->c1, &n); fp_neg(&o->c1, &t);
}
/* multiply by xi = 1 + u */
static inline void fp2_mul_xi(fp2_t *o, const fp2_t *a) { fp_t t0, t1; fp_sub(&t0, &a->c0, &a->c1); fp_add(&t1, &a->c0, &a->c1); o->c0 = t0; o->c1 = t1; }
/* square root by the norm ("complex") method; returns 1 on success */
static inline int fp2_sqrt(fp2_t *o, const fp2_t *a) {
    if (fp2_is_zero(a)) { fp2_zero(o); return 1; }
    fp_t two, inv2; fp_set_u64(&two, 2); fp_inv(&inv2, &two);
    if (fp_is_zero(&a->c1)) {
        fp_t s;
        if (fp_sqrt(&s, &a->c0)) { o->c0 = s; fp_zero(&o->c1); return 1; }
        fp_t na; fp_neg(&na, &a->c0);
        if (fp_sqrt(&s, &na)) { fp_zero(&o->c0); o->c1 = s; return 1; } /* (s u)^2 = -s^2 = a0 */
        return 0;
    }
    fp_t n, t, s, d, x0, x1;
    fp_sqr(&n, &a->c0); fp_sqr(&t, &a->c1); fp_add(&n, &n, &t);
    if (!fp_sqrt(&s, &n)) return 0;
    fp_add(&d, &a->c0, &s); fp_mul(&d, &d, &inv2);
    if (!fp_sqrt(&x0, &d)) {
        fp_sub(&d, &a->c0, &s); fp_mul(&d, &d, &inv2);
        if (!fp_sqrt(&x0, &d)) return 0;
    }
    fp_dbl(&t, &x0); fp_inv(&t, &t); fp_mul(&x1, &a->c1, &t);
    fp2_t r = {x0, x1}, c; fp2_sqr(&c, &r);
    if (!fp2_eq(&c, a)) return 0;
    *o = r; return 1;
}
static inline int fp2_lex_largest(const fp2_t *a) {
    if (!fp_is_zero(&a->c1)) return fp_lex_largest(&a->c1);
    return fp_lex_largest(&a->c0);
}

/* ---------- G2, affine with field inversions (simple and slow on purpose) ---------- */
typedef struct { fp2_t x, y; int inf; } g2a_t;
static inline void g2a_set_inf(g2a_t *p) { fp2_zero(&p->x); fp2_zero(&p->y); p->inf = 1; }
static inline void g2a_neg(g2a_t *o, const g2a_t *p) { *o = *p; fp2_neg(&o->y, &p->y); }
static inline void g2a_dbl(g2a_t *o, const g2a_t *p) {
    if (p->inf || fp2_is_zero(&p->y)) { g2a_set_inf(o); return; }
    fp2_t l, t, x3, y3;
    fp2_sqr(&l, &p->x); fp2_dbl(&t, &l); fp2_add(&l, &l, &t);           /* 3x^2 */
    fp2_dbl(&t, &p->y); fp2_inv(&t, &t); fp2_mul(&l, &l, &t);           /* / 2y */
    fp2_sqr(&x3, &l); fp2_sub(&x3, &x3, &p->x); fp2_sub(&x3, &x3, &p->x);
    fp2_sub(&t, &p->x, &x3); fp2_mul(&y3, &l, &t); fp2_sub(&y3, &y3, &p->y);
    o->x = x3; o->y = y3; o->inf = 0;
}
static inline void g2a_add(g2a_t *o, const g2a_t *p, const g2a_t *q) {
    if (p->inf) { *o = *q; return; }
    if (q->inf) { *o = *p; return; }
    if (fp2_eq(&p->x, &q->x)) {
        if (fp2_eq(&p->y, &q->y)) { g2a_dbl(o, p); return; }
        g2a_set_inf(o); return;
    }
    fp2_t l, t, x3, y3;
    fp2_sub(&l, &q->y, &p->y); fp2_sub(&t, &q->x, &p->x); fp2_inv(&t, &t); fp2_mul(&l, &l, &t);
    fp2_sqr(&x3, &l); fp2_sub(&x3, &x3, &p->x); fp2_sub(&x3, &x3, &q->x);
    fp2_sub(&t, &p->x, &x3); fp2_mul(&y3, &l, &t); fp2_sub(&y3, &y3, &p->y);
    o->x = x3; o->y = y3; o->inf = 0;
}
static inline void g2a_mul_fr(g2a_t *o, const g2a_t *p, const fr_t *k) {
    u64 v[4]; fr_to_plain(v, k);
    g2a_t acc; g2a_set_inf(&acc);
    for (int i = 255; i >= 0; i--) {
        g2a_dbl(&acc, &acc);
        if ((v[i >> 6] >> (i & 63)) & 1) g2a_add(&acc, &acc, p);
    }
    *o = acc;
}
/* 96-byte zcash encoding: x.c1 (with flags) || x.c0.  No subgroup check (reference
 * trusted_setup.go:125-136 decodes G2 with NoSubgroupChecks). Returns 0 on success. */
static inline int g2a_decompress(g2a_t *o, const uint8_t in[96]) {
    uint8_t flags = in[0] & 0xe0;
    if (!(flags & 0x80) || flags == 0xe0) return 1;
    if (flags & 0x40) {
        if (in[0] & 0x1f) return 1;
        for (int i = 1; i < 96; i++) if (in[i]) return 1;
        g2a_set_inf(o); return 0;
    }
    uint8_t xb[48]; memcpy(xb, in, 48); xb[0] &= 0x1f;
    fp2_t x, y, rhs, b;
    if (!fp_from_be48(&x.c1, xb)) return 1;
    if (!fp_from_be48(&x.c0, in + 48)) return 1;
    fp2_sqr(&rhs, &x); fp2_mul(&rhs, &rhs, &x);
    fp_set_u64(&b.c0, 4); fp_set_u64(&b.c1, 4); fp2_add(&rhs, &rhs, &b);
    if (!fp2_sqrt(&y, &rhs)) return 2;
    if (fp2_lex_largest(&y) != ((flags & 0x20) != 0)) fp2_neg(&y, &y);
    o->x = x; o->y = y; o->inf = 0;
    return 0;
}

#endif
