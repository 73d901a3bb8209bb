/*
 * oracle/pairing.h -- Fp6 / Fp12 tower and the (optimal-)ate pairing product check for
 * BLS12-381.  Plain C, deliberately textbook: affine Miller loop with dense Fp12
 * multiplications and a brute-force final exponentiation f^(p^6-1) then ^((p^6+1)/r) by
 * square-and-multiply -- no Frobenius constants, nothing to get subtly wrong.
 * TEST INFRASTRUCTURE ONLY (see field.h header).
 *
 * Stands in for gnark-crypto's bls12381.PairingCheck (not in /root/reference), called at
 * reference internal/kzg/kzg_verify.go:88,190 and internal/kzg_multi/kzg_verify.go:94.
 * Only the predicate  prod e(P_i, Q_i) == 1  is needed; it is invariant under the choice
 * of any non-degenerate power of the pairing, so no canonical GT value is pinned.
 *
 * Tower: Fp6 = Fp2[v]/(v^3 - xi), xi = 1+u;  Fp12 = Fp6[w]/(w^2 - v).
 * Twist (M-type): E'(Fp2): y^2 = x^3 + 4 xi, untwist (x',y') -> (x'/w^2, y'/w^3).
 */
#ifndef KZGO_PAIRING_H
#define KZGO_PAIRING_H
#include "curve.h"

typedef struct { fp2_t c0, c1, c2; } fp6_t;
typedef struct { fp6_t c0, c1; } fp12_t;

static inline void fp6_add(fp6_t *o, const fp6_t *a, const fp6_t *b) { fp2_add(&o->c0, &a->c0, &b->c0); fp2_add(&o->c1, &a->c1, &b->c1); fp2_add(&o->c2, &a->c2, &b->c2); }
static inline void fp6_sub(fp6_t *o, const fp6_t *a, const fp6_t *b) { fp2_sub(&o->c0, &a->c0, &b->c0); fp2_sub(&o->c1, &a->c1, &b->c1); fp2_sub(&o->c2, &a->c2, &b->c2); }
static inline void fp6_neg(fp6_t *o, const fp6_t *a) { fp2_neg(&o->c0, &a->c0); fp2_neg(&o->c1, &a->c1); fp2_neg(&o->c2, &a->c2); }
static inline void fp6_zero(fp6_t *o) { fp2_zero(&o->c0); fp2_zero(&o->c1); fp2_zero(&o->c2); }
static inline void fp6_one(fp6_t *o) { fp2_one(&o->c0); fp2_zero(&o->c1); fp2_zero(&o->c2); }
static inline int fp6_eq(const fp6_t *a, const fp6_t *b) { return fp2_eq(&a->c0, &b->c0) && fp2_eq(&a->c1, &b->c1) && fp2_eq(&a->c2, &b->c2); }
/* schoolbook product, reduced with v^3 = xi */
static inline void fp6_mul(fp6_t *o, const fp6_t *a, const fp6_t *b) {
    fp2_t a0b0, a0b1, a0b2, a1b0, a1b1, a1b2, a2b0, a2b1, a2b2, t, r0, r1, r2;
    fp2_mul(&a0b0, &a->c0, &b->c0); fp2_mul(&a0b1, &a->c0, &b->c1); fp2_mul(&a0b2, &a->c0, &b->c2);
    fp2_mul(&a1b0, &a->c1, &b->c0); fp2_mul(&a1b1, &a->c1, &b->c1); fp2_mul(&a1b2, &a->c1, &b->c2);
    fp2_mul(&a2b0, &a->c2, &b->c0); fp2_mul(&a2b1, &a->c2, &b->c1); fp2_mul(&a2b2, &a->c2, &b->c2);
    fp2_add(&t, &a1b2, &a2b1); fp2_mul_xi(&t, &t); fp2_add(&r0, &a0b0, &t);          /* v^3 terms */
    fp2_mul_xi(&t, &a2b2); fp2_add(&r1, &a0b1, &a1b0); fp2_add(&r1, &r1, &t);         /* v^4 term  */
    fp2_add(&r2, &a0b2, &a1b1); fp2_add(&r2, &r2, &a2b0);
    o->c0 = r0; o->c1 = r1; o->c2 = r2;
}
static inline void fp6_mul_v(fp6_t *o, const fp6_t *a) { fp2_t t; fp2_mul_xi(&t, &a->c2); fp2_t c0 = a->c0, c1 = a->c1; o->c0 = t; o->c1 = c0; o->c2 = c1; }
static inline void fp6_inv(fp6_t *o, const fp6_t *a) {
    fp2_t A, B, C, t, F;
    fp2_sqr(&A, &a->c0); fp2_mul(&t, &a->c1, &a->c2); fp2_mul_xi(&t, &t); fp2_sub(&A, &A, &t);
    fp2_sqr(&B, &a->c2); fp2_mul_xi(&B, &B); fp2_mul(&t, &a->c0, &a->c1); fp2_sub(&B, &B, &t);
    fp2_sqr(&C, &a->c1); fp2_mul(&t, &a->c0, &a->c2); fp2_sub(&C, &C, &t);
    fp2_mul(&F, &a->c2, &B); fp2_mul(&t, &a->c1, &C); fp2_add(&F, &F, &t); fp2_mul_xi(&F, &F);
    fp2_mul(&t, &a->c0, &A); fp2_add(&F, &F, &t);
    fp2_inv(&F, &F);
    fp2_mul(&o->c0, &A, &F); fp2_mul(&o->c1, &B, &F); fp2_mul(&o->c2, &C, &F);
}

static inline void fp12_one(fp12_t *o) { fp6_one(&o->c0); fp6_zero(&o->c1); }
static inline int fp12_is_one(const fp12_t *a) { fp12_t one; fp12_one(&one); return fp6_eq(&a->c0, &one.c0) && fp6_eq(&a->c1, &one.c1); }
static inline void fp12_mul(fp12_t *o, const fp12_t *a, const fp12_t *b) {
    fp6_t t0, t1, t2, t3, r0, r1;
    fp6_mul(&t0, &a->c0, &b->c0); fp6_mul(&t1, &a->c1, &b->c1);
    fp6_mul(&t2, &a->c0, &b->c1); fp6_mul(&t3, &a->c1, &b->c0);
    fp6_mul_v(&t1, &t1); fp6_add(&r0, &t0, &t1); fp6_add(&r1, &t2, &t3);
    o->c0 = r0; o->c1 = r1;
}
static inline void fp12_sqr(fp12_t *o, const fp12_t *a) { fp12_mul(o, a, a); }
static inline void fp12_conj(fp12_t *o, const fp12_t *a) { o->c0 = a->c0; fp6_neg(&o->c1, &a->c1); }
static inline void fp12_inv(fp12_t *o, const fp12_t *a) {
    fp6_t t0, t1; fp6_mul(&t0, &a->c0, &a->c0); fp6_mul(&t1, &a->c1, &a->c1); fp6_mul_v(&t1, &t1); fp6_sub(&t0, &t0, &t1);
    fp6_inv(&t0, &t0);
    fp6_mul(&o->c0, &a->c0, &t0); fp6_mul(&t1, &a->c1, &t0); fp6_neg(&o->c1, &t1);
}

/* line through T with slope lam (both on the twist), evaluated at P in E(Fp), scaled by xi
 * (an Fp2 factor, removed by the final exponentiation):
 *   l = xi*yP  +  (lam*xT - yT) w^3  -  (lam*xP) w^5 ,  w^3 = v w, w^5 = v^2 w            */
static inline void line_eval(fp12_t *l, const fp2_t *lam, const g2a_t *T, const g1a_t *P) {
    fp2_t t;
    fp6_zero(&l->c0); fp6_zero(&l->c1);
    l->c0.c0.c0 = P->y; l->c0.c0.c1 = P->y;                 /* xi * yP = yP + yP u */
    fp2_mul(&t, lam, &T->x); fp2_sub(&l->c1.c1, &t, &T->y);
    fp2_mul_fp(&t, lam, &P->x); fp2_neg(&l->c1.c2, &t);
}
/* f_{|x|,Q}(P), affine Miller loop; P, Q not infinity */
static inline void miller_loop(fp12_t *f, const g1a_t *P, const g2a_t *Q) {
    g2a_t T = *Q; fp12_t l; fp2_t lam, t;
    fp12_one(f);
    for (int i = 62; i >= 0; i--) {   /* BLS_X_ABS has its top bit at position 63 */
        fp12_sqr(f, f);
        /* tangent */
        fp2_sqr(&lam, &T.x); fp2_dbl(&t, &lam); fp2_add(&lam, &lam, &t);
        fp2_dbl(&t, &T.y); fp2_inv(&t, &t); fp2_mul(&lam, &lam, &t);
        line_eval(&l, &lam, &T, P); fp12_mul(f, f, &l);
        g2a_dbl(&T, &T);
        if ((BLS_X_ABS >> i) & 1) {
            fp2_sub(&lam, &Q->y, &T.y); fp2_sub(&t, &Q->x, &T.x); fp2_inv(&t, &t); fp2_mul(&lam, &lam, &t);
            line_eval(&l, &lam, &T, P); fp12_mul(f, f, &l);
            g2a_add(&T, &T, Q);
        }
    }
    /* x < 0: f <- f^-1 ~ conj(f) after the easy part; conj is applied for form's sake */
    fp12_conj(f, f);
}
static inline void final_exp(fp12_t *o, const fp12_t *f) {
    fp12_t a, b, acc;
    fp12_conj(&a, f); fp12_inv(&b, f); fp12_mul(&a, &a, &b);       /* f^(p^6-1) */
    fp12_one(&acc);
    for (int i = FINAL_EXP_LIMBS * 64 - 1; i >= 0; i--) {
        fp12_sqr(&acc, &acc);
        if ((FINAL_EXP_P6P1_DIV_R[i >> 6] >> (i & 63)) & 1) fp12_mul(&acc, &acc, &a);
    }
    *o = acc;
}
/* prod_i e(P_i, Q_i) == 1 ; pairs with an infinity member contribute 1 */
static inline int pairing_check(const g1a_t *Ps, const g2a_t *Qs, int n) {
    fp12_t f, m; fp12_one(&f);
    for (int i = 0; i < n; i++) {
        if (g1a_is_inf(&Ps[i]) || Qs[i].inf) continue;
        miller_loop(&m, &Ps[i], &Qs[i]); fp12_mul(&f, &f, &m);
    }
    final_exp(&f, &f);
    return fp12_is_one(&f);
}

#endif
