// host_pairing.hpp -- product host code: Fp2/Fp6/Fp12 tower, G2 decoding and the optimal-ate
// pairing *product check* for BLS12-381 with PREPARED second arguments.
//
// Replaces gnark-crypto's bls12381.PairingCheck as it is used by the reference's verifiers
// (internal/kzg/kzg_verify.go:88-99,190-199; internal/kzg_multi/kzg_verify.go:94-104).  In every
// one of those calls the G2 arguments are Context constants ([1]G2, [alpha]G2, [alpha^64]G2 and
// negations), so their Miller-loop line coefficients are computed once at context creation
// (G2Prepared) and a verification only evaluates the lines at the two G1 points that the GPU
// produced.  The pairing stays on the host by design: it is ~25k field multiplications of
// inherently serial work per *batch*, a poor fit for a 64-wide SIMD machine.
//
// Tower: Fp2 = Fp[u]/(u^2+1); Fp6 = Fp2[v]/(v^3-xi), xi = 1+u; Fp12 = Fp6[w]/(w^2-v).
// Twist (M-type): E'(Fp2): y^2 = x^3 + 4 xi, untwist (x',y') -> (x'/w^2, y'/w^3).
// Final exponentiation hard part: 3 (p^4-p^2+1)/r = (x-1)^2 (x+p) (x^2+p^2-1) + 3
// (identity checked numerically in gen_constants.py's companion test tests/test_host_pairing.py).
#pragma once
#include <thread>
#include <vector>
#include "host_field.hpp"
#include "host_pool.hpp"
#include "host_cyclo_ifma.hpp"

namespace kzg {

using F = HostFp;
using fe = HostFp::El;

struct Fp2 {
    fe c0, c1;
};
namespace fp2 {
inline void add(Fp2& r, const Fp2& a, const Fp2& b) { F::add(r.c0, a.c0, b.c0); F::add(r.c1, a.c1, b.c1); }
inline void sub(Fp2& r, const Fp2& a, const Fp2& b) { F::sub<0>(r.c0, a.c0, b.c0); F::sub<0>(r.c1, a.c1, b.c1); }
inline void neg(Fp2& r, const Fp2& a) { F::neg<0>(r.c0, a.c0); F::neg<0>(r.c1, a.c1); }
inline void dbl(Fp2& r, const Fp2& a) { add(r, a, a); }
inline void conj(Fp2& r, const Fp2& a) { r.c0 = a.c0; F::neg<0>(r.c1, a.c1); }
inline void mul(Fp2& r, const Fp2& a, const Fp2& b) {   // Karatsuba: 3 Fp multiplications
    fe t0, t1, s0, s1, m;
    F::mul(t0, a.c0, b.c0);
    F::mul(t1, a.c1, b.c1);
    F::add(s0, a.c0, a.c1);
    F::add(s1, b.c0, b.c1);
    F::mul(m, s0, s1);
    F::sub<0>(m, m, t0);
    F::sub<0>(r.c1, m, t1);
    F::sub<0>(r.c0, t0, t1);
}
inline void sqr(Fp2& r, const Fp2& a) {   // (a0+a1)(a0-a1), 2 a0 a1
    fe s, d, m;
    F::add(s, a.c0, a.c1);
    F::sub<0>(d, a.c0, a.c1);
    F::mul(m, a.c0, a.c1);
    F::mul(r.c0, s, d);
    F::dbl(r.c1, m);
}
inline void mul_fp(Fp2& r, const Fp2& a, const fe& b) { F::mul(r.c0, a.c0, b); F::mul(r.c1, a.c1, b); }
inline void mul_xi(Fp2& r, const Fp2& a) {   // (1+u)
    fe t0, t1;
    F::sub<0>(t0, a.c0, a.c1);
    F::add(t1, a.c0, a.c1);
    r.c0 = t0;
    r.c1 = t1;
}
inline bool is_zero(const Fp2& a) { return F::limbs_all_zero(a.c0) && F::limbs_all_zero(a.c1); }
inline bool eq(const Fp2& a, const Fp2& b) { return F::limbs_equal(a.c0, b.c0) && F::limbs_equal(a.c1, b.c1); }
inline void zero(Fp2& r) { F::set_zero(r.c0); F::set_zero(r.c1); }
inline void one(Fp2& r) { F::set_one(r.c0); F::set_zero(r.c1); }
inline void inv(Fp2& r, const Fp2& a) {
    fe n, t;
    F::sqr(n, a.c0);
    F::sqr(t, a.c1);
    F::add(n, n, t);
    F::inv(n, n);
    F::mul(r.c0, a.c0, n);
    F::mul(t, a.c1, n);
    F::neg<0>(r.c1, t);
}
inline void pow(Fp2& r, const Fp2& a, const uint64_t* e, int nbits) {
    Fp2 acc;
    one(acc);
    for (int i = nbits - 1; i >= 0; i--) {
        sqr(acc, acc);
        if ((e[i >> 6] >> (i & 63)) & 1) mul(acc, acc, a);
    }
    r = acc;
}
// square root via the norm; returns false if a is not a square
inline bool sqrt(Fp2& r, const Fp2& a) {
    if (is_zero(a)) { zero(r); return true; }
    fe two, inv2;
    F::set_one(two);
    F::dbl(two, two);
    F::inv(inv2, two);
    if (F::limbs_all_zero(a.c1)) {
        fe s;
        if (F::sqrt(s, a.c0)) { r.c0 = s; F::set_zero(r.c1); return true; }
        fe na;
        F::neg<0>(na, a.c0);
        if (F::sqrt(s, na)) { F::set_zero(r.c0); r.c1 = s; return true; }
        return false;
    }
    fe n, t, s, d, x0, x1;
    F::sqr(n, a.c0);
    F::sqr(t, a.c1);
    F::add(n, n, t);
    if (!F::sqrt(s, n)) return false;
    F::add(d, a.c0, s);
    F::mul(d, d, inv2);
    if (!F::sqrt(x0, d)) {
        F::sub<0>(d, a.c0, s);
        F::mul(d, d, inv2);
        if (!F::sqrt(x0, d)) return false;
    }
    F::dbl(t, x0);
    F::inv(t, t);
    F::mul(x1, a.c1, t);
    Fp2 cand{x0, x1}, chk;
    sqr(chk, cand);
    if (!eq(chk, a)) return false;
    r = cand;
    return true;
}
inline bool lex_largest(const Fp2& a) {
    if (!F::limbs_all_zero(a.c1)) return F::lex_largest(a.c1);
    return F::lex_largest(a.c0);
}
}  // namespace fp2

struct Fp6 {
    Fp2 c0, c1, c2;
};
namespace fp6 {
inline void add(Fp6& r, const Fp6& a, const Fp6& b) { fp2::add(r.c0, a.c0, b.c0); fp2::add(r.c1, a.c1, b.c1); fp2::add(r.c2, a.c2, b.c2); }
inline void sub(Fp6& r, const Fp6& a, const Fp6& b) { fp2::sub(r.c0, a.c0, b.c0); fp2::sub(r.c1, a.c1, b.c1); fp2::sub(r.c2, a.c2, b.c2); }
inline void neg(Fp6& r, const Fp6& a) { fp2::neg(r.c0, a.c0); fp2::neg(r.c1, a.c1); fp2::neg(r.c2, a.c2); }
inline void zero(Fp6& r) { fp2::zero(r.c0); fp2::zero(r.c1); fp2::zero(r.c2); }
inline void one(Fp6& r) { fp2::one(r.c0); fp2::zero(r.c1); fp2::zero(r.c2); }
inline bool eq(const Fp6& a, const Fp6& b) { return fp2::eq(a.c0, b.c0) && fp2::eq(a.c1, b.c1) && fp2::eq(a.c2, b.c2); }
// Karatsuba-style product (6 Fp2 multiplications), reduced with v^3 = xi
inline void mul(Fp6& r, const Fp6& a, const Fp6& b) {
    Fp2 t0, t1, t2, s, u, x0, x1, x2;
    fp2::mul(t0, a.c0, b.c0);
    fp2::mul(t1, a.c1, b.c1);
    fp2::mul(t2, a.c2, b.c2);
    // c0 = t0 + xi((a1+a2)(b1+b2) - t1 - t2)
    fp2::add(s, a.c1, a.c2);
    fp2::add(u, b.c1, b.c2);
    fp2::mul(x0, s, u);
    fp2::sub(x0, x0, t1);
    fp2::sub(x0, x0, t2);
    fp2::mul_xi(x0, x0);
    fp2::add(x0, x0, t0);
    // c1 = (a0+a1)(b0+b1) - t0 - t1 + xi t2
    fp2::add(s, a.c0, a.c1);
    fp2::add(u, b.c0, b.c1);
    fp2::mul(x1, s, u);
    fp2::sub(x1, x1, t0);
    fp2::sub(x1, x1, t1);
    fp2::mul_xi(s, t2);
    fp2::add(x1, x1, s);
    // c2 = (a0+a2)(b0+b2) - t0 - t2 + t1
    fp2::add(s, a.c0, a.c2);
    fp2::add(u, b.c0, b.c2);
    fp2::mul(x2, s, u);
    fp2::sub(x2, x2, t0);
    fp2::sub(x2, x2, t2);
    fp2::add(x2, x2, t1);
    r.c0 = x0;
    r.c1 = x1;
    r.c2 = x2;
}
inline void mul_v(Fp6& r, const Fp6& a) {   // times v
    Fp2 t;
    fp2::mul_xi(t, a.c2);
    Fp2 c0 = a.c0, c1 = a.c1;
    r.c0 = t;
    r.c1 = c0;
    r.c2 = c1;
}
inline void inv(Fp6& r, const Fp6& a) {
    Fp2 A, B, C, t, Fd;
    fp2::sqr(A, a.c0);
    fp2::mul(t, a.c1, a.c2);
    fp2::mul_xi(t, t);
    fp2::sub(A, A, t);
    fp2::sqr(B, a.c2);
    fp2::mul_xi(B, B);
    fp2::mul(t, a.c0, a.c1);
    fp2::sub(B, B, t);
    fp2::sqr(C, a.c1);
    fp2::mul(t, a.c0, a.c2);
    fp2::sub(C, C, t);
    fp2::mul(Fd, a.c2, B);
    fp2::mul(t, a.c1, C);
    fp2::add(Fd, Fd, t);
    fp2::mul_xi(Fd, Fd);
    fp2::mul(t, a.c0, A);
    fp2::add(Fd, Fd, t);
    fp2::inv(Fd, Fd);
    fp2::mul(r.c0, A, Fd);
    fp2::mul(r.c1, B, Fd);
    fp2::mul(r.c2, C, Fd);
}
}  // namespace fp6

struct Fp12 {
    Fp6 c0, c1;
};
namespace fp12 {
inline void one(Fp12& r) { fp6::one(r.c0); fp6::zero(r.c1); }
inline bool is_one(const Fp12& a) { Fp12 o; one(o); return fp6::eq(a.c0, o.c0) && fp6::eq(a.c1, o.c1); }
inline void mul(Fp12& r, const Fp12& a, const Fp12& b) {   // Karatsuba: 3 Fp6 multiplications
    Fp6 t0, t1, s, u, m;
    fp6::mul(t0, a.c0, b.c0);
    fp6::mul(t1, a.c1, b.c1);
    fp6::add(s, a.c0, a.c1);
    fp6::add(u, b.c0, b.c1);
    fp6::mul(m, s, u);
    fp6::sub(m, m, t0);
    fp6::sub(m, m, t1);
    fp6::mul_v(t1, t1);
    fp6::add(r.c0, t0, t1);
    r.c1 = m;
}
inline void sqr(Fp12& r, const Fp12& a) {   // complex squaring: 2 Fp6 multiplications
    Fp6 ab, s, t, m;
    fp6::mul(ab, a.c0, a.c1);
    fp6::add(s, a.c0, a.c1);
    fp6::mul_v(t, a.c1);
    fp6::add(t, t, a.c0);
    fp6::mul(m, s, t);          // (a0+a1)(a0+v a1) = a0^2 + v a1^2 + (1+v) a0 a1
    fp6::sub(m, m, ab);
    fp6::mul_v(t, ab);
    fp6::sub(r.c0, m, t);
    fp6::add(r.c1, ab, ab);
}
inline void conj(Fp12& r, const Fp12& a) { r.c0 = a.c0; fp6::neg(r.c1, a.c1); }
inline void inv(Fp12& r, const Fp12& a) {
    Fp6 t0, t1;
    fp6::mul(t0, a.c0, a.c0);
    fp6::mul(t1, a.c1, a.c1);
    fp6::mul_v(t1, t1);
    fp6::sub(t0, t0, t1);
    fp6::inv(t0, t0);
    fp6::mul(r.c0, a.c0, t0);
    fp6::mul(t1, a.c1, t0);
    fp6::neg(r.c1, t1);
}
}  // namespace fp12

// ---------------------------------------------------------------------------------------------
// G2 (affine, on the twist) and prepared Miller-loop lines
// ---------------------------------------------------------------------------------------------
struct G2Aff {
    Fp2 x, y;
    bool inf;
};
// 96-byte zcash encoding: x.c1 (with flags) || x.c0; no subgroup check (reference
// trusted_setup.go:125-136 uses NoSubgroupChecks).  Returns false on a malformed encoding.
inline bool g2_decompress(G2Aff& o, const uint8_t in[96]) {
    uint8_t flags = in[0] & 0xe0;
    if (!(flags & 0x80) || flags == 0xe0) return false;
    if (flags & 0x40) {
        if (in[0] & 0x1f) return false;
        for (int i = 1; i < 96; i++) if (in[i]) return false;
        fp2::zero(o.x);
        fp2::zero(o.y);
        o.inf = true;
        return true;
    }
    uint8_t xb[48];
    memcpy(xb, in, 48);
    xb[0] &= 0x1f;
    Fp2 x, y, rhs, b;
    if (!F::from_be48(x.c1, xb)) return false;
    if (!F::from_be48(x.c0, in + 48)) return false;
    fp2::sqr(rhs, x);
    fp2::mul(rhs, rhs, x);
    fe four;
    F::set_one(four);
    F::dbl(four, four);
    F::dbl(four, four);
    b.c0 = four;
    b.c1 = four;
    fp2::add(rhs, rhs, b);
    if (!fp2::sqrt(y, rhs)) return false;
    if (fp2::lex_largest(y) != ((flags & 0x20) != 0)) fp2::neg(y, y);
    o.x = x;
    o.y = y;
    o.inf = false;
    return true;
}
inline void g2_neg(G2Aff& r, const G2Aff& a) { r = a; fp2::neg(r.y, a.y); }

// one line of the Miller loop for a fixed Q:  l(P) = xi*yP + (lam*xT - yT) w^3 - (lam * xP) w^5
struct G2Line {
    Fp2 lam, c;   // slope, and lam*xT - yT
};
struct G2Prepared {
    std::vector<G2Line> lines;   // in loop order: per bit one tangent line, plus a chord when the bit is set
    bool inf = false;
};
inline void g2_prepare(G2Prepared& out, const G2Aff& Q) {
    out.lines.clear();
    out.inf = Q.inf;
    if (Q.inf) return;
    Fp2 Tx = Q.x, Ty = Q.y, lam, t, x3, y3;
    for (int i = 62; i >= 0; i--) {
        // tangent at T
        fp2::sqr(lam, Tx);
        fp2::dbl(t, lam);
        fp2::add(lam, lam, t);
        fp2::dbl(t, Ty);
        fp2::inv(t, t);
        fp2::mul(lam, lam, t);
        G2Line ln;
        ln.lam = lam;
        fp2::mul(t, lam, Tx);
        fp2::sub(ln.c, t, Ty);
        out.lines.push_back(ln);
        fp2::sqr(x3, lam);
        fp2::sub(x3, x3, Tx);
        fp2::sub(x3, x3, Tx);
        fp2::sub(t, Tx, x3);
        fp2::mul(y3, lam, t);
        fp2::sub(y3, y3, Ty);
        Tx = x3;
        Ty = y3;
        if ((BLS_X_ABS >> i) & 1) {
            // chord through T and Q
            fp2::sub(lam, Q.y, Ty);
            fp2::sub(t, Q.x, Tx);
            fp2::inv(t, t);
            fp2::mul(lam, lam, t);
            ln.lam = lam;
            fp2::mul(t, lam, Tx);
            fp2::sub(ln.c, t, Ty);
            out.lines.push_back(ln);
            fp2::sqr(x3, lam);
            fp2::sub(x3, x3, Tx);
            fp2::sub(x3, x3, Q.x);
            fp2::sub(t, Tx, x3);
            fp2::mul(y3, lam, t);
            fp2::sub(y3, y3, Ty);
            Tx = x3;
            Ty = y3;
        }
    }
}

struct G1AffH {   // host affine G1 point, Montgomery form
    fe x, y;
    bool inf;
};

// f *= l, l = (a, 0, 0) + (0, c, d) w  with a in Fp embedded as xi*yP = (yP, yP)
// f *= l for the sparse line value l = l0 + l1 w,  l0 = (yP (1+u), 0, 0),  l1 = (0, c, -lam xP):
//   f0 l0 = yP * xi * f0 (6 Fp multiplications), f1 l1 a sparse Fp6 product (6 Fp2 multiplications), and one dense Fp6 product
//   for the Karatsuba cross term: 42 Fp multiplications instead of the 54 of a general Fp12 product.
inline void mul_by_line(Fp12& f, const fe& yP, const fe& xP, const G2Line& ln) {
    Fp2 b = ln.c, c, t, u;
    fp2::mul_fp(t, ln.lam, xP);
    fp2::neg(c, t);
    // t0 = f0 * l0
    Fp6 t0, t1, s, m, l01;
    fp2::mul_xi(t, f.c0.c0); fp2::mul_fp(t0.c0, t, yP);
    fp2::mul_xi(t, f.c0.c1); fp2::mul_fp(t0.c1, t, yP);
    fp2::mul_xi(t, f.c0.c2); fp2::mul_fp(t0.c2, t, yP);
    // t1 = f1 * (0, b, c) = xi (f11 c + f12 b) + (f10 b + xi f12 c) v + (f10 c + f11 b) v^2
    fp2::mul(t, f.c1.c1, c); fp2::mul(u, f.c1.c2, b); fp2::add(t, t, u); fp2::mul_xi(t1.c0, t);
    fp2::mul(t, f.c1.c2, c); fp2::mul_xi(t, t); fp2::mul(u, f.c1.c0, b); fp2::add(t1.c1, t, u);
    fp2::mul(t, f.c1.c0, c); fp2::mul(u, f.c1.c1, b); fp2::add(t1.c2, t, u);
    // m = (f0 + f1)(l0 + l1) - t0 - t1
    fp6::add(s, f.c0, f.c1);
    F::set_zero(l01.c0.c0); F::set_zero(l01.c0.c1);
    l01.c0.c0 = yP; l01.c0.c1 = yP;
    l01.c1 = b;
    l01.c2 = c;
    fp6::mul(m, s, l01);
    fp6::sub(m, m, t0);
    fp6::sub(m, m, t1);
    fp6::mul_v(t1, t1);
    fp6::add(f.c0, t0, t1);
    f.c1 = m;
}

// Frobenius: f = sum_i c_i w^i -> sum_i conj(c_i) gamma^i w^i, gamma = xi^((p-1)/6)
struct FrobConst {
    Fp2 g[6];
    FrobConst() {
        Fp2 xi;
        F::set_one(xi.c0);
        F::set_one(xi.c1);
        fp2::one(g[0]);
        fp2::pow(g[1], xi, HostFpParams::EXP_PM1_DIV6, 381);
        for (int i = 2; i < 6; i++) fp2::mul(g[i], g[i - 1], g[1]);
    }
};
inline const FrobConst& frob_const() {
    static const FrobConst fc;
    return fc;
}
inline void frobenius(Fp12& r, const Fp12& a) {
    const FrobConst& fc = frob_const();
    // tower slots as powers of w:  c0.c0=w^0  c1.c0=w^1  c0.c1=w^2  c1.c1=w^3  c0.c2=w^4  c1.c2=w^5
    Fp2 t;
    fp2::conj(r.c0.c0, a.c0.c0);
    fp2::conj(t, a.c1.c0); fp2::mul(r.c1.c0, t, fc.g[1]);
    fp2::conj(t, a.c0.c1); fp2::mul(r.c0.c1, t, fc.g[2]);
    fp2::conj(t, a.c1.c1); fp2::mul(r.c1.c1, t, fc.g[3]);
    fp2::conj(t, a.c0.c2); fp2::mul(r.c0.c2, t, fc.g[4]);
    fp2::conj(t, a.c1.c2); fp2::mul(r.c1.c2, t, fc.g[5]);
}
// a^2 for a in the cyclotomic subgroup (a^(p^6+1) = 1, i.e. after the easy part of the final exponentiation):
// Granger-Scott, "Faster squaring in the cyclotomic subgroup of sixth degree extensions" -- three Fp4 squarings
// (6 Fp2 multiplications) instead of the 12 of a general Fp12 squaring.  With Fp12 = Fp6[w]/(w^2 - v), Fp6 = Fp2[v]/(v^3 - xi):
inline void cyclotomic_sqr(Fp12& r, const Fp12& a) {
    const Fp2 &r0 = a.c0.c0, &r4 = a.c0.c1, &r3 = a.c0.c2, &r2 = a.c1.c0, &r1 = a.c1.c1, &r5 = a.c1.c2;
    Fp2 t0, t1, t2, t3, t4, t5, tmp, s, u;
    auto fp4_sqr = [&](Fp2& lo, Fp2& hi, const Fp2& x, const Fp2& y) {   // (x + y Y)^2 = lo + hi Y with Y^2 = xi
        fp2::mul(tmp, x, y);
        fp2::add(s, x, y);
        fp2::mul_xi(u, y);
        fp2::add(u, u, x);
        fp2::mul(lo, s, u);
        fp2::sub(lo, lo, tmp);
        fp2::mul_xi(u, tmp);
        fp2::sub(lo, lo, u);
        fp2::add(hi, tmp, tmp);
    };
    fp4_sqr(t0, t1, r0, r1);
    fp4_sqr(t2, t3, r2, r3);
    fp4_sqr(t4, t5, r4, r5);
    Fp2 z0, z1, z2, z3, z4, z5;
    fp2::sub(z0, t0, r0); fp2::add(z0, z0, z0); fp2::add(z0, z0, t0);       // 3 t0 - 2 z0
    fp2::add(z1, t1, r1); fp2::add(z1, z1, z1); fp2::add(z1, z1, t1);       // 3 t1 + 2 z1
    fp2::mul_xi(tmp, t5);
    fp2::add(z2, tmp, r2); fp2::add(z2, z2, z2); fp2::add(z2, z2, tmp);     // 3 xi t5 + 2 z2
    fp2::sub(z3, t4, r3); fp2::add(z3, z3, z3); fp2::add(z3, z3, t4);       // 3 t4 - 2 z3
    fp2::sub(z4, t2, r4); fp2::add(z4, z4, z4); fp2::add(z4, z4, t2);       // 3 t2 - 2 z4
    fp2::add(z5, t3, r5); fp2::add(z5, z5, z5); fp2::add(z5, z5, t3);       // 3 t3 + 2 z5
    r.c0.c0 = z0; r.c0.c1 = z4; r.c0.c2 = z3; r.c1.c0 = z2; r.c1.c1 = z1; r.c1.c2 = z5;
}
// a^|x| for a in the cyclotomic subgroup, scalar tower (also the reference the IFMA chain below is tested against)
inline void exp_by_x_abs_scalar(Fp12& r, const Fp12& a) {
    Fp12 acc = a;
    for (int i = 62; i >= 0; i--) {
        cyclotomic_sqr(acc, acc);
        if ((BLS_X_ABS >> i) & 1) fp12::mul(acc, acc, a);
    }
    r = acc;
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
// The same on AVX-512 IFMA (host_cyclo_ifma.hpp): the chain of 63 squarings on eight-lane 52-bit-limb arithmetic, the six powers
// a^(2^k), k = 16, 48, 57, 60, 62, 63 (the set bits of |x|) handed back to the scalar tower and multiplied there.
namespace ifma {
// canonical 6 x 64-bit Montgomery (R = 2^384) words <-> 8 limbs of 52 bits
static inline void to52(uint64_t out[8], const uint64_t w[6]) {
    const uint64_t M = (1ull << 52) - 1;
    out[0] = w[0] & M;
    out[1] = ((w[0] >> 52) | (w[1] << 12)) & M;
    out[2] = ((w[1] >> 40) | (w[2] << 24)) & M;
    out[3] = ((w[2] >> 28) | (w[3] << 36)) & M;
    out[4] = ((w[3] >> 16) | (w[4] << 48)) & M;
    out[5] = (w[4] >> 4) & M;
    out[6] = ((w[4] >> 56) | (w[5] << 8)) & M;
    out[7] = w[5] >> 44;
}
static inline void from52(uint64_t w[6], const uint64_t l[8]) {          // value below 2^384
    w[0] = l[0] | (l[1] << 52);
    w[1] = (l[1] >> 12) | (l[2] << 40);
    w[2] = (l[2] >> 24) | (l[3] << 28);
    w[3] = (l[3] >> 36) | (l[4] << 16);
    w[4] = (l[4] >> 48) | (l[5] << 4) | (l[6] << 56);
    w[5] = (l[6] >> 8) | (l[7] << 44);
}
// the Fp4 pairs of the Granger-Scott formula: (r0, r1), (r2, r3), (r4, r5) = (c0.c0, c1.c1), (c1.c0, c0.c2), (c0.c1, c1.c2)
static inline Fp2* pair_x(Fp12& a, int k) { return k == 0 ? &a.c0.c0 : (k == 1 ? &a.c1.c0 : &a.c0.c1); }
static inline Fp2* pair_y(Fp12& a, int k) { return k == 0 ? &a.c1.c1 : (k == 1 ? &a.c0.c2 : &a.c1.c2); }
__attribute__((target("avx512f,avx512ifma"))) static inline void load_state(V8 V[3], const Fp12& a_in) {
    Fp12 a = a_in;
    V8 c448;
    splat(c448, C448);
    for (int k = 0; k < 3; k++) {
        alignas(64) uint64_t buf[8][8];
        memset(buf, 0, sizeof buf);
        const fe* co[4] = {&pair_x(a, k)->c0, &pair_x(a, k)->c1, &pair_y(a, k)->c0, &pair_y(a, k)->c1};
        for (int lane = 0; lane < 4; lane++) {
            uint64_t l[8];
            to52(l, co[lane]->l);
            for (int q = 0; q < 8; q++) buf[q][lane] = l[q];
        }
        V8 raw;
        for (int q = 0; q < 8; q++) raw.l[q] = _mm512_load_si512((const void*)buf[q]);
        mul8(V[k], raw, c448);                      // v 2^384 -> v 2^416 (and below 2p)
    }
}
__attribute__((target("avx512f,avx512ifma"))) static inline void store_state(Fp12& out, const V8 V[3]) {
    V8 c384;
    splat(c384, C384);
    for (int k = 0; k < 3; k++) {
        V8 w;
        mul8(w, V[k], c384);                        // v 2^416 -> v 2^384, below 2p
        alignas(64) uint64_t buf[8][8];
        for (int q = 0; q < 8; q++) _mm512_store_si512((void*)buf[q], w.l[q]);
        fe* co[4] = {&pair_x(out, k)->c0, &pair_x(out, k)->c1, &pair_y(out, k)->c0, &pair_y(out, k)->c1};
        for (int lane = 0; lane < 4; lane++) {
            uint64_t l[8], t[6];
            for (int q = 0; q < 8; q++) l[q] = buf[q][lane];
            from52(t, l);
            if (HostFp::geq_mod(t)) HostFp::sub_raw(t, t, HostFp::P::MOD);
            memcpy(co[lane]->l, t, 48);
        }
    }
}
// n cyclotomic squarings of a (tests)
__attribute__((target("avx512f,avx512ifma"))) static inline void cyclo_sqr_n(Fp12& r, const Fp12& a, int n) {
    V8 V[3];
    load_state(V, a);
    for (int i = 0; i < n; i++) cyclo_sqr(V[0], V[1], V[2]);
    store_state(r, V);
}
__attribute__((target("avx512f,avx512ifma"))) static inline void exp_by_x_abs(Fp12& r, const Fp12& a) {
    V8 V[3];
    load_state(V, a);
    bool first = true;
    for (int s = 1; s <= 63; s++) {
        cyclo_sqr(V[0], V[1], V[2]);
        if ((BLS_X_ABS >> s) & 1) {
            Fp12 snap;
            store_state(snap, V);
            if (first) { r = snap; first = false; }
            else fp12::mul(r, r, snap);
        }
    }
}
}  // namespace ifma
#endif
inline void exp_by_x_abs(Fp12& r, const Fp12& a) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    if (ifma::available()) { ifma::exp_by_x_abs(r, a); return; }
#endif
    Fp12 acc = a;
    for (int i = 62; i >= 0; i--) {
        cyclotomic_sqr(acc, acc);
        if ((BLS_X_ABS >> i) & 1) fp12::mul(acc, acc, a);
    }
    r = acc;
}
// a^x, x = -|x|: inverse = conjugate in the cyclotomic subgroup
inline void exp_by_x(Fp12& r, const Fp12& a) {
    exp_by_x_abs(r, a);
    fp12::conj(r, r);
}
inline void final_exponentiation(Fp12& r, const Fp12& f0) {
    // easy part: f^((p^6-1)(p^2+1))
    Fp12 f, t, a, b, c;
    fp12::conj(t, f0);
    fp12::inv(f, f0);
    fp12::mul(f, t, f);
    frobenius(t, f);
    frobenius(t, t);
    fp12::mul(f, t, f);
    // hard part: f^((x-1)^2 (x+p) (x^2+p^2-1) + 3)
    exp_by_x(a, f);
    fp12::conj(t, f);
    fp12::mul(a, a, t);           // f^(x-1)
    exp_by_x(b, a);
    fp12::conj(t, a);
    fp12::mul(a, b, t);           // f^((x-1)^2)
    exp_by_x(b, a);
    frobenius(t, a);
    fp12::mul(b, b, t);           // a^(x+p)
    exp_by_x(c, b);
    exp_by_x(c, c);               // b^(x^2)
    frobenius(t, b);
    frobenius(t, t);
    fp12::mul(c, c, t);           // * b^(p^2)
    fp12::conj(t, b);
    fp12::mul(c, c, t);           // * b^-1
    fp12::sqr(t, f);
    fp12::mul(t, t, f);           // f^3
    fp12::mul(r, c, t);
}
// Miller loop of ONE pair (f not yet conjugated for x < 0)
inline void miller_loop_prepared(Fp12& f, const G1AffH& P, const G2Prepared& Q) {
    fp12::one(f);
    size_t idx = 0;
    for (int i = 62; i >= 0; i--) {
        fp12::sqr(f, f);
        mul_by_line(f, P.y, P.x, Q.lines[idx++]);
        if ((BLS_X_ABS >> i) & 1) mul_by_line(f, P.y, P.x, Q.lines[idx++]);
    }
}
// A SEGMENT of the Miller loop of one pair: the steps of bits hi..lo of |x| started from f = 1, then squared lo more times -- with the
// prepared lines every step's line value is independent of f, so  f_loop = prod_segments segment(hi_j, lo_j)  and the segments of a
// loop can run on different threads (each pays its trailing squarings: 1.8 us each against 3.7 us for a step).
inline void miller_segment_prepared(Fp12& f, const G1AffH& P, const G2Prepared& Q, int hi, int lo) {
    fp12::one(f);
    size_t idx = (size_t)(62 - hi);
    for (int i = 62; i > hi; i--) idx += (BLS_X_ABS >> i) & 1;
    bool started = false;
    for (int i = hi; i >= lo; i--) {
        if (started) fp12::sqr(f, f);
        mul_by_line(f, P.y, P.x, Q.lines[idx++]);
        if ((BLS_X_ABS >> i) & 1) mul_by_line(f, P.y, P.x, Q.lines[idx++]);
        started = true;
    }
    for (int i = 0; i < lo; i++) fp12::sqr(f, f);
}
// e(P_0, Q_0) e(P_1, Q_1) == 1 with the Miller loops on host threads: one loop per thread (the shared-squaring loop below does 63
// squarings + 136 line products in sequence; two threads do 63 + 68 each and one extra product), or, when the pool has threads to spare
// (`segments` 2 or 3: a lone caller waits for this), every loop cut into segments of equal duration on a thread each
// (240 us -> ~155 / ~130 us per loop on an EPYC 9575F for 60 % / 100 % more CPU time)
inline bool pairing_check_prepared_2mt(const G1AffH* Ps, const G2Prepared* const* Qs, HostPool* pool = nullptr, int segments = 1);

// prod_i e(P_i, Q_i) == 1 for prepared Q_i; pairs with an infinity member contribute 1
inline bool pairing_check_prepared(const G1AffH* Ps, const G2Prepared* const* Qs, int n) {
    Fp12 f;
    fp12::one(f);
    std::vector<int> live;
    for (int k = 0; k < n; k++) if (!Ps[k].inf && !Qs[k]->inf) live.push_back(k);
    if (!live.empty()) {
        size_t idx = 0;
        for (int i = 62; i >= 0; i--) {
            fp12::sqr(f, f);
            for (int k : live) mul_by_line(f, Ps[k].y, Ps[k].x, Qs[k]->lines[idx]);
            idx++;
            if ((BLS_X_ABS >> i) & 1) {
                for (int k : live) mul_by_line(f, Ps[k].y, Ps[k].x, Qs[k]->lines[idx]);
                idx++;
            }
        }
        fp12::conj(f, f);   // x < 0
    }
    Fp12 e;
    final_exponentiation(e, f);
    return fp12::is_one(e);
}

inline bool pairing_check_prepared_2mt(const G1AffH* Ps, const G2Prepared* const* Qs, HostPool* pool, int segments) {
    if (Ps[0].inf || Qs[0]->inf || Ps[1].inf || Qs[1]->inf) return pairing_check_prepared(Ps, Qs, 2);
    Fp12 fs[6], f, e;
    int nf = 2;
    if (pool && segments >= 2) {
        // segment bounds (bits of |x|, 62..0) that equalise  3.7 us x steps + 1.8 us x trailing squarings
        static const int B2[3] = {62, 40, -1}, B3[4] = {62, 53, 36, -1};
        const int K = segments >= 3 ? 3 : 2;
        const int* B = K == 3 ? B3 : B2;
        nf = 2 * K;
        pool->parallel(nf, [&](int t) { const int pair = t / K, j = t % K; miller_segment_prepared(fs[t], Ps[pair], *Qs[pair], B[j], B[j + 1] + 1); });
    } else if (pool) {
        pool->parallel(2, [&](int i) { miller_loop_prepared(fs[i], Ps[i], *Qs[i]); });
    } else {
        std::thread t([&] { miller_loop_prepared(fs[1], Ps[1], *Qs[1]); });
        miller_loop_prepared(fs[0], Ps[0], *Qs[0]);
        t.join();
    }
    f = fs[0];
    for (int i = 1; i < nf; i++) fp12::mul(f, f, fs[i]);
    fp12::conj(f, f);   // x < 0
    final_exponentiation(e, f);
    return fp12::is_one(e);
}

}  // namespace kzg
