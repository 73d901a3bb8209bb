// g1.hpp -- BLS12-381 G1 arithmetic for the device kernels: affine points and extended
// Jacobian ("XYZZ": x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) accumulators over the loose radix-2^29
// field of mont29.hpp.
//
// Replaces, on the device, what the reference gets from gnark-crypto's G1Affine/G1Jac/
// g1JacExtended (bucket accumulators of G1Affine.MultiExp, reached from
// internal/multiexp/multiexp.go:25).  Formulas: EFD "xyzz" madd-2008-s / add-2008-s /
// dbl-2008-s-1 / mdbl-2008-s-1 for a = 0, with every exceptional case handled exactly
// (infinity operands, P + P, P - P), because the reference's vectors exercise them (zero blob,
// constant polynomials, infinity commitments).
//
// Loose-value bounds maintained for an accumulator (see mont29.hpp): X < 8p, Y < 4p, ZZ, ZZZ < 1.01p.
#pragma once
#include "mont29.hpp"

namespace kzg {

struct G1Aff {   // affine, Montgomery form; inf != 0 marks the identity
    fp_t x, y;
    uint32_t inf;
};
struct G1X {     // XYZZ; ZZ == all-zero limbs marks the identity
    fp_t x, y, zz, zzz;
};

KZ_HD bool g1x_is_inf(const G1X& p) { return Fp::limbs_all_zero(p.zz); }
KZ_HD void g1x_set_inf(G1X& p) {
    Fp::set_one(p.x);
    Fp::set_one(p.y);
    Fp::set_zero(p.zz);
    Fp::set_zero(p.zzz);
}
KZ_HD void g1x_from_affine(G1X& r, const fp_t& x, const fp_t& y) {
    r.x = x;
    r.y = y;
    Fp::set_one(r.zz);
    Fp::set_one(r.zzz);
}

// r = 2*(x,y) for an affine point (mdbl-2008-s-1, a = 0)
KZ_HD void g1x_dbl_affine(G1X& r, const fp_t& x, const fp_t& y) {
    fp_t U, V, W, S, M, t, X3;
    Fp::dbl(U, y);
    Fp::sqr(V, U);
    Fp::mul(W, U, V);
    Fp::mul(S, x, V);
    Fp::sqr(t, x);
    Fp::dbl(M, t);
    Fp::add(M, M, t);
    Fp::sqr(X3, M);
    Fp::dbl(t, S);
    Fp::template sub<2>(X3, X3, t);        // M^2 - 2S + 4p        (< 5.01p)
    Fp::template sub<3>(t, S, X3);         // S - X3 + 8p
    Fp::mul(t, M, t);
    Fp::mul(S, W, y);
    Fp::template sub<1>(r.y, t, S);        // < 3.01p
    r.x = X3;
    r.zz = V;
    r.zzz = W;
}
// r = 2*p (dbl-2008-s-1, a = 0)
KZ_HD void g1x_dbl(G1X& r, const G1X& p) {
    if (g1x_is_inf(p)) { r = p; return; }
    fp_t U, V, W, S, M, t, X3, Y3;
    Fp::dbl(U, p.y);
    Fp::sqr(V, U);
    Fp::mul(W, U, V);
    Fp::mul(S, p.x, V);
    Fp::sqr(t, p.x);
    Fp::dbl(M, t);
    Fp::add(M, M, t);
    Fp::sqr(X3, M);
    Fp::dbl(t, S);
    Fp::template sub<2>(X3, X3, t);
    Fp::template sub<3>(t, S, X3);
    Fp::mul(t, M, t);
    Fp::mul(S, W, p.y);
    Fp::template sub<1>(Y3, t, S);
    Fp::mul(r.zz, V, p.zz);
    Fp::mul(r.zzz, W, p.zzz);
    r.x = X3;
    r.y = Y3;
}

// acc += (x2, y2) with (x2, y2) affine and not the identity (madd-2008-s)
KZ_HD void g1x_madd(G1X& acc, const fp_t& x2, const fp_t& y2) {
    if (g1x_is_inf(acc)) { g1x_from_affine(acc, x2, y2); return; }
    fp_t U2, S2, Pd, Rd, PP, PPP, Q, t, X3;
    Fp::mul(U2, x2, acc.zz);
    Fp::mul(S2, y2, acc.zzz);
    Fp::template sub<3>(Pd, U2, acc.x);    // U2 - X1 + 8p   in (0.99p, 9.01p)
    Fp::template sub<2>(Rd, S2, acc.y);    // S2 - Y1 + 4p   in (0.99p, 5.01p)
    if (Fp::template is_zero_mod<16>(Pd)) {
        if (Fp::template is_zero_mod<16>(Rd)) g1x_dbl_affine(acc, x2, y2);
        else g1x_set_inf(acc);
        return;
    }
    Fp::sqr(PP, Pd);
    Fp::mul(PPP, Pd, PP);
    Fp::mul(Q, acc.x, PP);
    Fp::sqr(X3, Rd);
    Fp::template sub<1>(X3, X3, PPP);      // R^2 - PPP + 2p        (< 3.01p)
    Fp::dbl(t, Q);
    Fp::template sub<2>(X3, X3, t);        // ... - 2Q + 4p         (< 7.01p)
    Fp::template sub<3>(t, Q, X3);         // Q - X3 + 8p
    Fp::mul(t, Rd, t);
    Fp::mul(S2, acc.y, PPP);
    Fp::template sub<1>(acc.y, t, S2);     // < 3.01p
    acc.x = X3;
    Fp::mul(acc.zz, acc.zz, PP);
    Fp::mul(acc.zzz, acc.zzz, PPP);
}

// r = a + b, both XYZZ (add-2008-s)
KZ_HD void g1x_add(G1X& r, const G1X& a, const G1X& b) {
    if (g1x_is_inf(a)) { r = b; return; }
    if (g1x_is_inf(b)) { r = a; return; }
    fp_t U1, U2, S1, S2, Pd, Rd, PP, PPP, Q, t, X3, Y3;
    Fp::mul(U1, a.x, b.zz);
    Fp::mul(U2, b.x, a.zz);
    Fp::mul(S1, a.y, b.zzz);
    Fp::mul(S2, b.y, a.zzz);
    Fp::template sub<1>(Pd, U2, U1);
    Fp::template sub<1>(Rd, S2, S1);
    if (Fp::template is_zero_mod<16>(Pd)) {
        if (Fp::template is_zero_mod<16>(Rd)) g1x_dbl(r, a);
        else g1x_set_inf(r);
        return;
    }
    Fp::sqr(PP, Pd);
    Fp::mul(PPP, Pd, PP);
    Fp::mul(Q, U1, PP);
    Fp::sqr(X3, Rd);
    Fp::template sub<1>(X3, X3, PPP);
    Fp::dbl(t, Q);
    Fp::template sub<2>(X3, X3, t);
    Fp::template sub<3>(t, Q, X3);
    Fp::mul(t, Rd, t);
    Fp::mul(S1, S1, PPP);
    Fp::template sub<1>(Y3, t, S1);
    Fp::mul(t, a.zz, b.zz);
    Fp::mul(r.zz, t, PP);
    Fp::mul(t, a.zzz, b.zzz);
    Fp::mul(r.zzz, t, PPP);
    r.x = X3;
    r.y = Y3;
}
KZ_HD void g1x_neg(G1X& r, const G1X& a) {
    r = a;
    Fp::template neg<2>(r.y, a.y);   // 4p - Y (Y < 4p)
}

// XYZZ -> canonical affine (x, y in [0,p), Montgomery form).  One field inversion.
KZ_HD void g1x_to_affine(G1Aff& r, const G1X& p) {
    if (g1x_is_inf(p)) {
        Fp::set_zero(r.x);
        Fp::set_zero(r.y);
        r.inf = 1;
        return;
    }
    // 1/ZZ = ZZZ/(ZZ*ZZZ) ... with ZZ^3 = ZZZ^2 : 1/ZZZ = (ZZ/ZZZ)^3 /... keep it simple: invert t = ZZ*ZZZ
    fp_t t, ti, izz, izzz;
    Fp::mul(t, p.zz, p.zzz);
    Fp::pow_words(ti, t, FpParams::EXP_INV, FpParams::BITS);
    Fp::mul(izz, ti, p.zzz);
    Fp::mul(izzz, ti, p.zz);
    Fp::mul(t, p.x, izz);
    Fp::canon(r.x, t);
    Fp::mul(t, p.y, izzz);
    Fp::canon(r.y, t);
    r.inf = 0;
}

// zcash-format compression of a canonical affine point into 12 big-endian-ordered words
// (reference serialization.go:98-100 -> gnark G1Affine.Bytes).  out[0] is the most significant
// word; the caller stores the 48 bytes with a byte swap per word.
KZ_HD void g1_compress_words(uint32_t out_be[12], const G1Aff& a) {
    if (a.inf) {
#pragma unroll
        for (int i = 0; i < 12; i++) out_be[i] = 0;
        out_be[0] = 0xc0000000u;
        return;
    }
    fp_t xp, yp;
    uint32_t xw[12], yw[12];
    Fp::to_plain(xp, a.x);
    Fp::to_plain(yp, a.y);
    Fp::to_words32(xw, xp);
    Fp::to_words32(yw, yp);
    // y > (p-1)/2 ?
    bool gt = false, lt = false;
#pragma unroll
    for (int j = 11; j >= 0; j--) {
        if (!gt && !lt) {
            if (yw[j] > FpParams::HALF[j]) gt = true;
            else if (yw[j] < FpParams::HALF[j]) lt = true;
        }
    }
#pragma unroll
    for (int i = 0; i < 12; i++) out_be[i] = xw[11 - i];
    out_be[0] |= 0x80000000u | (gt ? 0x20000000u : 0u);
}

}  // namespace kzg
