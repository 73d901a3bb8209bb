// g1.hpp -- BLS12-381 G1 arithmetic shared by the device kernels and the product's host code:
// affine points and extended Jacobian ("XYZZ": x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) accumulators,
// templated on the field implementation F:
//     F = kzg::Fp      (mont29.hpp)     loose radix-2^29 field, device kernels (and host tests)
//     F = kzg::HostFp  (host_field.hpp) canonical 6 x 64-bit field, host-side finishing steps
//
// Replaces what the reference gets from gnark-crypto's G1Affine/G1Jac/g1JacExtended (bucket
// accumulators of G1Affine.MultiExp, reached from internal/multiexp/multiexp.go:25; the group
// operations of internal/kzg/kzg_verify.go:150-188).  Formulas: EFD "xyzz" madd-2008-s /
// add-2008-s / dbl-2008-s-1 / mdbl-2008-s-1 for a = 0, with every exceptional case handled
// exactly (infinity operands, P + P, P - P), because the reference's vectors exercise them (zero
// blob, constant polynomials, infinity commitments).
//
// Loose-value bounds maintained for an accumulator when F = Fp (see mont29.hpp):
//     X < 8p, Y < 4p, ZZ, ZZZ < 1.01p.   The sub<KLOG> template arguments encode the bias needed.
#pragma once
#include "mont29.hpp"

namespace kzg {

template <class F>
struct G1T {
    using fe = typename F::El;
    struct Aff {   // affine, Montgomery form; inf != 0 marks the identity
        fe x, y;
        uint32_t inf;
    };
    struct XYZZ {  // ZZ == all-zero limbs marks the identity
        fe x, y, zz, zzz;
    };

    static KZ_HD bool is_inf(const XYZZ& p) { return F::limbs_all_zero(p.zz); }
    static KZ_HD void opaque(XYZZ& p) {   // see mont29.hpp KZ_OPAQUE32
        F::opaque(p.x);
        F::opaque(p.y);
        F::opaque(p.zz);
        F::opaque(p.zzz);
    }
    static KZ_HD void set_inf(XYZZ& p) {
        F::set_one(p.x);
        F::set_one(p.y);
        F::set_zero(p.zz);
        F::set_zero(p.zzz);
    }
    static KZ_HD void from_affine(XYZZ& r, const fe& x, const fe& y) {
        r.x = x;
        r.y = y;
        F::set_one(r.zz);
        F::set_one(r.zzz);
    }
    static KZ_HD void from_aff(XYZZ& r, const Aff& a) {
        if (a.inf) set_inf(r);
        else from_affine(r, a.x, a.y);
    }

    // r = 2*(x,y) for an affine point (mdbl-2008-s-1, a = 0)
    static KZ_HD void dbl_affine(XYZZ& r, const fe& x, const fe& y) {
        fe U, V, W, S, M, t, X3;
        F::dbl(U, y);
        F::sqr(V, U);
        F::mul(W, U, V);
        F::mul(S, x, V);
        F::sqr(t, x);
        F::triple(M, t);
        F::sqr(X3, M);
        F::template sub_2c<2>(X3, X3, S);     // M^2 - 2S + 4p        (< 5.01p)
        F::template sub<3>(t, S, X3);         // S - X3 + 8p
        F::template neg<1>(S, y);             // 2p - y
        F::mul2(r.y, M, t, W, S);             // M (S - X3) - W y      (< 1.01p)
        r.x = X3;
        r.zz = V;
        r.zzz = W;
    }
    // r = 2*p (dbl-2008-s-1, a = 0).  No special case for the identity: ZZ = 0 gives ZZ3 = V*0 = 0 (all-zero limbs) again.
    static KZ_HD void dbl(XYZZ& r, const XYZZ& p_) {
        XYZZ p = p_;
        opaque(p);
        fe U, V, W, S, M, t, X3, Y3;
        F::dbl(U, p.y);
        F::sqr(V, U);
        F::mul(W, U, V);
        F::mul(S, p.x, V);
        F::sqr(t, p.x);
        F::triple(M, t);
        F::sqr(X3, M);
        F::template sub_2c<2>(X3, X3, S);
        F::template sub<3>(t, S, X3);
        F::template neg<2>(S, p.y);           // 4p - Y
        F::mul2(Y3, M, t, W, S);              // M (S - X3) - W Y      (< 1.01p)
        F::mul(r.zz, V, p.zz);
        F::mul(r.zzz, W, p.zzz);
        r.x = X3;
        r.y = Y3;
    }

    // acc += (x2, y2) with (x2, y2) affine and not the identity (madd-2008-s)
    static KZ_HD void madd(XYZZ& acc, const fe& x2_, const fe& y2_) {
        // the F::opaque calls sit at the head of each basic block that multiplies values defined in an earlier
        // block (instruction selection only sees 32-bit-ness inside one block; see mont29.hpp KZ_OPAQUE32)
        fe x2 = x2_, y2 = y2_;
        if (is_inf(acc)) { from_affine(acc, x2, y2); return; }
        opaque(acc);
        F::opaque(x2);
        F::opaque(y2);
        fe U2, S2, Pd, Rd, PP, PPP, Q, t, X3;
        F::mul(U2, x2, acc.zz);
        F::mul(S2, y2, acc.zzz);
        F::template sub<3>(Pd, U2, acc.x);    // U2 - X1 + 8p   in (0.99p, 9.01p)
        F::template sub<2>(Rd, S2, acc.y);    // S2 - Y1 + 4p   in (0.99p, 5.01p)
        if (F::template is_zero_mod<16>(Pd)) {
            if (F::template is_zero_mod<16>(Rd)) dbl_affine(acc, x2, y2);
            else set_inf(acc);
            return;
        }
        opaque(acc);
        F::opaque(Pd);
        F::opaque(Rd);
        F::sqr(PP, Pd);
        F::mul(PPP, Pd, PP);
        F::mul(Q, acc.x, PP);
        F::mul(acc.zz, acc.zz, PP);           // PP, Pd and X1 are dead from here on (keeps the live set small for R^2)
        F::mul(acc.zzz, acc.zzz, PPP);
        F::sqr(X3, Rd);
        F::template sub_b_2c<2>(X3, X3, PPP, Q);   // R^2 - PPP - 2Q + 4p   (PPP + 2Q < 3.03p; result < 5.01p)
        F::template sub<3>(t, Q, X3);         // Q - X3 + 8p
        F::template neg<2>(S2, acc.y);        // 4p - Y1
        F::mul2(acc.y, Rd, t, S2, PPP);       // R (Q - X3) - Y1 PPP   (< 1.01p), one reduction for both products
        acc.x = X3;
    }

    // r = a + b, both XYZZ (add-2008-s).  Identity operands are resolved with per-limb selects at the end, not with early
    // returns that copy whole structs: LLVM merges such copies into one memcpy from a phi of pointers, which pins every
    // XYZZ involved to scratch memory (seen as 220 B of scratch per lane in any kernel that inlines this).
    static KZ_HD void add(XYZZ& r, const XYZZ& a_, const XYZZ& b_) {
        XYZZ a = a_, b = b_;
        opaque(a);
        opaque(b);
        const bool a_inf = is_inf(a), b_inf = is_inf(b);
        fe U1, U2, S1, S2, Pd, Rd, PP, PPP, Q, t, X3, Y3;
        XYZZ res;
        F::mul(U1, a.x, b.zz);
        F::mul(U2, b.x, a.zz);
        F::mul(S1, a.y, b.zzz);
        F::mul(S2, b.y, a.zzz);
        F::template sub<1>(Pd, U2, U1);
        F::template sub<1>(Rd, S2, S1);
        if (!a_inf && !b_inf && F::template is_zero_mod<16>(Pd)) {
            if (F::template is_zero_mod<16>(Rd)) dbl(res, a);
            else set_inf(res);
        } else {
            opaque(a);
            opaque(b);
            F::opaque(Pd);
            F::opaque(Rd);
            F::opaque(U1);
            F::opaque(S1);
            F::sqr(PP, Pd);
            F::mul(PPP, Pd, PP);
            F::mul(Q, U1, PP);
            F::sqr(X3, Rd);
            F::template sub_b_2c<2>(X3, X3, PPP, Q);
            F::template sub<3>(t, Q, X3);
            F::template neg<1>(S1, S1);           // 2p - S1
            F::mul2(Y3, Rd, t, S1, PPP);          // R (Q - X3) - S1 PPP   (< 1.01p)
            F::mul(t, a.zz, b.zz);
            F::mul(res.zz, t, PP);
            F::mul(t, a.zzz, b.zzz);
            F::mul(res.zzz, t, PPP);
            res.x = X3;
            res.y = Y3;
        }
        select(res, b_inf, a, res);
        select(res, a_inf, b, res);
        r = res;
    }
    // acc += b with b read THROUGH AN ACCESSOR (B::get(fe&, coordinate 0..3 = X, Y, ZZ, ZZZ): an LDS snapshot in k_vmsm_window), each
    // coordinate fetched where it is used, ZZ and ZZZ twice -- so that the second operand never occupies 56 registers beside the
    // first (round 5: the suffix network of the window sums held 344 registers + 688 B of scratch with add() on two register
    // operands).  Exact like add(): identity operands and P + P / P - P are resolved by (divergent) branches that touch limbs only.
    template <class B>
    static KZ_HD void add_from(XYZZ& acc, const B& b) {
        fe bzz, t;
        b.get(bzz, 2);
        if (F::limbs_all_zero(bzz)) return;                   // b is the identity
        if (is_inf(acc)) {                                    // acc is: the sum is b
            b.get(acc.x, 0);
            b.get(acc.y, 1);
            b.get(acc.zz, 2);
            b.get(acc.zzz, 3);
            return;
        }
        opaque(acc);
        F::opaque(bzz);
        fe U1, S1, Pd, Rd;
        F::mul(U1, acc.x, bzz);                               // X1 ZZ2
        b.get(t, 0);
        F::mul(Pd, t, acc.zz);                                // X2 ZZ1
        F::template sub<1>(Pd, Pd, U1);
        b.get(t, 3);
        F::mul(S1, acc.y, t);                                 // Y1 ZZZ2
        b.get(t, 1);
        F::mul(Rd, t, acc.zzz);                               // Y2 ZZZ1
        F::template sub<1>(Rd, Rd, S1);
        if (F::template is_zero_mod<16>(Pd)) {
            if (F::template is_zero_mod<16>(Rd)) dbl(acc, acc);
            else set_inf(acc);
            return;
        }
        opaque(acc);
        F::opaque(Pd);
        F::opaque(Rd);
        F::opaque(U1);
        F::opaque(S1);
        fe PP, PPP, Q, X3;
        F::sqr(PP, Pd);
        F::mul(PPP, Pd, PP);
        F::mul(Q, U1, PP);
        F::mul(t, acc.zz, PP);
        b.get(bzz, 2);
        F::mul(acc.zz, t, bzz);                               // ZZ1 ZZ2 PP
        F::mul(t, acc.zzz, PPP);
        b.get(bzz, 3);
        F::mul(acc.zzz, t, bzz);                              // ZZZ1 ZZZ2 PPP
        F::sqr(X3, Rd);
        F::template sub_b_2c<2>(X3, X3, PPP, Q);
        F::template sub<3>(t, Q, X3);
        F::template neg<1>(S1, S1);                           // 2p - S1
        F::mul2(acc.y, Rd, t, S1, PPP);                       // R (Q - X3) - S1 PPP   (< 1.01p)
        acc.x = X3;
    }
    static KZ_HD void select(XYZZ& r, bool c, const XYZZ& a, const XYZZ& b) {   // r = c ? a : b, limb by limb
        F::select(r.x, c, a.x, b.x);
        F::select(r.y, c, a.y, b.y);
        F::select(r.zz, c, a.zz, b.zz);
        F::select(r.zzz, c, a.zzz, b.zzz);
    }
    static KZ_HD void neg(XYZZ& r, const XYZZ& a) {
        r = a;
        F::template neg<2>(r.y, a.y);   // 4p - Y (Y < 4p)
    }

    // XYZZ -> canonical affine (x, y in [0,p), Montgomery form).  One field inversion.
    static KZ_HD void to_affine(Aff& r, const XYZZ& p) {
        if (is_inf(p)) {
            F::set_zero(r.x);
            F::set_zero(r.y);
            r.inf = 1;
            return;
        }
        // with t = ZZ*ZZZ:  1/ZZ = ZZZ/t,  1/ZZZ = ZZ/t
        fe t, ti, izz, izzz;
        F::mul(t, p.zz, p.zzz);
        F::inv(ti, t);
        F::mul(izz, ti, p.zzz);
        F::mul(izzz, ti, p.zz);
        F::mul(t, p.x, izz);
        F::canon(r.x, t);
        F::mul(t, p.y, izzz);
        F::canon(r.y, t);
        r.inf = 0;
    }

    // ------------------------------------------------------------------------------------------
    // Jacobian accumulator (x = X/Z^2, y = Y/Z^3) for the doubling-dominated chains -- the GLV twiddle multiplications of
    // the FK20 transforms and the subgroup test: a = 0 doubling "dbl-2009-l" is 5 squarings + 2 multiplications against
    // 3 + 6 (one of them a fused pair) for the XYZZ doubling; the addition takes an operand with cached Z^2, Z^3 and costs
    // the same as the XYZZ one.  Same role as gnark-crypto's G1Jac under the reference's subgroup checks (reached from
    // internal/kzg/kzg_verify.go through G1Affine.IsInSubGroup) -- formulas from the EFD, not from that code.
    // Loose bounds when F = Fp:  X < 17.01p, Y < 5.01p, Z < 1.01p.
    // ------------------------------------------------------------------------------------------
    struct Jac {   // Z == all-zero limbs marks the identity
        fe x, y, z;
    };
    struct JacC {  // addition operand: Jacobian with zz = Z^2, zzz = Z^3 cached
        fe x, y, z, zz, zzz;
    };
    static KZ_HD bool is_inf(const Jac& p) { return F::limbs_all_zero(p.z); }
    static KZ_HD void opaque(Jac& p) {
        F::opaque(p.x);
        F::opaque(p.y);
        F::opaque(p.z);
    }
    static KZ_HD void set_inf(Jac& p) {
        F::set_one(p.x);
        F::set_one(p.y);
        F::set_zero(p.z);
    }
    static KZ_HD void select(Jac& r, bool c, const Jac& a, const Jac& b) {
        F::select(r.x, c, a.x, b.x);
        F::select(r.y, c, a.y, b.y);
        F::select(r.z, c, a.z, b.z);
    }
    // XYZZ -> Jacobian with Z = ZZ:  (X ZZ, Y ZZZ, ZZ)   [x = X/ZZ = X ZZ/ZZ^2,  y = Y/ZZZ = Y ZZZ/ZZ^3 since ZZZ^2 = ZZ^3]
    static KZ_HD void jac_from_xyzz(Jac& r, const XYZZ& p) {
        F::mul(r.x, p.x, p.zz);
        F::mul(r.y, p.y, p.zzz);
        r.z = p.zz;
    }
    // Jacobian -> XYZZ (X, Y, Z^2, Z^3); X and Y pass through a multiplication by one so that the XYZZ bounds hold again
    static KZ_HD void jac_to_xyzz(XYZZ& r, const Jac& p) {
        fe one;
        F::set_one(one);
        F::sqr(r.zz, p.z);
        F::mul(r.zzz, r.zz, p.z);
        F::mul(r.x, p.x, one);
        F::mul(r.y, p.y, one);
    }
    static KZ_HD void jac_cache(JacC& r, const Jac& p) {
        r.x = p.x;
        r.y = p.y;
        r.z = p.z;
        F::sqr(r.zz, p.z);
        F::mul(r.zzz, r.zz, p.z);
    }
    static KZ_HD void jac_cache_affine(JacC& r, const fe& x, const fe& y) {
        r.x = x;
        r.y = y;
        F::set_one(r.z);
        F::set_one(r.zz);
        F::set_one(r.zzz);
    }
    // r = 2p (dbl-2009-l with B doubled up front, so that 4XY^2 = (X + 2Y^2)^2 - X^2 - 4Y^4 needs no further scaling).
    // The identity needs no branch: Z = 0 gives Z3 = 2Y * 0 = 0 (all-zero limbs).
    static KZ_HD void jdbl(Jac& r, const Jac& p_) {
        Jac p = p_;
        opaque(p);
        fe A, B2, C4, t, D, E, X3, Y3, U;
        F::sqr(A, p.x);
        F::sqr(B2, p.y);
        F::dbl(B2, B2);                           // 2 Y^2            (< 2.01p)
        F::sqr(C4, B2);                           // 4 Y^4
        F::add(t, p.x, B2);                       //                   (< 19.1p)
        F::sqr(t, t);
        F::template sub_b_c<2>(D, t, A, C4);      // D = 4 X Y^2 + 4p  (< 5.01p)
        F::triple(E, A);                          // 3 X^2
        F::sqr(X3, E);
        F::template sub_2c<4>(X3, X3, D);         // E^2 - 2D + 16p    (< 17.01p)
        F::template sub<5>(t, D, X3);             // D - X3 + 32p      (< 37.1p)
        F::mul(Y3, E, t);
        F::template sub_2c<2>(Y3, Y3, C4);        // E (D - X3) - 8 Y^4 + 4p   (< 5.01p)
        F::dbl(U, p.y);
        F::mul(r.z, U, p.z);
        r.x = X3;
        r.y = Y3;
    }
    // r = a + b (add-2007-bl with b's Z^2, Z^3 cached); every exceptional case exact, resolved with selects as in add() above
    static KZ_HD void jadd(Jac& r, const Jac& a_, const JacC& b_) {
        Jac a = a_;
        JacC b = b_;
        opaque(a);
        F::opaque(b.x);
        F::opaque(b.y);
        F::opaque(b.z);
        F::opaque(b.zz);
        F::opaque(b.zzz);
        const bool a_inf = is_inf(a), b_inf = F::limbs_all_zero(b.z);
        fe Z1Z1, Z1c, U1, U2, S1, S2, H, Rd, HH, HHH, Q, t, X3, Y3;
        Jac res, bj;
        bj.x = b.x;
        bj.y = b.y;
        bj.z = b.z;
        F::sqr(Z1Z1, a.z);
        F::mul(Z1c, Z1Z1, a.z);
        F::mul(U1, a.x, b.zz);
        F::mul(U2, b.x, Z1Z1);
        F::mul(S1, a.y, b.zzz);
        F::mul(S2, b.y, Z1c);
        F::template sub<1>(H, U2, U1);
        F::template sub<1>(Rd, S2, S1);
        if (!a_inf && !b_inf && F::template is_zero_mod<16>(H)) {
            if (F::template is_zero_mod<16>(Rd)) jdbl(res, a);
            else set_inf(res);
        } else {
            opaque(a);
            F::opaque(b.z);
            F::opaque(H);
            F::opaque(Rd);
            F::opaque(U1);
            F::opaque(S1);
            F::sqr(HH, H);
            F::mul(HHH, H, HH);
            F::mul(Q, U1, HH);
            F::sqr(X3, Rd);
            F::template sub_b_2c<2>(X3, X3, HHH, Q);   // R^2 - H^3 - 2 U1 H^2 + 4p   (< 5.01p)
            F::template sub<3>(t, Q, X3);
            F::template neg<1>(S1, S1);
            F::mul2(Y3, Rd, t, S1, HHH);                // R (U1 H^2 - X3) - S1 H^3     (< 1.01p)
            F::mul(t, a.z, b.z);
            F::mul(res.z, t, H);
            res.x = X3;
            res.y = Y3;
        }
        select(res, b_inf, a, res);
        select(res, a_inf, bj, res);
        r = res;
    }

    // ------------------------------------------------------------------------------------------
    // r-torsion test of a finite affine point of the curve: phi(P) == -[x^2]P (Scott 2021), x = -BLS_X_ABS; what gnark's
    // G1Affine.IsInSubGroup decides under the reference's point decoding (serialization.go:108-115).
    // ------------------------------------------------------------------------------------------
    static KZ_HD bool in_subgroup(const fe& px, const fe& py, const fe& beta) {
        // T = [|x|]P, Q = [|x|]T in ONE rolled loop (pass 0, pass 1) so that the running point stays in registers through the 126 doublings
        Jac Q;
        JacC base;
        jac_cache_affine(base, px, py);
        Q.x = px;
        Q.y = py;
        F::set_one(Q.z);
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll 1
#endif
        for (int pass = 0; pass < 2; pass++) {
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll 1
#endif
            for (int bit = 62; bit >= 0; bit--) {
                jdbl(Q, Q);
                if ((BLS_X_ABS >> bit) & 1) jadd(Q, Q, base);
            }
            if (pass == 0) jac_cache(base, Q);
        }
        // phi(P) = (beta x, y) must equal -Q = (X, -Y, Z):   beta x Z^2 == X  and  y Z^3 == -Y
        if (is_inf(Q)) return false;
        fe l, r, zz, zzz;
        F::sqr(zz, Q.z);
        F::mul(zzz, zz, Q.z);
        F::mul(l, beta, px);
        F::mul(l, l, zz);
        bool ok = F::eq_mod(l, Q.x);
        F::mul(l, py, zzz);
        F::template neg<3>(r, Q.y);
        return ok && F::eq_mod(l, r);
    }

    // The same test WITHOUT the y coordinate, from x and rhs = x^3 + 4 = y^2 alone -- so that it can run beside the square root that
    // recovers y (both are ~1 ms latency chains per lane; k_g1_validate gives them one wavefront each).  [k](x, y) = (X_k(x), y Y_k(x)):
    // every multiple of P is kept as (X, Y, z) meaning the Jacobian point (X, Y, Z = z y).  Then
    //   * the doubling never sees y:  X3, Y3 depend on X, Y only and Z3 = 2 Y Z = (2 Y z) y  -- jdbl() as it is;
    //   * the addition of two such points, written out with Z_i = z_i y and y^2 = rhs, yields (X3, y3' y, Z3) with Z3 free of y; scaling
    //     that Jacobian triple by lambda = y gives (rhs X3, rhs^2 y3', Z3 y), the same form again;
    //   * P itself is (x rhs, rhs^2, z = 1), and the final comparison beta x Z^2 == X, y Z^3 == -Y reads beta x z^2 rhs == X, z^3 rhs^2 == -Y.
    // Exact for every point of the curve (identity operands, P + P, P - P handled as in jadd); y != 0 on this curve (odd group order).
    struct JacYC {   // addition operand with zzr = z^2 rhs (= Z^2), zzzr = z zzr (Z^3 = zzzr y), zr = z rhs cached
        fe x, y, z, zzr, zzzr, zr;
    };
    static KZ_HD void jy_cache(JacYC& r, const Jac& p, const fe& rhs) {
        fe t;
        r.x = p.x;
        r.y = p.y;
        r.z = p.z;
        F::sqr(t, p.z);
        F::mul(r.zzr, t, rhs);
        F::mul(r.zzzr, r.zzr, p.z);
        F::mul(r.zr, p.z, rhs);
    }
    static KZ_HD void jadd_y(Jac& r, const Jac& a_, const JacYC& b_, const fe& rhs_, const fe& rhs2_) {
        Jac a = a_;
        JacYC b = b_;
        fe rhs = rhs_, rhs2 = rhs2_;
        opaque(a);
        F::opaque(b.x);
        F::opaque(b.y);
        F::opaque(b.z);
        F::opaque(b.zzr);
        F::opaque(b.zzzr);
        F::opaque(b.zr);
        F::opaque(rhs);
        F::opaque(rhs2);
        const bool a_inf = is_inf(a), b_inf = F::limbs_all_zero(b.z);
        fe A1, zA1, U1, U2, s1, s2, H, Rd, HH, HHH, Q, t, X3, Y3;
        Jac res, bj;
        bj.x = b.x;
        bj.y = b.y;
        bj.z = b.z;
        F::sqr(t, a.z);
        F::mul(A1, t, rhs);                 // Z1^2
        F::mul(zA1, A1, a.z);               // Z1^3 / y
        F::mul(U1, a.x, b.zzr);
        F::mul(U2, b.x, A1);
        F::mul(s1, a.y, b.zzzr);            // S1 / y
        F::mul(s2, b.y, zA1);               // S2 / y
        F::template sub<1>(H, U2, U1);
        F::template sub<1>(Rd, s2, s1);     // R / y
        if (!a_inf && !b_inf && F::template is_zero_mod<16>(H)) {
            if (F::template is_zero_mod<16>(Rd)) jdbl(res, a);
            else set_inf(res);
        } else {
            opaque(a);
            F::opaque(b.zr);
            F::opaque(H);
            F::opaque(Rd);
            F::opaque(U1);
            F::opaque(s1);
            F::opaque(rhs);
            F::opaque(rhs2);
            F::sqr(HH, H);
            F::mul(HHH, H, HH);
            F::mul(Q, U1, HH);
            F::sqr(t, Rd);
            F::mul(X3, t, rhs);                              // R^2 = (R/y)^2 rhs
            F::template sub_b_2c<2>(X3, X3, HHH, Q);          // - H^3 - 2 U1 H^2 + 4p   (< 5.01p)
            F::template sub<3>(t, Q, X3);
            F::template neg<1>(s1, s1);
            F::mul2(Y3, Rd, t, s1, HHH);                      // Y3 / y
            F::mul(t, a.z, b.zr);
            F::mul(res.z, t, H);                              // Z3 = z1 z2 rhs H  (no y): becomes the new z after the scaling by y
            F::mul(res.x, X3, rhs);
            F::mul(res.y, Y3, rhs2);
        }
        select(res, b_inf, a, res);
        select(res, a_inf, bj, res);
        r = res;
    }
    // the chain's start (P as (x rhs, rhs^2, z = 1) with its cached operand) and its verdict, shared with the lane-quad form of the
    // chain in verify_kernels.hip (k_g1_validate)
    static KZ_HD void yfree_start(Jac& Q, JacYC& base, fe& rhs2, const fe& px, const fe& rhs) {
        F::sqr(rhs2, rhs);
        F::mul(Q.x, px, rhs);
        Q.y = rhs2;
        F::set_one(Q.z);
        jy_cache(base, Q, rhs);
    }
    static KZ_HD bool yfree_verdict(const Jac& Q, const fe& px, const fe& rhs, const fe& rhs2, const fe& beta) {
        if (is_inf(Q)) return false;
        fe l, r, zz, t;
        F::sqr(zz, Q.z);
        F::mul(t, zz, rhs);                 // Z^2
        F::mul(l, beta, px);
        F::mul(l, l, t);
        bool ok = F::eq_mod(l, Q.x);
        F::mul(t, zz, Q.z);
        F::mul(l, t, rhs2);                 // y Z^3 = z^3 rhs^2
        F::template neg<3>(r, Q.y);
        return ok && F::eq_mod(l, r);
    }
    static KZ_HD bool in_subgroup_yfree(const fe& px, const fe& rhs, const fe& beta) {
        fe rhs2;
        Jac Q;
        JacYC base;
        yfree_start(Q, base, rhs2, px, rhs);
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll 1
#endif
        for (int pass = 0; pass < 2; pass++) {
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll 1
#endif
            for (int bit = 62; bit >= 0; bit--) {
                jdbl(Q, Q);
                if ((BLS_X_ABS >> bit) & 1) jadd_y(Q, Q, base, rhs, rhs2);
            }
            if (pass == 0) jy_cache(base, Q, rhs);
        }
        return yfree_verdict(Q, px, rhs, rhs2, beta);
    }

    // ------------------------------------------------------------------------------------------
    // The y-free test a third time, for a SMALL REGISTER FOOTPRINT (round 5: k_g1_validate held 474-497 registers, one wavefront per
    // SIMD and nothing beside it).  Same verdict as in_subgroup() for every finite point of the curve; what changes is how it is
    // carried:
    //   * the group law on y^2 = x^3 + b never reads b, so the chain may run on ANY curve isomorphic to E.  (x, y) -> (x l^2, y l^3)
    //     maps E to E_l: y^2 = x^3 + 4 l^6.  With l = y the image of P is P' = (x rhs, rhs^2), rhs = y^2 = x^3 + 4: an AFFINE point
    //     free of y, and every multiple of it computed with the ordinary Jacobian formulas is free of y -- the form above reached
    //     the same representation but paid for it inside its addition (17 products where a mixed addition takes 10);
    //   * pass 0 leaves T' = [|x|]P' = (X, Y, Z) on E_y.  One more isomorphism, l = Z, makes (X, Y) an AFFINE point of E_{yZ}:
    //     pass 1 is the same loop from another start, no Jacobian operand with cached powers (six field elements above), no inversion.
    //     Q' = [|x|](X, Y) = (X', Y', Z') on E_{yZ} is the Jacobian point (X', Y', Z' Z y) of E, so phi(P) = (beta x, y) == -Q reads
    //     X' == Z'^2 (beta x rhs Z^2) and -Y' == Z'^3 (rhs^2 Z^3): both constants are formed between the passes;
    //   * the addition is INCOMPLETE on purpose.  Its exceptional cases are [k]B = +-B or [k]B = identity for some 0 < k < 2^64 with
    //     B = P' or T'; either makes the order of P a divisor of a number below r (gcd(|x|, r) = 1: T has the order of P whenever
    //     that order is r), so P is not in the r-torsion -- and in each of them the formulas yield Z3 = Z1 H = 0 (mod p), which
    //     every later doubling (Z3 = 2 Y Z) and addition keeps.  The verdict of an exceptional chain is therefore "Z = 0 at the end"
    //     = not in the subgroup, with no branch, no select and no copy of the operands in the loop;
    //   * what the ten additions read -- the start point of the pass -- and the two constants of the verdict live in a STASH the
    //     caller provides (S::put / S::get: LDS on the device, 14 KB per wavefront; plain members on the host), so that the 126
    //     doublings run with the running point alone in registers.
    // Bounds (F = Fp): after jdbl X < 17.01p, Y < 5.01p, Z < 1.01p; after the addition X < 5.01p, Y, Z < 1.01p.
    // ------------------------------------------------------------------------------------------
    template <class S>
    static KZ_HD void jmadd_incomplete(Jac& r, const Jac& a_, S& st) {   // r = a + (stash 2, stash 3)
        Jac a = a_;
        opaque(a);
        fe Z1Z1, U2, H, t, S2, Rd, Z3, HH, HHH, V, X3, Y3, x2, y2;
        F::sqr(Z1Z1, a.z);
        st.get(x2, 2);
        F::mul(U2, x2, Z1Z1);
        F::template sub<5>(H, U2, a.x);               // U2 - X1 + 32p        (< 33.01p)
        F::mul(t, a.z, Z1Z1);
        st.get(y2, 3);
        F::mul(S2, y2, t);
        F::template sub<3>(Rd, S2, a.y);              // S2 - Y1 + 8p         (< 9.01p)
        F::mul(Z3, a.z, H);                           // 0 (mod p) in every exceptional case, see above
        F::sqr(HH, H);
        F::mul(HHH, H, HH);
        F::mul(V, a.x, HH);
        F::sqr(X3, Rd);
        F::template sub_b_2c<2>(X3, X3, HHH, V);      // R^2 - H^3 - 2V + 4p  (< 5.01p)
        F::template sub<3>(t, V, X3);                 // V - X3 + 8p          (< 9.01p)
        F::template neg<3>(S2, a.y);                  // 8p - Y1
        F::mul2(Y3, Rd, t, S2, HHH);                  // R (V - X3) - Y1 H^3  (< 1.01p)
        r.x = X3;
        r.y = Y3;
        r.z = Z3;
    }
    template <class S>
    static KZ_HD bool in_subgroup_yfree_lean(const fe& px, const fe& rhs, const fe& beta, S& st) {
        Jac Q;
        {
            fe t;
            F::mul(Q.x, px, rhs);                     // P' = (x y^2, y^4) on E_y
            F::sqr(Q.y, rhs);
            F::set_one(Q.z);
            st.put(2, Q.x);
            st.put(3, Q.y);
            F::mul(t, beta, Q.x);
            st.put(0, t);                             // beta x rhs
        }
        bool bad = false;
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll 1
#endif
        for (int pass = 0; pass < 2; pass++) {
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll 1
#endif
            for (int bit = 62; bit >= 0; bit--) {
                jdbl(Q, Q);
                if ((BLS_X_ABS >> bit) & 1) jmadd_incomplete(Q, Q, st);
            }
            if (pass == 0) {                          // T' = Q: the verdict's constants, then (X, Y) as the affine start of pass 1
                fe zz, zzz, c;
                bad = F::template is_zero_mod<16>(Q.z);
                F::sqr(zz, Q.z);
                F::mul(zzz, zz, Q.z);
                st.get(c, 0);
                F::mul(c, c, zz);
                st.put(0, c);                         // beta x rhs Z^2
                st.get(c, 3);
                F::mul(c, c, zzz);
                st.put(1, c);                         // rhs^2 Z^3
                st.put(2, Q.x);
                st.put(3, Q.y);
                F::set_one(Q.z);
            }
        }
        if (bad || F::template is_zero_mod<16>(Q.z)) return false;
        fe zz, l, c;
        F::sqr(zz, Q.z);
        st.get(c, 0);
        F::mul(l, zz, c);
        bool ok = F::eq_mod(l, Q.x);
        F::mul(l, zz, Q.z);
        st.get(c, 1);
        F::mul(l, l, c);
        F::template neg<3>(c, Q.y);
        return ok && F::eq_mod(l, c);
    }
    // ------------------------------------------------------------------------------------------
    // [k]P for a wavefront-uniform k given as the width-5 NAF strings of its GLV halves (fk20_toom.h FK20_TOOM_WNAF: digit codes d1 |
    // d2 << 8 per bit position, two positions per word; k P = k1 P + k2 (-phi(P)), phi(x, y) = (beta x, y)) -- the FK20 transform
    // stage's 686 fixed-scalar products, round 5 form:
    //   * the eight odd multiples P, 3P, ..., 15P by CO-Z arithmetic (Meloni): one doubling that leaves P with 2P's Z, then seven
    //     additions R = D + T_j of two points with the SAME Z (5M + 2S, no Z^2 / Z^3 of either operand) that also re-express D = 2P
    //     with R's Z.  Z_{j+1} = Z_j (X_D - X_Tj), so T_j is brought to the last Z by the suffix products of those seven factors;
    //   * all eight then share one Z: on the isomorphic curve E_Z (see in_subgroup_yfree_lean) they are AFFINE, the 43 additions of
    //     the chain are mixed additions (10 products instead of 16 with a cached-Z Jacobian operand, two coordinates per table entry
    //     instead of five: 896 B of scratch per lane instead of 2240), and the result's Z is multiplied by the common Z at the end;
    //   * the additions are incomplete.  P has order r or is the identity (every input of this stage is a sum of SRS multiples), the
    //     scalars are 686 constants, and gen_fk20_toom.py walks every chain modulo r and asserts that no addition meets acc = +-operand
    //     or acc = identity after the first one; the identity as INPUT makes every Z product exactly zero, which is the identity out.
    // Cost 90 + 131 x 7 + 43 x 10.5 ~ 1460 field multiplications against 1740.
    // ------------------------------------------------------------------------------------------
    static KZ_HD void jmadd_inc(Jac& r, const Jac& a_, const fe& x2, const fe& y2) {   // r = a + (x2, y2); as jmadd_incomplete, operands in registers
        Jac a = a_;
        opaque(a);
        fe Z1Z1, U2, H, t, S2, Rd, Z3, HH, HHH, V, X3, Y3;
        F::sqr(Z1Z1, a.z);
        F::mul(U2, x2, Z1Z1);
        F::template sub<5>(H, U2, a.x);
        F::mul(t, a.z, Z1Z1);
        F::mul(S2, y2, t);
        F::template sub<3>(Rd, S2, a.y);
        F::mul(Z3, a.z, H);
        F::sqr(HH, H);
        F::mul(HHH, H, HH);
        F::mul(V, a.x, HH);
        F::sqr(X3, Rd);
        F::template sub_b_2c<2>(X3, X3, HHH, V);
        F::template sub<3>(t, V, X3);
        F::template neg<3>(S2, a.y);
        F::mul2(Y3, Rd, t, S2, HHH);
        r.x = X3;
        r.y = Y3;
        r.z = Z3;
    }
    // the table of mul_wnaf_coz: (tx[j], ty[j]) = (2j + 1) P as affine points of E_zc (every entry with the Jacobian Z = zc)
    static KZ_HD void coz_odd_multiples(fe tx[8], fe ty[8], fe& zc, const XYZZ& p_) {
        fe lam[7];
        {
            XYZZ p = p_;
            opaque(p);
            fe X, Y, Z, A, B2, C4, t, D, E, dx, dy, Dx, Dy;
            F::mul(X, p.x, p.zz);                       // Jacobian (X ZZ, Y ZZZ, ZZ)
            F::mul(Y, p.y, p.zzz);
            Z = p.zz;
            // DBLU: (Dx, Dy) = 2P and P re-expressed with its Z (jdbl's values: 4 X Y^2 and 8 Y^4)
            F::sqr(A, X);
            F::sqr(B2, Y);
            F::dbl(B2, B2);
            F::sqr(C4, B2);
            F::add(t, X, B2);
            F::sqr(t, t);
            F::template sub_b_c<2>(D, t, A, C4);        // 4 X Y^2  (< 5.01p)
            F::triple(E, A);
            F::sqr(Dx, E);
            F::template sub_2c<4>(Dx, Dx, D);           // < 17.01p
            F::template sub<5>(t, D, Dx);
            F::mul(Dy, E, t);
            F::template sub_2c<2>(Dy, Dy, C4);          // < 5.01p
            F::dbl(t, Y);
            F::mul(zc, t, Z);                           // Z of both
            tx[0] = D;
            F::dbl(ty[0], C4);                          // 8 Y^4   (< 2.02p)
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll 1
#endif
            for (int j = 0; j < 7; j++) {               // ZADDU: T_{j+1} = D + T_j, D re-expressed with the new Z
                fe X2 = tx[j], Y2 = ty[j], Cq, W1, W2, A1, Dq, X3, Y3;
                F::opaque(Dx);
                F::opaque(Dy);
                F::opaque(X2);
                F::opaque(Y2);
                F::opaque(zc);
                F::template sub<3>(dx, Dx, X2);          // X2 < 5.01p <= 8p
                F::template sub<2>(dy, Dy, Y2);          // Y2 < 3.01p <= 4p
                F::sqr(Cq, dx);
                F::mul(W1, Dx, Cq);
                F::mul(W2, X2, Cq);
                F::sqr(Dq, dy);
                F::template sub<1>(t, W1, W2);
                F::mul(A1, Dy, t);
                F::template sub_b_c<2>(X3, Dq, W1, W2);  // < 5.01p
                F::template sub<3>(t, W1, X3);
                F::mul(Y3, dy, t);
                F::template sub<1>(Y3, Y3, A1);          // < 3.01p
                F::mul(zc, zc, dx);
                lam[j] = dx;
                tx[j + 1] = X3;
                ty[j + 1] = Y3;
                Dx = W1;
                Dy = A1;
            }
            // every T_j to the last Z: times (prod_{i >= j} lam_i)^2 and ^3
            fe f = lam[6], f2, f3;
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll 1
#endif
            for (int j = 6; j >= 0; j--) {
                F::opaque(f);
                if (j < 6) F::mul(f, f, lam[j]);
                F::sqr(f2, f);
                F::mul(f3, f2, f);
                F::mul(tx[j], tx[j], f2);
                F::mul(ty[j], ty[j], f3);
            }
        }
    }
    static KZ_HD void mul_wnaf_coz(XYZZ& r, const XYZZ& p, const uint32_t* row, const fe& beta) {
        fe tx[8], ty[8], zc;
        coz_odd_multiples(tx, ty, zc, p);
        Jac acc;
        F::set_zero(acc.x);
        F::set_zero(acc.y);
        F::set_zero(acc.z);
        bool started = false;
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll 1
#endif
        for (int i = 130; i >= 0; i--) {
            if (started) jdbl(acc, acc);
            const uint32_t code = (row[i >> 1] >> (16 * (i & 1))) & 0xffffu;
            if (code == 0) continue;
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll 1
#endif
            for (int half = 0; half < 2; half++) {
                const int d = (int)(int8_t)(half ? code >> 8 : code & 0xffu);
                if (d) {
                    const int e = ((d < 0 ? -d : d) - 1) >> 1;
                    fe qx = tx[e], qy = ty[e];
                    // second half: -phi(.) = (beta X, -Y); a negative digit flips the sign once more
                    if (half) F::mul(qx, qx, beta);
                    if ((d < 0) != (half != 0)) F::template neg<2>(qy, qy);      // 4p - y  (y < 3.01p: the last entry is not rescaled)
                    if (started) {
                        jmadd_inc(acc, acc, qx, qy);
                    } else {
                        acc.x = qx;
                        acc.y = qy;
                        F::set_one(acc.z);
                        started = true;
                    }
                }
            }
        }
        F::mul(acc.z, acc.z, zc);                       // back from E_Z: (X', Y', Z' Z)
        jac_to_xyzz(r, acc);
    }
    struct MemberStash {   // the stash as plain members: host tests, and device callers that can afford the registers
        fe s[4];
        KZ_HD void put(int i, const fe& v) { s[i] = v; }
        KZ_HD void get(fe& v, int i) const { v = s[i]; }
    };
};

// ---- device / radix-2^29 instantiation, under the names the kernels use
using G1D = G1T<Fp>;
using G1Aff = G1D::Aff;
using G1X = G1D::XYZZ;
KZ_HD bool g1x_is_inf(const G1X& p) { return G1D::is_inf(p); }
KZ_HD void g1x_set_inf(G1X& p) { G1D::set_inf(p); }
KZ_HD void g1x_from_affine(G1X& r, const fp_t& x, const fp_t& y) { G1D::from_affine(r, x, y); }
KZ_HD void g1x_dbl_affine(G1X& r, const fp_t& x, const fp_t& y) { G1D::dbl_affine(r, x, y); }
KZ_HD void g1x_dbl(G1X& r, const G1X& p) { G1D::dbl(r, p); }
KZ_HD void g1x_madd(G1X& acc, const fp_t& x2, const fp_t& y2) { G1D::madd(acc, x2, y2); }
KZ_HD void g1x_add(G1X& r, const G1X& a, const G1X& b) { G1D::add(r, a, b); }
KZ_HD void g1x_neg(G1X& r, const G1X& a) { G1D::neg(r, a); }
KZ_HD void g1x_to_affine(G1Aff& r, const G1X& p) { G1D::to_affine(r, p); }
using G1J = G1D::Jac;
using G1JC = G1D::JacC;

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
