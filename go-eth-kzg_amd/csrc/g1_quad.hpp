// g1_quad.hpp -- the Jacobian group law of g1.hpp on a QUAD of lanes per point (device only).
//
// Where a call brings too few points to fill the chip, what the caller waits for is the length of ONE dependent chain of field
// multiplications: the 126 doublings of the r-torsion test behind every decoded point (reference serialization.go:108-115 ->
// gnark IsInSubGroup), the 130 doublings of each fixed-scalar product of the FK20 network for a single blob (fk20.go:76-124).
// A doubling is 7 multiplications but only 3 deep, an addition 13 but only 5 deep.  So four adjacent lanes take one point: all four
// hold the same coordinates, every level each lane forms ONE product -- operands picked by its role with v_cndmask -- and the
// results travel with quad_perm DPP moves (full rate, no LDS).  Same values as G1D::jdbl / G1D::jadd (up to the bound of Y3 in
// the addition, noted there); the exceptional cases are decided on replicated values, so a quad never diverges.
// Requires all four lanes of a quad to be active together (the kernels run FULL_WAVES).  A pair-of-lanes form follows the quad one.
#pragma once
#include "g1.hpp"

namespace kzg {

template <int K>
__device__ __forceinline__ void quad_bcast(fp_t& r, const fp_t& a) {
#pragma unroll
    for (int q = 0; q < 14; q++) r.l[q] = (uint32_t)__builtin_amdgcn_update_dpp((int)a.l[q], (int)a.l[q], K * 0x55, 0xf, 0xf, false);
}
struct QuadRole {
    bool r0, r1, r2, r3;
    __device__ __forceinline__ explicit QuadRole(int lane) : r0((lane & 3) == 0), r1((lane & 3) == 1), r2((lane & 3) == 2), r3((lane & 3) == 3) {}
};
// r = role 0 ? v0 : role 1 ? v1 : role 2 ? v2 : v3
__device__ __forceinline__ void quad_pick(fp_t& r, const QuadRole& q, const fp_t& v0, const fp_t& v1, const fp_t& v2, const fp_t& v3) {
    Fp::select(r, q.r2, v2, v3);
    Fp::select(r, q.r1, v1, r);
    Fp::select(r, q.r0, v0, r);
}

// p = 2p: G1D::jdbl (dbl-2009-l with 2Y^2 formed up front) in three levels {X^2, Y^2, 2Y Z}, {(3X^2)^2, (2Y^2)^2, (X + 2Y^2)^2},
// {3X^2 (D - X3)}; the identity (Z = all-zero limbs) stays the identity
__device__ __forceinline__ void jdbl_quad(G1J& p, const QuadRole& q) {
    G1D::opaque(p);
    fp_t U, a, b, M1, M2, t, B2, T, E, A, EE, C4, D, X3, Y3;
    Fp::dbl(U, p.y);
    Fp::select(a, q.r1, p.y, p.x);
    Fp::select(a, q.r2, U, a);
    Fp::select(b, q.r1, p.y, p.x);
    Fp::select(b, q.r2, p.z, b);
    Fp::mul(M1, a, b);                        // lanes 0, 3: X^2      lane 1: Y^2      lane 2: 2Y Z
    quad_bcast<1>(t, M1);
    Fp::dbl(B2, t);                           // 2 Y^2            (< 2.01p)
    Fp::add(T, p.x, B2);                      //                   (< 19.1p)
    Fp::triple(E, M1);                        // lanes 0, 3: 3 X^2
    Fp::select(a, q.r1, B2, E);
    Fp::select(a, q.r2, T, a);
    Fp::sqr(M2, a);                           // lanes 0, 3: E^2      lane 1: 4 Y^4    lane 2: (X + 2Y^2)^2
    quad_bcast<0>(A, M1);
    quad_bcast<0>(EE, M2);
    quad_bcast<1>(C4, M2);
    quad_bcast<2>(T, M2);
    quad_bcast<2>(p.z, M1);
    Fp::template sub_b_c<2>(D, T, A, C4);     // D = 4 X Y^2 + 4p  (< 5.01p)
    Fp::triple(E, A);
    Fp::template sub_2c<4>(X3, EE, D);        // E^2 - 2D + 16p    (< 17.01p)
    Fp::template sub<5>(t, D, X3);            // D - X3 + 32p      (< 37.1p)
    Fp::mul(Y3, E, t);
    Fp::template sub_2c<2>(p.y, Y3, C4);      // E (D - X3) - 8 Y^4 + 4p   (< 5.01p)
    p.x = X3;
}

// a = a + b: G1D::jadd (add-2007-bl, b's Z^2 and Z^3 cached) in five levels
//   {Z1^2, X1 Z2^2, Y1 Z2^3, Z1 Z2}, {Z1^3, X2 Z1^2}, {Y2 Z1^3, H^2, Z1 Z2 H}, {H^3, U1 H^2, R^2}, {R (U1 H^2 - X3), -S1 H^3};
// Y3 is the sum of the last two products (< 2.02p where G1D::jadd's fused product gives < 1.01p; the Jacobian bound is Y < 5.01p).
__device__ __forceinline__ void jadd_quad(G1J& a, const G1JC& b_, const QuadRole& q) {
    G1JC b = b_;
    G1D::opaque(a);
    Fp::opaque(b.x);
    Fp::opaque(b.y);
    Fp::opaque(b.z);
    Fp::opaque(b.zz);
    Fp::opaque(b.zzz);
    const bool a_inf = G1D::is_inf(a), b_inf = Fp::limbs_all_zero(b.z);
    fp_t oa, ob, M1, M2, M3, M4, M5, Z1Z1, U1, U2, S1, S2, H, Rd, HH, HHH, Q, RR, t, X3;
    G1J res, bj;
    bj.x = b.x;
    bj.y = b.y;
    bj.z = b.z;
    quad_pick(oa, q, a.z, a.x, a.y, a.z);
    quad_pick(ob, q, a.z, b.zz, b.zzz, b.z);
    Fp::mul(M1, oa, ob);                      // Z1^2 | U1 = X1 Z2^2 | S1 = Y1 Z2^3 | Z1 Z2
    quad_bcast<0>(Z1Z1, M1);
    Fp::select(oa, q.r0, a.z, b.x);
    Fp::mul(M2, oa, Z1Z1);                    // Z1^3 | U2 = X2 Z1^2 (lanes 1..3)
    quad_bcast<1>(U2, M2);
    quad_bcast<1>(U1, M1);
    Fp::template sub<1>(H, U2, U1);
    Fp::select(oa, q.r0, b.y, H);
    Fp::select(oa, q.r3, M1, oa);
    Fp::select(ob, q.r0, M2, H);
    Fp::mul(M3, oa, ob);                      // S2 = Y2 Z1^3 | H^2 | H^2 | Z1 Z2 H
    quad_bcast<0>(S2, M3);
    quad_bcast<2>(S1, M1);
    Fp::template sub<1>(Rd, S2, S1);
    if (!a_inf && !b_inf && Fp::template is_zero_mod<16>(H)) {
        if (Fp::template is_zero_mod<16>(Rd)) {
            res = a;
            jdbl_quad(res, q);
        } else {
            G1D::set_inf(res);
        }
    } else {
        Fp::opaque(H);
        Fp::opaque(Rd);
        Fp::opaque(U1);
        Fp::opaque(S1);
        quad_bcast<1>(HH, M3);
        Fp::select(oa, q.r0, H, U1);
        Fp::select(oa, q.r2, Rd, oa);
        Fp::select(ob, q.r2, Rd, HH);
        Fp::mul(M4, oa, ob);                  // H^3 | Q = U1 H^2 | R^2 | (Q again)
        quad_bcast<0>(HHH, M4);
        quad_bcast<1>(Q, M4);
        quad_bcast<2>(RR, M4);
        Fp::template sub_b_2c<2>(X3, RR, HHH, Q);   // R^2 - H^3 - 2 U1 H^2 + 4p   (< 5.01p)
        Fp::template sub<3>(t, Q, X3);
        Fp::template neg<1>(S1, S1);
        Fp::select(oa, q.r0, Rd, S1);
        Fp::select(ob, q.r0, t, HHH);
        Fp::mul(M5, oa, ob);                  // R (U1 H^2 - X3) | -S1 H^3 (lanes 1..3)
        quad_bcast<0>(t, M5);
        quad_bcast<1>(S2, M5);
        Fp::add(res.y, t, S2);
        quad_bcast<3>(res.z, M3);
        res.x = X3;
    }
    G1D::select(res, b_inf, a, res);
    G1D::select(res, a_inf, bj, res);
    a = res;
}

// ---- extended-Jacobian (XYZZ) accumulators on a quad: the addition networks around the FK20 products (fk20_kernels.hip toom_run) are
// chains of ~10 additions / doublings of whole points.  add-2008-s is 14 products in four levels
//   {X1 ZZ2, X2 ZZ1, Y1 ZZZ2, Y2 ZZZ1}, {P^2, ZZ1 ZZ2, ZZZ1 ZZZ2, R^2}, {P^3, ZZ1 ZZ2 P^2, U1 P^2}, {R (Q - X3), -S1 P^3, ZZZ1 ZZZ2 P^3}
// and dbl-2008-s-1 nine in three: {(2Y)^2, X^2}, {U V, X V, (3X^2)^2, V ZZ}, {M (S - X3), -W Y, W ZZZ}.  Same values as G1D::add / G1D::dbl
// except that Y3 is the sum of two reduced products (< 2.02p instead of < 1.01p; the XYZZ bound is Y < 4p).
__device__ __forceinline__ void xdbl_quad(G1X& p, const QuadRole& q) {
    G1D::opaque(p);
    fp_t U, a, b, M1, M2, M3, V, T, M, W, S, MM, X3, t, nY;
    Fp::dbl(U, p.y);
    Fp::select(a, q.r0, U, p.x);
    Fp::sqr(M1, a);                           // lane 0: V = (2Y)^2       lanes 1..3: X^2
    quad_bcast<0>(V, M1);
    quad_bcast<1>(T, M1);
    Fp::triple(M, T);                         // 3 X^2
    quad_pick(a, q, U, p.x, M, V);
    quad_pick(b, q, V, V, M, p.zz);
    Fp::mul(M2, a, b);                        // W = U V | S = X V | M^2 | ZZ3 = V ZZ   (ZZ = 0 stays all-zero limbs)
    quad_bcast<0>(W, M2);
    quad_bcast<1>(S, M2);
    quad_bcast<2>(MM, M2);
    Fp::template sub_2c<2>(X3, MM, S);        // M^2 - 2S + 4p        (< 5.01p)
    Fp::template sub<3>(t, S, X3);            // S - X3 + 8p
    Fp::template neg<2>(nY, p.y);             // 4p - Y
    quad_pick(a, q, M, W, W, W);
    quad_pick(b, q, t, nY, p.zzz, p.zzz);
    Fp::mul(M3, a, b);                        // M (S - X3) | -W Y | ZZZ3 = W ZZZ | (again)
    quad_bcast<0>(t, M3);
    quad_bcast<1>(S, M3);
    Fp::add(p.y, t, S);
    quad_bcast<3>(p.zz, M2);
    quad_bcast<2>(p.zzz, M3);
    p.x = X3;
}
__device__ __forceinline__ void xadd_quad(G1X& a, const G1X& b_, const QuadRole& q) {
    G1X b = b_;
    G1D::opaque(a);
    G1D::opaque(b);
    const bool a_inf = G1D::is_inf(a), b_inf = G1D::is_inf(b);
    fp_t oa, ob, M1, M2, M3, M4, U1, U2, S1, S2, Pd, Rd, PP, PPP, Q, RR, X3, t;
    G1X res;
    quad_pick(oa, q, a.x, b.x, a.y, b.y);
    quad_pick(ob, q, b.zz, a.zz, b.zzz, a.zzz);
    Fp::mul(M1, oa, ob);                      // U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1
    quad_bcast<0>(U1, M1);
    quad_bcast<1>(U2, M1);
    quad_bcast<2>(S1, M1);
    quad_bcast<3>(S2, M1);
    Fp::template sub<1>(Pd, U2, U1);
    Fp::template sub<1>(Rd, S2, S1);
    if (!a_inf && !b_inf && Fp::template is_zero_mod<16>(Pd)) {
        if (Fp::template is_zero_mod<16>(Rd)) {
            res = a;
            xdbl_quad(res, q);
        } else {
            G1D::set_inf(res);
        }
    } else {
        Fp::opaque(Pd);
        Fp::opaque(Rd);
        Fp::opaque(U1);
        Fp::opaque(S1);
        quad_pick(oa, q, Pd, a.zz, a.zzz, Rd);
        quad_pick(ob, q, Pd, b.zz, b.zzz, Rd);
        Fp::mul(M2, oa, ob);                  // PP = P^2 | ZZ1 ZZ2 | ZZZ1 ZZZ2 | R^2
        quad_bcast<0>(PP, M2);
        quad_pick(oa, q, Pd, M2, U1, U1);
        Fp::mul(M3, oa, PP);                  // PPP = P^3 | ZZ3 = ZZ1 ZZ2 PP | Q = U1 PP | (Q again)
        quad_bcast<0>(PPP, M3);
        quad_bcast<2>(Q, M3);
        quad_bcast<3>(RR, M2);
        Fp::template sub_b_2c<2>(X3, RR, PPP, Q);   // R^2 - PPP - 2Q + 4p   (< 5.01p)
        Fp::template sub<3>(t, Q, X3);
        Fp::template neg<1>(S1, S1);          // 2p - S1
        quad_pick(oa, q, Rd, S1, M2, M2);
        quad_pick(ob, q, t, PPP, PPP, PPP);
        Fp::mul(M4, oa, ob);                  // R (Q - X3) | -S1 PPP | ZZZ3 = ZZZ1 ZZZ2 PPP | (garbage on lane 3: R^2 PPP, unused)
        quad_bcast<0>(t, M4);
        quad_bcast<1>(S2, M4);
        Fp::add(res.y, t, S2);
        quad_bcast<1>(res.zz, M3);
        quad_bcast<2>(res.zzz, M4);
        res.x = X3;
    }
    G1D::select(res, b_inf, a, res);
    G1D::select(res, a_inf, b, res);
    a = res;
}

// acc += (x2, y2), affine and not the identity: G1D::madd (madd-2008-s, 10 products) in four levels
//   {X2 ZZ1, Y2 ZZZ1}, {P^2, R^2}, {P^3, X1 P^2, ZZ1 P^2}, {R (Q - X3), -Y1 P^3, ZZZ1 P^3}
__device__ __forceinline__ void xmadd_quad(G1X& acc, const fp_t& x2_, const fp_t& y2_, const QuadRole& q) {
    fp_t x2 = x2_, y2 = y2_;
    if (G1D::is_inf(acc)) { G1D::from_affine(acc, x2, y2); return; }
    G1D::opaque(acc);
    Fp::opaque(x2);
    Fp::opaque(y2);
    fp_t oa, ob, M1, M2, M3, M4, U2, S2, Pd, Rd, PP, RR, PPP, Q, X3, t, nY;
    Fp::select(oa, q.r0, x2, y2);
    Fp::select(ob, q.r0, acc.zz, acc.zzz);
    Fp::mul(M1, oa, ob);                      // lane 0: U2 = X2 ZZ1      lanes 1..3: S2 = Y2 ZZZ1
    quad_bcast<0>(U2, M1);
    quad_bcast<1>(S2, M1);
    Fp::template sub<3>(Pd, U2, acc.x);       // U2 - X1 + 8p
    Fp::template sub<2>(Rd, S2, acc.y);       // S2 - Y1 + 4p
    if (Fp::template is_zero_mod<16>(Pd)) {
        if (Fp::template is_zero_mod<16>(Rd)) G1D::dbl_affine(acc, x2, y2);
        else G1D::set_inf(acc);
        return;
    }
    G1D::opaque(acc);
    Fp::opaque(Pd);
    Fp::opaque(Rd);
    Fp::select(oa, q.r0, Pd, Rd);
    Fp::sqr(M2, oa);                          // lane 0: PP = P^2         lanes 1..3: R^2
    quad_bcast<0>(PP, M2);
    quad_bcast<1>(RR, M2);
    quad_pick(oa, q, Pd, acc.x, acc.zz, acc.zz);
    Fp::mul(M3, oa, PP);                      // PPP | Q = X1 PP | ZZ3 = ZZ1 PP | (again)
    quad_bcast<0>(PPP, M3);
    quad_bcast<1>(Q, M3);
    Fp::template sub_b_2c<2>(X3, RR, PPP, Q); // R^2 - PPP - 2Q + 4p   (< 5.01p)
    Fp::template sub<3>(t, Q, X3);            // Q - X3 + 8p
    Fp::template neg<2>(nY, acc.y);           // 4p - Y1
    quad_pick(oa, q, Rd, nY, acc.zzz, acc.zzz);
    quad_pick(ob, q, t, PPP, PPP, PPP);
    Fp::mul(M4, oa, ob);                      // R (Q - X3) | -Y1 PPP | ZZZ3 = ZZZ1 PPP | (again)
    quad_bcast<0>(t, M4);
    quad_bcast<1>(S2, M4);
    Fp::add(acc.y, t, S2);
    quad_bcast<2>(acc.zz, M3);
    quad_bcast<2>(acc.zzz, M4);
    acc.x = X3;
}

// ---- the same law on a PAIR of lanes (lanes 2i and 2i + 1): for calls with too many points for a quad each but room for two lanes --
// 17 to 32 blobs in the 686 products of the FK20 network.  The seven products of a doubling go in four rounds
//   {X^2 | Y^2}, {2Y Z | (2Y^2)^2}, {(3X^2)^2 | (X + 2Y^2)^2}, {3X^2 (D - X3)}
// and the thirteen of an addition in seven
//   {Z1^2 | X1 Z2^2}, {Y1 Z2^3 | Z1 Z2}, {Z1^3 | X2 Z1^2}, {Y2 Z1^3 | H^2}, {H^3 | Z1 Z2 H}, {U1 H^2 | R^2}, {R (U1 H^2 - X3) | -S1 H^3}
// (lane A | lane B): ~2,200 and ~4,100 instructions against ~3,100 and ~6,400 on one lane.  Same values as jdbl_quad / jadd_quad.
template <bool FROM_B>
__device__ __forceinline__ void pair_bcast(fp_t& r, const fp_t& a) {
#pragma unroll
    for (int q = 0; q < 14; q++) r.l[q] = (uint32_t)__builtin_amdgcn_update_dpp((int)a.l[q], (int)a.l[q], FROM_B ? 0xF5 : 0xA0, 0xf, 0xf, false);   // quad_perm [1,1,3,3] / [0,0,2,2]
}
__device__ __forceinline__ void jdbl_pair(G1J& p, const bool isB) {
    G1D::opaque(p);
    fp_t a, b, M1, M2, M3, U, B2, T, E, A, EE, C4, TT, D, X3, Y3, t;
    Fp::select(a, isB, p.y, p.x);
    Fp::sqr(M1, a);                           // A: X^2           B: Y^2
    pair_bcast<true>(t, M1);
    Fp::dbl(B2, t);                           // 2 Y^2            (< 2.01p)
    Fp::dbl(U, p.y);
    Fp::select(a, isB, B2, U);
    Fp::select(b, isB, B2, p.z);
    Fp::mul(M2, a, b);                        // A: 2Y Z          B: 4 Y^4
    Fp::add(T, p.x, B2);                      //                   (< 19.1p)
    Fp::triple(E, M1);                        // A: 3 X^2
    Fp::select(a, isB, T, E);
    Fp::sqr(M3, a);                           // A: E^2           B: (X + 2Y^2)^2
    pair_bcast<false>(A, M1);
    pair_bcast<false>(EE, M3);
    pair_bcast<true>(C4, M2);
    pair_bcast<true>(TT, M3);
    pair_bcast<false>(p.z, M2);
    Fp::template sub_b_c<2>(D, TT, A, C4);    // D = 4 X Y^2 + 4p  (< 5.01p)
    Fp::triple(E, A);
    Fp::template sub_2c<4>(X3, EE, D);        // E^2 - 2D + 16p    (< 17.01p)
    Fp::template sub<5>(t, D, X3);            // D - X3 + 32p      (< 37.1p)
    Fp::mul(Y3, E, t);
    Fp::template sub_2c<2>(p.y, Y3, C4);      // E (D - X3) - 8 Y^4 + 4p   (< 5.01p)
    p.x = X3;
}
__device__ __forceinline__ void jadd_pair(G1J& a, const G1JC& b_, const bool isB) {
    G1JC b = b_;
    G1D::opaque(a);
    Fp::opaque(b.x);
    Fp::opaque(b.y);
    Fp::opaque(b.z);
    Fp::opaque(b.zz);
    Fp::opaque(b.zzz);
    const bool a_inf = G1D::is_inf(a), b_inf = Fp::limbs_all_zero(b.z);
    fp_t oa, ob, M1, M2, M3, M4, M5, M6, M7, Z1Z1, U1, U2, S1, S2, H, Rd, HH, HHH, Q, RR, t, X3;
    G1J res, bj;
    bj.x = b.x;
    bj.y = b.y;
    bj.z = b.z;
    Fp::select(oa, isB, a.x, a.z);
    Fp::select(ob, isB, b.zz, a.z);
    Fp::mul(M1, oa, ob);                      // A: Z1^2          B: U1 = X1 Z2^2
    Fp::select(oa, isB, a.z, a.y);
    Fp::select(ob, isB, b.z, b.zzz);
    Fp::mul(M2, oa, ob);                      // A: S1 = Y1 Z2^3  B: Z1 Z2
    pair_bcast<false>(Z1Z1, M1);
    Fp::select(oa, isB, b.x, a.z);
    Fp::mul(M3, oa, Z1Z1);                    // A: Z1^3          B: U2 = X2 Z1^2
    pair_bcast<true>(U2, M3);
    pair_bcast<true>(U1, M1);
    Fp::template sub<1>(H, U2, U1);
    Fp::select(oa, isB, H, b.y);
    Fp::select(ob, isB, H, M3);
    Fp::mul(M4, oa, ob);                      // A: S2 = Y2 Z1^3  B: H^2
    pair_bcast<false>(S2, M4);
    pair_bcast<false>(S1, M2);
    Fp::template sub<1>(Rd, S2, S1);
    if (!a_inf && !b_inf && Fp::template is_zero_mod<16>(H)) {
        if (Fp::template is_zero_mod<16>(Rd)) {
            res = a;
            jdbl_pair(res, isB);
        } else {
            G1D::set_inf(res);
        }
    } else {
        Fp::opaque(H);
        Fp::opaque(Rd);
        Fp::opaque(U1);
        Fp::opaque(S1);
        pair_bcast<true>(HH, M4);
        Fp::select(oa, isB, M2, H);
        Fp::select(ob, isB, H, HH);
        Fp::mul(M5, oa, ob);                  // A: H^3           B: Z1 Z2 H
        Fp::select(oa, isB, Rd, U1);
        Fp::select(ob, isB, Rd, HH);
        Fp::mul(M6, oa, ob);                  // A: Q = U1 H^2    B: R^2
        pair_bcast<false>(HHH, M5);
        pair_bcast<false>(Q, M6);
        pair_bcast<true>(RR, M6);
        Fp::template sub_b_2c<2>(X3, RR, HHH, Q);   // R^2 - H^3 - 2 U1 H^2 + 4p   (< 5.01p)
        Fp::template sub<3>(t, Q, X3);
        Fp::template neg<1>(S1, S1);
        Fp::select(oa, isB, S1, Rd);
        Fp::select(ob, isB, HHH, t);
        Fp::mul(M7, oa, ob);                  // A: R (U1 H^2 - X3)   B: -S1 H^3
        pair_bcast<false>(t, M7);
        pair_bcast<true>(S2, M7);
        Fp::add(res.y, t, S2);
        pair_bcast<true>(res.z, M5);
        res.x = X3;
    }
    G1D::select(res, b_inf, a, res);
    G1D::select(res, a_inf, bj, res);
    a = res;
}

}  // namespace kzg
