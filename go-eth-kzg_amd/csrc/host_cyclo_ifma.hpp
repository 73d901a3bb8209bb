// host_cyclo_ifma.hpp -- the squaring chains of the final exponentiation on AVX-512 IFMA (product host code).
//
// Every verifier call ends with one final exponentiation (host_pairing.hpp; reference internal/kzg/kzg_verify.go:190-199,
// internal/kzg_multi/kzg_verify.go:94-104 through gnark's PairingCheck), and its hard part is five exponentiations by the curve
// parameter |x| = 0xd201000000010000: 5 x 63 cyclotomic squarings, a chain no second thread can share.  With the scalar tower (ADX
// Montgomery multiplication, 26 ns) a Granger-Scott squaring is 18 field multiplications and ~120 additions = 1.1 us, and the 315 of them
// are half of the 0.7 ms the host adds to EVERY verifier call.  The EPYC 9005 hosts of MI355X nodes execute 512-bit integer FMAs at full
// width, so here the squaring runs EIGHT field multiplications per instruction stream:
//   * a field element is 8 limbs of 52 bits (R = 2^416), and a register holds limb k of eight different elements (one per 64-bit lane);
//     mul8 is the operand-scanning Montgomery product on vpmadd52luq / vpmadd52huq: 8 x (16 + 1 + 16) IFMAs for eight products;
//   * values are kept LOOSE: sums are formed limb by limb without carries (limbs may exceed 52 bits until `norm`), subtraction adds a
//     fixed multiple of p whose limbs dominate the subtrahend's, and nothing is reduced mod p except by mul8 itself (its output is
//     below 2p for inputs up to 2^12 p: R leaves 35 bits above p);
//   * the state is three registers-of-limbs V0, V1, V2 holding the Fp4 pairs (r0, r1), (r2, r3), (r4, r5) of the Granger-Scott formula in
//     lanes [x0, x1, y0, y1]; one squaring is three mul8 for the three Fp4 squarings (schoolbook Fp2 products: exactly 8 per Fp4, the
//     operand vectors built with lane permutes), linear combinations with lane permutes, and two mul8 by the Montgomery one that bring
//     the 12 new coefficients back below 2p (the term 2 r_k of z_k = 3 t_k +- 2 r_k would otherwise double the bound every squaring).
// a^|x| needs no multiplication inside the chain: |x| = 2^63 + 2^62 + 2^60 + 2^57 + 2^48 + 2^16, so the chain of 63 squarings hands out
// six snapshots a^(2^k) and the scalar tower multiplies them (5 products, as in square-and-multiply).
// Checked against the scalar cyclotomic_sqr / exp_by_x_abs on random cyclotomic elements by tests/test_host_pairing.py, and through every
// verifier vector.  Without AVX-512 IFMA (or with MKZG_NO_IFMA in the environment) the scalar chain of host_pairing.hpp runs.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace kzg {
namespace ifma {

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define MKZG_IFMA_TARGET __attribute__((target("avx512f,avx512ifma")))

// p, -1/p mod 2^52, 2^416 mod p, and the conversion constants between Montgomery forms (host_field.hpp keeps R = 2^384):
// in:  mul8(v, 2^448) = v 2^32,  out: mul8(v, 2^384) = v 2^-32.  OFF8 / OFF128: 8p and 128p with "spread" limbs (limbs 0..6 at least
// 2^52 - 1), so that a limb-wise subtraction of a normalised value below 4p / 64p never underflows a limb.
static const uint64_t P52[8] = {0xeffffffffaaabull, 0xfeb153ffffb9full, 0x6b0f6241eabffull, 0x12bf6730d2a0full, 0x764774b84f385ull, 0x1ba7b6434bacdull, 0x1ea397fe69a4bull, 0x1a011ull};
static const uint64_t N0 = 0x3fffcfffcfffdull;
static const uint64_t ONE52[8] = {0x6480ea8e9b9afull, 0x65766c8fe444full, 0x8b540fea96f7dull, 0x3b2ee82efd422ull, 0xa6723e5f0ade5ull, 0xff6eb6fdd4230ull, 0xe06ef23c24a25ull, 0x14c8eull};
static const uint64_t C448[8] = {0x7fde37dba9366ull, 0x4e27525bc342bull, 0x1f5b1e9778489ull, 0xb872b2b91b9dcull, 0xb206f497dfcafull, 0x4137cc89a9b0bull, 0xd9d20d7e39959ull, 0x411cull};
static const uint64_t C384[8] = {0x900000002fffdull, 0xbc40c0002760ull, 0x3c758baebf400ull, 0x57455f4898575ull, 0xd77ce58537052ull, 0x71a97a256ec6ull, 0xec3fa80e4935cull, 0x15f65ull};
static const uint64_t OFF8[8] = {0x17fffffffd5558ull, 0x1f58a9ffffdcfeull, 0x1587b120f55ffeull, 0x195fb39869507aull, 0x1b23ba5c279c27ull, 0x1dd3db21a5d66aull, 0x1f51cbff34d257ull, 0xd0087ull};
static const uint64_t OFF128[8] = {0x1fffffffd55580ull, 0x158a9ffffdcff6ull, 0x187b120f55fffeull, 0x15fb39869507b4ull, 0x123ba5c279c288ull, 0x1d3db21a5d66baull, 0x151cbff34d258cull, 0xd0088eull};

struct V8 { __m512i l[8]; };               // limb k of eight field elements

static inline bool available() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && !getenv("MKZG_NO_IFMA");
    return ok;
}

MKZG_IFMA_TARGET static inline __m512i bc(uint64_t v) { return _mm512_set1_epi64((long long)v); }
MKZG_IFMA_TARGET static inline void splat(V8& r, const uint64_t c[8]) { for (int k = 0; k < 8; k++) r.l[k] = bc(c[k]); }
// carries: limbs below 2^52 afterwards (the top limb takes what is left: values stay far below 2^416)
MKZG_IFMA_TARGET static inline void norm(V8& a) {
    const __m512i M = bc((1ull << 52) - 1);
    for (int k = 0; k < 7; k++) {
        const __m512i c = _mm512_srli_epi64(a.l[k], 52);
        a.l[k] = _mm512_and_si512(a.l[k], M);
        a.l[k + 1] = _mm512_add_epi64(a.l[k + 1], c);
    }
}
MKZG_IFMA_TARGET static inline void add(V8& r, const V8& a, const V8& b) { for (int k = 0; k < 8; k++) r.l[k] = _mm512_add_epi64(a.l[k], b.l[k]); }
MKZG_IFMA_TARGET static inline void dbl(V8& r, const V8& a) { for (int k = 0; k < 8; k++) r.l[k] = _mm512_slli_epi64(a.l[k], 1); }
// off - a, a normalised and below half of off's value
MKZG_IFMA_TARGET static inline void neg(V8& r, const V8& a, const uint64_t off[8]) { for (int k = 0; k < 8; k++) r.l[k] = _mm512_sub_epi64(bc(off[k]), a.l[k]); }
// lane j of r = lane idx[j] of a (idx < 8) or of b (idx >= 8); lanes outside keep are zero; then shifted left by sh[j]
MKZG_IFMA_TARGET static inline void pick(V8& r, const V8& a, const V8& b, __m512i idx, __mmask8 keep, __m512i sh) {
    for (int k = 0; k < 8; k++) r.l[k] = _mm512_sllv_epi64(_mm512_maskz_permutex2var_epi64(keep, a.l[k], idx, b.l[k]), sh);
}
MKZG_IFMA_TARGET static inline void acc_pick(V8& r, const V8& a, const V8& b, __m512i idx, __mmask8 keep, __m512i sh) {
    for (int k = 0; k < 8; k++) r.l[k] = _mm512_add_epi64(r.l[k], _mm512_sllv_epi64(_mm512_maskz_permutex2var_epi64(keep, a.l[k], idx, b.l[k]), sh));
}
// the same without a shift
MKZG_IFMA_TARGET static inline void pick0(V8& r, const V8& a, const V8& b, __m512i idx, __mmask8 keep) {
    for (int k = 0; k < 8; k++) r.l[k] = _mm512_maskz_permutex2var_epi64(keep, a.l[k], idx, b.l[k]);
}
MKZG_IFMA_TARGET static inline void acc_pick0(V8& r, const V8& a, const V8& b, __m512i idx, __mmask8 keep) {
    for (int k = 0; k < 8; k++) r.l[k] = _mm512_add_epi64(r.l[k], _mm512_maskz_permutex2var_epi64(keep, a.l[k], idx, b.l[k]));
}
// r = a b / 2^416 mod p lane by lane; a, b normalised and below 2^12 p; r normalised and below 2p
MKZG_IFMA_TARGET static inline void mul8(V8& r, const V8& a, const V8& b) {
    const __m512i n0 = bc(N0);
    __m512i P[8], acc[9];
    for (int j = 0; j < 8; j++) P[j] = bc(P52[j]);
    for (int j = 0; j < 9; j++) acc[j] = _mm512_setzero_si512();
#pragma GCC unroll 8
    for (int i = 0; i < 8; i++) {
        const __m512i bi = b.l[i];
#pragma GCC unroll 8
        for (int j = 0; j < 8; j++) {
            acc[j] = _mm512_madd52lo_epu64(acc[j], a.l[j], bi);
            acc[j + 1] = _mm512_madd52hi_epu64(acc[j + 1], a.l[j], bi);
        }
        const __m512i q = _mm512_madd52lo_epu64(_mm512_setzero_si512(), acc[0], n0);
#pragma GCC unroll 8
        for (int j = 0; j < 8; j++) {
            acc[j] = _mm512_madd52lo_epu64(acc[j], q, P[j]);
            acc[j + 1] = _mm512_madd52hi_epu64(acc[j + 1], q, P[j]);
        }
        const __m512i c = _mm512_srli_epi64(acc[0], 52);          // acc[0] is 0 mod 2^52 now: drop it, keep its carry
        acc[0] = _mm512_add_epi64(acc[1], c);
#pragma GCC unroll 8
        for (int j = 1; j < 8; j++) acc[j] = acc[j + 1];
        acc[8] = _mm512_setzero_si512();
    }
    for (int k = 0; k < 8; k++) r.l[k] = acc[k];
    norm(r);
}

// one Granger-Scott squaring of the state (V0, V1, V2), lanes [x0, x1, y0, y1] of the pairs (r0, r1), (r2, r3), (r4, r5); every V normalised and below 2p
MKZG_IFMA_TARGET static inline void cyclo_sqr(V8& V0, V8& V1, V8& V2) {
    const __m512i S0011 = _mm512_set_epi64(0, 0, 0, 0, 1, 1, 0, 0), S1100 = _mm512_set_epi64(0, 0, 0, 0, 0, 0, 1, 1), S1111 = _mm512_set_epi64(0, 0, 0, 0, 1, 1, 1, 1);
    // _mm512_set_epi64 lists lane 7 first
    const __m512i IA1 = _mm512_set_epi64(1, 0, 1, 0, 1, 0, 1, 0), IA2 = _mm512_set_epi64(3, 2, 3, 2, 0, 0, 0, 0);
    const __m512i IB1 = _mm512_set_epi64(0, 1, 1, 0, 2, 3, 3, 2), IB2 = _mm512_set_epi64(2, 2, 2, 2, 0, 0, 0, 0), IB3 = _mm512_set_epi64(8 + 3, 3, 3, 8 + 3, 0, 0, 0, 0);
    const __m512i IT1 = _mm512_set_epi64(0, 0, 0, 0, 2, 0, 6, 4), IT2 = _mm512_set_epi64(0, 0, 0, 0, 3, 8 + 1, 7, 8 + 5), IT3 = _mm512_set_epi64(0, 0, 0, 0, 0, 0, 8 + 2, 8 + 0);
    const __m512i IT4 = _mm512_set_epi64(0, 0, 0, 0, 0, 0, 8 + 3, 1), IT5 = _mm512_set_epi64(0, 0, 0, 0, 0, 0, 8 + 0, 2), IT6 = _mm512_set_epi64(0, 0, 0, 0, 0, 0, 1, 3);
    const __m512i IVA = _mm512_set_epi64(0, 0, 0, 0, 3, 2, 8 + 1, 8 + 0), IVB = _mm512_set_epi64(0, 0, 0, 0, 8 + 3, 8 + 2, 1, 0);
    const __m512i IG1 = _mm512_set_epi64(0, 0, 0, 0, 1, 0, 2, 2), IG2 = _mm512_set_epi64(0, 0, 0, 0, 0, 0, 3, 8 + 3);
    const __m512i IPK = _mm512_set_epi64(8 + 3, 8 + 2, 8 + 1, 8 + 0, 3, 2, 1, 0), IHI = _mm512_set_epi64(0, 0, 0, 0, 7, 6, 5, 4);
    V8 T[3], Vn[3];
    V8* V[3] = {&V0, &V1, &V2};
    for (int k = 0; k < 3; k++) {
        const V8& v = *V[k];
        neg(Vn[k], v, OFF8);
        // A = [x0, x1, x0, x1, s0, s1, s0, s1], B = [y0, y1, y1, y0, u0, u1, u1, u0] with s = x + y, u = x + xi y = (x0 + y0 - y1, x1 + y0 + y1)
        V8 A, B, M, Mn;
        pick0(A, v, v, IA1, 0xff);
        acc_pick0(A, v, v, IA2, 0xf0);
        pick0(B, v, v, IB1, 0xff);
        acc_pick0(B, v, v, IB2, 0xf0);
        acc_pick0(B, v, Vn[k], IB3, 0xf0);
        norm(A);
        norm(B);
        mul8(M, A, B);                                  // [x0y0, x1y1, x0y1, x1y0, s0u0, s1u1, s0u1, s1u0], each below 2p
        neg(Mn, M, OFF8);
        // the Fp4 square (lo, hi): tmp = x y = (m0 - m1, m2 + m3); lo = s u - tmp - xi tmp; hi = 2 tmp
        //   lo0 = m4 - m5 - 2 m0 + 2 m1 + m2 + m3     lo1 = m6 + m7 - 2 m2 - 2 m3 - m0 + m1     hi0 = 2 m0 - 2 m1     hi1 = 2 m2 + 2 m3
        pick(T[k], M, Mn, IT1, 0x0f, S0011);            // [ m4,   m6,  2m0, 2m2]
        acc_pick(T[k], M, Mn, IT2, 0x0f, S0011);        // [-m5,   m7, -2m1, 2m3]
        acc_pick(T[k], M, Mn, IT3, 0x03, S1100);        // [-2m0, -2m2]
        acc_pick(T[k], M, Mn, IT4, 0x03, S1100);        // [ 2m1, -2m3]
        acc_pick0(T[k], M, Mn, IT5, 0x03);            // [ m2,  -m0]
        acc_pick0(T[k], M, Mn, IT6, 0x03);            // [ m3,   m1]
    }
    // z0 = 3 t0 - 2 r0, z1 = 3 t1 + 2 r1 (from T0, V0);  z4 = 3 t2 - 2 r4, z5 = 3 t3 + 2 r5 (from T1, V2);
    // z2 = 3 xi t5 + 2 r2, z3 = 3 t4 - 2 r3 (from T2, V1), xi (h0, h1) = (h0 - h1, h0 + h1)
    V8 N0v, N1v, N2v, W, G, T2n;
    dbl(N0v, T[0]); add(N0v, N0v, T[0]);
    pick(W, V0, Vn[0], IVA, 0x0f, S1111);               // [-2x0, -2x1, 2y0, 2y1]
    add(N0v, N0v, W);
    dbl(N2v, T[1]); add(N2v, N2v, T[1]);
    pick(W, V2, Vn[2], IVA, 0x0f, S1111);
    add(N2v, N2v, W);
    norm(T[2]);                                         // below 32p: the offset of its negation is 128p
    neg(T2n, T[2], OFF128);
    pick0(G, T[2], T[2], IG1, 0x0f);                  // [h0, h0, l0, l1]
    acc_pick0(G, T[2], T2n, IG2, 0x03);               // [-h1, h1]
    dbl(N1v, G); add(N1v, N1v, G);
    pick(W, V1, Vn[1], IVB, 0x0f, S1111);               // [2x0, 2x1, -2y0, -2y1]
    add(N1v, N1v, W);
    // back below 2p: (N0 | N1) and N2 times the Montgomery one
    V8 one, Pk, R1, R2;
    splat(one, ONE52);
    pick0(Pk, N0v, N1v, IPK, 0xff);
    norm(Pk);
    mul8(R1, Pk, one);
    norm(N2v);
    mul8(R2, N2v, one);
    V0 = R1;
    pick0(V1, R1, R1, IHI, 0x0f);
    V2 = R2;
}

#undef MKZG_IFMA_TARGET
#else
static inline bool available() { return false; }
#endif

}  // namespace ifma
}  // namespace kzg
