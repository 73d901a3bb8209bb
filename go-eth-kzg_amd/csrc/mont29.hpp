// mont29.hpp -- Montgomery arithmetic in a redundant radix-2^29 representation, written for
// gfx950 (MI355X) VALU.  Shared by every kernel of the product; also compiles as plain C++
// for host-side unit tests (tests/test_mont29_host.py drives it through a tiny C shim).
//
// Why radix 2^29 and not 12 x 32-bit limbs (the in-memory layout of gnark's fp.Element that the
// reference uses through internal/multiexp/multiexp.go:20-26):
//   * gfx950 issues v_mad_u64_u32 (32x32+64 -> 64) at the same 4 cycles/wave as v_add_co_u32
//     (profiles/r01_ubench_valu.txt), and a VCC carry link (v_add_co -> v_addc_co) needs an extra
//     wait state.  So the cheapest big multiply is the one with the fewest instructions and no
//     carry chains: 29-bit limbs leave 6 spare bits per 64-bit column sum, enough to add all
//     28 partial products of a 14-limb Montgomery multiply with one v_mad_u64_u32 each
//     (28 * 2^58 < 2^63) and propagate carries once at the end.  hipcc emits 392 v_mad_u64_u32
//     + ~95 shift/and/add per Fp multiply, against ~1230 instructions for 32-bit CIOS.
//   * R = 2^(29*14) = 2^406 leaves 25 spare bits above p: values only need to stay below
//     ~2^393 for a product to come out below p + epsilon, so additions and subtractions never
//     reduce modulo p ("loose" values); subtraction adds a fixed multiple K*p as bias.
//
// Conventions
//   El.l[i] : limb i, weight 2^(29 i).  "normalized" = every limb < 2^29 except possibly the top one.
//   loose   : any normalized value < 2^393 (Fp) representing its residue mod M.
//   canon   : the unique representative in [0, M).
#pragma once
#include <stdint.h>
#include "constants29.h"

#if defined(__HIPCC__)
#define KZ_HD __host__ __device__ __forceinline__
#define KZ_HD_NOINLINE __host__ __device__ __noinline__
#else
#define KZ_HD inline
#define KZ_HD_NOINLINE inline
#endif

// Loop-carried field elements pass through an empty asm on the device (Mont29::opaque, called at the top of the
// group-law routines): without it LLVM widens loop-carried limb registers to 64 bits (zext folded into the phi)
// and every limb product becomes a 64x32-bit multiply (two v_mad_u64_u32 + moves instead of one).
#if defined(__HIP_DEVICE_COMPILE__)
#define KZ_OPAQUE32(x) asm("" : "+v"(x))
#else
#define KZ_OPAQUE32(x) do { } while (0)
#endif

namespace kzg {

constexpr int W29 = 29;
constexpr uint32_t MASK29 = (1u << 29) - 1;

template <class P>
struct ModInv30;   // modinv30.hpp (included at the end of this header)

template <class P>
struct Mont29 {
    static constexpr int N = P::N;
    struct El {
        uint32_t l[N];
    };

    static KZ_HD void opaque(El& r) {
#pragma unroll
        for (int i = 0; i < N; i++) KZ_OPAQUE32(r.l[i]);
    }
    static KZ_HD void select(El& r, bool c, const El& a, const El& b) {
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = c ? a.l[i] : b.l[i];
    }
    static KZ_HD void set_zero(El& r) {
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
    }
    static KZ_HD void set(El& r, const uint32_t* c) {
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = c[i];
    }
    static KZ_HD void set_one(El& r) {
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::ONE[i];
    }
    static KZ_HD bool limbs_all_zero(const El& a) {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < N; i++) t |= a.l[i];
        return t == 0;
    }
    static KZ_HD bool limbs_equal(const El& a, const El& b) {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < N; i++) t |= a.l[i] ^ b.l[i];
        return t == 0;
    }

    // r = a*b/R mod M.  a, b normalized (one of them may carry limbs up to 2^30).
    // Output normalized, value < a*b/R + M.
    static KZ_HD void mul(El& r, const El& a, const El& b) {
        uint64_t t[N + 1];
#pragma unroll
        for (int i = 0; i < N; i++) {
#pragma unroll
            for (int j = 0; j < N; j++) t[j] = (i == 0 ? 0 : t[j]) + (uint64_t)a.l[j] * b.l[i];
            if (i == 0) t[N] = 0;
            uint32_t q = ((uint32_t)t[0] * P::NINV) & MASK29;
#pragma unroll
            for (int j = 0; j < N; j++) t[j] += (uint64_t)q * P::MOD[j];
            t[1] += t[0] >> W29;
#pragma unroll
            for (int j = 0; j < N; j++) t[j] = t[j + 1];
            t[N] = 0;
        }
#pragma unroll
        for (int j = 0; j < N - 1; j++) {
            t[j + 1] += t[j] >> W29;
            r.l[j] = (uint32_t)t[j] & MASK29;
        }
        r.l[N - 1] = (uint32_t)t[N - 1];
    }
    // r = a*a/R mod M: the 91 off-diagonal products are formed once against the doubled operand
    // (105 + 196 multiply-adds instead of 392).  a normalized; output as mul.
    static KZ_HD void sqr(El& r, const El& a) {
        uint64_t t[2 * N];
#pragma unroll
        for (int k = 0; k < 2 * N; k++) t[k] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            t[2 * i] += (uint64_t)a.l[i] * a.l[i];
            const uint32_t a2 = a.l[i] << 1;
#pragma unroll
            for (int j = i + 1; j < N; j++) t[i + j] += (uint64_t)a2 * a.l[j];
        }
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t q = ((uint32_t)t[i] * P::NINV) & MASK29;
#pragma unroll
            for (int j = 0; j < N; j++) t[i + j] += (uint64_t)q * P::MOD[j];
            t[i + 1] += t[i] >> W29;
        }
#pragma unroll
        for (int j = N; j < 2 * N - 1; j++) {
            t[j + 1] += t[j] >> W29;
            r.l[j - N] = (uint32_t)t[j] & MASK29;
        }
        r.l[N - 1] = (uint32_t)t[2 * N - 1];
    }
    // r = (a*b + c*d)/R mod M with ONE Montgomery reduction.  All four operands normalized with limbs < 2^29
    // (42 * 2^58 < 2^64 per column).  Output normalized, value < (a*b + c*d)/R + M.
    static KZ_HD void mul2(El& r, const El& a, const El& b, const El& c, const El& d) {
        uint64_t t[N + 1];
#pragma unroll
        for (int i = 0; i < N; i++) {
#pragma unroll
            for (int j = 0; j < N; j++) t[j] = (i == 0 ? 0 : t[j]) + (uint64_t)a.l[j] * b.l[i];
#pragma unroll
            for (int j = 0; j < N; j++) t[j] += (uint64_t)c.l[j] * d.l[i];
            if (i == 0) t[N] = 0;
            uint32_t q = ((uint32_t)t[0] * P::NINV) & MASK29;
#pragma unroll
            for (int j = 0; j < N; j++) t[j] += (uint64_t)q * P::MOD[j];
            t[1] += t[0] >> W29;
#pragma unroll
            for (int j = 0; j < N; j++) t[j] = t[j + 1];
            t[N] = 0;
        }
#pragma unroll
        for (int j = 0; j < N - 1; j++) {
            t[j + 1] += t[j] >> W29;
            r.l[j] = (uint32_t)t[j] & MASK29;
        }
        r.l[N - 1] = (uint32_t)t[N - 1];
    }

    // r = a + b, carry-normalized, no modular reduction
    static KZ_HD void add(El& r, const El& a, const El& b) {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint32_t v = a.l[i] + b.l[i] + c;
            if (i < N - 1) {
                r.l[i] = v & MASK29;
                c = v >> W29;
            } else {
                r.l[i] = v;
            }
        }
    }
    static KZ_HD void dbl(El& r, const El& a) { add(r, a, a); }

    // r = a - b + K*M with K = 2^KLOG (KLOG in 0..5); requires b <= K*M.  No modular reduction.
    template <int KLOG>
    static KZ_HD void sub(El& r, const El& a, const El& b) {
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            int32_t v = (int32_t)(a.l[i] - b.l[i] + P::KMOD[KLOG][i]) + c;
            if (i < N - 1) {
                r.l[i] = (uint32_t)v & MASK29;
                c = v >> W29;
            } else {
                r.l[i] = (uint32_t)v;
            }
        }
    }
    // r = a - b - 2c + K*M with K = 2^KLOG; requires b + 2c <= K*M.  One carry sweep instead of three (X3 = R^2 - PPP - 2Q).
    template <int KLOG>
    static KZ_HD void sub_b_2c(El& r, const El& a, const El& b, const El& c2) {
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            int32_t v = (int32_t)(a.l[i] - b.l[i] - 2 * c2.l[i] + P::KMOD[KLOG][i]) + c;
            if (i < N - 1) {
                r.l[i] = (uint32_t)v & MASK29;
                c = v >> W29;
            } else {
                r.l[i] = (uint32_t)v;
            }
        }
    }
    // r = a - b - c + K*M with K = 2^KLOG; requires b + c <= K*M  (Jacobian doubling: (X + 2B)^2 - A - 4C)
    template <int KLOG>
    static KZ_HD void sub_b_c(El& r, const El& a, const El& b, const El& c1) {
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            int32_t v = (int32_t)(a.l[i] - b.l[i] - c1.l[i] + P::KMOD[KLOG][i]) + c;
            if (i < N - 1) {
                r.l[i] = (uint32_t)v & MASK29;
                c = v >> W29;
            } else {
                r.l[i] = (uint32_t)v;
            }
        }
    }
    // r = a - 2c + K*M (requires 2c <= K*M) and r = 3a: fused forms for the doubling formulas, one carry sweep each
    template <int KLOG>
    static KZ_HD void sub_2c(El& r, const El& a, const El& c2) {
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            int32_t v = (int32_t)(a.l[i] - 2 * c2.l[i] + P::KMOD[KLOG][i]) + c;
            if (i < N - 1) {
                r.l[i] = (uint32_t)v & MASK29;
                c = v >> W29;
            } else {
                r.l[i] = (uint32_t)v;
            }
        }
    }
    static KZ_HD void triple(El& r, const El& a) {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint32_t v = 3 * a.l[i] + c;
            if (i < N - 1) {
                r.l[i] = v & MASK29;
                c = v >> W29;
            } else {
                r.l[i] = v;
            }
        }
    }
    // r = K*M - a
    template <int KLOG>
    static KZ_HD void neg(El& r, const El& a) {
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            int32_t v = (int32_t)(P::KMOD[KLOG][i] - a.l[i]) + c;
            if (i < N - 1) {
                r.l[i] = (uint32_t)v & MASK29;
                c = v >> W29;
            } else {
                r.l[i] = (uint32_t)v;
            }
        }
    }
    // if a >= M then a - M else a   (one conditional subtraction)
    static KZ_HD void reduce_once(El& r, const El& a) {
        El d;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            int32_t v = (int32_t)(a.l[i] - P::MOD[i]) + c;
            if (i < N - 1) {
                d.l[i] = (uint32_t)v & MASK29;
                c = v >> W29;
            } else {
                d.l[i] = (uint32_t)v;
            }
        }
        bool lt = (int32_t)d.l[N - 1] < 0;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = lt ? a.l[i] : d.l[i];
    }
    // canonical representative of a loose value (same residue, still Montgomery form)
    static KZ_HD void canon(El& r, const El& a) {
        El one, t;
        set_one(one);
        mul(t, a, one);
        reduce_once(r, t);
    }
    // Montgomery form -> plain integer (canonical, limbs29)
    static KZ_HD void to_plain(El& r, const El& a) {
        El one, t;
        set_zero(one);
        one.l[0] = 1;
        mul(t, a, one);
        reduce_once(r, t);
    }
    // plain integer (normalized limbs, any value < 2^393) -> Montgomery form (loose)
    static KZ_HD void from_plain(El& r, const El& a) {
        El r2;
        set(r2, P::R2);
        mul(r, a, r2);
    }
    // a == 0 (mod M) for a loose a known to satisfy 0 <= a < BOUND*M, BOUND < 2^29.
    // Fast filter: if a = k*M then k = a_0 * M^-1 mod 2^29; almost every non-multiple fails it.
    template <uint32_t BOUND>
    static KZ_HD bool is_zero_mod(const El& a) {
        uint32_t k = (a.l[0] * P::PINV) & MASK29;
        if (k >= BOUND) return false;
        El c;
        canon(c, a);
        return limbs_all_zero(c);
    }
    static KZ_HD bool eq_mod(const El& a, const El& b) {
        El ca, cb;
        canon(ca, a);
        canon(cb, b);
        return limbs_equal(ca, cb);
    }

    // ---- packing: 32-bit little-endian words <-> 29-bit limbs (plain integers)
    static KZ_HD void from_words32(El& r, const uint32_t* w) {  // w has P::N32 words
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int bit = W29 * i, wi = bit >> 5, off = bit & 31;
            uint32_t v = 0;
            if (wi < P::N32) v = w[wi] >> off;
            if (off + W29 > 32 && wi + 1 < P::N32) v |= w[wi + 1] << (32 - off);
            r.l[i] = v & MASK29;
        }
    }
    static KZ_HD void to_words32(uint32_t* w, const El& a) {  // a canonical (fits N32 words)
#pragma unroll
        for (int j = 0; j < P::N32; j++) w[j] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int bit = W29 * i, wi = bit >> 5, off = bit & 31;
            if (wi < P::N32) w[wi] |= a.l[i] << off;
            if (off + W29 > 32 && wi + 1 < P::N32) w[wi + 1] |= a.l[i] >> (32 - off);
        }
    }
    // words32 (little-endian) >= modulus ?
    static KZ_HD bool words32_geq_mod(const uint32_t* w) {
        bool gt = false, lt = false;
#pragma unroll
        for (int j = P::N32 - 1; j >= 0; j--) {
            if (!gt && !lt) {
                if (w[j] > P::MOD32[j]) gt = true;
                else if (w[j] < P::MOD32[j]) lt = true;
            }
        }
        return !lt;  // equal counts as >=
    }

    // r = a^e, e given as N32 little-endian 32-bit words (a loose; r loose).  Rolled loop.
    // a^e with fixed 4-bit windows: nbits squarings + one multiplication per non-zero nibble (+ 14 for the table), against one per
    // set bit of the plain binary method (the square-root exponent (p+1)/4 has ~190 of them).  Inputs/outputs loose.
    static KZ_HD_NOINLINE void pow_words(El& r, const El& a, const uint32_t* e, int nbits) {
        El tbl[15];
        tbl[0] = a;
        for (int i = 1; i < 15; i++) mul(tbl[i], tbl[i - 1], a);
        El acc;
        set_one(acc);
        const int nw = (nbits + 3) / 4;
        for (int w = nw - 1; w >= 0; w--) {
            if (w != nw - 1) {
                sqr(acc, acc);
                sqr(acc, acc);
                sqr(acc, acc);
                sqr(acc, acc);
            }
            const uint32_t d = (e[w >> 3] >> (4 * (w & 7))) & 15u;
            if (d) mul(acc, acc, tbl[d - 1]);
        }
        r = acc;
    }
    static KZ_HD void inv_fermat(El& r, const El& a) { pow_words(r, a, P::EXP_INV, P::BITS); }   // a^(M-2); 0 -> 0
    static KZ_HD void inv(El& r, const El& a) { ModInv30<P>::inv_mont(r, a); }                      // divsteps; 0 -> 0
};

using Fp = Mont29<FpParams>;
using Fr = Mont29<FrParams>;
using fp_t = Fp::El;
using fr_t = Fr::El;

// ---- Fr helpers that keep values canonical (cheap, and makes equality / serialisation trivial)
struct FrC {
    static KZ_HD void add(fr_t& r, const fr_t& a, const fr_t& b) {
        fr_t t;
        Fr::add(t, a, b);
        Fr::reduce_once(r, t);
    }
    static KZ_HD void sub(fr_t& r, const fr_t& a, const fr_t& b) {
        fr_t t;
        Fr::sub<0>(t, a, b);
        Fr::reduce_once(r, t);
    }
    static KZ_HD void mul(fr_t& r, const fr_t& a, const fr_t& b) {
        fr_t t;
        Fr::mul(t, a, b);
        Fr::reduce_once(r, t);
    }
    static KZ_HD void neg(fr_t& r, const fr_t& a) {
        fr_t t;
        Fr::neg<0>(t, a);
        Fr::reduce_once(r, t);
    }
};

}  // namespace kzg

#include "modinv30.hpp"
