// modinv30.hpp -- modular inversion by Bernstein-Yang "divsteps" (safegcd) in batches of 30 steps on
// signed 30-bit limbs, variable time.  ~20k integer instructions for a 381-bit modulus against
// ~285k for the Fermat exponentiation a^(p-2) -- it removes the longest serial chain of the
// commitment path's tail (one inversion per blob to leave XYZZ coordinates) and of every to_affine.
//
// Algorithm as published (Bernstein & Yang, "Fast constant-time gcd computation and modular
// inversion", 2019; the limb-level batching follows the well-known 2x2 transition-matrix scheme):
//   divstep(delta, f, g) = (1-delta, g, (g-f)/2)           if delta > 0 and g odd
//                        = (1+delta, f, (g + (g mod 2) f)/2) otherwise
// starting from (1, M, a) with d = 0, e = 1 and the invariants d*a = f, e*a = g (mod M).
// When g reaches 0, f = +-1 and a^-1 = +-d.
#pragma once
#include <stdint.h>
#include "mont29.hpp"

namespace kzg {

template <class P>
struct ModInv30 {
    static constexpr int L = P::N;                 // 30-bit limbs: 30*N >= 29*N bits, plenty of headroom
    static constexpr int32_t M30 = (int32_t)(0xffffffffu >> 2);
    struct S30 { int32_t v[L]; };

    // 30 divsteps on the low words; returns the new delta and the transition matrix t with
    // 2^30 [f'; g'] = t [f; g]
    static KZ_HD int32_t divsteps30(int32_t delta, uint32_t f, uint32_t g, int32_t t[4]) {
        int32_t u = 1, v = 0, q = 0, r = 1;
        for (int i = 0; i < 30; i++) {
            if (delta > 0 && (g & 1u)) {
                delta = 1 - delta;
                uint32_t nf = g, ng = (g - f) >> 1;
                int32_t nu = q * 2, nv = r * 2, nq = q - u, nr = r - v;      // (* 2, not << 1: the entries are signed)
                f = nf; g = ng; u = nu; v = nv; q = nq; r = nr;
            } else {
                delta = 1 + delta;
                if (g & 1u) { g += f; q += u; r += v; }
                g >>= 1;
                u *= 2;
                v *= 2;
            }
        }
        t[0] = u; t[1] = v; t[2] = q; t[3] = r;
        return delta;
    }
    // [f; g] <- t [f; g] / 2^30 (exact)
    static KZ_HD void update_fg(S30& f, S30& g, const int32_t t[4]) {
        const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
        int64_t cf = u * f.v[0] + v * g.v[0];
        int64_t cg = q * f.v[0] + r * g.v[0];
        cf >>= 30;
        cg >>= 30;
#pragma unroll
        for (int i = 1; i < L; i++) {
            cf += u * f.v[i] + v * g.v[i];
            cg += q * f.v[i] + r * g.v[i];
            f.v[i - 1] = (int32_t)cf & M30; cf >>= 30;
            g.v[i - 1] = (int32_t)cg & M30; cg >>= 30;
        }
        f.v[L - 1] = (int32_t)cf;
        g.v[L - 1] = (int32_t)cg;
    }
    // [d; e] <- t [d; e] / 2^30 mod M, values kept in (-2M, M)
    static KZ_HD void update_de(S30& d, S30& e, const int32_t t[4]) {
        const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
        const int32_t sd = d.v[L - 1] >> 31, se = e.v[L - 1] >> 31;
        int32_t md = (u & sd) + (v & se);
        int32_t me = (q & sd) + (r & se);
        int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0];
        int64_t ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
        md -= (int32_t)((P::MINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
        me -= (int32_t)((P::MINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
        cd += (int64_t)P::MOD30[0] * md;
        ce += (int64_t)P::MOD30[0] * me;
        cd >>= 30;
        ce >>= 30;
#pragma unroll
        for (int i = 1; i < L; i++) {
            cd += (int64_t)u * d.v[i] + (int64_t)v * e.v[i];
            ce += (int64_t)q * d.v[i] + (int64_t)r * e.v[i];
            cd += (int64_t)P::MOD30[i] * md;
            ce += (int64_t)P::MOD30[i] * me;
            d.v[i - 1] = (int32_t)cd & M30; cd >>= 30;
            e.v[i - 1] = (int32_t)ce & M30; ce >>= 30;
        }
        d.v[L - 1] = (int32_t)cd;
        e.v[L - 1] = (int32_t)ce;
    }
    // r in (-2M, M) -> sign * r mod M in [0, M), limbs normalized
    static KZ_HD void normalize(S30& r, int32_t sign) {
        int32_t cond_add = r.v[L - 1] >> 31;
        int32_t cond_neg = sign >> 31;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            int32_t x = r.v[i] + (P::MOD30[i] & cond_add);
            x = (x ^ cond_neg) - cond_neg;
            x += c;
            if (i < L - 1) { c = x >> 30; x &= M30; }
            r.v[i] = x;
        }
        cond_add = r.v[L - 1] >> 31;
        c = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            int32_t x = r.v[i] + (P::MOD30[i] & cond_add) + c;
            if (i < L - 1) { c = x >> 30; x &= M30; }
            r.v[i] = x;
        }
    }
    // plain canonical integer (32-bit words) -> its inverse mod M (32-bit words); 0 -> 0
    static KZ_HD_NOINLINE void inv_words(uint32_t out[P::N32], const uint32_t in[P::N32]) {
        S30 f, g, d, e;
#pragma unroll
        for (int i = 0; i < L; i++) {
            f.v[i] = P::MOD30[i];
            d.v[i] = 0;
            e.v[i] = i == 0 ? 1 : 0;
            const int bit = 30 * i, wi = bit >> 5, off = bit & 31;
            uint32_t x = 0;
            if (wi < P::N32) x = in[wi] >> off;
            if (off + 30 > 32 && wi + 1 < P::N32) x |= in[wi + 1] << (32 - off);
            g.v[i] = (int32_t)(x & (uint32_t)M30);
        }
        int32_t delta = 1;
        for (int iter = 0; iter < 64; iter++) {     // <= (49*bits+57)/17/30 batches; 64 is a safe cap
            int32_t gz = 0;
#pragma unroll
            for (int i = 0; i < L; i++) gz |= g.v[i];
            if (gz == 0) break;
            int32_t t[4];
            const uint32_t f0 = (uint32_t)f.v[0] | ((uint32_t)f.v[1] << 30);
            const uint32_t g0 = (uint32_t)g.v[0] | ((uint32_t)g.v[1] << 30);
            delta = divsteps30(delta, f0, g0, t);
            update_de(d, e, t);
            update_fg(f, g, t);
        }
        // f = +-1 (or M when the input was 0): result = sign(f) * d
        normalize(d, f.v[L - 1]);
        // was the input 0 ?  then f = +-M and d = 0 already
#pragma unroll
        for (int j = 0; j < P::N32; j++) out[j] = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            const int bit = 30 * i, wi = bit >> 5, off = bit & 31;
            const uint32_t x = (uint32_t)d.v[i];
            if (wi < P::N32) out[wi] |= x << off;
            if (off + 30 > 32 && wi + 1 < P::N32) out[wi + 1] |= x >> (32 - off);
        }
    }
    // Montgomery form in, Montgomery form out (loose in, loose out): a^-1 R = modinv(a R) * R^2
    static KZ_HD void inv_mont(typename Mont29<P>::El& r, const typename Mont29<P>::El& a) {
        using F = Mont29<P>;
        typename F::El c, pl, r3;
        uint32_t w[P::N32], o[P::N32];
        F::canon(c, a);
        F::to_words32(w, c);
        inv_words(o, w);
        F::from_words32(pl, o);
        F::set(r3, P::R3);
        F::mul(r, pl, r3);
    }
};

}  // namespace kzg
