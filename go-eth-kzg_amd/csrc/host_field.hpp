// host_field.hpp -- host-side (CPU) field arithmetic of the PRODUCT: Fp with 6 x 64-bit limbs,
// Montgomery R = 2^384, always canonical.  Used only for the O(1)-per-call finishing steps that
// stay on the host by design (DESIGN.md "what runs where"): the Horner combination of the
// variable-base MSM window sums, one fixed-base scalar multiplication and the final pairing
// check of the batch verifiers (reference internal/kzg/kzg_verify.go:160-199).  It offers the
// same static interface as kzg::Fp (mont29.hpp) so that the group law of g1.hpp is shared.
// This is not a CPU fallback for any kernel: nothing here computes an MSM, NTT or hash.
#pragma once
#include <stdint.h>
#include <string.h>
#include "constants29.h"

namespace kzg {

struct HostFp {
    using P = HostFpParams;
    typedef unsigned __int128 u128;
    struct El {
        uint64_t l[6];
    };
    static inline void set_zero(El& r) { memset(r.l, 0, 48); }
    static inline void set_one(El& r) { memcpy(r.l, P::ONE, 48); }
    static inline void set(El& r, const uint64_t* c) { memcpy(r.l, c, 48); }
    static inline bool limbs_all_zero(const El& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3] | a.l[4] | a.l[5]) == 0; }
    static inline bool limbs_equal(const El& a, const El& b) { return memcmp(a.l, b.l, 48) == 0; }
    static inline bool geq_mod(const uint64_t* t) {
        for (int i = 5; i >= 0; i--) {
            if (t[i] > P::MOD[i]) return true;
            if (t[i] < P::MOD[i]) return false;
        }
        return true;
    }
    static inline void mul(El& r, const El& a, const El& b) {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 6; i++) {
            uint64_t bi = b.l[i];
            u128 x = (u128)a.l[0] * bi + t[0];
            uint64_t A = (uint64_t)(x >> 64);
            uint64_t q = (uint64_t)x * P::NINV;
            u128 y = (u128)q * P::MOD[0] + (uint64_t)x;
            uint64_t C = (uint64_t)(y >> 64);
            for (int j = 1; j < 6; j++) {
                x = (u128)a.l[j] * bi + t[j] + A;
                A = (uint64_t)(x >> 64);
                y = (u128)q * P::MOD[j] + (uint64_t)x + C;
                C = (uint64_t)(y >> 64);
                t[j - 1] = (uint64_t)y;
            }
            t[5] = C + A;
        }
        if (geq_mod(t)) sub_raw(t, t, P::MOD);
        memcpy(r.l, t, 48);
    }
    static inline void sqr(El& r, const El& a) { mul(r, a, a); }
    static inline void opaque(El&) {}
    template <int KLOG>
    static inline void sub_2c(El& r, const El& a, const El& c2) {
        El t;
        sub<KLOG>(t, a, c2);
        sub<KLOG>(r, t, c2);
    }
    static inline void triple(El& r, const El& a) {
        El t;
        add(t, a, a);
        add(r, t, a);
    }
    template <int KLOG>
    static inline void sub_b_2c(El& r, const El& a, const El& b, const El& c2) {
        El t;
        sub<KLOG>(t, a, b);
        sub<KLOG>(t, t, c2);
        sub<KLOG>(r, t, c2);
    }
    template <int KLOG>
    static inline void sub_b_c(El& r, const El& a, const El& b, const El& c1) {
        El t;
        sub<KLOG>(t, a, b);
        sub<KLOG>(r, t, c1);
    }
    static inline void select(El& r, bool c, const El& a, const El& b) { r = c ? a : b; }
    static inline void mul2(El& r, const El& a, const El& b, const El& c, const El& d) {   // a*b + c*d
        El t1, t2;
        mul(t1, a, b);
        mul(t2, c, d);
        add(r, t1, t2);
    }
    static inline uint64_t add_raw(uint64_t* o, const uint64_t* a, const uint64_t* b) {
        uint64_t c = 0;
        for (int i = 0; i < 6; i++) { u128 s = (u128)a[i] + b[i] + c; o[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        return c;
    }
    static inline uint64_t sub_raw(uint64_t* o, const uint64_t* a, const uint64_t* b) {
        uint64_t br = 0;
        for (int i = 0; i < 6; i++) { u128 s = (u128)a[i] - b[i] - br; o[i] = (uint64_t)s; br = (uint64_t)(s >> 64) & 1; }
        return br;
    }
    static inline void add(El& r, const El& a, const El& b) {
        uint64_t t[6];
        add_raw(t, a.l, b.l);
        if (geq_mod(t)) sub_raw(t, t, P::MOD);
        memcpy(r.l, t, 48);
    }
    static inline void dbl(El& r, const El& a) { add(r, a, a); }
    template <int KLOG>
    static inline void sub(El& r, const El& a, const El& b) {
        uint64_t t[6];
        if (sub_raw(t, a.l, b.l)) add_raw(t, t, P::MOD);
        memcpy(r.l, t, 48);
    }
    template <int KLOG>
    static inline void neg(El& r, const El& a) {
        if (limbs_all_zero(a)) { r = a; return; }
        uint64_t t[6];
        sub_raw(t, P::MOD, a.l);
        memcpy(r.l, t, 48);
    }
    static inline void canon(El& r, const El& a) { r = a; }
    template <uint32_t BOUND>
    static inline bool is_zero_mod(const El& a) { return limbs_all_zero(a); }
    static inline bool eq_mod(const El& a, const El& b) { return limbs_equal(a, b); }
    static inline void pow(El& r, const El& a, const uint64_t* e, int nbits) {
        El acc;
        set_one(acc);
        for (int i = nbits - 1; i >= 0; i--) {
            sqr(acc, acc);
            if ((e[i >> 6] >> (i & 63)) & 1) mul(acc, acc, a);
        }
        r = acc;
    }
    static inline void inv(El& r, const El& a) { pow(r, a, P::EXP_INV, 381); }
    static inline bool sqrt(El& r, const El& a) {
        El s, c;
        pow(s, a, P::EXP_SQRT, 381);
        sqr(c, s);
        if (!limbs_equal(c, a)) return false;
        r = s;
        return true;
    }
    // plain little-endian 64-bit words (value < p) <-> Montgomery form
    static inline void from_plain(El& r, const uint64_t* w) {
        El t, r2;
        memcpy(t.l, w, 48);
        set(r2, P::R2);
        mul(r, t, r2);
    }
    static inline void to_plain(uint64_t* w, const El& a) {
        El one, t;
        set_zero(one);
        one.l[0] = 1;
        mul(t, a, one);
        memcpy(w, t.l, 48);
    }
    static inline bool from_be48(El& r, const uint8_t* b) {   // false if >= p
        uint64_t v[6];
        for (int i = 0; i < 6; i++) { uint64_t w = 0; for (int j = 0; j < 8; j++) w = (w << 8) | b[(5 - i) * 8 + j]; v[i] = w; }
        if (geq_mod(v)) return false;
        from_plain(r, v);
        return true;
    }
    static inline bool lex_largest(const El& a) {   // a > (p-1)/2
        uint64_t v[6];
        to_plain(v, a);
        for (int i = 5; i >= 0; i--) {
            if (v[i] > P::HALF[i]) return true;
            if (v[i] < P::HALF[i]) return false;
        }
        return false;
    }
};

}  // namespace kzg
