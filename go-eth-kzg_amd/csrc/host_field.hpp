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
#include "mont29.hpp"

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
    // r = a b / 2^384 mod p, inputs and output canonical.  On x86-64 with BMI2 + ADX (every CPU an MI355X ships in): the CIOS loop
    // unrolled with mulx and the two carry chains of adcx / adox -- a product's low halves ride the OF chain, its high halves the CF
    // chain -- 6 x (12 mulx + 26 add-with-carry).  p < 2^383, so the running sum never needs an eighth word ("no-carry" CIOS) and ends
    // below 2p.  2.1x the speed of the portable loop below on the verifiers' host finish (tools/ubench_host_finish.cpp); the portable
    // loop stays for other hosts and as the reference the unit tests compare with (tests/test_host_pairing.py).
    static inline void mul(El& r, const El& a, const El& b) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__) && !defined(MKZG_NO_ADX)
        if (have_adx()) { mul_adx(r, a, b); return; }
#endif
        mul_portable(r, a, b);
    }
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__) && !defined(MKZG_NO_ADX)
    static inline bool have_adx() {
        static const bool ok = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx");
        return ok;
    }
#define MKZG_MM_ROW(SRC, T0, T1, T2, T3, T4, T5, T6)                                        \
    "xorl %%eax, %%eax\n\t"                                                                  \
    "mulxq " SRC(0) ", %[lo], %[hi]\n\t adoxq %[lo], %[" T0 "]\n\t adcxq %[hi], %[" T1 "]\n\t" \
    "mulxq " SRC(1) ", %[lo], %[hi]\n\t adoxq %[lo], %[" T1 "]\n\t adcxq %[hi], %[" T2 "]\n\t" \
    "mulxq " SRC(2) ", %[lo], %[hi]\n\t adoxq %[lo], %[" T2 "]\n\t adcxq %[hi], %[" T3 "]\n\t" \
    "mulxq " SRC(3) ", %[lo], %[hi]\n\t adoxq %[lo], %[" T3 "]\n\t adcxq %[hi], %[" T4 "]\n\t" \
    "mulxq " SRC(4) ", %[lo], %[hi]\n\t adoxq %[lo], %[" T4 "]\n\t adcxq %[hi], %[" T5 "]\n\t" \
    "mulxq " SRC(5) ", %[lo], %[hi]\n\t adoxq %[lo], %[" T5 "]\n\t adcxq %[hi], %[" T6 "]\n\t" \
    "movl $0, %%eax\n\t adoxq %%rax, %[" T6 "]\n\t"
#define MKZG_MM_A(j) #j "*8(%[a])"
#define MKZG_MM_P(j) "%[p" #j "]"
    // one CIOS iteration: t += a * b[I]; m = t0 * NINV; t += m * p; the registers then rotate by one (t0 has become zero)
#define MKZG_MM_ITER(I, T0, T1, T2, T3, T4, T5, T6)                                        \
    "movq " #I "*8(%[b]), %%rdx\n\t" MKZG_MM_ROW(MKZG_MM_A, T0, T1, T2, T3, T4, T5, T6)    \
    "movq %[" T0 "], %%rdx\n\t imulq %[ninv], %%rdx\n\t" MKZG_MM_ROW(MKZG_MM_P, T0, T1, T2, T3, T4, T5, T6)
    static inline void mul_adx(El& r, const El& a, const El& b) {
        static const uint64_t PM[6] = {P::MOD[0], P::MOD[1], P::MOD[2], P::MOD[3], P::MOD[4], P::MOD[5]};
        static const uint64_t NI = P::NINV;
        uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, lo, hi;
        asm(MKZG_MM_ITER(0, "t0", "t1", "t2", "t3", "t4", "t5", "t6")
            MKZG_MM_ITER(1, "t1", "t2", "t3", "t4", "t5", "t6", "t0")
            MKZG_MM_ITER(2, "t2", "t3", "t4", "t5", "t6", "t0", "t1")
            MKZG_MM_ITER(3, "t3", "t4", "t5", "t6", "t0", "t1", "t2")
            MKZG_MM_ITER(4, "t4", "t5", "t6", "t0", "t1", "t2", "t3")
            MKZG_MM_ITER(5, "t5", "t6", "t0", "t1", "t2", "t3", "t4")
            : [t0] "+&r"(t0), [t1] "+&r"(t1), [t2] "+&r"(t2), [t3] "+&r"(t3), [t4] "+&r"(t4), [t5] "+&r"(t5), [t6] "+&r"(t6), [lo] "=&r"(lo), [hi] "=&r"(hi)
            : [a] "r"(a.l), [b] "r"(b.l), [ninv] "m"(NI), [p0] "m"(PM[0]), [p1] "m"(PM[1]), [p2] "m"(PM[2]), [p3] "m"(PM[3]), [p4] "m"(PM[4]), [p5] "m"(PM[5]),
              "m"(*(const uint64_t(*)[6])a.l), "m"(*(const uint64_t(*)[6])b.l)
            : "rax", "rdx", "cc");
        // after six rotations the value sits in t6, t0, t1, t2, t3, t4 (t5 is the zeroed word)
        const uint64_t t[6] = {t6, t0, t1, t2, t3, t4};
        uint64_t u[6];
        const uint64_t keep = 0 - sub_raw(u, t, P::MOD);    // all-ones if t < p
        for (int i = 0; i < 6; i++) r.l[i] = (t[i] & keep) | (u[i] & ~keep);
    }
#undef MKZG_MM_ITER
#undef MKZG_MM_P
#undef MKZG_MM_A
#undef MKZG_MM_ROW
#endif
    static inline void mul_portable(El& r, const El& a, const El& b) {
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
    // branch-free: the tower arithmetic of the pairing does about as many additions as multiplications, on unpredictable values
    static inline void add(El& r, const El& a, const El& b) {
        uint64_t t[6], u[6];
        add_raw(t, a.l, b.l);                               // < 2p < 2^384: no carry out
        const uint64_t keep = 0 - sub_raw(u, t, P::MOD);    // all-ones if t < p
        for (int i = 0; i < 6; i++) r.l[i] = (t[i] & keep) | (u[i] & ~keep);
    }
    static inline void dbl(El& r, const El& a) { add(r, a, a); }
    template <int KLOG>
    static inline void sub(El& r, const El& a, const El& b) {
        uint64_t t[6], c = 0;
        const uint64_t fix = 0 - sub_raw(t, a.l, b.l);      // all-ones if a < b: add p back
        for (int i = 0; i < 6; i++) { u128 s = (u128)t[i] + (P::MOD[i] & fix) + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
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
    static inline void inv_fermat(El& r, const El& a) { pow(r, a, P::EXP_INV, 381); }
    // a^-1 (0 -> 0) through the divsteps inversion the kernels use (modinv30.hpp, on the 29-bit-limb form of the same field): 3 field
    // multiplications of conversions around ~20 k simple integer operations, where a^(p-2) is 381 squarings and ~190 multiplications
    // (measured on the GPU box's host: 15 us).  Three of them sit on every verifier call's host finish (two conversions to affine, one in
    // the easy part of the final exponentiation) and 136 in every prepared G2 point; tests/test_host_pairing.py compares the two forms.
    static inline void inv(El& r, const El& a) {
        uint64_t pl[6];
        to_plain(pl, a);
        uint32_t w[12];
        for (int i = 0; i < 6; i++) { w[2 * i] = (uint32_t)pl[i]; w[2 * i + 1] = (uint32_t)(pl[i] >> 32); }
        fp_t x, xm, im, ip;
        Fp::from_words32(x, w);
        Fp::from_plain(xm, x);
        ModInv30<FpParams>::inv_mont(im, xm);
        Fp::to_plain(ip, im);
        Fp::to_words32(w, ip);
        for (int i = 0; i < 6; i++) pl[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
        from_plain(r, pl);
    }
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
