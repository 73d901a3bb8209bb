// tests/host_shim/mont29_shim.cpp -- exposes the product's header-only field / curve code
// (go-eth-kzg_amd/csrc/mont29.hpp, g1.hpp) to Python so that it can be unit-tested on the CPU
// against big-integer arithmetic (tests/test_mont29_host.py).  Host logic test only; the
// kernels use the very same templates on the device.
#include "../../go-eth-kzg_amd/csrc/g1.hpp"
#include "../../go-eth-kzg_amd/csrc/modinv30.hpp"
#include <cstring>
using namespace kzg;
extern "C" {
void shim_fp_mul(const uint32_t* a, const uint32_t* b, uint32_t* o) { fp_t x, y, r; Fp::set(x, a); Fp::set(y, b); Fp::mul(r, x, y); memcpy(o, r.l, 56); }
void shim_fp_sqr(const uint32_t* a, uint32_t* o) { fp_t x, r; Fp::set(x, a); Fp::sqr(r, x); memcpy(o, r.l, 56); }
void shim_fp_mul2(const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* o) {
    fp_t x, y, z, w, r; Fp::set(x, a); Fp::set(y, b); Fp::set(z, c); Fp::set(w, d); Fp::mul2(r, x, y, z, w); memcpy(o, r.l, 56);
}
void shim_fp_sub_b_2c(const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t* o) { fp_t x, y, z, r; Fp::set(x, a); Fp::set(y, b); Fp::set(z, c); Fp::sub_b_2c<2>(r, x, y, z); memcpy(o, r.l, 56); }
void shim_fp_sub_2c(const uint32_t* a, const uint32_t* c, uint32_t* o) { fp_t x, z, r; Fp::set(x, a); Fp::set(z, c); Fp::sub_2c<2>(r, x, z); memcpy(o, r.l, 56); }
void shim_fp_triple(const uint32_t* a, uint32_t* o) { fp_t x, r; Fp::set(x, a); Fp::triple(r, x); memcpy(o, r.l, 56); }
void shim_fr_sqr(const uint32_t* a, uint32_t* o) { fr_t x, r; Fr::set(x, a); Fr::sqr(r, x); memcpy(o, r.l, 36); }
void shim_fp_add(const uint32_t* a, const uint32_t* b, uint32_t* o) { fp_t x, y, r; Fp::set(x, a); Fp::set(y, b); Fp::add(r, x, y); memcpy(o, r.l, 56); }
void shim_fp_sub(int klog, const uint32_t* a, const uint32_t* b, uint32_t* o) {
    fp_t x, y, r; Fp::set(x, a); Fp::set(y, b);
    switch (klog) { case 0: Fp::sub<0>(r, x, y); break; case 1: Fp::sub<1>(r, x, y); break; case 2: Fp::sub<2>(r, x, y); break; case 3: Fp::sub<3>(r, x, y); break; default: Fp::sub<4>(r, x, y); }
    memcpy(o, r.l, 56);
}
void shim_fp_canon(const uint32_t* a, uint32_t* o) { fp_t x, r; Fp::set(x, a); Fp::canon(r, x); memcpy(o, r.l, 56); }
void shim_fp_to_plain(const uint32_t* a, uint32_t* o) { fp_t x, r; Fp::set(x, a); Fp::to_plain(r, x); memcpy(o, r.l, 56); }
void shim_fp_from_plain(const uint32_t* a, uint32_t* o) { fp_t x, r; Fp::set(x, a); Fp::from_plain(r, x); memcpy(o, r.l, 56); }
int shim_fp_is_zero16(const uint32_t* a) { fp_t x; Fp::set(x, a); return Fp::is_zero_mod<16>(x); }
void shim_fp_inv(const uint32_t* a, uint32_t* o) { fp_t x, r; Fp::set(x, a); Fp::pow_words(r, x, FpParams::EXP_INV, FpParams::BITS); memcpy(o, r.l, 56); }
void shim_fp_inv_fast(const uint32_t* a, uint32_t* o) { fp_t x, r; Fp::set(x, a); ModInv30<FpParams>::inv_mont(r, x); memcpy(o, r.l, 56); }
void shim_fr_inv_fast(const uint32_t* a, uint32_t* o) { fr_t x, r; Fr::set(x, a); ModInv30<FrParams>::inv_mont(r, x); memcpy(o, r.l, 36); }
void shim_fp_words(const uint32_t* w12, uint32_t* o14, uint32_t* back12) { fp_t x; Fp::from_words32(x, w12); memcpy(o14, x.l, 56); Fp::to_words32(back12, x); }
void shim_fr_mul(const uint32_t* a, const uint32_t* b, uint32_t* o) { fr_t x, y, r; Fr::set(x, a); Fr::set(y, b); FrC::mul(r, x, y); memcpy(o, r.l, 36); }
void shim_fr_add(const uint32_t* a, const uint32_t* b, uint32_t* o) { fr_t x, y, r; Fr::set(x, a); Fr::set(y, b); FrC::add(r, x, y); memcpy(o, r.l, 36); }
void shim_fr_sub(const uint32_t* a, const uint32_t* b, uint32_t* o) { fr_t x, y, r; Fr::set(x, a); Fr::set(y, b); FrC::sub(r, x, y); memcpy(o, r.l, 36); }
// points: acc = 4 x 14 limbs (x, y, zz, zzz)
void shim_g1_madd(uint32_t* acc, const uint32_t* x, const uint32_t* y) { G1X a; memcpy(&a, acc, 224); fp_t px, py; Fp::set(px, x); Fp::set(py, y); g1x_madd(a, px, py); memcpy(acc, &a, 224); }
void shim_g1_add(const uint32_t* a, const uint32_t* b, uint32_t* o) { G1X x, y, r; memcpy(&x, a, 224); memcpy(&y, b, 224); g1x_add(r, x, y); memcpy(o, &r, 224); }
// the same through the accessor form (g1.hpp add_from: the second operand fetched coordinate by coordinate, as k_vmsm_window reads its LDS snapshot)
struct XyzzRef { const G1X* p; void get(fp_t& v, int c) const { v = c == 0 ? p->x : c == 1 ? p->y : c == 2 ? p->zz : p->zzz; } };
void shim_g1_add_from(const uint32_t* a, const uint32_t* b, uint32_t* o) { G1X x, y; memcpy(&x, a, 224); memcpy(&y, b, 224); XyzzRef r{&y}; G1D::add_from(x, r); memcpy(o, &x, 224); }
// [k]P by the co-Z table / mixed-addition chain of the FK20 transform stage (g1.hpp mul_wnaf_coz); row: 66 words of digit codes
void shim_g1_mul_wnaf_coz(const uint32_t* a, const uint32_t* row66, uint32_t* o) {
    G1X x, r; memcpy(&x, a, 224); fp_t beta; Fp::set(beta, FP_BETA); G1D::mul_wnaf_coz(r, x, row66, beta); memcpy(o, &r, 224);
}
void shim_g1_dbl(const uint32_t* a, uint32_t* o) { G1X x, r; memcpy(&x, a, 224); g1x_dbl(r, x); memcpy(o, &r, 224); }
void shim_g1_set_inf(uint32_t* o) { G1X r; g1x_set_inf(r); memcpy(o, &r, 224); }
// Jacobian accumulators: 3 x 14 limbs (x, y, z)
void shim_g1j_from_xyzz(const uint32_t* a, uint32_t* o) { G1X x; memcpy(&x, a, 224); G1J r; G1D::jac_from_xyzz(r, x); memcpy(o, &r, 168); }
void shim_g1j_to_xyzz(const uint32_t* a, uint32_t* o) { G1J x; memcpy(&x, a, 168); G1X r; G1D::jac_to_xyzz(r, x); memcpy(o, &r, 224); }
void shim_g1j_set_inf(uint32_t* o) { G1J r; G1D::set_inf(r); memcpy(o, &r, 168); }
void shim_g1j_dbl(const uint32_t* a, uint32_t* o) { G1J x, r; memcpy(&x, a, 168); G1D::jdbl(r, x); memcpy(o, &r, 168); }
void shim_g1j_add(const uint32_t* a, const uint32_t* b, uint32_t* o) {   // b: another Jacobian accumulator
    G1J x, y, r; memcpy(&x, a, 168); memcpy(&y, b, 168); G1JC c; G1D::jac_cache(c, y); G1D::jadd(r, x, c); memcpy(o, &r, 168);
}
void shim_g1j_add_affine(const uint32_t* a, const uint32_t* px, const uint32_t* py, uint32_t* o) {
    G1J x, r; memcpy(&x, a, 168); fp_t fx, fy; Fp::set(fx, px); Fp::set(fy, py); G1JC c; G1D::jac_cache_affine(c, fx, fy); G1D::jadd(r, x, c); memcpy(o, &r, 168);
}
// r-torsion test of the affine point (x, y) (Montgomery limbs): bit 0 = with y, bit 1 = from x and y^2 alone, bit 2 = the register-lean
// form of the latter (isomorphic curves, incomplete additions, stash)
int shim_g1_in_subgroup(const uint32_t* x, const uint32_t* y) {
    fp_t fx, fy, beta, rhs;
    Fp::set(fx, x); Fp::set(fy, y); Fp::set(beta, FP_BETA);
    Fp::sqr(rhs, fy);
    G1D::MemberStash st;
    return (G1D::in_subgroup(fx, fy, beta) ? 1 : 0) | (G1D::in_subgroup_yfree(fx, rhs, beta) ? 2 : 0) | (G1D::in_subgroup_yfree_lean(fx, rhs, beta, st) ? 4 : 0);
}
// returns compressed 48 bytes
void shim_g1_compress(const uint32_t* a, uint8_t* out48) {
    G1X x; memcpy(&x, a, 224); G1Aff aff; g1x_to_affine(aff, x);
    uint32_t w[12]; g1_compress_words(w, aff);
    for (int i = 0; i < 12; i++) { out48[4 * i] = w[i] >> 24; out48[4 * i + 1] = w[i] >> 16; out48[4 * i + 2] = w[i] >> 8; out48[4 * i + 3] = w[i]; }
}
}
