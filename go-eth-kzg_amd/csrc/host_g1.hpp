// host_g1.hpp -- product host code: G1 helpers over HostFp for the verifiers' finishing steps
// (window-sum Horner combination, [e]G, point negation/addition, affine conversion) -- the
// handful of group operations per *batch* that reference internal/kzg/kzg_verify.go:160-188
// performs after its MSMs.  Shares the group law with the device through g1.hpp.
#pragma once
#include <vector>
#include "g1.hpp"
#include "host_pairing.hpp"

namespace kzg {

using G1H = G1T<HostFp>;
using G1HX = G1H::XYZZ;

// device export format: 4 coordinates (X, Y, ZZ, ZZZ) as canonical plain integers, 12 LE 32-bit words each
inline void g1h_from_plain_words(G1HX& r, const uint32_t* w48) {
    fe* dst[4] = {&r.x, &r.y, &r.zz, &r.zzz};
    for (int c = 0; c < 4; c++) {
        uint64_t v[6];
        for (int i = 0; i < 6; i++) v[i] = (uint64_t)w48[12 * c + 2 * i] | ((uint64_t)w48[12 * c + 2 * i + 1] << 32);
        HostFp::from_plain(*dst[c], v);
    }
}
inline void g1h_generator(G1HX& r) {
    fe x, y;
    HostFp::set(x, HostFpParams::GEN_X);
    HostFp::set(y, HostFpParams::GEN_Y);
    G1H::from_affine(r, x, y);
}
// r = [k]p, k as 4 little-endian 64-bit words
inline void g1h_mul(G1HX& r, const G1HX& p, const uint64_t k[4]) {
    G1HX acc;
    G1H::set_inf(acc);
    for (int i = 255; i >= 0; i--) {
        G1H::dbl(acc, acc);
        if ((k[i >> 6] >> (i & 63)) & 1) G1H::add(acc, acc, p);
    }
    r = acc;
}
inline void g1h_to_affine(G1AffH& r, const G1HX& p) {
    G1H::Aff a;
    G1H::to_affine(a, p);
    r.x = a.x;
    r.y = a.y;
    r.inf = a.inf != 0;
}
// sum_w 2^(c w) S_w  (Horner from the top window)
inline void g1h_horner(G1HX& r, const G1HX* sums, int nwin, int c);
// sum_w 2^(c w) W_w over the device's window sums (nwin x 48 words: X, Y, ZZ, ZZZ as canonical plain integers; ZZ == 0 marks the identity)
inline void g1h_horner_words(G1HX& r, const uint32_t* wsum, int nwin, int c) {
    std::vector<G1HX> win((size_t)nwin);
    for (int w = 0; w < nwin; w++) {
        const uint32_t* p = wsum + (size_t)w * 48;
        bool zero = true;
        for (int q = 24; q < 36; q++) zero = zero && p[q] == 0;
        if (zero) G1H::set_inf(win[(size_t)w]); else g1h_from_plain_words(win[(size_t)w], p);
    }
    g1h_horner(r, win.data(), nwin, c);
}
// fixed-base multiplication for the generator: 32 byte-windows x 255 multiples (XYZZ, built once per context):
// [k]G is 32 additions instead of 256 doublings + ~128 additions
struct G1HFixedBase {
    std::vector<G1HX> t;
    bool ready = false;
};
inline void g1h_fixed_build(G1HFixedBase& fb, const G1HX& g) {
    fb.t.resize(32 * 255);
    G1HX base = g;
    for (int w = 0; w < 32; w++) {
        G1HX run = base;
        for (int d = 1; d <= 255; d++) {
            fb.t[(size_t)w * 255 + d - 1] = run;
            G1H::add(run, run, base);
        }
        base = run;                                  // 256 * base
    }
    fb.ready = true;
}
inline void g1h_fixed_mul(G1HX& r, const G1HFixedBase& fb, const uint64_t k[4]) {
    G1HX acc;
    G1H::set_inf(acc);
    for (int w = 0; w < 32; w++) {
        const unsigned d = (unsigned)(k[w >> 3] >> (8 * (w & 7))) & 0xffu;
        if (d) G1H::add(acc, acc, fb.t[(size_t)w * 255 + d - 1]);
    }
    r = acc;
}
inline void g1h_horner(G1HX& r, const G1HX* sums, int nwin, int c) {
    G1HX acc;
    G1H::set_inf(acc);
    for (int w = nwin - 1; w >= 0; w--) {
        for (int q = 0; q < c; q++) G1H::dbl(acc, acc);
        G1H::add(acc, acc, sums[w]);
    }
    r = acc;
}
// compressed 48 bytes -> affine, curve check only (test helper and G1 inputs of host-only paths)
inline bool g1h_decompress(G1AffH& o, const uint8_t in[48]) {
    uint8_t flags = in[0] & 0xe0;
    if (!(flags & 0x80) || flags == 0xe0) return false;
    if (flags & 0x40) {
        if (in[0] & 0x1f) return false;
        for (int i = 1; i < 48; i++) if (in[i]) return false;
        HostFp::set_zero(o.x);
        HostFp::set_zero(o.y);
        o.inf = true;
        return true;
    }
    uint8_t xb[48];
    memcpy(xb, in, 48);
    xb[0] &= 0x1f;
    fe x, y, rhs, four;
    if (!HostFp::from_be48(x, xb)) return false;
    HostFp::sqr(rhs, x);
    HostFp::mul(rhs, rhs, x);
    HostFp::set_one(four);
    HostFp::dbl(four, four);
    HostFp::dbl(four, four);
    HostFp::add(rhs, rhs, four);
    if (!HostFp::sqrt(y, rhs)) return false;
    if (HostFp::lex_largest(y) != ((flags & 0x20) != 0)) HostFp::neg<0>(y, y);
    o.x = x;
    o.y = y;
    o.inf = false;
    return true;
}

// affine -> zcash compressed 48 bytes
inline void g1h_compress(uint8_t out[48], const G1AffH& a) {
    if (a.inf) { memset(out, 0, 48); out[0] = 0xc0; return; }
    uint64_t v[6];
    HostFp::to_plain(v, a.x);
    for (int i = 0; i < 6; i++) for (int j = 0; j < 8; j++) out[(5 - i) * 8 + j] = (uint8_t)(v[i] >> (56 - 8 * j));
    out[0] |= 0x80;
    if (HostFp::lex_largest(a.y)) out[0] |= 0x20;
}
// (a + b) mod r on 8-word little-endian plain scalars
inline void fr_words_add(uint32_t out[8], const uint32_t a[8], const uint32_t b[8]) {
    uint64_t c = 0;
    uint32_t t[8];
    for (int i = 0; i < 8; i++) { uint64_t s = (uint64_t)a[i] + b[i] + c; t[i] = (uint32_t)s; c = s >> 32; }
    bool geq = c != 0;
    if (!geq) {
        geq = true;
        for (int i = 7; i >= 0; i--) {
            if (t[i] > FrParams::MOD32[i]) break;
            if (t[i] < FrParams::MOD32[i]) { geq = false; break; }
        }
    }
    if (geq) {
        uint64_t br = 0;
        for (int i = 0; i < 8; i++) { uint64_t s = (uint64_t)t[i] - FrParams::MOD32[i] - br; t[i] = (uint32_t)s; br = (s >> 32) & 1; }
    }
    memcpy(out, t, 32);
}

}  // namespace kzg
