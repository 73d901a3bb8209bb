// tests/host_shim/pairing_shim.cpp -- exposes the product's HOST pairing / G1 code
// (go-eth-kzg_amd/csrc/host_pairing.hpp, host_g1.hpp) to Python for CPU-side unit tests.
#include "../../go-eth-kzg_amd/csrc/host_g1.hpp"
#include "../../go-eth-kzg_amd/csrc/host_validate.hpp"
using namespace kzg;
extern "C" {
// prod e(P_i, Q_i) == 1 ?  P: n x 48 B compressed, Q: n x 96 B compressed.  -1 on decode error.
int shim_pairing_check(const uint8_t* p48, const uint8_t* q96, int n) {
    std::vector<G1AffH> ps(n);
    std::vector<G2Prepared> prep(n);
    std::vector<const G2Prepared*> qp(n);
    for (int i = 0; i < n; i++) {
        if (!g1h_decompress(ps[i], p48 + 48 * i)) return -1;
        G2Aff q;
        if (!g2_decompress(q, q96 + 96 * i)) return -2;
        g2_prepare(prep[i], q);
        qp[i] = &prep[i];
    }
    if (n == 2) {                    // the two-thread variant must agree with the shared-squaring loop
        const bool a = pairing_check_prepared(ps.data(), qp.data(), 2), b = pairing_check_prepared_2mt(ps.data(), qp.data());
        if (a != b) return -3;
        static HostPool pool(6);     // the segmented loops (two and three segments per loop on pool threads) must agree as well
        if (pairing_check_prepared_2mt(ps.data(), qp.data(), &pool, 1) != a || pairing_check_prepared_2mt(ps.data(), qp.data(), &pool, 2) != a ||
            pairing_check_prepared_2mt(ps.data(), qp.data(), &pool, 3) != a)
            return -4;
        return a ? 1 : 0;
    }
    return pairing_check_prepared(ps.data(), qp.data(), n) ? 1 : 0;
}
// miller_segment_prepared: the product of the segments of a loop, for several cuts (every bit position of |x| that is set lies inside,
// at the start and at the end of some segment), equals miller_loop_prepared's value exactly.  Returns the number of mismatches.
int shim_miller_segments_check(const uint8_t* p48, const uint8_t* q96) {
    G1AffH P;
    G2Aff Q;
    if (!g1h_decompress(P, p48) || !g2_decompress(Q, q96)) return -1;
    G2Prepared pr;
    g2_prepare(pr, Q);
    Fp12 whole;
    miller_loop_prepared(whole, P, pr);
    auto same = [](const Fp12& a, const Fp12& b) { return fp6::eq(a.c0, b.c0) && fp6::eq(a.c1, b.c1); };
    static const int cuts[][5] = {{62, -1, -1, -1, -1}, {62, 40, -1, -1, -1}, {62, 53, 36, -1, -1}, {62, 61, 48, 16, -1}, {62, 60, 57, 47, 15}, {62, 0, -1, -1, -1}};
    int bad = 0;
    for (const auto& cut : cuts) {
        Fp12 acc, seg;
        fp12::one(acc);
        for (int j = 0; j < 5 && (j == 0 || cut[j] >= 0); j++) {
            const int hi = cut[j], lo = (j + 1 < 5 && cut[j + 1] >= 0) ? cut[j + 1] + 1 : 0;
            miller_segment_prepared(seg, P, pr, hi, lo);
            fp12::mul(acc, acc, seg);
        }
        if (!same(acc, whole)) bad++;
    }
    return bad;
}
// The AVX-512 IFMA squaring chain (host_cyclo_ifma.hpp) against the scalar tower on elements of the cyclotomic subgroup: 1, 2, 3, 16, 63
// squarings in a row (every squaring's output is the next one's loose input) and the whole a^|x|.  Returns the number of mismatches,
// -1 on a decoding error, -2 when the CPU has no AVX-512 IFMA (nothing to compare).
int shim_ifma_chain_check(const uint8_t* p48, const uint8_t* q96) {
#if defined(__x86_64__)
    if (!ifma::available()) return -2;
    G1AffH P;
    G2Aff Q;
    if (!g1h_decompress(P, p48) || !g2_decompress(Q, q96)) return -1;
    G2Prepared pr;
    g2_prepare(pr, Q);
    Fp12 f, t, g;
    miller_loop_prepared(f, P, pr);
    fp12::conj(t, f);
    fp12::inv(g, f);
    fp12::mul(f, t, g);
    frobenius(t, f);
    frobenius(t, t);
    fp12::mul(f, t, f);                       // cyclotomic now
    int bad = 0;
    auto same = [](const Fp12& a, const Fp12& b) { return fp6::eq(a.c0, b.c0) && fp6::eq(a.c1, b.c1); };
    for (int round = 0; round < 6; round++) {
        for (int n : {1, 2, 3, 16, 63}) {
            Fp12 a = f, b;
            for (int i = 0; i < n; i++) cyclotomic_sqr(a, a);
            ifma::cyclo_sqr_n(b, f, n);
            if (!same(a, b)) bad++;
        }
        Fp12 a, b;
        exp_by_x_abs_scalar(a, f);
        ifma::exp_by_x_abs(b, f);
        if (!same(a, b)) bad += 100;
        cyclotomic_sqr(t, f);
        fp12::mul(f, t, f);                   // another element of the subgroup: f^3
    }
    // the identity and an element with zero coefficients
    Fp12 one, r1;
    fp12::one(one);
    ifma::cyclo_sqr_n(r1, one, 5);
    if (!fp12::is_one(r1)) bad += 1000;
    return bad;
#else
    return -2;
#endif
}
// cyclotomic_sqr against the general squaring on an element of the cyclotomic subgroup (f^((p^6-1)(p^2+1)) of a Miller-loop
// value) and on its powers; returns the number of mismatches.  Also checks that it is NOT a squaring outside the subgroup
// (so the test would notice if the check were vacuous).
int shim_cyclotomic_sqr_check(const uint8_t* p48, const uint8_t* q96) {
    G1AffH P;
    G2Aff Q;
    if (!g1h_decompress(P, p48) || !g2_decompress(Q, q96)) return -1;
    G2Prepared pr;
    g2_prepare(pr, Q);
    Fp12 f;
    fp12::one(f);
    size_t idx = 0;
    for (int i = 62; i >= 0; i--) {
        fp12::sqr(f, f);
        mul_by_line(f, P.y, P.x, pr.lines[idx++]);
        if ((BLS_X_ABS >> i) & 1) mul_by_line(f, P.y, P.x, pr.lines[idx++]);
    }
    Fp12 raw = f, t, g;
    fp12::conj(t, f);
    fp12::inv(g, f);
    fp12::mul(f, t, g);
    frobenius(t, f);
    frobenius(t, t);
    fp12::mul(f, t, f);                       // cyclotomic now
    int bad = 0;
    for (int k = 0; k < 20; k++) {
        Fp12 a, b;
        fp12::sqr(a, f);
        cyclotomic_sqr(b, f);
        if (!(fp6::eq(a.c0, b.c0) && fp6::eq(a.c1, b.c1))) bad++;
        fp12::mul(f, a, raw.c0.c0.c0.l[0] & 1 ? f : a);      // move to another element of the subgroup
        fp12::conj(t, f);
        fp12::mul(t, t, f);                   // x * conj(x) = 1 inside the subgroup
        if (!fp12::is_one(t)) bad += 100;
    }
    Fp12 a, b;
    fp12::sqr(a, raw);
    cyclotomic_sqr(b, raw);
    if (fp6::eq(a.c0, b.c0) && fp6::eq(a.c1, b.c1)) bad += 1000;   // must differ outside the subgroup
    return bad;
}
// HostPool: every index of every parallel() call runs exactly once, also with several submitting threads and more tasks than
// workers; returns the number of mistakes
int shim_pool_selftest() {
    HostPool pool(3);
    std::atomic<int> bad(0);
    auto submitter = [&](int seed) {
        for (int rep = 0; rep < 300; rep++) {
            const int k = 1 + (rep + seed) % 7;
            std::vector<int> hits(k, 0);
            pool.parallel(k, [&](int i) { hits[i] += 1 + i; });
            for (int i = 0; i < k; i++) if (hits[i] != 1 + i) bad++;
        }
    };
    std::thread a(submitter, 0), b(submitter, 3);
    submitter(5);
    a.join();
    b.join();
    HostPool none(0);                      // no workers: runs inline
    int inl = 0;
    none.parallel(4, [&](int i) { inl += i; });
    if (inl != 6) bad++;
    return bad.load();
}
// [k]P for compressed P, k = 32-byte little-endian words; returns affine x,y as 2x48 big-endian + inf flag
int shim_g1_mul(const uint8_t* p48, const uint64_t* k, uint8_t* out_xy96) {
    G1AffH a;
    if (!g1h_decompress(a, p48)) return -1;
    G1HX p, r;
    if (a.inf) G1H::set_inf(p); else G1H::from_affine(p, a.x, a.y);
    g1h_mul(r, p, k);
    G1AffH o;
    g1h_to_affine(o, r);
    if (o.inf) return 1;
    uint64_t v[6];
    HostFp::to_plain(v, o.x);
    for (int i = 0; i < 6; i++) for (int j = 0; j < 8; j++) out_xy96[(5 - i) * 8 + j] = (uint8_t)(v[i] >> (56 - 8 * j));
    HostFp::to_plain(v, o.y);
    for (int i = 0; i < 6; i++) for (int j = 0; j < 8; j++) out_xy96[48 + (5 - i) * 8 + j] = (uint8_t)(v[i] >> (56 - 8 * j));
    return 0;
}

// the ADX form of HostFp::mul against the portable loop on n pseudo-random canonical pairs plus the edge values; returns the number of
// mismatches, -1 if this CPU has no BMI2 + ADX (nothing to compare)
// HostFp::inv (divsteps through the 29-bit-limb field) against a^(p-2), and a * a^-1 == 1; returns the number of mismatches
int shim_hostfp_inv_check(int n, uint64_t seed) {
    int bad = 0;
    HostFp::El a, i1, i2, one, t;
    HostFp::set_one(one);
    uint64_t s = seed;
    auto next = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int k = 0; k < n; k++) {
        uint64_t w[6];
        for (int q = 0; q < 6; q++) w[q] = next();
        w[5] &= 0x0fffffffffffffffull;                       // < 2^380 < p
        if (k == 0) memset(w, 0, 48);                          // 0 -> 0 in both forms
        if (k == 1) { memset(w, 0, 48); w[0] = 1; }
        HostFp::from_plain(a, w);
        HostFp::inv(i1, a);
        HostFp::inv_fermat(i2, a);
        if (!HostFp::limbs_equal(i1, i2)) bad++;
        if (k > 0) { HostFp::mul(t, a, i1); if (!HostFp::limbs_equal(t, one)) bad++; }
    }
    return bad;
}
int shim_hostfp_mul_adx_check(int n, uint64_t seed) {
#if defined(__x86_64__) && !defined(MKZG_NO_ADX)
    if (!HostFp::have_adx()) return -1;
    auto next = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
    std::vector<HostFp::El> v;
    HostFp::El e;
    HostFp::set_zero(e); v.push_back(e);
    e.l[0] = 1; v.push_back(e);
    HostFp::set_one(e); v.push_back(e);
    for (int k = 1; k <= 3; k++) {                       // p - k
        memcpy(e.l, HostFpParams::MOD, 48);
        e.l[0] -= (uint64_t)k;
        v.push_back(e);
    }
    for (int i = 0; i < 6; i++) e.l[i] = ~0ull;          // (2^384 - 1) mod p: at most nine subtractions
    while (HostFp::geq_mod(e.l)) HostFp::sub_raw(e.l, e.l, HostFpParams::MOD);
    v.push_back(e);
    for (int i = 0; i < n; i++) {
        for (int q = 0; q < 6; q++) e.l[q] = next();
        e.l[5] &= 0x0fffffffffffffffull;                  // < 2^380 < p
        v.push_back(e);
    }
    int bad = 0;
    for (size_t i = 0; i < v.size(); i++)
        for (size_t j = 0; j < (i < 8 ? v.size() : (size_t)16); j++) {
            HostFp::El x, y;
            HostFp::mul_adx(x, v[i], v[j]);
            HostFp::mul_portable(y, v[i], v[j]);
            if (!HostFp::limbs_equal(x, y) || HostFp::geq_mod(x.l)) bad++;
            HostFp::mul_adx(x, v[i], v[i]);              // in-place aliasing as sqr() uses it
            HostFp::El z = v[i];
            HostFp::mul_adx(z, z, z);
            if (!HostFp::limbs_equal(x, z)) bad++;
        }
    return bad;
#else
    (void)n; (void)seed;
    return -1;
#endif
}
// host_validate.hpp: status of one compressed point and its affine record (32 words) as the kernels consume it
int shim_g1_validate_row(const uint8_t* in48, uint32_t* row32) { return g1h_validate_row(in48, row32); }
// the record back to compressed bytes through the device's limb templates compiled for the host (round trip of the conversion)
void shim_row_compress(const uint32_t* row32, uint8_t* out48) {
    G1Aff a;
    for (int i = 0; i < 14; i++) { a.x.l[i] = row32[i]; a.y.l[i] = row32[14 + i]; }
    a.inf = row32[28];
    uint32_t w12[12];
    g1_compress_words(w12, a);
    for (int i = 0; i < 12; i++) { const uint32_t v = __builtin_bswap32(w12[i]); memcpy(out48 + 4 * i, &v, 4); }
}
}
