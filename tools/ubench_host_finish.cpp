// tools/ubench_host_finish.cpp -- what the verifiers' host finish (DESIGN.md "what runs where") spends its ~0.9 ms on: the product's
// own host code timed piece by piece on the CPU it runs on.   g++ -O3 -std=c++17 -pthread tools/ubench_host_finish.cpp -o /tmp/ubench_host_finish
#include <chrono>
#include <cstdio>
#include "../go-eth-kzg_amd/csrc/host_g1.hpp"
using namespace kzg;
template <class F>
static double time_us(int reps, F f) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) f();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}
int main() {
    // a G1 point and a G2 point: the generators, from their standard compressed encodings
    static const uint8_t g1c[48] = {0x97, 0xf1, 0xd3, 0xa7, 0x31, 0x97, 0xd7, 0x94, 0x26, 0x95, 0x63, 0x8c, 0x4f, 0xa9, 0xac, 0x0f, 0xc3, 0x68, 0x8c, 0x4f, 0x97, 0x74, 0xb9, 0x05,
                                    0xa1, 0x4e, 0x3a, 0x3f, 0x17, 0x1b, 0xac, 0x58, 0x6c, 0x55, 0xe8, 0x3f, 0xf9, 0x7a, 0x1a, 0xef, 0xfb, 0x3a, 0xf0, 0x0a, 0xdb, 0x22, 0xc6, 0xbb};
    static const uint8_t g2c[96] = {0x93, 0xe0, 0x2b, 0x60, 0x52, 0x71, 0x9f, 0x60, 0x7d, 0xac, 0xd3, 0xa0, 0x88, 0x27, 0x4f, 0x65, 0x59, 0x6b, 0xd0, 0xd0, 0x99, 0x20, 0xb6, 0x1a,
                                    0xb5, 0xda, 0x61, 0xbb, 0xdc, 0x7f, 0x50, 0x49, 0x33, 0x4c, 0xf1, 0x12, 0x13, 0x94, 0x5d, 0x57, 0xe5, 0xac, 0x7d, 0x05, 0x5d, 0x04, 0x2b, 0x7e,
                                    0x02, 0x4a, 0xa2, 0xb2, 0xf0, 0x8f, 0x0a, 0x91, 0x26, 0x08, 0x05, 0x27, 0x2d, 0xc5, 0x10, 0x51, 0xc6, 0xe4, 0x7a, 0xd4, 0xfa, 0x40, 0x3b, 0x02,
                                    0xb4, 0x51, 0x0b, 0x64, 0x7a, 0xe3, 0xd1, 0x77, 0x0b, 0xac, 0x03, 0x26, 0xa8, 0x05, 0xbb, 0xef, 0xd4, 0x80, 0x56, 0xc8, 0xc1, 0x21, 0xbd, 0xb8};
    G1AffH P;
    G2Aff Q;
    if (!g1h_decompress(P, g1c) || !g2_decompress(Q, g2c)) { printf("decode failed\n"); return 1; }
    G2Prepared prep;
    g2_prepare(prep, Q);
    HostFp::El a = P.x, b = P.y, r;
    volatile uint64_t sink = 0;
    printf("Fp mul            %8.1f ns\n", 1e3 * time_us(2000000, [&] { HostFp::mul(r, a, b); a = r; }));
    sink += a.l[0];
    printf("Fp inv (divsteps) %8.2f us\n", time_us(20000, [&] { HostFp::inv(r, a); a = r; }));
    printf("Fp inv (Fermat)   %8.2f us\n", time_us(2000, [&] { HostFp::inv_fermat(r, a); a = r; }));
    sink += a.l[0];
    Fp12 f, g;
    miller_loop_prepared(f, P, prep);
    printf("Miller loop       %8.1f us\n", time_us(200, [&] { miller_loop_prepared(g, P, prep); }));
    printf("Fp12 mul          %8.2f us\n", time_us(20000, [&] { fp12::mul(g, g, f); }));
    printf("Fp12 sqr          %8.2f us\n", time_us(20000, [&] { fp12::sqr(g, g); }));
    Fp12 e;
    final_exponentiation(e, f);
    printf("final exp         %8.1f us\n", time_us(200, [&] { final_exponentiation(g, f); }));
    printf("cyclotomic sqr    %8.2f us\n", time_us(20000, [&] { cyclotomic_sqr(e, e); }));
    printf("exp_by_x          %8.1f us\n", time_us(500, [&] { exp_by_x(g, e); }));
    printf("Fp12 inv          %8.2f us\n", time_us(5000, [&] { fp12::inv(g, e); }));
    const G2Prepared* qs[2] = {&prep, &prep};
    G1AffH ps[2] = {P, P};
    printf("2-pairing check   %8.1f us (one thread)\n", time_us(100, [&] { sink += pairing_check_prepared(ps, qs, 2); }));
    HostPool pool(4);
    printf("2-pairing check   %8.1f us (Miller loops on two pool threads)\n", time_us(100, [&] { sink += pairing_check_prepared_2mt(ps, qs, &pool); }));
    {
        HostPool pool8(8);
        printf("2-pairing check   %8.1f us (every loop in two segments, four threads)\n", time_us(100, [&] { sink += pairing_check_prepared_2mt(ps, qs, &pool8, 2); }));
        printf("2-pairing check   %8.1f us (every loop in three segments, six threads)\n", time_us(100, [&] { sink += pairing_check_prepared_2mt(ps, qs, &pool8, 3); }));
    }
    // G1 side: Horner over 32 window sums of 8 bits, the fixed-base multiplication, the conversion to affine
    uint32_t wsum[32 * 48];
    {
        G1HX x;
        G1H::from_affine(x, P.x, P.y);
        for (int w = 0; w < 32; w++) {
            G1AffH af;
            g1h_to_affine(af, x);
            uint64_t xp[6], yp[6];
            HostFp::to_plain(xp, af.x);
            HostFp::to_plain(yp, af.y);
            for (int q = 0; q < 12; q++) {
                wsum[w * 48 + q] = (uint32_t)(xp[q / 2] >> (32 * (q & 1)));
                wsum[w * 48 + 12 + q] = (uint32_t)(yp[q / 2] >> (32 * (q & 1)));
            }
            for (int q = 24; q < 48; q++) wsum[w * 48 + q] = 0;
            G1H::dbl(x, x);
        }
    }
    G1HX h;
    printf("Horner 32 x 8 bit %8.1f us\n", time_us(300, [&] { g1h_horner_words(h, wsum, 32, 8); }));
    G1AffH af;
    printf("to affine         %8.2f us\n", time_us(3000, [&] { g1h_to_affine(af, h); }));
    printf("pool.parallel(3) of empty tasks %6.2f us\n", time_us(20000, [&] { pool.parallel(3, [&](int) {}); }));
    return (int)(sink & 1) * 0;
}
