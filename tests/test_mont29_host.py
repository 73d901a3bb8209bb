"""Host-side unit tests of the product's radix-2^29 field / G1 templates
(go-eth-kzg_amd/csrc/mont29.hpp, g1.hpp) against Python big-integer arithmetic.

The kernels instantiate exactly these templates on the device; testing them on the CPU (g++)
catches arithmetic and bound mistakes without a GPU.  The G1 results are also compared with
the CPU oracle (compressed encodings are canonical, so any correct implementation agrees).
"""
import ctypes as C
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_DIR = os.path.join(ROOT, "tests", "host_shim")
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
RP = 1 << (29 * 14)
RF = 1 << (29 * 9)
G1X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1


@pytest.fixture(scope="module")
def shim():
    from conftest import build_host_shim
    return C.CDLL(build_host_shim("mont29", ("mont29.hpp", "g1.hpp", "constants29.h", "modinv30.hpp")))


def limbs(v, n=14):
    assert 0 <= v < 1 << (29 * n)
    return (C.c_uint32 * n)(*[(v >> (29 * i)) & ((1 << 29) - 1) for i in range(n)])


def val(arr):
    return sum(int(x) << (29 * i) for i, x in enumerate(arr))


def out(n=14):
    return (C.c_uint32 * n)()


def normalized(arr):
    return all(int(x) < (1 << 29) for x in arr[:-1])


def test_fp_mul_add_sub(shim):
    rng = random.Random(1)
    for it in range(300):
        # loose inputs up to 2^392
        a = rng.randrange(0, 1 << rng.choice([381, 385, 390, 392]))
        b = rng.randrange(0, 1 << rng.choice([381, 385, 390, 392]))
        o = out()
        shim.shim_fp_mul(limbs(a), limbs(b), o)
        v = val(o)
        assert normalized(o)
        assert (v * RP - a * b) % P == 0
        assert v < a * b // RP + P + 1
        shim.shim_fp_add(limbs(a), limbs(b), o)
        assert val(o) == a + b and normalized(o)
        for klog in range(5):
            bb = rng.randrange(0, (1 << klog) * P + 1)
            shim.shim_fp_sub(klog, limbs(a), limbs(bb), o)
            assert val(o) == a - bb + (1 << klog) * P and normalized(o)


def test_fp_sqr_mul2(shim):
    """dedicated squaring and the fused two-product multiply (one Montgomery reduction), loose inputs, worst-case limbs"""
    rng = random.Random(11)
    allones = (1 << 392) - 1                                       # every limb 2^29 - 1: the column-sum worst case
    cases = [(allones, allones, allones, allones)]
    cases += [tuple(rng.randrange(0, 1 << rng.choice([381, 385, 390, 392])) for _ in range(4)) for _ in range(300)]
    for a, b, c, d in cases:
        o = out()
        shim.shim_fp_sqr(limbs(a), o)
        v = val(o)
        assert normalized(o) and (v * RP - a * a) % P == 0 and v < a * a // RP + P + 1
        shim.shim_fp_mul2(limbs(a), limbs(b), limbs(c), limbs(d), o)
        v = val(o)
        assert normalized(o) and (v * RP - a * b - c * d) % P == 0 and v < (a * b + c * d) // RP + P + 1
    for a in [0, 1, R - 1] + [rng.randrange(0, R) for _ in range(100)]:
        o = out(9)
        shim.shim_fr_sqr(limbs(a, 9), o)
        assert val(o) % R == a * a * pow(RF, -1, R) % R


def test_fp_fused_linear_forms(shim):
    """a - b - 2c + 4p, a - 2c + 4p and 3a: exact integer results, normalized limbs"""
    rng = random.Random(12)
    for it in range(300):
        a = rng.randrange(0, 1 << rng.choice([381, 383, 385]))
        b = rng.randrange(0, P + P // 100)
        c = rng.randrange(0, P + P // 100)
        o = out()
        shim.shim_fp_sub_b_2c(limbs(a), limbs(b), limbs(c), o)
        assert val(o) == a - b - 2 * c + 4 * P and normalized(o)
        shim.shim_fp_sub_2c(limbs(a), limbs(c), o)
        assert val(o) == a - 2 * c + 4 * P and normalized(o)
        shim.shim_fp_triple(limbs(a), o)
        assert val(o) == 3 * a and normalized(o)


def test_fp_canon_plain_inv(shim):
    rng = random.Random(2)
    for it in range(100):
        a = rng.randrange(0, 1 << 392)
        o = out()
        shim.shim_fp_canon(limbs(a), o)
        assert val(o) == a % P
        shim.shim_fp_to_plain(limbs(a), o)
        assert val(o) == a * pow(RP, -1, P) % P
        x = rng.randrange(0, P)
        shim.shim_fp_from_plain(limbs(x), o)
        assert val(o) % P == x * RP % P
    for k in range(16):
        assert shim.shim_fp_is_zero16(limbs(k * P)) == 1
        assert shim.shim_fp_is_zero16(limbs(k * P + 1)) == 0
    assert shim.shim_fp_is_zero16(limbs(5 * P + (1 << 29) * 12345)) == 0
    for it in range(5):
        a = rng.randrange(1, P)
        o = out()
        shim.shim_fp_inv(limbs(a * RP % P), o)
        assert val(o) % P == pow(a, -1, P) * RP % P


def test_divsteps_inversion(shim):
    """modinv30.hpp (Bernstein-Yang divsteps) against pow(a, -1, M), Fp and Fr, loose inputs included"""
    rng = random.Random(9)
    vals = [1, 2, P - 1, P - 2, (P + 1) // 2, 3 * P + 5] + [rng.randrange(1, 1 << 392) for _ in range(200)]
    for a in vals:
        o = out()
        shim.shim_fp_inv_fast(limbs(a), o)
        x = a * pow(RP, -1, P) % P
        if x == 0:
            assert val(o) % P == 0
        else:
            assert val(o) % P == pow(x, -1, P) * RP % P, hex(a)
            assert val(o) < 2 * P
    o = out()
    shim.shim_fp_inv_fast(limbs(0), o)
    assert val(o) % P == 0
    shim.shim_fp_inv_fast(limbs(P), o)
    assert val(o) % P == 0
    for a in [1, 2, R - 1, (R + 1) // 2] + [rng.randrange(1, R) for _ in range(200)]:
        o = out(9)
        shim.shim_fr_inv_fast(limbs(a, 9), o)
        x = a * pow(RF, -1, R) % R
        assert val(o) % R == pow(x, -1, R) * RF % R


def test_fp_words(shim):
    rng = random.Random(3)
    for it in range(50):
        a = rng.randrange(0, P)
        w = (C.c_uint32 * 12)(*[(a >> (32 * i)) & 0xFFFFFFFF for i in range(12)])
        o, back = out(), (C.c_uint32 * 12)()
        shim.shim_fp_words(w, o, back)
        assert val(o) == a
        assert list(back) == list(w)


def test_fr_ops(shim):
    rng = random.Random(4)
    for it in range(300):
        a, b = rng.randrange(0, R), rng.randrange(0, R)
        if it == 0:
            a, b = R - 1, R - 1
        o = out(9)
        shim.shim_fr_mul(limbs(a, 9), limbs(b, 9), o)
        assert val(o) == a * b * pow(RF, -1, R) % R
        shim.shim_fr_add(limbs(a, 9), limbs(b, 9), o)
        assert val(o) == (a + b) % R
        shim.shim_fr_sub(limbs(a, 9), limbs(b, 9), o)
        assert val(o) == (a - b) % R


# ---- G1: affine big-int reference
def ec_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return x3, (lam * (x1 - x3) - y1) % P


def ec_mul(k, p):
    acc = None
    while k:
        if k & 1:
            acc = ec_add(acc, p)
        p = ec_add(p, p)
        k >>= 1
    return acc


def compress(pt):
    if pt is None:
        return bytes([0xC0] + [0] * 47)
    x, y = pt
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if y > (P - 1) // 2 else 0)
    return bytes(b)


def xyzz_from_affine(pt):
    arr = (C.c_uint32 * 56)()
    x, y = pt
    for j, v in enumerate([x * RP % P, y * RP % P, RP % P, RP % P]):
        for i in range(14):
            arr[14 * j + i] = (v >> (29 * i)) & ((1 << 29) - 1)
    return arr


def shim_compress(shim, acc):
    b = C.create_string_buffer(48)
    shim.shim_g1_compress(acc, b)
    return b.raw


def test_g1_ops(shim):
    G = (G1X, G1Y)
    rng = random.Random(5)
    pts = [ec_mul(rng.randrange(1, R), G) for _ in range(6)]
    # madd chain incl. start from infinity, doubling and cancellation
    acc = (C.c_uint32 * 56)()
    shim.shim_g1_set_inf(acc)
    ref = None
    assert shim_compress(shim, acc) == compress(None)
    seq = [pts[0], pts[1], pts[2], pts[0], pts[3]]
    for pt in seq:
        shim.shim_g1_madd(acc, limbs(pt[0] * RP % P), limbs(pt[1] * RP % P))
        ref = ec_add(ref, pt)
        assert shim_compress(shim, acc) == compress(ref)
    # P + P through madd (acc == point)
    acc = xyzz_from_affine(pts[4])
    shim.shim_g1_madd(acc, limbs(pts[4][0] * RP % P), limbs(pts[4][1] * RP % P))
    assert shim_compress(shim, acc) == compress(ec_add(pts[4], pts[4]))
    # P - P through madd
    acc = xyzz_from_affine(pts[4])
    shim.shim_g1_madd(acc, limbs(pts[4][0] * RP % P), limbs((P - pts[4][1]) * RP % P))
    assert shim_compress(shim, acc) == compress(None)
    # full add / dbl, including non-trivial ZZ on both sides
    a = xyzz_from_affine(pts[0])
    shim.shim_g1_madd(a, limbs(pts[1][0] * RP % P), limbs(pts[1][1] * RP % P))    # a = p0+p1
    b = xyzz_from_affine(pts[2])
    shim.shim_g1_madd(b, limbs(pts[3][0] * RP % P), limbs(pts[3][1] * RP % P))    # b = p2+p3
    o = (C.c_uint32 * 56)()
    shim.shim_g1_add(a, b, o)
    r_ab = ec_add(ec_add(pts[0], pts[1]), ec_add(pts[2], pts[3]))
    assert shim_compress(shim, o) == compress(r_ab)
    shim.shim_g1_add(a, a, o)                                                     # add -> doubling path
    assert shim_compress(shim, o) == compress(ec_mul(2, ec_add(pts[0], pts[1])))
    shim.shim_g1_dbl(a, o)
    assert shim_compress(shim, o) == compress(ec_mul(2, ec_add(pts[0], pts[1])))
    inf = (C.c_uint32 * 56)()
    shim.shim_g1_set_inf(inf)
    shim.shim_g1_add(a, inf, o)
    assert shim_compress(shim, o) == compress(ec_add(pts[0], pts[1]))
    shim.shim_g1_add(inf, a, o)
    assert shim_compress(shim, o) == compress(ec_add(pts[0], pts[1]))
    # the accessor form of the addition (add_from, what k_vmsm_window's suffix network runs): general, P + P, P - P, identity on
    # either side, identity + identity -- same group element as add() every time
    neg_a = (C.c_uint32 * 56)(*a)
    for i in range(14):
        neg_a[14 + i] = limbs((4 * P - sum(int(a[14 + k]) << (29 * k) for k in range(14))))[i]
    for x_, y_, want in ((a, b, r_ab), (a, a, ec_mul(2, ec_add(pts[0], pts[1]))), (a, neg_a, None), (a, inf, ec_add(pts[0], pts[1])),
                         (inf, a, ec_add(pts[0], pts[1])), (inf, inf, None), (b, a, r_ab)):
        shim.shim_g1_add_from(x_, y_, o)
        assert shim_compress(shim, o) == compress(want)
        shim.shim_g1_add(x_, y_, o)
        assert shim_compress(shim, o) == compress(want)
    # long chain: bounds stay valid over many operations
    acc = (C.c_uint32 * 56)()
    shim.shim_g1_set_inf(acc)
    ref = None
    for i in range(40):
        pt = pts[i % 6]
        shim.shim_g1_madd(acc, limbs(pt[0] * RP % P), limbs(pt[1] * RP % P))
        ref = ec_add(ref, pt)
        if i % 7 == 3:
            shim.shim_g1_dbl(acc, acc)
            ref = ec_add(ref, ref)
        if i % 5 == 4:
            (shim.shim_g1_add if i % 10 == 4 else shim.shim_g1_add_from)(acc, a, acc)
            ref = ec_add(ref, ec_add(pts[0], pts[1]))
    assert shim_compress(shim, acc) == compress(ref)


def jac_value(acc, k):
    return sum(int(acc[14 * k + i]) << (29 * i) for i in range(14))


def jac_check_bounds(acc):
    # the loose bounds g1.hpp states for a Jacobian accumulator
    assert jac_value(acc, 0) < 17.01 * P and jac_value(acc, 1) < 5.01 * P and jac_value(acc, 2) < 1.01 * P


def jac_compress(shim, j):
    x = (C.c_uint32 * 56)()
    shim.shim_g1j_to_xyzz(j, x)
    # X < 8p, Y < 4p, ZZ, ZZZ < 1.01p again
    v = [sum(int(x[14 * k + i]) << (29 * i) for i in range(14)) for k in range(4)]
    assert v[0] < 8 * P and v[1] < 4 * P and v[2] < 1.01 * P and v[3] < 1.01 * P
    return shim_compress(shim, x)


def test_g1_jacobian_ops(shim):
    """Jacobian doubling / addition against affine arithmetic in Python, every exceptional case, bounds after every step"""
    G = (G1X, G1Y)
    rng = random.Random(11)
    pts = [ec_mul(rng.randrange(1, R), G) for _ in range(5)]
    aff = lambda pt: (limbs(pt[0] * RP % P), limbs(pt[1] * RP % P))
    J = lambda: (C.c_uint32 * 42)()
    # XYZZ with a non-trivial ZZ -> Jacobian -> back
    a = xyzz_from_affine(pts[0])
    shim.shim_g1_madd(a, *aff(pts[1]))
    ja = J()
    shim.shim_g1j_from_xyzz(a, ja)
    ref_a = ec_add(pts[0], pts[1])
    assert jac_compress(shim, ja) == compress(ref_a)
    # identity handling
    inf = J()
    shim.shim_g1j_set_inf(inf)
    assert jac_compress(shim, inf) == compress(None)
    o = J()
    shim.shim_g1j_dbl(inf, o)
    assert jac_compress(shim, o) == compress(None)
    shim.shim_g1j_add(inf, ja, o)
    assert jac_compress(shim, o) == compress(ref_a)
    shim.shim_g1j_add(ja, inf, o)
    assert jac_compress(shim, o) == compress(ref_a)
    shim.shim_g1j_add(inf, inf, o)
    assert jac_compress(shim, o) == compress(None)
    shim.shim_g1j_add_affine(inf, *aff(pts[2]), o)
    assert jac_compress(shim, o) == compress(pts[2])
    # P + P and P - P through the addition
    shim.shim_g1j_add(ja, ja, o)
    assert jac_compress(shim, o) == compress(ec_mul(2, ref_a))
    shim.shim_g1j_add_affine(o, *aff(ec_mul(2, ref_a)), o)
    assert jac_compress(shim, o) == compress(ec_mul(4, ref_a))
    neg = (ref_a[0], P - ref_a[1])
    shim.shim_g1j_add_affine(ja, *aff(neg), o)
    assert jac_compress(shim, o) == compress(None)
    # long mixed chain (what a windowed multiplication does): 5 doublings, then additions of accumulators and affine points
    acc, ref = J(), None
    shim.shim_g1j_set_inf(acc)
    for i in range(60):
        for _ in range(5):
            shim.shim_g1j_dbl(acc, acc)
            ref = ec_add(ref, ref) if ref else None
            jac_check_bounds(acc)
        if i % 3 != 2:
            shim.shim_g1j_add(acc, ja, acc)
            ref = ec_add(ref, ref_a)
        else:
            shim.shim_g1j_add_affine(acc, *aff(pts[i % 5]), acc)
            ref = ec_add(ref, pts[i % 5])
        jac_check_bounds(acc)
    assert jac_compress(shim, acc) == compress(ref)
    # 126 doublings in a row (the subgroup test's chain)
    acc = J()
    shim.shim_g1j_from_xyzz(xyzz_from_affine(pts[3]), acc)
    for _ in range(126):
        shim.shim_g1j_dbl(acc, acc)
        jac_check_bounds(acc)
    assert jac_compress(shim, acc) == compress(ec_mul(1 << 126, pts[3]))


def test_g1_subgroup_tests(shim):
    """phi(P) == -[x^2]P with and without the y coordinate (g1.hpp in_subgroup / in_subgroup_yfree) on points inside and outside
    the prime-order subgroup, including the order-3 points (0, +-2) that drive the chain through its exceptional cases"""
    G = (G1X, G1Y)
    rng = random.Random(21)
    aff = lambda pt: (limbs(pt[0] * RP % P), limbs(pt[1] * RP % P))
    for _ in range(4):
        pt = ec_mul(rng.randrange(1, R), G)
        assert shim.shim_g1_in_subgroup(*aff(pt)) == 7
    outside = [(0, 2), (0, P - 2)]
    while len(outside) < 6:                                     # random points of the curve: in G1 with probability 1/cofactor
        x = rng.randrange(P)
        rhs = (x * x * x + 4) % P
        y = pow(rhs, (P + 1) // 4, P)
        if y * y % P == rhs:
            outside.append((x, y))
    for pt in outside:
        assert ec_mul(R, pt) is not None                        # not of order r
        assert shim.shim_g1_in_subgroup(*aff(pt)) == 0, pt
    # h * Q lies in G1 for any curve point Q (h = cofactor): both tests must accept it
    H_COF = 0x396c8c005555e1568c00aaab0000aaab
    q = ec_mul(H_COF, outside[3])
    assert q is not None and ec_mul(R, q) is None
    assert shim.shim_g1_in_subgroup(*aff(q)) == 7
    # points of SMALL order (h = 3 * 11^2 * 10177^2 * 859267^2 * 52437899^2): the chains of all three forms meet [k]B = +-B and
    # [k]B = identity inside their loops -- the register-lean form has no branch for these, its additions return Z = 0 (mod p)
    # and the verdict reads that as "not in the subgroup" (g1.hpp in_subgroup_yfree_lean)
    assert H_COF == 3 * 11 ** 2 * 10177 ** 2 * 859267 ** 2 * 52437899 ** 2
    small = []
    for base in outside[2:]:
        for f, e in ((3, 1), (11, 2), (10177, 2), (859267, 2), (52437899, 2)):
            pt = ec_mul(H_COF * R // f ** e, base)              # order divides f^e
            while pt is not None:
                small.append(pt)
                pt = ec_mul(f, pt)
    small += [ec_add(a, b) for a, b in zip(small, small[1:]) if ec_add(a, b) is not None]      # mixed small orders (33, 3 * 10177, ...)
    seen = set(small)
    assert len(seen) >= 20, len(seen)
    for pt in seen:
        assert ec_mul(R, pt) is not None
        assert shim.shim_g1_in_subgroup(*aff(pt)) == 0, pt
    # a point of G1 plus a point of order 3: outside, and the three forms agree
    mixed = ec_add(ec_mul(12345, G), (0, 2))
    assert shim.shim_g1_in_subgroup(*aff(mixed)) == 0


def test_g1_fixed_scalar_products_by_coz_table(shim):
    """g1.hpp mul_wnaf_coz (the FK20 transform stage's 686 fixed-scalar products): co-Z table of the odd multiples, mixed additions on
    the isomorphic curve, incomplete by design -- against big-integer scalar multiplication for random scalars in the generator's digit
    format (gen_fk20_toom.py wnaf_row: k = k1 + k2 x^2, k P = k1 P + k2 (-phi(P))), for affine and non-trivially scaled inputs, the
    identity, and with the accumulator bounds checked at the output"""
    X2 = 0xD201000000010000 ** 2
    G = (G1X, G1Y)
    rng = random.Random(77)

    def wnaf5(k):
        ds = []
        while k:
            if k & 1:
                d = k & 31
                if d >= 16:
                    d -= 32
                k -= d
            else:
                d = 0
            ds.append(d)
            k >>= 1
        return ds

    def row(k):
        d1, d2 = wnaf5(k % X2), wnaf5(k // X2)
        assert len(d1) <= 131 and len(d2) <= 131
        d1 += [0] * (132 - len(d1))
        d2 += [0] * (132 - len(d2))
        codes = [(a & 0xFF) | ((b & 0xFF) << 8) for a, b in zip(d1, d2)]
        return (C.c_uint32 * 66)(*[codes[2 * i] | codes[2 * i + 1] << 16 for i in range(66)])

    o = (C.c_uint32 * 56)()
    for trial in range(12):
        pt = ec_mul(rng.randrange(1, R), G)
        a = xyzz_from_affine(pt)
        if trial % 2:                                    # a non-trivial ZZ / ZZZ on the input: p0 + p1 through the mixed addition
            other = ec_mul(rng.randrange(1, R), G)
            shim.shim_g1_madd(a, limbs(other[0] * RP % P), limbs(other[1] * RP % P))
            pt = ec_add(pt, other)
        k = rng.randrange(1, R) if trial else 1
        if trial == 3:
            k = R - 1
        if trial == 5:
            k = X2                                       # k1 = 0: the chain starts in the second half
        shim.shim_g1_mul_wnaf_coz(a, row(k), o)
        assert shim_compress(shim, o) == compress(ec_mul(k % R, pt)), (trial, k)
        v = [sum(int(o[14 * q + i]) << (29 * i) for i in range(14)) for q in range(4)]
        assert v[0] < 8 * P and v[1] < 4 * P and v[2] < 1.01 * P and v[3] < 1.01 * P
    inf = (C.c_uint32 * 56)()
    shim.shim_g1_set_inf(inf)
    shim.shim_g1_mul_wnaf_coz(inf, row(rng.randrange(1, R)), o)
    assert shim_compress(shim, o) == compress(None)


def test_g1_matches_oracle(shim):
    """cross-check with the CPU oracle on k*G"""
    import oracle
    G = (G1X, G1Y)
    for k in (1, 2, 3, 0xDEADBEEF, R - 1):
        assert compress(ec_mul(k, G)) == oracle.g1_gen_mul(k)
