"""CPU unit tests of the product's host-side pairing and G1 code
(go-eth-kzg_amd/csrc/host_pairing.hpp, host_g1.hpp) -- bilinearity on the trusted-setup powers
[tau^i]G1 / [tau^i]G2 and the identities the final exponentiation relies on."""
import ctypes as C
import json
import os
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_DIR = os.path.join(ROOT, "tests", "host_shim")
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
X = -0xD201000000010000


@pytest.fixture(scope="module")
def shim():
    from conftest import build_host_shim
    return C.CDLL(build_host_shim("pairing", ("host_pairing.hpp", "host_cyclo_ifma.hpp", "host_field.hpp", "host_g1.hpp", "host_validate.hpp", "host_pool.hpp", "g1.hpp", "mont29.hpp", "constants29.h")))


@pytest.fixture(scope="module")
def setup():
    with open(os.path.join(ROOT, "go-eth-kzg_amd", "trusted_setup.json")) as fh:
        js = json.load(fh)
    g1 = [bytes.fromhex(x[2:]) for x in js["g1_monomial"][:16]]
    g2 = [bytes.fromhex(x[2:]) for x in js["g2_monomial"][:16]]
    return g1, g2


def neg(p48):
    b = bytearray(p48)
    if not (b[0] & 0x40):
        b[0] ^= 0x20
    return bytes(b)


def check(shim, ps, qs):
    return shim.shim_pairing_check(b"".join(ps), b"".join(qs), len(ps))


def test_hard_part_identity():
    h = (P**4 - P**2 + 1) // R
    assert (X - 1) ** 2 * (X + P) * (X * X + P * P - 1) + 3 == 3 * h
    assert (P - 1) % 6 == 0


def test_bilinearity(shim, setup):
    g1, g2 = setup
    # e(tau^a G1, tau^b G2) e(-tau^c G1, tau^d G2) == 1  <=>  a+b == c+d
    t0 = time.time()
    assert check(shim, [g1[1], neg(g1[0])], [g2[0], g2[1]]) == 1
    dt = time.time() - t0
    assert check(shim, [g1[5], neg(g1[7])], [g2[3], g2[1]]) == 1
    assert check(shim, [g1[2], neg(g1[0])], [g2[0], g2[1]]) == 0
    assert check(shim, [g1[0]], [g2[0]]) == 0
    assert check(shim, [g1[3], g1[3]], [g2[2], g2[2]]) == 0
    # three factors: tau^(1+2) * tau^(0+4) * tau^-(...)
    assert check(shim, [g1[1], neg(g1[3])], [g2[2], g2[0]]) == 1
    assert check(shim, [g1[1], g1[2], neg(g1[3]), neg(g1[5])], [g2[2], g2[3], g2[0], g2[0]]) == 1
    # infinity members contribute 1
    inf = bytes([0xC0] + [0] * 47)
    assert check(shim, [inf, g1[1], neg(g1[0])], [g2[5], g2[0], g2[1]]) == 1
    assert check(shim, [inf], [g2[0]]) == 1
    print("2-pairing check incl. line preparation: %.1f ms" % (dt * 1e3))


def test_cyclotomic_squaring(shim, setup):
    """Granger-Scott squaring == general Fp12 squaring on 20 elements of the cyclotomic subgroup, and differs outside it"""
    g1, g2 = setup
    assert shim.shim_cyclotomic_sqr_check(g1[3], g2[2]) == 0
    assert shim.shim_cyclotomic_sqr_check(g1[1], g2[0]) == 0


def test_g1_mul_matches_oracle(shim, setup):
    import oracle
    g1, _ = setup
    for k in (1, 2, 0xDEADBEEFCAFE, R - 1):
        karr = (C.c_uint64 * 4)(*[(k >> (64 * i)) & (2**64 - 1) for i in range(4)])
        out = C.create_string_buffer(96)
        rc = shim.shim_g1_mul(g1[0], karr, out)
        assert rc == 0
        x = int.from_bytes(out.raw[:48], "big")
        y = int.from_bytes(out.raw[48:], "big")
        exp = bytearray(x.to_bytes(48, "big"))
        exp[0] |= 0x80 | (0x20 if y > (P - 1) // 2 else 0)
        assert bytes(exp) == oracle.g1_gen_mul(k)


def test_host_pool(shim):
    """the persistent host threads the verifiers' host steps run on (host_pool.hpp)"""
    assert shim.shim_pool_selftest() == 0


def test_hostfp_inv_divsteps_matches_fermat(shim):
    """HostFp::inv goes through the divsteps inversion of the 29-bit-limb field (modinv30.hpp, the one the kernels use): same
    values as a^(p-2) on random elements, on 0 and on 1, and a * a^-1 == 1"""
    shim.shim_hostfp_inv_check.argtypes = [C.c_int, C.c_uint64]
    assert shim.shim_hostfp_inv_check(3000, 0x243f6a8885a308d3) == 0


def test_hostfp_mul_adx_matches_portable(shim):
    """HostFp::mul has an x86-64 form with mulx / adcx / adox (host_field.hpp) beside the portable loop: both must give the same canonical
    limbs on random operands and on the edge values 0, 1, R mod p, p - 1, p - 2, p - 3, 2^384 - 1 mod p, also when the output aliases
    an input"""
    bad = shim.shim_hostfp_mul_adx_check(4000, 0x9e3779b97f4a7c15)
    if bad < 0:
        pytest.skip("this CPU has no BMI2 + ADX")
    assert bad == 0


def test_ifma_squaring_chain_matches_scalar_tower(shim, setup):
    """the AVX-512 IFMA cyclotomic squaring chain of the final exponentiation (host_cyclo_ifma.hpp: eight 52-bit-limb Montgomery
    multiplications per instruction stream, loose values, lane permutes for the Fp4 squarings) against the scalar Granger-Scott squaring:
    1, 2, 3, 16 and 63 squarings in a row and the whole a^|x| on several elements of the cyclotomic subgroup, for several pairs"""
    g1, g2 = setup
    ran = 0
    for i, j in ((1, 0), (5, 3), (0, 7), (9, 9)):
        rc = shim.shim_ifma_chain_check(g1[i], g2[j])
        if rc == -2:
            pytest.skip("no AVX-512 IFMA on this CPU")
        assert rc == 0, (i, j, rc)
        ran += 1
    assert ran == 4


def test_miller_loop_segments_multiply_to_the_whole_loop(shim, setup):
    """host_pairing.hpp miller_segment_prepared: a Miller loop cut into segments that start from 1 and carry their trailing squarings
    (what a lone verifier call spreads over pool threads) multiplies back to the value of the whole loop, for cuts at, before and
    after every set bit of |x|; the pairing checks built on two and three segments agree with the plain ones (shim_pairing_check)"""
    g1, g2 = setup
    for i, j in ((1, 0), (5, 3), (0, 7), (9, 9)):
        assert shim.shim_miller_segments_check(g1[i], g2[j]) == 0, (i, j)


def test_host_point_validation_matches_oracle(shim, setup):
    """host_validate.hpp g1h_validate_row (gnark SetBytes on the calling thread for the single-proof entry points, reference
    serialization.go:108-115): the status of every kind of encoding equals the oracle's deserialisation -- trusted-setup points, multiples
    of the generator, the identity, bad flag combinations, x >= p, x not on the curve, curve points OUTSIDE the r-torsion (orders with
    small factors among them), both sign bits -- and the affine record it leaves compresses back to the input (the 29-bit-limb conversion)."""
    import random
    import sys
    sys.path.insert(0, ROOT)
    import oracle
    g1, _ = setup
    rng = random.Random(11)
    cases = list(g1[:8]) + [oracle.g1_gen_mul(k) for k in (1, 2, 3, R - 1, 0x1234567890abcdef, rng.randrange(R))]
    cases.append(bytes([0xc0] + [0] * 47))                                  # identity
    cases.append(bytes([0xc0] + [0] * 46 + [1]))                            # identity flag with payload
    cases.append(bytes([0xe0] + [0] * 47))                                  # all three flags
    cases.append(bytes([0x40] + [0] * 47))                                  # not compressed
    cases.append(bytes(48))                                                 # no flags
    cases.append(bytes([0x9f] + [0xff] * 47))                               # x >= p
    cases.append(bytes([0x80]) + (P - 1).to_bytes(48, "big")[1:])           # x = p - 1 with the flag byte
    # curve points that are (almost surely) not in the subgroup, and x values that are not on the curve, with either sign bit
    n_out = n_off = 0
    x = 5
    while n_out < 12 or n_off < 6:
        x += 1
        rhs = (x * x * x + 4) % P
        on = pow(rhs, (P - 1) // 2, P) == 1
        for sign in (0x80, 0xa0):
            b = bytearray(x.to_bytes(48, "big"))
            b[0] |= sign
            cases.append(bytes(b))
        n_out += on
        n_off += not on
    # a point of order 3: (0, 2) -- x = 0
    cases.append(bytes([0x80] + [0] * 47))
    cases.append(bytes([0xa0] + [0] * 47))
    shim.shim_g1_validate_row.restype = C.c_int
    seen = set()
    for enc in cases:
        row = (C.c_uint32 * 32)()
        got = shim.shim_g1_validate_row(enc, row)
        want = oracle.g1_validate(enc)
        assert got == want, (enc.hex(), got, want)
        seen.add(got)
        if got in (0, 13) and not (enc[0] & 0x40):
            out = C.create_string_buffer(48)
            shim.shim_row_compress(row, out)
            assert out.raw == enc, enc.hex()
            assert row[28] == 0
        else:
            assert row[28] == 1 and not any(row[i] for i in range(28))
    assert seen == {0, 11, 12, 13}
