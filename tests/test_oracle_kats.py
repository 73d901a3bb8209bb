"""Known-answer tests of the reference's own unit tests, restated for the CPU oracle (second pin besides
the consensus-spec vectors).  Each test names the reference test it mirrors."""
import json
import os

import pytest

import oracle

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ROOT_2_32 = 10238227357739495823651030575849232062558860180284477541189508159991286009131
HERE = os.path.dirname(os.path.abspath(__file__))


def be32(x):
    return (x % R).to_bytes(32, "big")


def domain(n):
    w = pow(ROOT_2_32, (1 << 32) // n, R)
    return [pow(w, i, R) for i in range(n)]


def brp(xs):
    n = len(xs)
    lg = n.bit_length() - 1
    return [xs[int(format(i, "0%db" % lg)[::-1], 2)] if lg else xs[i] for i in range(n)]


def lagrange_at(dom, s):
    """L_i(s) for the domain `dom` (n-th roots of unity): L_i(s) = w_i (s^n - 1) / (n (s - w_i))"""
    n = len(dom)
    z = (pow(s, n, R) - 1) * pow(n, -1, R) % R
    return [w * z % R * pow(s - w, -1, R) % R for w in dom]


def test_constants():
    """api_test.go:18-27 TestModulus / TestPointAtInfinity"""
    assert R.to_bytes(32, "big").hex() == "73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001"
    assert oracle.g1_gen_mul(0) == bytes([0xC0] + [0] * 47)
    assert oracle.g1_gen_mul(R) == bytes([0xC0] + [0] * 47)
    # the standard generator (g1_monomial[0] of the trusted setup)
    assert oracle.g1_gen_mul(1).hex() == "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"


def test_compute_challenge_interop():
    """fiatshamir_test.go:14-25 TestComputeChallengeInterop"""
    got = oracle.compute_challenge(bytes(131072), bytes([0xC0] + [0] * 47))
    assert got.hex() == "04b7b22af63d2b2f1ced8d550560e5d1e4b01e355903dee22781e87826856096"


def test_commit_regression():
    """internal/kzg/srs_test.go:61-71 TestCommitRegression: Lagrange SRS on the 4-domain with secret 100"""
    dom = domain(4)
    L = lagrange_at(dom, 100)
    srs = b"".join(oracle.g1_gen_mul(l) for l in L)
    poly = [12345, 123456, 1234567, 12345678]
    rc, got = oracle.g1_msm(b"".join(be32(p) for p in poly), srs)
    assert rc == 0
    assert got.hex() == "85bdf872da5b8561d23055d32db3fc86c672b0be7543b8c1e48634af07231bf7ab6385b765750921017cbcdbcd14f8e0"


def test_multiexp_smoke():
    """internal/multiexp/multiexp_test.go:13-33,98-133: MSM(1234567^i, (i+1) G) == naive == closed form; zero length -> identity"""
    n = 256
    scalars = [pow(1234567, i, R) for i in range(n)]
    points = b"".join(oracle.g1_gen_mul(i + 1) for i in range(n))
    sb = b"".join(be32(s) for s in scalars)
    rc, fast = oracle.g1_msm(sb, points)
    rc2, slow = oracle.g1_msm(sb, points, naive=True)
    closed = oracle.g1_gen_mul(sum(s * (i + 1) for i, s in enumerate(scalars)) % R)
    assert rc == 0 and rc2 == 0 and fast == slow == closed
    rc, empty = oracle.g1_msm(b"", b"")
    assert rc == 0 and empty == bytes([0xC0] + [0] * 47)          # multiexp_test.go:57-66, 78-96


def naive_dft(xs, w):
    n = len(xs)
    return [sum(x * pow(w, i * k, R) for i, x in enumerate(xs)) % R for k in range(n)]


def test_fft_against_naive_and_roundtrip():
    """internal/domain/fft_test.go:9-102: FFT/IFFT and coset FFT/IFFT round trips (n = 8, coset generator 7),
    and the DIF kernel against a direct DFT"""
    for n in (8, 64, 128):
        xs = [(i * i * 7919 + 13) % R for i in range(n)]
        w = pow(ROOT_2_32, (1 << 32) // n, R)
        raw = b"".join(be32(x) for x in xs)
        rc, f = oracle.fr_fft(raw)
        assert rc == 0
        got = [int.from_bytes(f[32 * i:32 * i + 32], "big") for i in range(n)]
        assert got == naive_dft(xs, w)
        rc, back = oracle.fr_fft(f, inverse=True)
        assert rc == 0 and back == raw
        rc, cf = oracle.fr_coset_fft(raw, be32(7))
        assert rc == 0
        gotc = [int.from_bytes(cf[32 * i:32 * i + 32], "big") for i in range(n)]
        assert gotc == naive_dft([x * pow(7, i, R) % R for i, x in enumerate(xs)], w)
        rc, cback = oracle.fr_coset_fft(cf, be32(7), inverse=True)
        assert rc == 0 and cback == raw


def poly_eval(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


def test_fk20_against_naive_proofs(tmp_path):
    """internal/kzg_multi/kzg_prove_test.go:18-74: FK20 cells and proofs == per-coset evaluation and naive
    quotient commitments, on an insecure SRS with secret 1234 and the polynomial with coefficient i = i."""
    s = 1234
    n = 4096
    coeffs = list(range(n))
    # insecure SRS (internal/kzg_multi/srs.go:113-141): monomial [s^i]G, Lagrange [L_i(s)]G (natural root order)
    g1_mono = [oracle.g1_gen_mul(pow(s, i, R)) for i in range(n)]
    g1_lag = [oracle.g1_gen_mul(l) for l in lagrange_at(domain(n), s)]
    with open(os.path.join(os.path.dirname(HERE), "go-eth-kzg_amd", "trusted_setup.json")) as fh:
        g2 = json.load(fh)["g2_monomial"]          # unused by the proving path; only decoded
    setup = {"g1_monomial": ["0x" + p.hex() for p in g1_mono], "g1_lagrange": ["0x" + p.hex() for p in g1_lag], "g2_monomial": g2}
    path = tmp_path / "insecure_setup.json"
    path.write_text(json.dumps(setup))
    ctx = oracle.OracleContext(str(path))
    # blob = evaluations on the 4096-domain in bit-reversed order
    rc, ev = oracle.fr_fft(b"".join(be32(c) for c in coeffs))
    evals = [ev[32 * i:32 * i + 32] for i in range(n)]
    blob = b"".join(brp(evals))
    rc, commitment = ctx.blob_to_kzg_commitment(blob)
    assert rc == 0 and commitment == oracle.g1_gen_mul(poly_eval(coeffs, s))      # Lagrange commit == [f(s)]G
    rc, cells, proofs = ctx.compute_cells_and_kzg_proofs(blob)
    assert rc == 0
    ext = brp(domain(8192))
    for k in (0, 1, 5, 64, 127):
        coset = ext[64 * k:64 * k + 64]
        # cells: f on the coset
        want = b"".join(be32(poly_eval(coeffs, x)) for x in coset)
        assert cells[2048 * k:2048 * k + 2048] == want
        # proof: commitment to q(X) = (f(X) - I(X)) / (X^64 - h^64); division of f alone gives the same quotient
        c = pow(coset[0], 64, R)
        q = [0] * (n - 64)
        rem = list(coeffs)
        for i in range(n - 1, 63, -1):
            q[i - 64] = rem[i]
            rem[i - 64] = (rem[i - 64] + rem[i] * c) % R
        assert proofs[48 * k:48 * k + 48] == oracle.g1_gen_mul(poly_eval(q, s))
    ctx.close()
