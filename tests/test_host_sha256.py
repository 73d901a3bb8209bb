"""CPU tests of the product's host SHA-256 (go-eth-kzg_amd/csrc/host_sha256.hpp): both compression functions against hashlib over the
block-boundary lengths, and host_compute_challenge against the reference's own KAT (fiatshamir_test.go:14-25), against a
restatement of fiatshamir.go:22-33 with hashlib, and against the oracle."""
import ctypes as C
import hashlib
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_DIR = os.path.join(ROOT, "tests", "host_shim")
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


@pytest.fixture(scope="module")
def shim():
    from conftest import build_host_shim
    L = C.CDLL(build_host_shim("sha", ("host_sha256.hpp", "host_copy.hpp")))
    L.shim_copy_to_staging.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.shim_sha256.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
    L.shim_sha256_c.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
    L.shim_challenge.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.shim_challenge_x2.argtypes = [C.c_char_p] * 6
    L.shim_challenge_x16.argtypes = [C.c_char_p] * 3
    return L


def test_sha256_against_hashlib(shim):
    data = bytes((i * 131 + 7) & 0xFF for i in range(4096))
    for n in [0, 1, 3, 55, 56, 57, 63, 64, 65, 119, 120, 127, 128, 129, 1000, 4096]:
        for fn in (shim.shim_sha256, shim.shim_sha256_c):
            out = C.create_string_buffer(32)
            fn(data[:n], n, out)
            assert out.raw == hashlib.sha256(data[:n]).digest(), (n, fn)


def _challenge(blob, cm):
    """fiatshamir.go:22-33 restated with hashlib"""
    d = hashlib.sha256(b"FSBLOBVERIFY_V1_" + (0).to_bytes(8, "big") + (4096).to_bytes(8, "big") + blob + cm).digest()
    return (int.from_bytes(d, "big") % R).to_bytes(32, "big")


def test_challenge_kat_and_oracle(shim):
    import oracle
    out = C.create_string_buffer(32)
    # the reference's interop / regression value (fiatshamir_test.go:14-25): zero blob, commitment = the identity
    shim.shim_challenge(bytes(131072), bytes([0xC0]) + bytes(47), out)
    assert out.raw == bytes.fromhex("04b7b22af63d2b2f1ced8d550560e5d1e4b01e355903dee22781e87826856096")
    blobs = [bytes(131072), oracle.rand_blob(5), oracle.rand_blob(6), (R - 1).to_bytes(32, "big") * 4096]
    cms = [bytes([0xC0]) + bytes(47), bytes(range(48)), b"\xff" * 48, bytes(48)]
    for b, c in zip(blobs, cms):
        shim.shim_challenge(b, c, out)
        assert out.raw == _challenge(b, c)
        assert out.raw == oracle.compute_challenge(b, c)
        assert int.from_bytes(out.raw, "big") < R
    # digests at and above r are reduced: search a few inputs whose raw digest exceeds r (about half of all do)
    seen_big = 0
    for s in range(8):
        b = oracle.rand_blob(100 + s)
        d = hashlib.sha256(b"FSBLOBVERIFY_V1_" + (0).to_bytes(8, "big") + (4096).to_bytes(8, "big") + b + cms[1]).digest()
        if int.from_bytes(d, "big") >= R:
            seen_big += 1
            if int.from_bytes(d, "big") >= 2 * R:
                seen_big += 100
        shim.shim_challenge(b, cms[1], out)
        assert out.raw == _challenge(b, cms[1])
    assert seen_big >= 1


def test_challenge_pairs_match_single(shim):
    """host_compute_challenge_x2 (two hash chains interleaved per core) against the one-transcript form and hashlib"""
    import oracle
    blobs = [bytes(131072), oracle.rand_blob(11), oracle.rand_blob(12), (R - 1).to_bytes(32, "big") * 4096]
    cms = [bytes([0xC0]) + bytes(47), bytes(range(48)), b"\xff" * 48, bytes(48)]
    za, zb = C.create_string_buffer(32), C.create_string_buffer(32)
    for i in range(4):
        for j in range(4):
            shim.shim_challenge_x2(blobs[i], cms[i], za, blobs[j], cms[j], zb)
            assert za.raw == _challenge(blobs[i], cms[i]) and zb.raw == _challenge(blobs[j], cms[j]), (i, j)


def test_challenge_sixteen_lanes_match_single(shim):
    """host_compute_challenge_x16 (one transcript per AVX-512 lane) against hashlib: sixteen different blobs and commitments, among them the
    zero blob and the all-(r-1) blob"""
    import oracle
    blobs = [bytes(131072), (R - 1).to_bytes(32, "big") * 4096] + [oracle.rand_blob(40 + i) for i in range(14)]
    cms = [bytes([0xC0]) + bytes(47), b"\xff" * 48] + [bytes((7 * i + j) & 0xFF for j in range(48)) for i in range(14)]
    z = C.create_string_buffer(16 * 32)
    if not shim.shim_challenge_x16(b"".join(blobs), b"".join(cms), z):
        pytest.skip("no AVX-512 on this CPU")
    for i in range(16):
        assert z.raw[32 * i:32 * i + 32] == _challenge(blobs[i], cms[i]), i


def test_copy_to_staging_every_alignment(shim):
    """copy_to_staging (host_copy.hpp: the gathers' copy into pinned staging) for aligned destinations and sizes (the streaming-store
    path: blobs 131072 B, cells 2048 B), unaligned sources, and the shapes that fall back to memcpy (odd sizes, unaligned destinations);
    guard bytes around the destination stay untouched"""
    src_buf = (C.c_ubyte * (140000 + 64))(*[(i * 37 + 11) & 0xFF for i in range(140000 + 64)])
    for n in (0, 1, 127, 128, 129, 2048, 2049, 4096, 131072):
        for src_off in (0, 1, 13, 32):
            for dst_off in (0, 32, 8, 1):
                dst = (C.c_ubyte * (n + 256))(*([0xEE] * (n + 256)))
                base = C.addressof(dst)
                pad = (-base) % 64 + dst_off                  # a 64-byte aligned address plus the offset under test
                shim.shim_copy_to_staging(base + pad, C.addressof(src_buf) + src_off, n)
                got = bytes(dst)
                assert got[pad:pad + n] == bytes(src_buf[src_off:src_off + n]), (n, src_off, dst_off)
                assert set(got[:pad]) <= {0xEE} and set(got[pad + n:]) <= {0xEE}, (n, src_off, dst_off)
