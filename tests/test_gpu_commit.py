"""GPU parity tests for BlobToKZGCommitment (config 2 of BASELINE.json), through the C ABI.

Bar: bit-exact.  Compared against (a) the reference's consensus-spec vectors
(tests/blob_to_kzg_commitment, consensus_specs_test.go:32-70) and (b) the CPU oracle on seeded
GetRandBlob inputs (reference bench_test.go:38-46) and edge cases.
"""
import hashlib
import struct

import pytest

pytestmark = pytest.mark.gpu

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def rand_blob(seed):
    """GetRandBlob (reference bench_test.go:17-46), product-side generator (hashlib)"""
    out = bytearray()
    for i in range(0, 131072, 32):
        v = int.from_bytes(hashlib.sha256(struct.pack(">q", seed + i)).digest(), "big") % R
        out += v.to_bytes(32, "big")
    return bytes(out)


def test_consensus_vectors(vectors, gpu_ctx):
    from conftest import BadHex
    n_ok = 0
    for c in vectors.cases("blob_to_kzg_commitment"):
        blob = vectors.val(c["input"]["blob"])
        exp = vectors.val(c["output"])
        if isinstance(blob, BadHex) or len(blob) != 131072:
            assert exp is None
            continue
        rc, out = gpu_ctx.BlobToKZGCommitment(blob)
        if exp is None:
            assert rc == 3, c["name"]          # ErrNonCanonicalScalar
        else:
            assert rc == 0 and out == exp, c["name"]
            n_ok += 1
    assert n_ok == 7


def test_batch_vs_oracle(gpu_ctx, oracle_ctx):
    import oracle
    blobs = [rand_blob(s) for s in range(12)]
    assert blobs[3] == oracle.rand_blob(3)      # the two GetRandBlob restatements agree
    blobs.append(bytes(131072))                                       # zero polynomial -> infinity
    blobs.append((1).to_bytes(32, "big") * 4096)                      # constant polynomial 1
    blobs.append((R - 1).to_bytes(32, "big") * 4096)                  # all scalars r-1 (max canonical)
    blobs.append(bytes(32) * 4095 + (5).to_bytes(32, "big"))          # single non-zero scalar
    rc, outs, st = gpu_ctx.BlobToKZGCommitmentBatch(blobs)
    assert rc == 0 and all(s == 0 for s in st)
    for b, o in zip(blobs, outs):
        rco, exp = oracle_ctx.blob_to_kzg_commitment(b)
        assert rco == 0 and o == exp
    assert outs[12] == bytes([0xC0] + [0] * 47)


def test_batch_with_invalid_items(gpu_ctx, oracle_ctx):
    good = rand_blob(100)
    bad = bytearray(rand_blob(101))
    bad[2111 * 32:2111 * 32 + 32] = R.to_bytes(32, "big")             # scalar #2111 == r (non canonical)
    allbad = b"\xff" * 131072
    rc, outs, st = gpu_ctx.BlobToKZGCommitmentBatch([good, bytes(bad), allbad, good])
    assert rc == 3 and st == [0, 3, 3, 0]
    exp = oracle_ctx.blob_to_kzg_commitment(good)[1]
    assert outs[0] == exp and outs[3] == exp


def test_api_edges(gpu_ctx):
    blob = rand_blob(7)
    rc, _ = gpu_ctx.BlobToKZGCommitment(blob, num_goroutines=1024)    # multiexp_test.go:68-76
    assert rc == 14
    rc, _ = gpu_ctx.BlobToKZGCommitment(None)
    assert rc == 6
    rc, _, _ = gpu_ctx.BlobToKZGCommitmentBatch([])
    assert rc == 0
    rc, _ = gpu_ctx.CommitLagrange(b"")                               # srs.go:57 zero length
    assert rc == 15


def test_commit_lagrange_prefix(gpu_ctx, oracle_ctx):
    """Commit on a short polynomial == commitment of the zero-padded blob (srs.go:61 G1[:len(p)])"""
    blob = rand_blob(9)
    rc, out = gpu_ctx.CommitLagrange(blob[:32 * 100])
    exp = oracle_ctx.blob_to_kzg_commitment(blob[:3200] + bytes(131072 - 3200))[1]
    assert rc == 0 and out == exp


def test_windows_agree(product, oracle_ctx):
    """every supported window size gives the same bytes"""
    blob = rand_blob(11)
    exp = oracle_ctx.blob_to_kzg_commitment(blob)[1]
    for c in (11, 12, 13):
        ctx = product.Context()
        assert ctx.set_option("commit_direct_bits", 0) == 0       # the window width is the bucket method's
        assert ctx.set_option("msm_window", c) == 0
        assert ctx.set_option("max_batch", 4) == 0
        rc, out = ctx.BlobToKZGCommitment(blob)
        ctx.close()
        assert rc == 0 and out == exp, c


@pytest.mark.parametrize("bits", [12, 14, 15])
def test_direct_table_mode(product, oracle_ctx, bits):
    """commit_direct_bits: the direct-table kernels give the same bytes as the oracle.  15 divides 255, so that
    width has no spare top window and takes the r - s branch for scalars with bit 254 set"""
    import torch
    torch.cuda.init()
    top = b"".join(((1 << 254) + 977 * i).to_bytes(32, "big") for i in range(4096))         # every scalar has bit 254 set
    edge = b"".join(v.to_bytes(32, "big") for v in ([R - 1, 1 << 254, (1 << 254) - 1, 1, 0, (1 << 240) - 1, R - (1 << 254)] * 586)[:4096])
    blobs = [rand_blob(21), bytes(131072), (R - 1).to_bytes(32, "big") * 4096, rand_blob(22), top, edge]
    ctx = product.Context()
    assert ctx.set_option("max_batch", 4) == 0          # 6 blobs -> two chunks through both workspaces
    assert ctx.set_option("commit_direct_bits", bits) == 0
    rc, outs, st = ctx.BlobToKZGCommitmentBatch(blobs)
    assert rc == 0 and st == [0] * len(blobs)
    for b, o in zip(blobs, outs):
        assert o == oracle_ctx.blob_to_kzg_commitment(b)[1]
    bad = bytearray(blobs[0])
    bad[-32:] = R.to_bytes(32, "big")
    rc, outs, st = ctx.BlobToKZGCommitmentBatch([bytes(bad), blobs[3]])
    assert rc == 3 and st == [3, 0] and outs[1] == oracle_ctx.blob_to_kzg_commitment(blobs[3])[1]
    ctx.close()


def test_direct_table_lane_shapes(product, oracle_ctx):
    """The direct-table commitment spreads a blob over 512 lanes (8 scalars each) in a full batch and over 1024, 2048 or 4096 lanes
    in calls of fewer blobs (msm_kernels.hip direct_pick_shape: 1..8 blobs -> a PAIR of lanes per scalar, lower and upper windows, 9..32 -> 1
    scalar per lane, ..64 -> 2, ..128 -> 4; 129..255 -> 1 with a serial sum instead of the butterflies), with a fold of
    the partial sums in between: every shape gives the oracle's bytes, an invalid blob is reported at its index."""
    import torch
    torch.cuda.init()
    distinct = [rand_blob(400 + i) for i in range(5)] + [bytes(131072)]
    exp = [oracle_ctx.blob_to_kzg_commitment(b)[1] for b in distinct]
    ctx = product.Context()
    assert ctx.set_option("commit_direct_bits", 12) == 0
    for n in (1, 2, 8, 9, 20, 32, 33, 64, 65, 128, 129, 200, 255, 300, 3):      # 1..8: a pair of lanes per scalar (lower / upper windows); the last 3 with max_batch 320
        if n == 300:
            assert ctx.set_option("max_batch", 320) == 0          # one piece above 256 blobs: one scalar per lane again (a full round + 352 wavefronts otherwise)
        blobs = [distinct[i % 6] for i in range(n)]
        bad_at = n // 2 if n > 2 else None
        if bad_at is not None:
            bad = bytearray(blobs[bad_at])
            bad[32 * 4095:] = R.to_bytes(32, "big")
            blobs[bad_at] = bytes(bad)
        rc, outs, st = ctx.BlobToKZGCommitmentBatch(blobs)                     # host-pointer pipeline (chunked accumulate launches)
        assert rc == (3 if bad_at is not None else 0), n
        for i in range(n):
            if i == bad_at:
                assert st[i] == 3
            else:
                assert st[i] == 0 and outs[i] == exp[i % 6], (n, i)
        d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()       # device-resident entry
        d_out = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
        d_st = torch.ones(n, dtype=torch.int32, device="cuda")
        ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), n, d_out.data_ptr(), d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out_b, st_d = d_out.cpu().numpy().tobytes(), d_st.cpu().tolist()
        for i in range(n):
            if i == bad_at:
                assert st_d[i] == 3
            else:
                assert st_d[i] == 0 and out_b[48 * i:48 * i + 48] == exp[i % 6], (n, i)
    ctx.close()


def test_few_blob_shapes_with_a_small_workspace(product, oracle_ctx):
    """max_batch 8: the lane's buffer starts with 8 x 512 records, so a single blob (8192 records in the window-split shape) takes the
    one-lane-per-scalar shape and eight blobs grow the buffer: same bytes, with and without the host finish"""
    blobs = [rand_blob(770 + i) for i in range(8)]
    exp = [oracle_ctx.blob_to_kzg_commitment(b)[1] for b in blobs]
    ctx = product.Context()
    try:
        assert ctx.set_option("commit_direct_bits", 12) == 0 and ctx.set_option("max_batch", 8) == 0
        for hf in (1, 0):
            assert ctx.set_option("host_fin", hf) == 0
            for n in (1, 3, 4, 5, 8):
                rc, outs, st = ctx.BlobToKZGCommitmentBatch(blobs[:n])
                assert rc == 0 and outs == exp[:n] and st == [0] * n, (hf, n)
        assert ctx.set_option("max_batch", 16) == 0                           # 16 x 512 = 8192 records: one blob fits the split shape, two do not
        for n in (1, 2, 8):
            rc, outs, st = ctx.BlobToKZGCommitmentBatch(blobs[:n])
            assert rc == 0 and outs == exp[:n], n
    finally:
        ctx.close()


def test_options_and_calibration(product, oracle_ctx):
    """options only change how the work is cut (same bytes); unknown keys / values are MKZG_ERR_BAD_ARGS; mkzg_valu_calibrate returns
    a plausible issue rate"""
    import torch
    torch.cuda.init()
    blobs = [rand_blob(31 + i) for i in range(7)]
    exp = [oracle_ctx.blob_to_kzg_commitment(b)[1] for b in blobs]
    for chunk, batch, zc in ((1, 2, 0), (2, 4, 1), (3, 256, 0), (3, 256, 1), (64, 256, 1)):       # host-pointer pipeline: chunks per piece, pieces per call
        ctx = product.Context()
        assert ctx.set_option("max_batch", batch) == 0 and ctx.get_option("max_batch") == batch
        assert ctx.set_option("host_chunk", chunk) == 0 and ctx.get_option("host_chunk") == chunk
        assert ctx.set_option("host_zero_copy", zc) == 0 and ctx.get_option("host_zero_copy") == zc
        assert ctx.get_option("commit_direct_bits") == -1            # library default: auto-sized direct table
        assert ctx.set_option("commit_direct_bits", 12 if chunk == 3 else 0) == 0     # the chunked accumulate launches exist in direct-table mode only
        assert ctx.get_option("host_wait") == -1 and ctx.set_option("host_wait", 2) == 17 and ctx.set_option("host_wait", chunk & 1) == 0 and ctx.get_option("host_wait") == (chunk & 1)
        assert ctx.get_option("cpu_share") >= 2
        assert ctx.set_option("debug_serial", 1) == 0 and ctx.set_option("debug_serial", 0) == 0
        assert ctx.set_option("fk20_segments", 0) == 0 and ctx.get_option("fk20_segments") == 0 and ctx.set_option("fk20_segments", 1) == 0
        for key in ("verify_shape", "host_sha"):
            assert ctx.set_option(key, 2) == 17 and ctx.set_option(key, 1) == 0 and ctx.get_option(key) == 1 and ctx.set_option(key, -1) == 0
        assert ctx.set_option("host_chunk", 0) == 17                 # MKZG_ERR_BAD_ARGS
        assert ctx.set_option("no_such_option", 1) == 17 and ctx.get_option("no_such_option") is None
        rc, outs, st = ctx.BlobToKZGCommitmentBatch(blobs)
        assert rc == 0 and outs == exp and st == [0] * 7
        assert ctx.get_option("commit_direct_bits_effective") == (12 if chunk == 3 else 0) and ctx.get_option("devices") == 1 and ctx.get_option("lanes") == 32
        # lanes / trim: the memory the lanes hold goes back to the device and is rebuilt by the next call (same bytes)
        free0 = torch.cuda.mem_get_info()[0]
        assert ctx.set_option("trim", 1) == 0
        assert torch.cuda.mem_get_info()[0] > free0
        assert ctx.set_option("lanes", 0) == 17 and ctx.set_option("lanes", 33) == 17 and ctx.set_option("lanes", 2) == 0 and ctx.get_option("lanes") == 2
        rc, outs, st = ctx.BlobToKZGCommitmentBatch(blobs)
        assert rc == 0 and outs == exp and st == [0] * 7
        assert 2 <= ctx.get_option("host_threads") <= 16 and ctx.get_option("tables_on_own_device") == 1
        ctx.close()
    ctx = product.Context()
    rate = ctx.valu_calibrate()
    ctx.close()
    assert rate is not None and 1e11 < rate < 2e12                   # 100 G .. 2 T wave-instructions / s


def test_direct_table_degrades_instead_of_failing(product, oracle_ctx):
    """a drop-in must degrade, not fail: with most of the HBM taken the 146 GB table cannot be allocated -- the library steps down to
    a narrower table or to the bucket method and still returns the reference's bytes"""
    import torch
    torch.cuda.init()
    free, total = torch.cuda.mem_get_info()
    keep = 30 << 30                                                  # leave 30 GB: room for the 12-bit table (21.5 GB) at most
    hog = torch.empty(max(0, free - keep), dtype=torch.uint8, device="cuda") if free > keep else None
    blobs = [rand_blob(51), rand_blob(52)]
    ctx = product.Context()
    try:
        assert ctx.set_option("commit_direct_bits", 15) == 0
        rc, outs, st = ctx.BlobToKZGCommitmentBatch(blobs)
        assert rc == 0 and st == [0, 0]
        assert outs == [oracle_ctx.blob_to_kzg_commitment(b)[1] for b in blobs]
        eff = ctx.get_option("commit_direct_bits_effective")
        assert eff in (0, 12), eff                                   # never the width asked for: it cannot fit
        # auto sizing leaves room for the FK20 table, the lanes and some slack (62 GB with the defaults): with 30 GB free that means the bucket method
        assert ctx.set_option("commit_direct_bits", -1) == 0
        rc, outs2, st = ctx.BlobToKZGCommitmentBatch(blobs)
        assert rc == 0 and outs2 == outs and ctx.get_option("commit_direct_bits_effective") == 0
    finally:
        ctx.close()
        del hog
        torch.cuda.empty_cache()


def test_c_program_through_the_abi():
    """tests/c_abi/use_all.c (plain C99 against include/mi355kzg.h, as cgo would bind it) on the GPU: the zero blob, a consensus-spec
    commitment vector byte for byte, proof + verify accept / reject"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src, exe, libdir = os.path.join(root, "tests", "c_abi", "use_all.c"), os.path.join(root, "tests", "c_abi", "use_all"), os.path.join(root, "go-eth-kzg_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), src, "-o", exe, "-L", libdir, "-lmi355kzg",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe, os.path.join(root, "go-eth-kzg_amd", "trusted_setup.json"), os.path.join(root, "tests", "golden", "c_caller_vector.bin")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    for line in ("commitment of the zero blob: rc=0 ok", "commitment vector: rc=0 ok", "proof vector: rc=0 ok", "verify valid: rc=0 ok", "verify corrupted: rc=1 ok",
                 "batch verify valid: rc=0 ok", "batch verify corrupted: rc=1 ok"):
        assert line in r.stdout, r.stdout + r.stderr


def test_msm_g1_gnark_layouts(gpu_ctx):
    """mkzg_msm_g1, the multiexp.MultiExpG1 seam with gnark-crypto's Montgomery layouts, on the reference's own KAT
    (internal/multiexp/multiexp_test.go:13-33: scalars 1234567^i, points (i+1) G, 256 of them) plus identity points, zero scalars and a
    10000-point run that crosses the 8192-point piece size; expected values from Python big-integer curve arithmetic and the oracle"""
    import oracle
    from test_mont29_host import P, G1X, G1Y, ec_add, ec_mul, compress
    RP, RF = 1 << 384, 1 << 256
    G = (G1X, G1Y)
    n = 256
    pts, cur = [], None
    for i in range(n):
        cur = ec_add(cur, G)
        pts.append(cur)
    scal = [pow(1234567, i, R) for i in range(n)]

    def mont_pt(pt):
        return (0, 0) if pt is None else (pt[0] * RP % P, pt[1] * RP % P)

    def run(ss, pp):
        rc, (x, y) = gpu_ctx.msm_g1([s * RF % R for s in ss], [mont_pt(p) for p in pp])
        assert rc == 0
        if x == 0 and y == 0:
            return None
        inv = pow(RP, -1, P)
        return (x * inv % P, y * inv % P)

    def expect(ss, pp):
        keep = [(s, p) for s, p in zip(ss, pp) if p is not None]
        rc, out = oracle.g1_msm(b"".join(s.to_bytes(32, "big") for s, _ in keep), b"".join(compress(p) for _, p in keep))
        assert rc == 0
        return out

    got = run(scal, pts)
    assert compress(got) == expect(scal, pts)
    # identity points and zero scalars contribute nothing; all-zero input gives the identity
    scal2 = list(scal); pts2 = list(pts)
    pts2[5] = None; pts2[200] = None; scal2[7] = 0; scal2[0] = R - 1
    assert compress(run(scal2, pts2)) == expect(scal2, pts2)
    assert run([0] * 8, pts[:8]) is None and run([], []) is None
    # more than one 8192-point piece: 10000 points (multiples of the first 256, cheap to build) with small distinct scalars
    big_pts = [pts[i % 256] for i in range(10000)]
    big_sc = [(i * 2654435761 + 12345) % R for i in range(10000)]
    folded = [0] * 256
    for i, sv in enumerate(big_sc):
        folded[i % 256] = (folded[i % 256] + sv) % R
    assert compress(run(big_sc, big_pts)) == expect(folded, pts)


def test_every_documented_option_reads_back_and_sets(product):
    """every tuning knob that include/mi355kzg.h documents for mkzg_ctx_set_option can be read, and writing the value read back succeeds
    (a knob that sets its value and then reports MKZG_ERR_BAD_ARGS slipped through once)"""
    import os
    import re
    import torch
    torch.cuda.init()
    hdr = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "mi355kzg.h")).read()
    block = hdr[hdr.index("Tuning knobs"):hdr.index("int mkzg_ctx_set_option")]
    keys = re.findall(r'^ \*   "([a-z0-9_]+)"', block, flags=re.M)
    assert {"msm_window", "max_batch", "commit_direct_bits", "fk20_bits", "host_wait", "fk20_segments", "debug_serial", "lanes", "trim", "verify_shape"} <= set(keys)
    ctx = product.Context()
    for key in keys:
        if key == "trim":                                             # set only
            assert ctx.set_option(key, 1) == 0
            continue
        v = ctx.get_option(key)
        assert v is not None, key
        assert ctx.set_option(key, v) == 0, key
        assert ctx.get_option(key) == v, key
    for key in ("commit_direct_bits_effective", "fk20_bits_effective", "devices", "host_threads", "cpu_share", "tables_on_own_device", "pool_node_cpus"):
        assert ctx.get_option(key) is not None, key
    ctx.close()


def test_host_finish_of_few_blob_calls(product, oracle_ctx, vectors):
    """option host_fin (default 1): host-pointer commitment calls of up to four blobs and the single-proof entry points get the sum's XYZZ
    coordinates back and invert + compress on the calling thread (api.hip commit_host_finish: the device's own limb templates compiled for
    the host) instead of waiting for the device's one-wavefront finish.  Same bytes and codes either way: the consensus vectors, the
    identity (zero blob, constant polynomial), a non-canonical blob among four, five blobs (the device's finish), ComputeKZGProof at a
    point of the domain.  Reference: prove.go:13-33,85-111 -> serialization.go:98-100."""
    from conftest import BadHex
    R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    ctx = product.Context()
    try:
        assert ctx.set_option("commit_direct_bits", 12) == 0                    # 24 GB: runs beside the other modules' contexts
        assert ctx.get_option("host_fin") == 1 and ctx.set_option("host_fin", 2) == 17
        blobs = [rand_blob(900 + i) for i in range(3)] + [bytes(131072), (9).to_bytes(32, "big") * 4096]
        bad = bytearray(blobs[0])
        bad[64:96] = R_MOD.to_bytes(32, "big")
        bad = bytes(bad)
        exp = [oracle_ctx.blob_to_kzg_commitment(b)[1] for b in blobs]
        z_dom = (1).to_bytes(32, "big")                                          # omega^0
        z_any = (0x1234567).to_bytes(32, "big")
        got = {}
        for hf in (1, 0, 1):
            assert ctx.set_option("host_fin", hf) == 0 and ctx.get_option("host_fin") == hf
            res = []
            for b, e in zip(blobs, exp):
                rc, cm = ctx.BlobToKZGCommitment(b)
                assert rc == 0 and cm == e
            rc, outs, st = ctx.BlobToKZGCommitmentBatch([blobs[0], bad, blobs[3], blobs[1]])          # four: host finish, one item rejected
            assert rc == 3 and st == [0, 3, 0, 0] and outs[0] == exp[0] and outs[2] == exp[3] and outs[3] == exp[1]
            rc, outs, st = ctx.BlobToKZGCommitmentBatch(blobs)                                        # five: the device's finish whatever the option
            assert rc == 0 and outs == exp and st == [0] * 5
            assert ctx.BlobToKZGCommitment(bad)[0] == 3
            for b, e in zip(blobs[:2] + blobs[3:], exp[:2] + exp[3:]):
                res.append(ctx.ComputeBlobKZGProof(b, e))
                res.append(ctx.ComputeKZGProof(b, z_dom))
                res.append(ctx.ComputeKZGProof(b, z_any))
                assert res[-3][0] == 0 and res[-3][1] == oracle_ctx.compute_blob_kzg_proof(b, e)[1]
                assert res[-2] == tuple(oracle_ctx.compute_kzg_proof(b, z_dom)) and res[-1] == tuple(oracle_ctx.compute_kzg_proof(b, z_any))
            assert ctx.ComputeKZGProof(bad, z_any)[0] == 3
            got[hf] = res
            n_ok = 0
            for c in vectors.cases("compute_kzg_proof"):
                blob, z, e = vectors.val(c["input"]["blob"]), vectors.val(c["input"]["z"]), vectors.val(c["output"])
                if any(isinstance(v, BadHex) for v in (blob, z)) or len(blob) != 131072 or len(z) != 32:
                    continue
                rc, pr, y = ctx.ComputeKZGProof(blob, z)
                if e is None:
                    assert rc != 0, c["name"]
                else:
                    assert rc == 0 and pr == e[0] and y == e[1], c["name"]
                    n_ok += 1
            assert n_ok > 30
        assert got[0] == got[1]
    finally:
        ctx.close()
