"""GPU tests of what a caller gets WITHOUT touching an option (the cgo wrapper's NewContext4096 sets none either), of the
context constructor's failure paths, and of the resource fall-backs.

Reference behaviour being mirrored:
  NewContext4096 / parseTrustedSetup   api.go:90-117, trusted_setup.go:45-83,144-192   (bad setup -> error / panic; here rc 16, nothing leaked)
  BlobToKZGCommitment                  prove.go:13-33                                   (one path, no tuning)
  ComputeCells                         api_eip7594.go:12-26
  ComputeCellsAndKZGProofs / recovery  api_eip7594.go:28-52,144-161                     (never fail for resource reasons)
"""
import ctypes as C
import json
import os
import threading

import pytest

from test_gpu_commit import R, rand_blob

pytestmark = pytest.mark.gpu
GB = 1 << 30


def _auto_width(free_bytes):
    """what "commit_direct_bits" = -1 must pick (api.hip ensure_direct): the widest table that leaves 40 GB of the free HBM alone"""
    for c in (15, 14, 13, 12):
        nwin = -(-255 // c)
        if nwin * 4096 * (1 << (c - 1)) * 128 + 40 * GB <= free_bytes:
            return c
    return 0


@pytest.fixture(scope="module")
def default_ctx(product):
    """a context exactly as mkzg_ctx_new hands it out; closed with the module so that its table (146 GB on an empty device) is gone
    before the other test files build theirs"""
    import torch
    torch.cuda.init()
    torch.cuda.empty_cache()
    ctx = product.Context()
    ctx.free_before = torch.cuda.mem_get_info()[0]
    yield ctx
    ctx.close()


def test_default_is_the_auto_sized_direct_table(default_ctx, vectors, oracle_ctx):
    """no option set: commit_direct_bits reads -1, the first commitment builds the widest table that leaves 40 GB free -- the 15-bit one
    (146 GB) on a device that is otherwise empty -- and the consensus vectors come out byte for byte"""
    from conftest import BadHex
    ctx = default_ctx
    assert ctx.get_option("commit_direct_bits") == -1 and ctx.get_option("commit_direct_bits_effective") == 0      # nothing built yet
    want = _auto_width(ctx.free_before)
    n_ok = 0
    for c in vectors.cases("blob_to_kzg_commitment"):
        blob, exp = vectors.val(c["input"]["blob"]), vectors.val(c["output"])
        if isinstance(blob, BadHex) or len(blob) != 131072:
            continue
        rc, out = ctx.BlobToKZGCommitment(blob)
        if exp is None:
            assert rc == 3, c["name"]
        else:
            assert rc == 0 and out == exp, c["name"]
            n_ok += 1
    assert n_ok == 7
    eff = ctx.get_option("commit_direct_bits_effective")
    assert eff == want, (eff, want, ctx.free_before / GB)
    if ctx.free_before >= 200 * GB:
        assert eff == 15, "an (almost) empty MI355X must get the 15-bit table by default"
    # the proof MSM goes through the same table
    n_ok = 0
    for c in vectors.cases("compute_blob_kzg_proof"):
        blob, cm, exp = vectors.val(c["input"]["blob"]), vectors.val(c["input"]["commitment"]), vectors.val(c["output"])
        if any(isinstance(v, BadHex) for v in (blob, cm)) or len(blob) != 131072 or len(cm) != 48:
            continue
        rc, proof = ctx.ComputeBlobKZGProof(blob, cm)
        if exp is None:
            assert rc != 0, c["name"]
        else:
            assert rc == 0 and proof == exp, c["name"]
            n_ok += 1
    assert n_ok == 7


def test_default_options_batch_256(default_ctx, product, oracle_ctx):
    """configs[1] with the library's defaults: 256 device-resident blobs, every 8th commitment against the oracle and all of them
    against the bucket method of a second context (the two methods share no kernel); host-pointer entry point too"""
    import torch
    B = 256
    blobs = [rand_blob(s) for s in range(B)]
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_out = torch.zeros(B * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.ones(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert default_ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), B, d_out.data_ptr(), d_st.data_ptr(), 0) == 0
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    out = d_out.cpu().numpy().tobytes()
    idx = list(range(0, B, 8))
    rc, exp, _ = oracle_ctx.blob_to_kzg_commitment_batch(b"".join(blobs[i] for i in idx), len(idx))
    assert rc == 0
    for k, i in enumerate(idx):
        assert out[48 * i:48 * i + 48] == exp[48 * k:48 * k + 48], i
    ref = product.Context()
    try:
        assert ref.set_option("commit_direct_bits", 0) == 0
        d_ref = torch.zeros(B * 48, dtype=torch.uint8, device="cuda")
        assert ref.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), B, d_ref.data_ptr(), d_st.data_ptr(), 0) == 0
        torch.cuda.synchronize()
        assert bool((d_ref == d_out).all().item())
    finally:
        ref.close()
    rc, outs, st = default_ctx.BlobToKZGCommitmentBatch(blobs)
    assert rc == 0 and st == [0] * B and b"".join(outs) == out


def test_compute_cells_batch_device(gpu_ctx, oracle_ctx):
    """mkzg_compute_cells_batch_device (ComputeCells over a device-resident batch, api_eip7594.go:12-26): 64 blobs, three of them against
    the oracle, all of them against the cells of ComputeCellsAndKZGProofs; a non-canonical blob is reported at its index"""
    import torch
    n = 64
    blobs = [rand_blob(3000 + i) for i in range(n - 2)] + [bytes(131072), (R - 1).to_bytes(32, "big") * 4096]
    bad = bytearray(blobs[7])
    bad[32 * 100:32 * 101] = R.to_bytes(32, "big")
    blobs[7] = bytes(bad)
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(n * 262144, dtype=torch.uint8, device="cuda")
    d_st = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    L = gpu_ctx._L
    assert L.mkzg_compute_cells_batch_device(gpu_ctx._h, d_blobs.data_ptr(), n, d_cells.data_ptr(), d_st.data_ptr(), s.cuda_stream) == 0
    s.synchronize()
    st = d_st.cpu().tolist()
    assert st == [3 if i == 7 else 0 for i in range(n)]
    cells = d_cells.cpu().numpy().tobytes()
    for i in (0, n - 2, n - 1):
        rc, oc = oracle_ctx.compute_cells(blobs[i])
        assert rc == 0 and cells[i * 262144:(i + 1) * 262144] == oc, i
    d_cells2 = torch.zeros(n * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(n * 128 * 48, dtype=torch.uint8, device="cuda")
    d_st2 = torch.ones(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert gpu_ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), n, d_cells2.data_ptr(), d_proofs.data_ptr(), d_st2.data_ptr(), s.cuda_stream) == 0
    s.synchronize()
    assert d_st2.cpu().tolist() == st
    cells2 = d_cells2.cpu().numpy().tobytes()
    for i in range(n):
        if i != 7:
            assert cells[i * 262144:(i + 1) * 262144] == cells2[i * 262144:(i + 1) * 262144], i
    # argument checks and the empty batch
    assert L.mkzg_compute_cells_batch_device(gpu_ctx._h, d_blobs.data_ptr(), 0, None, None, 0) == 0
    assert L.mkzg_compute_cells_batch_device(gpu_ctx._h, None, 2, d_cells.data_ptr(), d_st.data_ptr(), 0) == 17


def _ctx_new(product, js, devices=(0,)):
    h = C.c_void_p()
    ids = (C.c_int * len(devices))(*devices)
    rc = product.lib().mkzg_ctx_new(js, len(js), ids, len(devices), C.byref(h))
    return rc, h


def test_bad_trusted_setup_is_refused_without_leaks(product):
    """NewContext4096 on a broken setup (api.go:90-117 returns errors, trusted_setup.go:45-83,144-192 panics): MKZG_ERR_BAD_TRUSTED_SETUP,
    no context, and the device memory and streams taken on the way are given back"""
    import torch
    torch.cuda.init()
    with open(product.DEFAULT_SETUP, "rb") as fh:
        raw = fh.read()
    js = json.loads(raw)
    variants = {}
    variants["truncated JSON"] = raw[: len(raw) // 2]
    variants["not JSON at all"] = b"trusted setup"
    j = dict(js, g1_lagrange=js["g1_lagrange"][:4095])
    variants["4095 Lagrange points"] = json.dumps(j).encode()
    j = dict(js, g1_monomial=js["g1_monomial"][:100])
    variants["100 monomial points"] = json.dumps(j).encode()
    j = dict(js, g2_monomial=js["g2_monomial"][:1])
    variants["one G2 point"] = json.dumps(j).encode()
    lag = list(js["g1_lagrange"])
    P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    enc = int(lag[5][2:], 16)
    flags, x = enc >> 381, enc & ((1 << 381) - 1)
    for d in range(1, 64):                                      # a neighbouring x with no point on the curve: x^3 + 4 is not a square
        if x + d < P and pow(((x + d) ** 3 + 4) % P, (P - 1) // 2, P) != 1:
            lag[5] = "0x%096x" % ((flags << 381) | (x + d))
            break
    else:
        raise AssertionError("no off-curve x found")
    variants["a g1_lagrange entry off the curve"] = json.dumps(dict(js, g1_lagrange=lag)).encode()
    lag = list(js["g1_lagrange"])
    lag[9] = "0x" + "00" * 48                                    # neither the compressed nor the infinity flag
    variants["a g1_lagrange entry with bad flags"] = json.dumps(dict(js, g1_lagrange=lag)).encode()
    g2 = list(js["g2_monomial"])
    g2[1] = "0x" + "ff" * 96
    variants["bad g2_monomial[1]"] = json.dumps(dict(js, g2_monomial=g2)).encode()
    lag = list(js["g1_lagrange"])
    lag[3] = lag[3][:-2]                                         # 47 bytes
    variants["a short hex string"] = json.dumps(dict(js, g1_lagrange=lag)).encode()
    lag = list(js["g1_lagrange"])
    lag[3] = lag[3][2:] + "00"                                   # no 0x prefix
    variants["no 0x prefix"] = json.dumps(dict(js, g1_lagrange=lag)).encode()
    # one good construction first, so that one-off allocations of the runtime do not count as leaks
    rc, h = _ctx_new(product, raw)
    assert rc == 0
    product.lib().mkzg_ctx_free(h)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for name, blob in variants.items():
        for rep in range(3):
            rc, h = _ctx_new(product, blob)
            assert rc == 16 and not h.value, (name, rc)
    assert product.lib().mkzg_ctx_new(None, 0, None, 0, C.byref(C.c_void_p())) == 17
    assert product.lib().mkzg_ctx_new(raw, len(raw), None, 0, None) == 17
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20), "device memory leaked by refused constructions"
    rc, h = _ctx_new(product, raw)                               # and the library still works
    assert rc == 0
    product.lib().mkzg_ctx_free(h)


def test_fk20_table_degrades_instead_of_failing(product, oracle_ctx):
    """ComputeCellsAndKZGProofs / recovery never fail for resource reasons in the reference (api_eip7594.go:28-52,144-161): with the HBM
    taken by another tenant the FK20 window table steps down 12 -> 11 -> 8 bits (47 -> 25.8 -> 4.3 GB) and the bytes stay the oracle's; only when
    not even that fits does the call report MKZG_ERR_HIP -- and succeeds again once memory is back"""
    import torch
    torch.cuda.init()
    torch.cuda.empty_cache()
    blob = rand_blob(4242)
    rc, oc, op = oracle_ctx.compute_cells_and_kzg_proofs(blob)
    assert rc == 0
    ctx = product.Context()
    hog = None
    try:
        # twiddles, monomial SRS and the lane's workspaces first (small), so that the hog below squeezes the table only
        rc, cells = ctx.ComputeCells(blob)
        assert rc == 0 and b"".join(cells) == oc
        free = torch.cuda.mem_get_info()[0]
        keep = 12 * GB                                              # room for the 4.3 GB table + workspaces, not for 25.8 GB
        hog = torch.empty(free - keep, dtype=torch.uint8, device="cuda")
        rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(blob)
        assert rc == 0 and b"".join(cells) == oc and b"".join(proofs) == op
        assert ctx.get_option("fk20_bits") == 12 and ctx.get_option("fk20_bits_effective") == 8
        ids = list(range(0, 128, 2))
        rc, c2, p2 = ctx.RecoverCellsAndComputeKZGProofs(ids, [cells[i] for i in ids])
        assert rc == 0 and c2 == cells and p2 == proofs
        del hog
        hog = None
        torch.cuda.empty_cache()
        # asking again rebuilds the table at the width that now fits
        assert ctx.set_option("fk20_bits", 13) == 0 and ctx.set_option("fk20_bits", 12) == 0
        rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(blob)
        assert rc == 0 and b"".join(proofs) == op and ctx.get_option("fk20_bits_effective") == 12
    finally:
        del hog
        ctx.close()
        torch.cuda.empty_cache()
    # nothing fits: a clean error, no abort, and a later call on the same context succeeds
    ctx = product.Context()
    hog = None
    try:
        rc, cells = ctx.ComputeCells(blob)
        assert rc == 0
        hog = []
        for _ in range(8):                                           # the driver gives large blocks back lazily: take what turns up until 2 GB are left
            free = torch.cuda.mem_get_info()[0]
            if free <= 3 * GB:
                break
            hog.append(torch.empty(free - 2 * GB, dtype=torch.uint8, device="cuda"))
            torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] <= 3 * GB
        rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(blob)
        assert rc == 100 and cells is None                          # MKZG_ERR_HIP
        del hog
        hog = None
        torch.cuda.empty_cache()
        rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(blob)
        assert rc == 0 and b"".join(proofs) == op
    finally:
        del hog
        ctx.close()
        torch.cuda.empty_cache()


def test_multi_device_context_on_two_gpus(product, oracle_ctx):
    """(needs two GPUs) a context over devices [0, 1] whose FIRST calls are the ones that build shared tables lazily, made from a thread
    whose current device is not the context's: every peer's tables must land on that peer's own device, the results must be the
    oracle's, and the caller's current device must be left alone"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box: covered by test_multi_device_context_on_one_gpu with devices=[0, 0]")
    torch.cuda.init()
    blobs = [rand_blob(8100 + i) for i in range(4)]
    exp = [oracle_ctx.compute_cells_and_kzg_proofs(b) for b in blobs[:2]]
    cms = [oracle_ctx.blob_to_kzg_commitment(b)[1] for b in blobs]
    ctx = product.Context(devices=[0, 1])
    errors = []

    def first_calls():
        try:
            torch.cuda.set_device(1)                                 # the context's first device is 0
            tuples = []
            for bi in (0, 1):
                cells = [exp[bi][1][i * 2048:(i + 1) * 2048] for i in range(128)]
                proofs = [exp[bi][2][i * 48:(i + 1) * 48] for i in range(128)]
                for ci in range(0, 128, 13):
                    tuples.append((cms[bi], ci, cells[ci], proofs[ci]))
            cols = lambda t: ([x[0] for x in t], [x[1] for x in t], [x[2] for x in t], [x[3] for x in t])
            assert ctx.VerifyCellKZGProofBatch(*cols(tuples)) == 0          # FIRST call on the fresh context: twiddles + monomial SRS on both peers
            assert torch.cuda.current_device() == 1
            assert ctx.get_option("tables_on_own_device") == 1
            for bi in (0, 1, 0, 1):                                  # round-robin: both peers
                rc, cells = ctx.ComputeCells(blobs[bi])
                assert rc == 0 and b"".join(cells) == exp[bi][1]
            rc, outs, st = ctx.BlobToKZGCommitmentBatch(blobs)
            assert rc == 0 and outs == cms
            for bi in (0, 1):
                rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(blobs[bi])
                assert rc == 0 and b"".join(proofs) == exp[bi][2]
            assert ctx.get_option("tables_on_own_device") == 1
            assert torch.cuda.current_device() == 1
        except Exception:      # noqa: BLE001
            import traceback
            errors.append(traceback.format_exc())

    t = threading.Thread(target=first_calls)
    t.start()
    t.join(timeout=900)
    ctx.close()
    assert not errors, errors[0]


def test_sleeping_waits_give_the_same_results(product):
    """option host_wait: every entry point that waits for the device does so by the runtime's polling wait (0) or by event queries and
    sleeps (1, what a crowded context does by itself) -- same bytes and verdicts either way (api.go / api_eip7594.go entry points)"""
    import torch
    torch.cuda.init()
    blobs = [rand_blob(900 + i) for i in range(3)]
    results = []
    for mode in (0, 1):
        ctx = product.Context()
        assert ctx.set_option("host_wait", mode) == 0
        out = []
        rc, cms, st = ctx.BlobToKZGCommitmentBatch(blobs)
        assert rc == 0 and st == [0] * 3
        out.append(cms)
        prs = []
        for b, cm in zip(blobs, cms):
            rc, pr = ctx.ComputeBlobKZGProof(b, cm)
            assert rc == 0
            prs.append(pr)
        out.append(prs)
        z = (12345).to_bytes(32, "big")
        rc, pz, y = ctx.ComputeKZGProof(blobs[0], z)
        assert rc == 0
        out.append((pz, y))
        assert ctx.VerifyKZGProof(cms[0], z, y, pz) == 0
        assert ctx.VerifyBlobKZGProof(blobs[0], cms[0], prs[0]) == 0
        assert ctx.VerifyBlobKZGProofBatch(blobs, cms, prs) == 0
        assert ctx.VerifyBlobKZGProofBatch(blobs, cms, prs[::-1]) != 0
        rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(blobs[1])
        assert rc == 0
        out.append((cells, proofs))
        assert ctx.VerifyCellKZGProofBatch([cms[1]] * 128, list(range(128)), cells, proofs) == 0
        assert ctx.VerifyCellKZGProofBatch([cms[0]] * 128, list(range(128)), cells, proofs) != 0
        ids = list(range(1, 128, 2))
        rc, rcells, rproofs = ctx.RecoverCellsAndComputeKZGProofs(ids, [cells[i] for i in ids])
        assert rc == 0 and rcells == cells and rproofs == proofs
        results.append(out)
        ctx.close()
    assert results[0] == results[1]


def test_pipelined_calls_on_one_stream_keep_one_lane(product):
    """Asynchronous device-pointer calls issued back to back on ONE stream are serialised by the stream anyway: they must reuse the lane
    (and the workspaces) of the previous call on that stream instead of taking a fresh lane each -- no new device memory, every call
    returns at once, same bytes (api.hip lane_acquire, rule 0)."""
    import time
    import torch
    torch.cuda.init()
    ctx = product.Context()
    n = 24
    blobs = [rand_blob(1200 + i) for i in range(n)]
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    outs = [(torch.zeros(n * 262144, dtype=torch.uint8, device="cuda"), torch.zeros(n * 6144, dtype=torch.uint8, device="cuda"),
             torch.ones(n, dtype=torch.int32, device="cuda")) for _ in range(5)]
    stream = torch.cuda.Stream()
    st = stream.cuda_stream

    def call(k):
        c, p, s = outs[k]
        assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), n, c.data_ptr(), p.data_ptr(), s.data_ptr(), st) == 0

    call(0)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    returned = []
    t0 = time.perf_counter()
    for k in range(1, 5):
        call(k)
        returned.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    assert free0 - torch.cuda.mem_get_info()[0] < (64 << 20), "back-to-back calls on one stream allocated new lane workspaces"
    assert returned[-1] < 0.5 * total, (returned, total)          # the calls were enqueued, not waited for
    for k in range(1, 5):
        assert int(outs[k][2].abs().sum().item()) == 0
        assert torch.equal(outs[k][0], outs[0][0]) and torch.equal(outs[k][1], outs[0][1])
    ctx.close()


def _direct_bytes(cb):
    return -(-255 // cb) * 4096 * (1 << (cb - 1)) * 128


def _expected_widths(free_bytes, order):
    """the library's sizing rule restated (api.hip direct_headroom / ensure_direct, api_cells.hip ensure_fk20_tables) -> (commit bits, fk20 bits,
    smallest distance of a decision from its threshold)"""
    fk12 = 8192 * 22 * 2048 * 128
    lanes = 32 * (340 << 20)
    slack = 4 << 30
    margin = 1 << 62

    def commit(free, fk_built):
        nonlocal margin
        head = slack + lanes + (0 if fk_built else fk12)
        for cb in (15, 14, 13, 12):
            margin = min(margin, abs(free - (_direct_bytes(cb) + head)))
            if _direct_bytes(cb) + head <= free:
                return cb
        return 0

    if order == "commit_first":
        cb = commit(free_bytes, False)
        left = free_bytes - (_direct_bytes(cb) if cb else 0)
        margin = min(margin, abs(left - fk12 - (1 << 30)))
        return cb, (12 if fk12 + (1 << 30) <= left else None), margin
    margin = min(margin, abs(free_bytes - fk12 - (1 << 30)))
    if fk12 + (1 << 30) > free_bytes:
        return None, None, margin
    return commit(free_bytes - fk12, True), 12, margin


def test_default_tables_do_not_push_each_other_down(product, oracle_ctx):
    """Both window tables at their DEFAULT options on a device that is NOT empty (a shared GPU, several ranks per GPU; here: 180 GiB free,
    or what the session's other contexts leave): the commitment table is sized so that the FK20 table still gets its default width
    afterwards -- with 180 GiB free 14 bits (82 GB) + 12 bits (47 GB), not 15 bits (146 GB) and then a degraded FK20 table -- and, built
    in the other order, the FK20 table already in place is not reserved twice.  `commit_direct_bits_effective` / `fk20_bits_effective`
    say what is in use; same bytes out as the oracle either way."""
    import torch
    torch.cuda.init()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    # 180 GiB is the telling size (15 bits would fit beside a fixed 40 GB reserve and then push the FK20 table down a width); further down
    # the list: what the session's other contexts leave, at a size where no decision sits within 3 GiB of its threshold
    target = next((t << 30 for t in (180, 150, 120, 95, 85, 70) if (t << 30) <= free - (2 << 30) and
                   all(_expected_widths(t << 30, o)[1] is not None and _expected_widths(t << 30, o)[2] >= (3 << 30) for o in ("commit_first", "fk20_first"))), None)
    if target is None:
        pytest.skip(f"needs 72 GiB of free HBM, {free >> 30} GiB are free")
    hog = torch.empty(free - target, dtype=torch.uint8, device="cuda")
    blob = rand_blob(77)
    exp_cm = oracle_ctx.blob_to_kzg_commitment(blob)[1]
    rc_o, exp_cells, exp_proofs = oracle_ctx.compute_cells_and_kzg_proofs(blob)
    checked = 0
    try:
        for order in ("commit_first", "fk20_first"):
            want_cb, want_fk, margin = _expected_widths(target, order)
            if want_fk is None or margin < (3 << 30):
                continue                                   # a decision within 3 GB of its threshold: the lanes' own allocations could tip it
            ctx = product.Context()
            try:
                def commit():
                    rc, cm = ctx.BlobToKZGCommitment(blob)
                    assert rc == 0 and cm == exp_cm

                def cells():
                    rc, cl, pr = ctx.ComputeCellsAndKZGProofs(blob)
                    assert rc == 0 and b"".join(cl) == exp_cells and b"".join(pr) == exp_proofs

                for step in ((commit, cells) if order == "commit_first" else (cells, commit)):
                    step()
                assert ctx.get_option("fk20_bits_effective") == want_fk, order
                assert ctx.get_option("commit_direct_bits_effective") == want_cb, (order, target >> 30, ctx.get_option("commit_direct_bits_effective"), want_cb)
                checked += 1
            finally:
                ctx.close()
    finally:
        del hog
        torch.cuda.empty_cache()
    assert checked >= 1
