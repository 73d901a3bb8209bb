"""GPU tests at BASELINE.json's full sizes (configs[1], configs[2], configs[3]) through the device-resident
entry points.  Parity: a seeded subset is bit-compared with the oracle; the whole batch is checked through
size-independent properties (batch == per-item calls, accept valid / reject one corrupted item, order
independence of the batch results)."""
import hashlib

import pytest

pytestmark = pytest.mark.gpu


def _blobs(n, seed0):
    import oracle
    return [oracle.rand_blob(seed0 + s) for s in range(n)]


def test_commitment_batch_256(gpu_ctx, oracle_ctx):
    import torch
    B = 256
    blobs = _blobs(B, 0)
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_out = torch.zeros(B * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.ones(B, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):      # repeated calls cycle through both workspaces
        assert gpu_ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), B, d_out.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    out = bytes(d_out.cpu().tolist())
    cms = [out[48 * i:48 * i + 48] for i in range(B)]
    # the oracle on ALL 256 blobs (SURVEY 8d cfg 2: seeds 0..255) through its OpenMP batch helper
    rc, exp, stl = oracle_ctx.blob_to_kzg_commitment_batch(b"".join(blobs), B)
    assert rc == 0
    for i in range(B):
        assert cms[i] == exp[48 * i:48 * i + 48], i
    # order independence + batch == single: the reversed batch gives the reversed outputs
    rev = torch.frombuffer(bytearray(b"".join(reversed(blobs))), dtype=torch.uint8).cuda()
    d_out2 = torch.zeros(B * 48, dtype=torch.uint8, device="cuda")
    assert gpu_ctx.blob_to_kzg_commitment_batch_device(rev.data_ptr(), B, d_out2.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    out2 = bytes(d_out2.cpu().tolist())
    assert [out2[48 * i:48 * i + 48] for i in range(B)] == cms[::-1]
    for i in (1, 77, 255):
        assert gpu_ctx.BlobToKZGCommitment(blobs[i]) == (0, cms[i])
    # all distinct (distinct seeds) -- a cheap guard against aliasing between batch items
    assert len(set(cms)) == B


def test_verify_batch_1024(gpu_ctx, oracle_ctx):
    import torch
    N = 1024
    blobs = _blobs(N, 0)                          # SURVEY 8d cfg 3, literally: seeds 0..1023, commitment_i / proof_i as bench_test.go:55-66
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cm = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(N, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert gpu_ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), N, d_cm.data_ptr(), d_st.data_ptr(), st) == 0
    assert gpu_ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), N, d_pr.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    # parity of the generated inputs: every 16th commitment + proof (64 of each, first and last included) against the oracle
    cm = bytes(d_cm.cpu().tolist())
    pr = bytes(d_pr.cpu().tolist())
    idx = list(range(0, N, 16))[:-1] + [N - 1]
    sub_b = b"".join(blobs[i] for i in idx)
    rc, exp_cm, _ = oracle_ctx.blob_to_kzg_commitment_batch(sub_b, len(idx))
    assert rc == 0
    rc, exp_pr, _ = oracle_ctx.compute_blob_kzg_proof_batch(sub_b, exp_cm, len(idx))
    assert rc == 0
    for k, i in enumerate(idx):
        assert cm[48 * i:48 * i + 48] == exp_cm[48 * k:48 * k + 48], i
        assert pr[48 * i:48 * i + 48] == exp_pr[48 * k:48 * k + 48], i
    badb = d_blobs.clone()
    badb[131072 * 900 + 32 * 17:131072 * 900 + 32 * 18] = 0xff           # blob 900 holds a non-canonical field element
    try:
        for shape in (0, 1):                 # the lone-call shape and the shape calls take when they share the chip (option verify_shape)
            assert gpu_ctx.set_option("verify_shape", shape) == 0
            assert gpu_ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, st) == 0
            bad = d_pr.clone()                   # cfg 3's negative run: proof_0 replaced by proof_1
            bad[0:48] = d_pr[48:96]
            assert gpu_ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), bad.data_ptr(), N, st) == 1
            for victim in (700, 1023):           # one corrupted proof anywhere must reject the whole batch
                bad = d_pr.clone()
                other = (victim + 1) % N
                bad[48 * victim:48 * victim + 48] = d_pr[48 * other:48 * other + 48]
                assert gpu_ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), bad.data_ptr(), N, st) == 1
            badc = d_cm.clone()
            badc[0] = 0x00                        # compressed flag cleared -> decoding error, not a verification failure
            assert gpu_ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), badc.data_ptr(), d_pr.data_ptr(), N, st) == 11
            assert gpu_ctx.verify_blob_kzg_proof_batch_device(badb.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, st) == 3
    finally:
        assert gpu_ctx.set_option("verify_shape", -1) == 0


def test_verify_batch_host_pointers_1024_and_beyond(gpu_ctx, product):
    """VerifyBlobKZGProofBatch as the reference API hands it over (verify.go:88: []*Blob, host memory) at the BASELINE size and beyond one
    staging piece, through every route of the host-pointer verifier: the lone caller's pipelined path with the challenges hashed in pairs
    on the host pool (host_sha -1 / 1), the device hash after the upload (host_sha 0), the forced throughput shape (shared hash rounds),
    and -- at 2100 blobs, a piece of 2048 + one of 52 -- the device taking the FIRST transcripts of the big piece.  Accept; one wrong
    proof at the first, a middle and the last position (in the device-hashed and in the host-hashed range); a non-canonical blob; an
    undecodable commitment; nil blobs with the reference's precedence (an error at a lower index wins)."""
    import ctypes as C
    import torch
    import numpy as np
    L = product.lib()
    D = 2100                                      # EVERY item of the 1024- and the 2100-blob batches is its own blob (64 until round 5: a
    #                                               bug that permutes items within a residue class was invisible); numpy Philox instead of
    #                                               4096 SHA-256 calls per blob, uniform below 2^254 < r
    raw = np.random.Generator(np.random.Philox(20_000)).integers(0, 256, size=(D, 4096, 32), dtype=np.uint8)
    raw[:, :, 0] &= 0x3F
    blobs = [raw[i].tobytes() for i in range(D)]
    d_blobs = torch.from_numpy(raw.reshape(-1)).cuda()
    d_cm = torch.zeros(D * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(D * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(D, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert gpu_ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), D, d_cm.data_ptr(), d_st.data_ptr(), st) == 0
    assert gpu_ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), D, d_pr.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    cm, pr = bytes(d_cm.cpu().tolist()), bytes(d_pr.cpu().tolist())
    keep = [C.create_string_buffer(b, 131072) for b in blobs]
    noncanon = bytearray(blobs[3])
    noncanon[32 * 4000:32 * 4001] = b"\xff" * 32
    keep_bad = C.create_string_buffer(bytes(noncanon), 131072)

    def run(n, proofs=None, cms_=None, swap=None):
        ptrs = [C.cast(keep[i % D], C.c_char_p) for i in range(n)]
        for i, what in (swap or {}).items():
            ptrs[i] = None if what is None else C.cast(what, C.c_char_p)
        arr = (C.c_char_p * n)(*ptrs)
        cj = cms_ if cms_ is not None else b"".join(cm[48 * (i % D):48 * (i % D) + 48] for i in range(n))
        pj = proofs if proofs is not None else b"".join(pr[48 * (i % D):48 * (i % D) + 48] for i in range(n))
        return L.mkzg_verify_blob_kzg_proof_batch(gpu_ctx._h, arr, cj, pj, n, n, n)

    def wrong_at(n, v):
        pj = bytearray(b"".join(pr[48 * (i % D):48 * (i % D) + 48] for i in range(n)))
        o = (v + 1) % D
        pj[48 * v:48 * v + 48] = pr[48 * o:48 * o + 48]
        return bytes(pj)

    try:
        for host_sha, shape in ((-1, -1), (1, -1), (0, -1), (-1, 1)):
            assert gpu_ctx.set_option("host_sha", host_sha) == 0 and gpu_ctx.set_option("verify_shape", shape) == 0
            n = 1024
            assert run(n) == 0, (host_sha, shape)
            for v in (0, 700, 1023):
                assert run(n, proofs=wrong_at(n, v)) == 1, (host_sha, shape, v)
            assert run(n, swap={3 + 64 * 14: keep_bad}) == 3                              # ErrNonCanonicalScalar
            badc = bytearray(b"".join(cm[48 * (i % D):48 * (i % D) + 48] for i in range(n)))
            badc[48 * 2] = 0x00
            assert run(n, cms_=bytes(badc)) == 11                                        # undecodable commitment
            assert run(n, swap={5: None}) == 6                                           # ErrDeserializeNilInput
            assert run(n, cms_=bytes(badc), swap={905: None}) == 11                      # the commitment at index 2 comes first
            assert run(n, swap={905: None, 3 + 64 * 14: keep_bad}) == 3                  # ... and so does the blob at index 899
        assert gpu_ctx.set_option("host_sha", -1) == 0 and gpu_ctx.set_option("verify_shape", -1) == 0
        n = 2100                                  # 2048 + 52: the big piece's first transcripts are hashed by the device, the rest by the host pool
        assert run(n) == 0
        for v in (100, 1500, 2047, 2048, 2099):
            assert run(n, proofs=wrong_at(n, v)) == 1, v
        assert run(n, swap={2090: None}) == 6
    finally:
        assert gpu_ctx.set_option("host_sha", -1) == 0 and gpu_ctx.set_option("verify_shape", -1) == 0


def test_verify_host_pointers_ragged_sizes(gpu_ctx, product):
    """the host-pointer verifier's chunk plan, task lists and hash forms at the sizes where they change: one chunk / several, a first chunk
    of 32, tails merged into the last chunk, fewer transcripts than sixteen per pool thread (SHA-NI pairs) and more (AVX-512 lanes), an odd
    transcript left over -- accept, and reject with the wrong proof at the LAST position (the tail of the last task of the last chunk)"""
    import ctypes as C
    import torch
    L = product.lib()
    D = 48
    blobs = _blobs(D, 21_000)
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cm = torch.zeros(D * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(D * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(D, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert gpu_ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), D, d_cm.data_ptr(), d_st.data_ptr(), st) == 0
    assert gpu_ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), D, d_pr.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    cm, pr = bytes(d_cm.cpu().tolist()), bytes(d_pr.cpu().tolist())
    keep = [C.create_string_buffer(b, 131072) for b in blobs]
    threads = gpu_ctx.get_option("host_threads")
    sizes = sorted({1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64, 65, 80, 95, 96, 97, 159, 160, 161, 176, 177, 255, 256, 257, 16 * threads - 1, 16 * threads, 16 * threads + 17, 300})
    for n in sizes:
        arr = (C.c_char_p * n)(*[C.cast(keep[i % D], C.c_char_p) for i in range(n)])
        cj = b"".join(cm[48 * (i % D):48 * (i % D) + 48] for i in range(n))
        pj = b"".join(pr[48 * (i % D):48 * (i % D) + 48] for i in range(n))
        assert L.mkzg_verify_blob_kzg_proof_batch(gpu_ctx._h, arr, cj, pj, n, n, n) == 0, n
        o = n % D                                                     # a proof of another blob at the last position
        bad = pj[:48 * (n - 1)] + pr[48 * o:48 * o + 48]
        assert L.mkzg_verify_blob_kzg_proof_batch(gpu_ctx._h, arr, cj, bad, n, n, n) == 1, n


def test_cells_and_proofs_batch_64(gpu_ctx, oracle_ctx):
    import torch
    NB = 64
    blobs = _blobs(NB, 0)
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.ones(NB, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert gpu_ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    cells = d_cells.cpu().numpy().tobytes()
    proofs = d_proofs.cpu().numpy().tobytes()
    for b in range(0, NB, 4):                     # every 4th blob (0.4 s each on the oracle) and the last
        rc, ec, ep = oracle_ctx.compute_cells_and_kzg_proofs(blobs[b])
        assert rc == 0 and cells[b * 262144:(b + 1) * 262144] == ec and proofs[b * 6144:(b + 1) * 6144] == ep, b
    rc, ec, ep = oracle_ctx.compute_cells_and_kzg_proofs(blobs[NB - 1])
    assert rc == 0 and cells[(NB - 1) * 262144:] == ec and proofs[(NB - 1) * 6144:] == ep
    # systematic code: the first 64 cells of every blob are the blob itself
    for b in range(NB):
        assert cells[b * 262144:b * 262144 + 131072] == blobs[b]
    digest = hashlib.sha256(proofs).hexdigest()
    # determinism across a second run (checksum of all 8192 proofs)
    assert gpu_ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert hashlib.sha256(d_proofs.cpu().numpy().tobytes()).hexdigest() == digest


def _cells_of_64_blobs(gpu_ctx):
    """64 GetRandBlob blobs (seeds 0..63) -> their commitments, 8192 cells and 8192 proofs, all produced on the device"""
    import torch
    NB = 64
    blobs = _blobs(NB, 0)
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
    d_cm = torch.zeros(NB * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.ones(NB, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert gpu_ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    assert gpu_ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), NB, d_cm.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    cms = d_cm.cpu().numpy().tobytes()
    cells = d_cells.cpu().numpy().tobytes()
    proofs = d_proofs.cpu().numpy().tobytes()
    return blobs, cms, cells, proofs


def test_verify_cell_batch_8192_and_recover_64(gpu_ctx, oracle_ctx):
    """configs[4] at BASELINE size on one GPU: VerifyCellKZGProofBatch over 64 blobs x 128 cells = 8192 (commitment, index, cell,
    proof) tuples in blob-major order (reference bench_eip7594_test.go:45-60 scaled to SURVEY 8d's cfg 5a) and
    RecoverCellsAndComputeKZGProofs of all 64 blobs from their even-indexed cells (bench_eip7594_test.go:62-76)."""
    NB = 64
    blobs, cms, cells, proofs = _cells_of_64_blobs(gpu_ctx)
    c_list, i_list, cell_list, p_list = [], [], [], []
    for b in range(NB):
        for k in range(128):
            c_list.append(cms[48 * b:48 * b + 48])
            i_list.append(k)
            cell_list.append(cells[b * 262144 + k * 2048:b * 262144 + (k + 1) * 2048])
            p_list.append(proofs[(b * 128 + k) * 48:(b * 128 + k + 1) * 48])
    assert len(c_list) == 8192
    # parity anchors: three blobs' commitments, cells and proofs against the oracle
    for b in (0, 31, 63):
        assert oracle_ctx.blob_to_kzg_commitment(blobs[b]) == (0, cms[48 * b:48 * b + 48])
        rc, ec, ep = oracle_ctx.compute_cells_and_kzg_proofs(blobs[b])
        assert rc == 0 and ec == cells[b * 262144:(b + 1) * 262144] and ep == proofs[b * 6144:(b + 1) * 6144]
    assert gpu_ctx.VerifyCellKZGProofBatch(c_list, i_list, cell_list, p_list) == 0
    # one swapped cell anywhere -> verification failure (not an input error)
    for victim in (0, 4100, 8191):
        bad = list(cell_list)
        other = victim ^ 1
        bad[victim], bad[other] = bad[other], bad[victim]
        assert gpu_ctx.VerifyCellKZGProofBatch(c_list, i_list, bad, p_list) == 1
    # a proof with the compression flag cleared -> decoding error
    badp = list(p_list)
    badp[5000] = bytes([badp[5000][0] & 0x7F]) + badp[5000][1:]
    assert gpu_ctx.VerifyCellKZGProofBatch(c_list, i_list, cell_list, badp) == 11
    # a cell holding a field element >= r -> non-canonical scalar
    badc = list(cell_list)
    badc[8000] = badc[8000][:32 * 17] + b"\xff" * 32 + badc[8000][32 * 18:]
    assert gpu_ctx.VerifyCellKZGProofBatch(c_list, i_list, badc, p_list) == 3
    # order independence: the same tuples cell-major (the commitments no longer arrive in runs) still verify
    perm = [b * 128 + k for k in range(128) for b in range(NB)]
    assert gpu_ctx.VerifyCellKZGProofBatch([c_list[i] for i in perm], [i_list[i] for i in perm], [cell_list[i] for i in perm], [p_list[i] for i in perm]) == 0

    # ---- recovery of all 64 blobs from their 64 even-indexed cells, one batch call
    ids = list(range(0, 128, 2))
    jobs = [(ids, [cell_list[b * 128 + i] for i in ids]) for b in range(NB)]
    rc, res = gpu_ctx.RecoverCellsAndComputeKZGProofsBatch(jobs)
    assert rc == 0 and len(res) == NB
    for b in range(NB):
        st, oc, op = res[b]
        assert st == 0
        assert b"".join(oc) == cells[b * 262144:(b + 1) * 262144], b          # round trip: erase -> recover == the encoder's output
        assert b"".join(op) == proofs[b * 6144:(b + 1) * 6144], b
    for b in (1, 40, 62):                                                      # and bit-exact against the oracle's own recovery
        rc, ec, ep = oracle_ctx.recover_cells_and_kzg_proofs(ids, jobs[b][1])
        assert rc == 0 and ec == b"".join(res[b][1]) and ep == b"".join(res[b][2])


def test_commitment_batch_256_direct_table_c15(product, oracle_ctx):
    """configs[1] exactly as bench.py runs it: 256 blobs through the 15-bit direct window table (146 GB), device-resident,
    on the two prioritised streams bench.py alternates over -- every 8th commitment against the oracle, all of them against the
    bucket method (the two methods share no kernel)."""
    import torch
    torch.cuda.init()
    B = 256
    blobs = _blobs(B, 0)
    ctx = product.Context()
    try:
        assert ctx.set_option("max_batch", B) == 0
        assert ctx.set_option("commit_direct_bits", 0) == 0
        d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
        d_ref = torch.zeros(B * 48, dtype=torch.uint8, device="cuda")
        d_st = torch.ones(B, dtype=torch.int32, device="cuda")
        assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), B, d_ref.data_ptr(), d_st.data_ptr(), 0) == 0      # bucket method
        torch.cuda.synchronize()
        assert int(d_st.abs().sum().item()) == 0
        assert ctx.set_option("commit_direct_bits", 15) == 0
        streams = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)]
        outs = [torch.zeros(B * 48, dtype=torch.uint8, device="cuda") for _ in streams]
        sts = [torch.ones(B, dtype=torch.int32, device="cuda") for _ in streams]
        torch.cuda.synchronize()            # the fills ran on the default stream; the calls below run on their own streams
        for rep in range(4):
            i = rep % 2
            assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), B, outs[i].data_ptr(), sts[i].data_ptr(), streams[i].cuda_stream) == 0
        torch.cuda.synchronize()
        assert ctx.get_option("commit_direct_bits_effective") == 15, "the 146 GB table was not available on this box"
        for i in range(2):
            assert int(sts[i].abs().sum().item()) == 0
            assert bool((outs[i] == d_ref).all().item())
        out = outs[0].cpu().numpy().tobytes()
        idx = list(range(0, B, 8))
        rc, exp, _ = oracle_ctx.blob_to_kzg_commitment_batch(b"".join(blobs[i] for i in idx), len(idx))
        assert rc == 0
        for k, i in enumerate(idx):
            assert out[48 * i:48 * i + 48] == exp[48 * k:48 * k + 48], i
    finally:
        ctx.close()


def test_batches_beyond_8192(gpu_ctx):
    """the reference's batch verifiers take any length (verify.go:88-143, api_eip7594.go:163-214); here batches above the MSM
    kernels' 8192-point jobs are cut into pieces that share ONE random scalar and carry global exponents.  8193 items: the last
    piece holds a single item."""
    import torch
    NB, N = 64, 8193
    blobs, cms, cells, proofs = _cells_of_64_blobs(gpu_ctx)
    # ---- VerifyBlobKZGProofBatch over 8193 device-resident (blob, commitment, proof) triples (the 64 blobs, repeated)
    d_blobs64 = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cm64 = torch.frombuffer(bytearray(cms), dtype=torch.uint8).cuda()
    d_pr64 = torch.zeros(NB * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(NB, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert gpu_ctx.compute_blob_kzg_proof_batch_device(d_blobs64.data_ptr(), d_cm64.data_ptr(), NB, d_pr64.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    idx = torch.arange(N, device="cuda") % NB
    d_blobs = d_blobs64.view(NB, 131072)[idx].contiguous().view(-1)
    d_cm = d_cm64.view(NB, 48)[idx].contiguous().view(-1)
    d_pr = d_pr64.view(NB, 48)[idx].contiguous().view(-1)
    torch.cuda.synchronize()
    assert gpu_ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, st) == 0
    for victim in (8192, 8191, 0):                      # the lone item of the last piece, the last of the first piece, the first
        bad = d_pr.clone()
        other = (victim + 1) % NB
        bad[48 * victim:48 * victim + 48] = d_pr64[48 * other:48 * other + 48]
        assert gpu_ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), bad.data_ptr(), N, st) == 1, victim
    badc = d_cm.clone()
    badc[48 * 8192] = 0x00
    assert gpu_ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), badc.data_ptr(), d_pr.data_ptr(), N, st) == 11
    # two items swapped ACROSS the piece boundary: every piece on its own would still see valid-looking inputs of the right kind;
    # only the global exponents / the single r make the batch fail
    sw = d_pr.clone()
    sw[48 * 8191:48 * 8192], sw[48 * 8192:48 * 8193] = d_pr[48 * 8192:48 * 8193].clone(), d_pr[48 * 8191:48 * 8192].clone()
    assert gpu_ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), sw.data_ptr(), N, st) == 1
    del d_blobs, bad, badc, sw
    # ---- the same through the host-pointer entry point (Go's []*Blob): 2100 items = two staging pieces
    host_blobs = [blobs[i % NB] for i in range(2100)]
    pr64 = d_pr64.cpu().numpy().tobytes()
    h_cms = [cms[48 * (i % NB):48 * (i % NB) + 48] for i in range(2100)]
    h_prs = [pr64[48 * (i % NB):48 * (i % NB) + 48] for i in range(2100)]
    assert gpu_ctx.VerifyBlobKZGProofBatch(host_blobs, h_cms, h_prs) == 0
    h_bad = list(h_prs)
    h_bad[2099] = h_prs[0]
    assert gpu_ctx.VerifyBlobKZGProofBatch(host_blobs, h_cms, h_bad) == 1
    h_nil = list(host_blobs)
    h_nil[2050] = None
    assert gpu_ctx.VerifyBlobKZGProofBatch(h_nil, h_cms, h_prs) == 6          # ErrDeserializeNilInput
    # ---- VerifyCellKZGProofBatch over 8193 cells: the 8192 of the 64 blobs plus one of them again
    c_list, i_list, cell_list, p_list = [], [], [], []
    for b in range(NB):
        for k in range(128):
            c_list.append(cms[48 * b:48 * b + 48])
            i_list.append(k)
            cell_list.append(cells[b * 262144 + k * 2048:b * 262144 + (k + 1) * 2048])
            p_list.append(proofs[(b * 128 + k) * 48:(b * 128 + k + 1) * 48])
    c_list.append(c_list[77]); i_list.append(i_list[77]); cell_list.append(cell_list[77]); p_list.append(p_list[77])
    assert len(c_list) == N
    assert gpu_ctx.VerifyCellKZGProofBatch(c_list, i_list, cell_list, p_list) == 0
    badl = list(cell_list)
    badl[8192] = cell_list[78]
    assert gpu_ctx.VerifyCellKZGProofBatch(c_list, i_list, badl, p_list) == 1
    badl = list(cell_list)
    badl[8191], badl[8192] = cell_list[8192], cell_list[8191]
    assert gpu_ctx.VerifyCellKZGProofBatch(c_list, i_list, badl, p_list) == 1
    badi = list(i_list)
    badi[8192] = 128
    assert gpu_ctx.VerifyCellKZGProofBatch(c_list, badi, cell_list, p_list) == 4           # ErrInvalidCellID


def test_multi_device_context_on_one_gpu(product, oracle_ctx):
    """mkzg_ctx_new with n_dev > 1: one context that shards host-pointer batches over the listed devices and combines the batch
    verifiers' partial sums in process.  A 1-GPU box lists device 0 twice -- two full per-device contexts on the same GPU, which
    exercises the whole orchestration (shard bounds, per-device threads, in-process fold / finish, error precedence)."""
    import torch
    torch.cuda.init()
    assert product.lib().mkzg_ctx_new(b"{}", 2, (product.C.c_int * 2)(0, 99), 2, product.C.byref(product.C.c_void_p())) == 17      # no such device
    ctx = product.Context(devices=[0, 0])
    try:
        assert ctx.get_option("devices") == 2
        assert ctx.set_option("commit_direct_bits", 12) == 0           # two full per-device contexts share this one GPU's HBM
        blobs = _blobs(7, 500)
        exp_cm = [oracle_ctx.blob_to_kzg_commitment(b)[1] for b in blobs]
        rc, outs, st = ctx.BlobToKZGCommitmentBatch(blobs)
        assert rc == 0 and outs == exp_cm and st == [0] * 7
        bad = bytearray(blobs[5])
        bad[64:96] = b"\xff" * 32
        rc, outs, st = ctx.BlobToKZGCommitmentBatch(blobs[:5] + [bytes(bad)] + blobs[6:])
        assert rc == 3 and st == [0, 0, 0, 0, 0, 3, 0] and outs[6] == exp_cm[6]
        exp_pr = []
        for b, c in zip(blobs, exp_cm):
            rc, pr = ctx.ComputeBlobKZGProof(b, c)                    # single-item calls go round-robin over the devices
            assert rc == 0 and pr == oracle_ctx.compute_blob_kzg_proof(b, c)[1]
            exp_pr.append(pr)
        assert ctx.VerifyBlobKZGProofBatch(blobs, exp_cm, exp_pr) == 0
        swapped = list(exp_pr)
        swapped[2], swapped[5] = swapped[5], swapped[2]                # one item on each device's shard
        assert ctx.VerifyBlobKZGProofBatch(blobs, exp_cm, swapped) == 1
        assert ctx.VerifyBlobKZGProofBatch(blobs[:5] + [bytes(bad)] + blobs[6:], exp_cm, exp_pr) == 3
        assert ctx.VerifyBlobKZGProofBatch(blobs, exp_cm, exp_pr[:6]) == 2
        for i in range(7):
            assert ctx.VerifyBlobKZGProof(blobs[i], exp_cm[i], exp_pr[i]) == 0
        # cells: two blobs, 20 tuples cut 10 / 10 over the devices
        tuples = []
        for bi in (0, 1):
            rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(blobs[bi])
            assert rc == 0
            rco, oc, op = oracle_ctx.compute_cells_and_kzg_proofs(blobs[bi])
            assert b"".join(cells) == oc and b"".join(proofs) == op
            for ci in (0, 5, 64, 127, 33, 2, 90, 91, 92, 1):
                tuples.append((exp_cm[bi], ci, cells[ci], proofs[ci]))
        cols = lambda t: ([x[0] for x in t], [x[1] for x in t], [x[2] for x in t], [x[3] for x in t])
        assert ctx.VerifyCellKZGProofBatch(*cols(tuples)) == 0
        badt = list(tuples)
        badt[3] = (badt[3][0], badt[3][1], badt[3][2], badt[14][3])
        assert ctx.VerifyCellKZGProofBatch(*cols(badt)) == 1
        # recovery batch sharded by blob
        ids = list(range(1, 128, 2))
        jobs = []
        full = []
        for bi in (0, 1, 2):
            rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(blobs[bi])
            full.append((cells, proofs))
            jobs.append((ids, [cells[i] for i in ids]))
        rc, res = ctx.RecoverCellsAndComputeKZGProofsBatch(jobs)
        assert rc == 0
        for bi in range(3):
            assert res[bi][0] == 0 and res[bi][1] == full[bi][0] and res[bi][2] == full[bi][1]
    finally:
        ctx.close()
