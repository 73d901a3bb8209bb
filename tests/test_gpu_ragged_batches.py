"""Batch sizes that do not fill the kernels' units (64 blobs per SHA-256 workgroup, 32 per consumer wavefront, slices of 4096 proofs):
device-resident proof batches of ragged sizes must equal the per-blob API, and verify (reference prove.go:46-67, verify.go:88-143)."""
import pytest

from test_gpu_commit import rand_blob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def table_ctx(product):
    """a second context whose proof MSMs go through the direct window table (12 bits: 24 GB), closed with the module"""
    import torch
    torch.cuda.init()
    ctx = product.Context()
    assert ctx.set_option("commit_direct_bits", 12) == 0
    yield ctx
    ctx.close()


@pytest.mark.parametrize("method", ["buckets", "table"])
@pytest.mark.parametrize("n", [1, 31, 33, 63, 65, 130, 200, 300])
def test_ragged_proof_batches_match_single_calls(gpu_ctx, table_ctx, n, method):
    import torch
    ctx = gpu_ctx if method == "buckets" else table_ctx          # 130, 200: the one-scalar-per-lane shape of 129..255 items; 300: 256 + 44
    blobs = [rand_blob(2000 + (i % 5)) for i in range(n)]           # five distinct blobs, repeated
    singles = {}
    for i in range(5):
        if i < n:
            rc, cm = ctx.BlobToKZGCommitment(blobs[i])
            assert rc == 0
            rc, pr = ctx.ComputeBlobKZGProof(blobs[i], cm)
            assert rc == 0
            singles[i] = (cm, pr)
    cms = b"".join(singles[i % 5][0] for i in range(n))
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cm = torch.frombuffer(bytearray(cms), dtype=torch.uint8).cuda()
    d_pr = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.ones(n, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), n, d_pr.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    got = bytes(d_pr.cpu().numpy().tobytes())
    for i in range(n):
        assert got[48 * i:48 * i + 48] == singles[i % 5][1], (n, i)
    assert ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), n, st) == 0
    # one wrong proof anywhere (here: the last lane pair of the last wavefront) must be rejected
    bad = bytearray(got)
    bad[48 * (n - 1):48 * n] = singles[(n) % 5][1] if n > 1 and singles[(n) % 5][1] != singles[(n - 1) % 5][1] else bytes(got[:48][::-1])
    d_bad = torch.frombuffer(bad, dtype=torch.uint8).cuda()
    assert ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_bad.data_ptr(), n, st) != 0


@pytest.mark.parametrize("n", [9, 16, 17, 32, 33, 40, 65, 100, 130, 300])
def test_ragged_cell_proof_batches_match_single_calls(gpu_ctx, n):
    """ComputeCellsAndKZGProofs for n blobs in one call (the 686 fixed-scalar products of the transform network with one LANE per
    blob above 32 blobs -- partially filled wavefronts, more than one wavefront of blobs --, a PAIR of lanes per blob up to 32 and a QUAD
    up to 16; the MSM with 8, 12 or 24 lanes per sum; 130 blobs as one piece, 300 as two of 150) against the per-blob entry point -- different code paths for fk20.go:76-124 that must
    agree byte for byte; one blob of the batch is invalid."""
    import torch
    ctx = gpu_ctx
    distinct = [rand_blob(3000 + i) for i in range(4)]
    singles = []
    for b in distinct:
        rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(b)
        assert rc == 0
        singles.append((b"".join(cells), b"".join(proofs)))
    blobs = [distinct[i % 4] for i in range(n)]
    bad_at = n // 2
    bad = bytearray(blobs[bad_at])
    bad[32 * 77:32 * 78] = b"\xff" * 32                                 # not a canonical field element
    blobs[bad_at] = bytes(bad)
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(n * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(n * 128 * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.ones(n, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), n, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st)
    torch.cuda.synchronize()
    status = d_st.cpu().tolist()
    cells = d_cells.cpu().numpy().tobytes()
    proofs = d_proofs.cpu().numpy().tobytes()
    for i in range(n):
        if i == bad_at:
            assert status[i] != 0
            continue
        assert status[i] == 0, i
        assert cells[i * 262144:(i + 1) * 262144] == singles[i % 4][0], (n, i)
        assert proofs[i * 6144:(i + 1) * 6144] == singles[i % 4][1], (n, i)


@pytest.mark.parametrize("n", [100, 150])
def test_segmented_and_whole_fk20_products_agree(gpu_ctx, n):
    """Calls of 65..192 blobs run the 686 fixed-scalar products of the FK20 transform stage in cost-equal segments over several launches
    (fk20_kernels.hip k_fk20_toom_seg: K = 2 at 100 blobs, K = 4 at 150); option fk20_segments 0 runs them whole, as calls of other
    sizes do.  Same proofs byte for byte (fk20.go:76-124), including a blob the batch rejects."""
    import torch
    ctx = gpu_ctx
    blobs = [rand_blob(4000 + (i % 7)) for i in range(n)]
    bad = bytearray(blobs[n - 3])
    bad[32 * 9:32 * 10] = b"\xff" * 32
    blobs[n - 3] = bytes(bad)
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    st = torch.cuda.current_stream().cuda_stream
    results = []
    for seg in (1, 0):
        assert ctx.set_option("fk20_segments", seg) == 0 and ctx.get_option("fk20_segments") == seg
        d_cells = torch.zeros(n * 262144, dtype=torch.uint8, device="cuda")
        d_proofs = torch.zeros(n * 6144, dtype=torch.uint8, device="cuda")
        d_st = torch.ones(n, dtype=torch.int32, device="cuda")
        assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), n, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st) == 0
        torch.cuda.synchronize()
        results.append((d_proofs.cpu(), d_cells.cpu(), d_st.cpu()))
    assert ctx.set_option("fk20_segments", 1) == 0
    status = results[0][2].tolist()
    assert status[n - 3] != 0 and sum(1 for v in status if v) == 1
    assert torch.equal(results[0][2], results[1][2]) and torch.equal(results[0][1], results[1][1])
    keep = torch.ones(n * 6144, dtype=torch.bool)
    keep[(n - 3) * 6144:(n - 2) * 6144] = False                      # the rejected blob's proofs are unspecified
    assert torch.equal(results[0][0][keep], results[1][0][keep])
    # and against the single-blob path for one of them
    rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(blobs[5])
    assert rc == 0 and bytes(results[0][0][5 * 6144:6 * 6144].numpy().tobytes()) == b"".join(proofs)
