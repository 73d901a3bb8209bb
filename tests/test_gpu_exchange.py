"""GPU test of the exchange step inside the library (go-eth-kzg_amd/csrc/api_exchange.hip, include/mi355kzg.h "multi-process jobs") on
REAL RCCL through the C ABI: a communicator of one rank (the box has one GPU) made by the library's own mkzg_comm_unique_id /
mkzg_comm_init, so ncclAllGather / ncclBroadcast run on device memory, and the verdicts of the three `_exchange` entry points
(reference verify.go:88-143, api_eip7594.go:163-214) are those of the single-process calls: accept, reject, the reference's input errors,
and the refusal of shards that do not add up to the announced batch.  The protocol with 2 and 8 ranks runs on the CPU under
ThreadSanitizer over a stand-in for librccl (tests/test_host_sanitizers.py)."""
import ctypes as C

import pytest

from test_gpu_commit import rand_blob

pytestmark = pytest.mark.gpu
VERIFY_FAILED = 1
R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


@pytest.fixture(scope="module")
def comm(gpu_ctx, product):
    L = product.lib()
    ident = C.create_string_buffer(128)
    assert L.mkzg_comm_unique_id(ident) == 0
    h = C.c_void_p()
    assert L.mkzg_comm_init(gpu_ctx._h, ident, 0, 1, C.byref(h)) == 0 and h.value
    yield h
    L.mkzg_comm_free(h)


def test_blob_exchange_world_1(gpu_ctx, product, comm):
    import torch
    L = product.lib()
    n = 9
    blobs = [rand_blob(7700 + s) for s in range(n)]
    rc, cms, _ = gpu_ctx.BlobToKZGCommitmentBatch(blobs)
    assert rc == 0
    prs = [gpu_ctx.ComputeBlobKZGProof(b, c)[1] for b, c in zip(blobs, cms)]
    keep = [C.create_string_buffer(b, 131072) for b in blobs]
    arr = (C.c_char_p * n)(*[C.cast(k, C.c_char_p) for k in keep])
    cj, pj = b"".join(cms), b"".join(prs)
    assert L.mkzg_verify_blob_kzg_proof_batch_exchange(comm, arr, cj, pj, n, n) == 0
    assert L.mkzg_comm_last_exchange_us(comm) > 0
    bad = pj[:48 * 4] + pj[48 * 5:48 * 6] + pj[48 * 5:]
    assert L.mkzg_verify_blob_kzg_proof_batch_exchange(comm, arr, cj, bad, n, n) == VERIFY_FAILED
    assert L.mkzg_verify_blob_kzg_proof_batch_exchange(comm, arr, cj, pj, n, n + 1) == product.ERR_BATCH_LENGTH        # the shards do not tile the batch
    assert L.mkzg_verify_blob_kzg_proof_batch_exchange(comm, arr, cj, pj, 0, 0) == 0                                   # an empty batch verifies (kzg_verify.go:121-123)
    # a non-canonical blob: the reference's input error, through the record
    nb = bytearray(blobs[3])
    nb[64:96] = R_MOD.to_bytes(32, "big")
    keep2 = list(keep)
    keep2[3] = C.create_string_buffer(bytes(nb), 131072)
    arr2 = (C.c_char_p * n)(*[C.cast(k, C.c_char_p) for k in keep2])
    want = gpu_ctx.VerifyBlobKZGProofBatch(blobs[:3] + [bytes(nb)] + blobs[4:], cms, prs)
    assert want not in (0, VERIFY_FAILED)
    assert L.mkzg_verify_blob_kzg_proof_batch_exchange(comm, arr2, cj, pj, n, n) == want
    # device-resident shard
    d_b = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_c = torch.frombuffer(bytearray(cj), dtype=torch.uint8).cuda()
    d_p = torch.frombuffer(bytearray(pj), dtype=torch.uint8).cuda()
    d_bad = torch.frombuffer(bytearray(bad), dtype=torch.uint8).cuda()
    st = torch.cuda.current_stream().cuda_stream
    assert L.mkzg_verify_blob_kzg_proof_batch_exchange_device(comm, d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, n, st) == 0
    assert L.mkzg_verify_blob_kzg_proof_batch_exchange_device(comm, d_b.data_ptr(), d_c.data_ptr(), d_bad.data_ptr(), n, n, st) == VERIFY_FAILED
    assert L.mkzg_verify_blob_kzg_proof_batch_exchange_device(comm, d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, n - 1, st) == product.ERR_BATCH_LENGTH


def test_cell_exchange_world_1(gpu_ctx, product, comm):
    L = product.lib()
    blobs = [rand_blob(7800 + s) for s in range(2)]
    rc, cms, _ = gpu_ctx.BlobToKZGCommitmentBatch(blobs)
    tuples = []
    for b, cm in zip(blobs, cms):
        rc, cells, proofs = gpu_ctx.ComputeCellsAndKZGProofs(b)
        assert rc == 0
        tuples += [(cm, i, cells[i], proofs[i]) for i in range(0, 128, 3)]
    n = len(tuples)
    keep = [C.create_string_buffer(t[2], 2048) for t in tuples]
    arr = (C.c_char_p * n)(*[C.cast(k, C.c_char_p) for k in keep])
    idx = (C.c_uint64 * n)(*[t[1] for t in tuples])
    cj, pj = b"".join(t[0] for t in tuples), b"".join(t[3] for t in tuples)
    assert L.mkzg_verify_cell_kzg_proof_batch_exchange(comm, cj, idx, arr, pj, n, n) == 0
    bad = pj[48:96] + pj[48:]
    assert L.mkzg_verify_cell_kzg_proof_batch_exchange(comm, cj, idx, arr, bad, n, n) == VERIFY_FAILED
    assert L.mkzg_verify_cell_kzg_proof_batch_exchange(comm, cj, idx, arr, pj, n, 2 * n) == product.ERR_BATCH_LENGTH
    idx2 = (C.c_uint64 * n)(*([t[1] for t in tuples[:-1]] + [128]))
    assert L.mkzg_verify_cell_kzg_proof_batch_exchange(comm, cj, idx2, arr, pj, n, n) == product.ERR_INVALID_CELL_ID
    # this rank's cells contiguous in HBM (round 5: mkzg_verify_cell_kzg_proof_batch_exchange_device)
    import torch
    d_cells = torch.frombuffer(bytearray(b"".join(t[2] for t in tuples)), dtype=torch.uint8).cuda()
    st = torch.cuda.current_stream().cuda_stream
    idx_p = C.cast(idx, C.c_void_p)
    assert L.mkzg_verify_cell_kzg_proof_batch_exchange_device(comm, cj, idx_p, d_cells.data_ptr(), pj, n, n, st) == 0
    assert L.mkzg_verify_cell_kzg_proof_batch_exchange_device(comm, cj, idx_p, d_cells.data_ptr(), bad, n, n, st) == VERIFY_FAILED
    assert L.mkzg_verify_cell_kzg_proof_batch_exchange_device(comm, cj, idx_p, d_cells.data_ptr(), pj, n, n + 3, st) == product.ERR_BATCH_LENGTH
    assert L.mkzg_verify_cell_kzg_proof_batch_exchange_device(comm, cj, C.cast(idx2, C.c_void_p), d_cells.data_ptr(), pj, n, n, st) == product.ERR_INVALID_CELL_ID
    assert L.mkzg_verify_cell_kzg_proof_batch_exchange_device(comm, cj, idx_p, None, pj, n, n, st) == product.ERR_BAD_ARGS     # a rank-local argument error, through the collectives
    d_cells[2048 * 5 + 64:2048 * 5 + 96] = torch.frombuffer(bytearray(R_MOD.to_bytes(32, "big")), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    assert L.mkzg_verify_cell_kzg_proof_batch_exchange_device(comm, cj, idx_p, d_cells.data_ptr(), pj, n, n, st) == 3                # a non-canonical cell: the reference's input error


def test_wrapping_a_foreign_communicator_is_refused_without_one(gpu_ctx, product):
    L = product.lib()
    h = C.c_void_p()
    assert L.mkzg_comm_from_nccl(gpu_ctx._h, None, 0, 1, C.byref(h)) == product.ERR_BAD_ARGS
    assert L.mkzg_comm_init(gpu_ctx._h, C.create_string_buffer(128), 3, 2, C.byref(h)) == product.ERR_BAD_ARGS
