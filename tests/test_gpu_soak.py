"""Every entry point of the reference's API called repeatedly on one context (results identical call after call, no device
memory growth): what a long-running consensus client does with the library (reference api.go Context is reused for the
lifetime of the process)."""
import pytest

from test_gpu_commit import rand_blob

pytestmark = pytest.mark.gpu


def test_repeated_calls_are_stable_and_leak_free(gpu_ctx):
    import torch
    ctx = gpu_ctx
    b0, b1 = rand_blob(900), rand_blob(901)
    rc, cm0 = ctx.BlobToKZGCommitment(b0)
    assert rc == 0
    rc, pr0 = ctx.ComputeBlobKZGProof(b0, cm0)
    assert rc == 0

    def one_round():
        rc, cm = ctx.BlobToKZGCommitment(b1)
        assert rc == 0
        rc, pr = ctx.ComputeBlobKZGProof(b1, cm)
        assert rc == 0
        assert ctx.VerifyBlobKZGProof(b1, cm, pr) == 0
        assert ctx.VerifyBlobKZGProofBatch([b0, b1], [cm0, cm], [pr0, pr]) == 0
        rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(b1)
        assert rc == 0
        ids = list(range(1, 128, 2))
        rc, rcells, rproofs = ctx.RecoverCellsAndComputeKZGProofs(ids, [cells[i] for i in ids])
        assert rc == 0 and rcells == cells and rproofs == proofs
        assert ctx.VerifyCellKZGProofBatch([cm] * 128, list(range(128)), cells, proofs) == 0
        bad = bytearray(pr)
        bad[7] ^= 1
        assert ctx.VerifyBlobKZGProof(b1, cm, bytes(bad)) != 0           # a failing call must not disturb the next ones
        return cm, pr, cells[5], proofs[77]

    first = one_round()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(12):
        assert one_round() == first
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert abs(free0 - free1) < (64 << 20), (free0, free1)
