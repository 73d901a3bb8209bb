"""mkzg_ctx_prepare: the reference's NewContext4096 precomputes everything at construction (api.go:90-149: the commit key, fk20.NewFK20
at :137, the recovery domain), so the FIRST call on a Go Context costs what every later one does; here the big tables are built by the
first call that needs them unless the caller asks for them up front.  Checked with small table widths (12-bit commitment table, 8-bit
FK20 table: 24 + 4.3 GB) so that the test runs beside the other modules' contexts: nothing is built before the call, the widths asked
for are built by it, a second call changes nothing, the first commitment / cells call afterwards is correct and no slower than the
second, and a mask with unknown bits is MKZG_ERR_BAD_ARGS."""
import time

import pytest

from test_gpu_commit import rand_blob

pytestmark = pytest.mark.gpu


def test_prepare_builds_the_tables_up_front(product, oracle_ctx):
    import torch
    torch.cuda.init()
    ctx = product.Context()
    try:
        assert ctx.set_option("commit_direct_bits", 12) == 0 and ctx.set_option("fk20_bits", 8) == 0
        assert ctx.get_option("commit_direct_bits_effective") == 0 and ctx.get_option("fk20_bits_effective") == 0
        assert ctx.prepare(8) == 17 and ctx.prepare(4 | 1) == 17                 # unknown bits
        free0 = torch.cuda.mem_get_info()[0]
        assert ctx.prepare(1) == 0
        assert ctx.get_option("commit_direct_bits_effective") == 12 and ctx.get_option("fk20_bits_effective") == 0
        free1 = torch.cuda.mem_get_info()[0]
        assert free0 - free1 > 20 << 30                                          # the 24 GB table is there
        assert ctx.prepare(2) == 0
        assert ctx.get_option("fk20_bits_effective") == 8
        free2 = torch.cuda.mem_get_info()[0]
        assert free1 - free2 > 4 << 30
        assert ctx.prepare(3) == 0 and ctx.prepare(0) == 0                       # idempotent; an empty mask is a no-op
        assert abs(torch.cuda.mem_get_info()[0] - free2) < (64 << 20)
        blob = rand_blob(4242)
        t0 = time.perf_counter()
        rc, cm = ctx.BlobToKZGCommitment(blob)
        t1 = time.perf_counter()
        rc2, cm2 = ctx.BlobToKZGCommitment(blob)
        t2 = time.perf_counter()
        assert rc == 0 and rc2 == 0 and cm == cm2 == oracle_ctx.blob_to_kzg_commitment(blob)[1]
        # the first call allocates its lane's workspaces (milliseconds); it does not build a 24 GB table (a quarter of a second and more)
        assert t1 - t0 < 0.1, (t1 - t0, t2 - t1)
        t0 = time.perf_counter()
        rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(blob)
        t1 = time.perf_counter()
        assert rc == 0 and t1 - t0 < 0.15, t1 - t0
        rco, oc, op = oracle_ctx.compute_cells_and_kzg_proofs(blob)
        assert rco == 0 and b"".join(cells) == oc and b"".join(proofs) == op
        half = list(range(0, 128, 2))
        rc, rcells, rproofs = ctx.RecoverCellsAndComputeKZGProofs(half, [cells[j] for j in half])
        assert rc == 0 and rcells == cells and rproofs == proofs
    finally:
        ctx.close()


def test_prepare_with_the_bucket_method(product, oracle_ctx):
    """commit_direct_bits 0: prepare builds the 10.5 MB window table of the bucket method; same bytes out"""
    ctx = product.Context()
    try:
        assert ctx.set_option("commit_direct_bits", 0) == 0
        assert ctx.prepare(1) == 0 and ctx.get_option("commit_direct_bits_effective") == 0
        blob = rand_blob(4243)
        rc, cm = ctx.BlobToKZGCommitment(blob)
        assert rc == 0 and cm == oracle_ctx.blob_to_kzg_commitment(blob)[1]
    finally:
        ctx.close()
