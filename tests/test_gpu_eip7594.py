"""GPU parity tests of the EIP-7594 paths through the C ABI (reference consensus_specs_test.go:361-519)."""
import pytest

from test_gpu_commit import rand_blob

pytestmark = pytest.mark.gpu


def _bad(*vals):
    from conftest import BadHex
    return any(isinstance(v, BadHex) for v in vals)


def test_compute_cells_vectors(vectors, gpu_ctx):
    n_ok = 0
    for c in vectors.cases("compute_cells_and_kzg_proofs"):
        blob = vectors.val(c["input"]["blob"])
        exp = vectors.val(c["output"])
        if _bad(blob) or len(blob) != 131072:
            assert exp is None
            continue
        rc, cells = gpu_ctx.ComputeCells(blob)
        if exp is None:
            assert rc != 0, c["name"]
        else:
            assert rc == 0 and cells == exp[0], c["name"]
            n_ok += 1
    assert n_ok == 7


def test_compute_cells_vs_oracle(gpu_ctx, oracle_ctx):
    for seed in (42, 43):
        blob = rand_blob(seed)
        rc, cells = gpu_ctx.ComputeCells(blob)
        rco, exp = oracle_ctx.compute_cells(blob)
        assert rc == 0 and rco == 0 and b"".join(cells) == exp


def test_compute_cells_and_kzg_proofs_vectors(vectors, gpu_ctx):
    n_ok = 0
    for c in vectors.cases("compute_cells_and_kzg_proofs"):
        blob = vectors.val(c["input"]["blob"])
        exp = vectors.val(c["output"])
        if _bad(blob) or len(blob) != 131072:
            assert exp is None
            continue
        rc, cells, proofs = gpu_ctx.ComputeCellsAndKZGProofs(blob)
        if exp is None:
            assert rc != 0, c["name"]
        else:
            assert rc == 0 and cells == exp[0], c["name"]
            assert proofs == exp[1], c["name"]
            n_ok += 1
    assert n_ok == 7


def test_compute_cells_and_kzg_proofs_vs_oracle(gpu_ctx, oracle_ctx):
    blob = rand_blob(42)                                    # reference bench_eip7594_test.go:11
    rc, cells, proofs = gpu_ctx.ComputeCellsAndKZGProofs(blob)
    rco, ec, ep = oracle_ctx.compute_cells_and_kzg_proofs(blob)
    assert rc == 0 and rco == 0
    assert b"".join(cells) == ec and b"".join(proofs) == ep
