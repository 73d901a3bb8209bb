"""Pin the CPU oracle on every consensus-spec vector the reference ships.

Mirrors the reference's consensus_specs_test.go:32-519 rule set:
  * output null            => the call must return an input error
  * compute APIs           => output bytes compared byte-for-byte
  * verifiers              => output true <=> rc == 0, false <=> rc == VERIFY_FAILED (1);
                              any other error must coincide with output null
                              (consensus_specs_test.go:219-226, 349-356, 457-464)
"""
import pytest

VERIFY_FAILED = 1


def _bad(*vals):
    """fixed-size byte arrays are part of the Go API's types; a wrong-length input cannot even
    be passed (the reference tests treat the hex-decode/length failure as the expected error)."""
    from conftest import BadHex
    return any(isinstance(v, BadHex) for v in vals)


def _check_verifier(rc, expected, name):
    if expected is None:
        assert rc not in (0, VERIFY_FAILED), f"{name}: expected input error, got rc={rc}"
    elif expected is True:
        assert rc == 0, f"{name}: expected accept, got rc={rc}"
    else:
        assert rc == VERIFY_FAILED, f"{name}: expected reject, got rc={rc}"


def test_blob_to_kzg_commitment(vectors, oracle_ctx):
    n_ok = 0
    for c in vectors.cases("blob_to_kzg_commitment"):
        blob = vectors.val(c["input"]["blob"])
        exp = vectors.val(c["output"])
        if _bad(blob):
            assert exp is None
            continue
        rc, out = oracle_ctx.blob_to_kzg_commitment(blob)
        if exp is None:
            assert rc != 0, c["name"]
        else:
            assert rc == 0 and out == exp, c["name"]
            n_ok += 1
    assert n_ok == 7


def test_compute_kzg_proof(vectors, oracle_ctx):
    n_ok = 0
    for c in vectors.cases("compute_kzg_proof"):
        blob, z = vectors.val(c["input"]["blob"]), vectors.val(c["input"]["z"])
        exp = vectors.val(c["output"])
        if _bad(blob, z):
            assert exp is None
            continue
        rc, proof, y = oracle_ctx.compute_kzg_proof(blob, z)
        if exp is None:
            assert rc != 0, c["name"]
        else:
            assert rc == 0 and proof == exp[0] and y == exp[1], c["name"]
            n_ok += 1
    assert n_ok > 30


def test_compute_blob_kzg_proof(vectors, oracle_ctx):
    n_ok = 0
    for c in vectors.cases("compute_blob_kzg_proof"):
        blob, cm = vectors.val(c["input"]["blob"]), vectors.val(c["input"]["commitment"])
        exp = vectors.val(c["output"])
        if _bad(blob, cm):
            assert exp is None
            continue
        rc, proof = oracle_ctx.compute_blob_kzg_proof(blob, cm)
        if exp is None:
            assert rc != 0, c["name"]
        else:
            assert rc == 0 and proof == exp, c["name"]
            n_ok += 1
    assert n_ok == 7


def test_verify_kzg_proof(vectors, oracle_ctx):
    for c in vectors.cases("verify_kzg_proof"):
        i = c["input"]
        cm, z, y, pr = (vectors.val(i[k]) for k in ("commitment", "z", "y", "proof"))
        exp = vectors.val(c["output"])
        if _bad(cm, z, y, pr):
            assert exp is None
            continue
        _check_verifier(oracle_ctx.verify_kzg_proof(cm, z, y, pr), exp, c["name"])


def test_verify_blob_kzg_proof(vectors, oracle_ctx):
    for c in vectors.cases("verify_blob_kzg_proof"):
        i = c["input"]
        blob, cm, pr = (vectors.val(i[k]) for k in ("blob", "commitment", "proof"))
        exp = vectors.val(c["output"])
        if _bad(blob, cm, pr):
            assert exp is None
            continue
        _check_verifier(oracle_ctx.verify_blob_kzg_proof(blob, cm, pr), exp, c["name"])


def test_verify_blob_kzg_proof_batch(vectors, oracle_ctx):
    for c in vectors.cases("verify_blob_kzg_proof_batch"):
        i = c["input"]
        blobs, cms, prs = (vectors.val(i[k]) for k in ("blobs", "commitments", "proofs"))
        exp = vectors.val(c["output"])
        if _bad(*blobs, *cms, *prs):
            assert exp is None
            continue
        _check_verifier(oracle_ctx.verify_blob_kzg_proof_batch(blobs, cms, prs), exp, c["name"])


def test_compute_cells_and_kzg_proofs(vectors, oracle_ctx):
    n_ok = 0
    for c in vectors.cases("compute_cells_and_kzg_proofs"):
        blob = vectors.val(c["input"]["blob"])
        exp = vectors.val(c["output"])
        if _bad(blob):
            assert exp is None
            continue
        rc, cells, proofs = oracle_ctx.compute_cells_and_kzg_proofs(blob)
        rc2, cells2 = oracle_ctx.compute_cells(blob)
        if exp is None:
            assert rc != 0 and rc2 != 0, c["name"]
        else:
            assert rc == 0 and rc2 == 0
            assert cells == b"".join(exp[0]), c["name"]
            assert proofs == b"".join(exp[1]), c["name"]
            assert cells2 == cells        # consensus_specs_test.go:389-397
            n_ok += 1
    assert n_ok == 7


def test_verify_cell_kzg_proof_batch(vectors, oracle_ctx):
    for c in vectors.cases("verify_cell_kzg_proof_batch"):
        i = c["input"]
        cms, idx, cells, prs = (vectors.val(i[k]) for k in ("commitments", "cell_indices", "cells", "proofs"))
        exp = vectors.val(c["output"])
        if _bad(*cms, *cells, *prs):
            assert exp is None
            continue
        _check_verifier(oracle_ctx.verify_cell_kzg_proof_batch(cms, idx, cells, prs), exp, c["name"])


def test_recover_cells_and_kzg_proofs(vectors, oracle_ctx):
    n_ok = 0
    for c in vectors.cases("recover_cells_and_kzg_proofs"):
        i = c["input"]
        ids, cells = vectors.val(i["cell_indices"]), vectors.val(i["cells"])
        exp = vectors.val(c["output"])
        if _bad(*cells):
            assert exp is None
            continue
        rc, oc, op = oracle_ctx.recover_cells_and_kzg_proofs(ids, cells)
        rc2, oc2 = oracle_ctx.recover_cells(ids, cells)
        if exp is None:
            assert rc != 0 and rc2 != 0, c["name"]
        else:
            assert rc == 0 and oc == b"".join(exp[0]) and op == b"".join(exp[1]), c["name"]
            assert rc2 == 0 and oc2 == oc
            n_ok += 1
    assert n_ok >= 4
