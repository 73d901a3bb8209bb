"""Single-blob calls of the reference API from many threads on one context (verify.go:152-169, examples_test.go:11): the library serves
calls that find others in flight as groups through its batch paths (go-eth-kzg_amd/csrc/host_combine.hpp).  Every member must get
exactly what its own call would have returned alone: the oracle's bytes, ITS verdict, ITS input error -- whatever else is in the group."""
import ctypes as C
import threading

import pytest

pytestmark = pytest.mark.gpu

K = 24          # caller threads: more than the two groups in flight, so groups of many members form
BLOB = 131072


def _run_threads(fn, k=K):
    errors = []

    def wrap(i):
        try:
            fn(i)
        except Exception:      # noqa: BLE001
            import traceback
            errors.append(traceback.format_exc())

    ts = [threading.Thread(target=wrap, args=(i,)) for i in range(k)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[0]


def test_combined_single_blob_calls_match_the_oracle(gpu_ctx, product, oracle_ctx):
    import oracle
    L = product.lib()
    blobs = [oracle.rand_blob(40_000 + i) for i in range(6)]
    exp_cm = [oracle_ctx.blob_to_kzg_commitment(b)[1] for b in blobs]
    exp_pr = [oracle_ctx.compute_blob_kzg_proof(b, c)[1] for b, c in zip(blobs, exp_cm)]
    exp_cells = [oracle_ctx.compute_cells_and_kzg_proofs(b) for b in blobs[:2]]
    noncanon = bytearray(blobs[1])
    noncanon[32 * 77:32 * 78] = b"\xff" * 32
    noncanon = bytes(noncanon)
    bad_cm = b"\x00" + exp_cm[2][1:]                     # compressed flag cleared: a decoding error
    assert gpu_ctx.get_option("combine_single_calls") == 1
    before = gpu_ctx.get_option("combined_calls")

    def commitments(k):
        out = C.create_string_buffer(48)
        for it in range(12):
            i = (k + it) % 6
            if (k + it) % 11 == 5:                        # a non-canonical blob among the group's members: ITS error, nobody else's
                assert L.mkzg_blob_to_kzg_commitment(gpu_ctx._h, noncanon, 0, out) == 3
            else:
                assert L.mkzg_blob_to_kzg_commitment(gpu_ctx._h, blobs[i], 0, out) == 0 and out.raw == exp_cm[i], (k, it)
        assert L.mkzg_blob_to_kzg_commitment(gpu_ctx._h, None, 0, out) == 6

    def verifications(k):
        for it in range(12):
            i = (k + it) % 6
            what = (k * 5 + it) % 9
            if what == 3:                                 # another blob's proof: this member fails, the rest of its group must not
                assert L.mkzg_verify_blob_kzg_proof(gpu_ctx._h, blobs[i], exp_cm[i], exp_pr[(i + 1) % 6]) == 1
            elif what == 6:
                assert L.mkzg_verify_blob_kzg_proof(gpu_ctx._h, noncanon, exp_cm[1], exp_pr[1]) == 3
            elif what == 8:
                assert L.mkzg_verify_blob_kzg_proof(gpu_ctx._h, blobs[2], bad_cm, exp_pr[2]) == 11
            else:
                assert L.mkzg_verify_blob_kzg_proof(gpu_ctx._h, blobs[i], exp_cm[i], exp_pr[i]) == 0, (k, it)

    def cells(k):
        c_out = C.create_string_buffer(128 * 2048)
        p_out = C.create_string_buffer(128 * 48)
        for it in range(4):
            i = (k + it) % 2
            if (k + it) % 7 == 3:
                assert L.mkzg_compute_cells_and_kzg_proofs(gpu_ctx._h, noncanon, 0, c_out, p_out) == 3
            else:
                assert L.mkzg_compute_cells_and_kzg_proofs(gpu_ctx._h, blobs[i], 0, c_out, p_out) == 0
                assert c_out.raw == exp_cells[i][1] and p_out.raw == exp_cells[i][2], (k, it)

    def mixed(k):
        (commitments, verifications, cells)[k % 3](k)

    for fn in (commitments, verifications, cells, mixed):
        _run_threads(fn)
    assert gpu_ctx.get_option("combined_calls") > before, "24 threads of single-blob calls never met in a group"
    # the same with the combiner switched off: every call on its own (what the library did until round 5)
    assert gpu_ctx.set_option("combine_single_calls", 0) == 0
    try:
        served = gpu_ctx.get_option("combined_calls")
        _run_threads(mixed, k=9)
        assert gpu_ctx.get_option("combined_calls") == served
    finally:
        assert gpu_ctx.set_option("combine_single_calls", 1) == 0


def test_a_lone_caller_is_a_group_of_one(gpu_ctx, product, oracle_ctx):
    """no other call in flight: the single-item path as before, nothing counted as combined"""
    import oracle
    L = product.lib()
    blob = oracle.rand_blob(41_000)
    rc, cm = gpu_ctx.BlobToKZGCommitment(blob)
    assert rc == 0 and cm == oracle_ctx.blob_to_kzg_commitment(blob)[1]
    rc, pr = gpu_ctx.ComputeBlobKZGProof(blob, cm)
    before = gpu_ctx.get_option("combined_calls")
    for _ in range(5):
        assert gpu_ctx.VerifyBlobKZGProof(blob, cm, pr) == 0
        assert gpu_ctx.VerifyBlobKZGProof(blob, cm, cm) == 1
        assert gpu_ctx.BlobToKZGCommitment(blob) == (0, cm)
    assert gpu_ctx.get_option("combined_calls") == before
