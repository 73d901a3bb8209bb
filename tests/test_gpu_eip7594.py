"""GPU parity tests of the EIP-7594 paths through the C ABI (reference consensus_specs_test.go:361-519)."""
import pytest

from test_gpu_commit import R, rand_blob

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


def _check_verifier(rc, expected, name):
    if expected is None:
        assert rc not in (0, 1), f"{name}: expected input error, got rc={rc}"
    elif expected is True:
        assert rc == 0, f"{name}: expected accept, got rc={rc}"
    else:
        assert rc == 1, f"{name}: expected reject, got rc={rc}"


@pytest.mark.parametrize("host_validate", [1, 0], ids=["host_validation", "device_validation"])
def test_verify_cell_kzg_proof_batch_vectors(vectors, gpu_ctx, host_validate):
    """option host_validate (round 6, default 1): calls whose distinct commitments + proofs are at most 96 points decode and subgroup-test
    them on the host pool (host_validate.hpp; a column sidecar is one cell of every blob of the block), larger ones and option 0 on the
    device -- the 30 vectors (invalid commitments and proofs among them) under both"""
    assert gpu_ctx.set_option("host_validate", host_validate) == 0
    try:
        for c in vectors.cases("verify_cell_kzg_proof_batch"):
            i = c["input"]
            cms, idx, cells, prs = (vectors.val(i[k]) for k in ("commitments", "cell_indices", "cells", "proofs"))
            exp = vectors.val(c["output"])
            if _bad(*cms, *cells, *prs) or any(len(x) != 48 for x in cms) or any(len(x) != 2048 for x in cells) or any(len(x) != 48 for x in prs):
                assert exp is None
                continue
            _check_verifier(gpu_ctx.VerifyCellKZGProofBatch(cms, idx, cells, prs), exp, c["name"])
    finally:
        assert gpu_ctx.set_option("host_validate", 1) == 0


def test_small_cell_batches_under_both_validations(gpu_ctx):
    """Column-sidecar-sized calls (EIP-7594: one cell of each of a block's 6 / 9 blobs; 8 columns of them; 95 + 1 and 96 + 1 points, the
    boundary of the host validation): accept, reject a swapped proof, and answer a curve point outside the r-torsion -- as a proof and as a
    commitment -- with the decoder's error, identically with the points validated on the host and on the device.  Reference:
    api_eip7594.go:163-214, serialization.go:108-115."""
    from test_gpu_commit import rand_blob
    from test_mont29_host import P, R, ec_mul, compress
    blobs = [rand_blob(6100 + i) for i in range(3)]
    data = []
    for b in blobs:
        rc, cm = gpu_ctx.BlobToKZGCommitment(b)
        rc2, cells, proofs = gpu_ctx.ComputeCellsAndKZGProofs(b)
        assert rc == 0 and rc2 == 0
        data.append((cm, cells, proofs))
    x = 3
    while True:
        x += 1
        rhs = (x * x * x + 4) % P
        y = pow(rhs, (P + 1) // 4, P)
        if y * y % P == rhs and ec_mul(R, (x, y)) is not None:
            break
    outside = compress((x, y))

    def batch(n_blobs, cols):
        c, idx, ce, pr = [], [], [], []
        for col in cols:
            for b in range(n_blobs):
                c.append(data[b][0]); idx.append(col); ce.append(data[b][1][col]); pr.append(data[b][2][col])
        return c, idx, ce, pr

    shapes = [batch(3, [9]), batch(3, list(range(0, 128, 16))), batch(1, list(range(95))), batch(1, list(range(96))), batch(3, list(range(31))), batch(3, list(range(32)))]
    results = {}
    for hv in (1, 0):
        assert gpu_ctx.set_option("host_validate", hv) == 0
        res = []
        for c, idx, ce, pr in shapes:
            res.append(gpu_ctx.VerifyCellKZGProofBatch(c, idx, ce, pr))
            res.append(gpu_ctx.VerifyCellKZGProofBatch(c, idx, ce, pr[1:] + pr[:1]))
            bad = list(pr)
            bad[len(bad) // 2] = outside
            res.append(gpu_ctx.VerifyCellKZGProofBatch(c, idx, ce, bad))
            badc = list(c)
            badc[-1] = outside
            res.append(gpu_ctx.VerifyCellKZGProofBatch(badc, idx, ce, pr))
        results[hv] = res
    assert gpu_ctx.set_option("host_validate", 1) == 0
    assert results[1] == results[0], (results[1], results[0])
    for k in range(len(shapes)):
        assert results[1][4 * k: 4 * k + 4] == [0, 1, 13, 13], (k, results[1][4 * k: 4 * k + 4])


def test_verify_cell_kzg_proof_batch_vectors_device_cells(vectors, gpu_ctx):
    """the same 30 vectors through mkzg_verify_cell_kzg_proof_batch_device: the cells contiguous in HBM, commitments / indices / proofs on
    the host; and for the valid ones the sharded record of the device form equals the host-pointer form's for a fixed r"""
    import torch
    n_run = 0
    for c in vectors.cases("verify_cell_kzg_proof_batch"):
        i = c["input"]
        cms, idx, cells, prs = (vectors.val(i[k]) for k in ("commitments", "cell_indices", "cells", "proofs"))
        exp = vectors.val(c["output"])
        if _bad(*cms, *cells, *prs) or any(len(x) != 48 for x in cms) or any(len(x) != 2048 for x in cells) or any(len(x) != 48 for x in prs):
            continue
        if len(cells) != len(idx) or not cells:
            continue                                          # length mismatches and the empty batch have no device form of their own
        d_cells = torch.frombuffer(bytearray(b"".join(cells)), dtype=torch.uint8).cuda()
        rc = gpu_ctx.verify_cell_kzg_proof_batch_device(cms, idx, d_cells.data_ptr(), prs, torch.cuda.current_stream().cuda_stream)
        _check_verifier(rc, exp, c["name"])
        assert rc == gpu_ctx.VerifyCellKZGProofBatch(cms, idx, cells, prs), c["name"]
        if exp is True:
            r32 = bytes(range(1, 33))
            rc1, rec1 = gpu_ctx.verify_cell_batch_partial_device(cms, idx, d_cells.data_ptr(), prs, 5, r32)
            rc2, rec2 = gpu_ctx.verify_cell_batch_partial(cms, idx, cells, prs, 5, r32)
            assert rc1 == 0 and rc2 == 0 and rec1 == rec2
        n_run += 1
    assert n_run >= 20


def test_verify_cells_two_blobs(gpu_ctx):
    """cells of two blobs, interleaved (dedup path), then a wrong cell must be rejected"""
    items = []
    for seed in (5, 6):
        blob = rand_blob(seed)
        rc, cm = gpu_ctx.BlobToKZGCommitment(blob)
        rc2, cells, proofs = gpu_ctx.ComputeCellsAndKZGProofs(blob)
        assert rc == 0 and rc2 == 0
        items += [(cm, i, cells[i], proofs[i]) for i in range(0, 128, 3)]
    items = items[::2] + items[1::2]
    cms, idx, cells, prs = map(list, zip(*items))
    assert gpu_ctx.VerifyCellKZGProofBatch(cms, idx, cells, prs) == 0
    cells[5], cells[6] = cells[6], cells[5]
    assert gpu_ctx.VerifyCellKZGProofBatch(cms, idx, cells, prs) == 1
    assert gpu_ctx.VerifyCellKZGProofBatch(cms, idx[:-1], cells, prs) == 2
    assert gpu_ctx.VerifyCellKZGProofBatch([], [], [], []) == 0


def test_recover_cells_and_kzg_proofs_vectors(vectors, gpu_ctx):
    n_ok = 0
    for c in vectors.cases("recover_cells_and_kzg_proofs"):
        i = c["input"]
        ids, cells = vectors.val(i["cell_indices"]), vectors.val(i["cells"])
        exp = vectors.val(c["output"])
        if _bad(*cells) or any(len(x) != 2048 for x in cells):
            assert exp is None
            continue
        rc, oc, op = gpu_ctx.RecoverCellsAndComputeKZGProofs(ids, cells)
        rc2, oc2 = gpu_ctx.RecoverCells(ids, cells)
        if exp is None:
            assert rc != 0 and rc2 != 0, c["name"]
        else:
            assert rc == 0 and oc == exp[0] and op == exp[1], c["name"]
            assert rc2 == 0 and oc2 == oc
            n_ok += 1
    assert n_ok >= 4


def test_recover_patterns_vs_oracle(gpu_ctx, oracle_ctx):
    """the reference bench pattern (even cells, bench_eip7594_test.go:64-69) and ragged subsets"""
    import random
    blob = rand_blob(42)
    rc, cells, proofs = gpu_ctx.ComputeCellsAndKZGProofs(blob)
    assert rc == 0
    rng = random.Random(3)
    subsets = [list(range(0, 128, 2)), list(range(64)), list(range(64, 128)), sorted(rng.sample(range(128), 64)), sorted(rng.sample(range(128), 100)), list(range(128))]
    for ids in subsets:
        rc, oc, op = gpu_ctx.RecoverCellsAndComputeKZGProofs(ids, [cells[i] for i in ids])
        assert rc == 0 and oc == cells and op == proofs, ids[:4]
    ids = subsets[0]
    assert gpu_ctx.RecoverCells(ids[:63], [cells[i] for i in ids[:63]])[0] == 10          # ErrNotEnoughCellsForReconstruction
    assert gpu_ctx.RecoverCells(ids[::-1], [cells[i] for i in ids[::-1]])[0] == 8          # ErrCellIDsNotOrdered
    assert gpu_ctx.RecoverCells(ids, [cells[i] for i in ids[:-1]])[0] == 7                 # ErrNumCellIDsNotEqualNumCells
    assert gpu_ctx.RecoverCells(ids[:-1] + [128], [cells[i] for i in ids])[0] == 9         # ErrFoundInvalidCellID


def test_sharded_cell_verifier_records(gpu_ctx, oracle_ctx):
    """mkzg_verify_cell_kzg_proof_batch_partial / _finish (multi-GPU form): with the same r and offsets the product's 512-byte
    records equal the oracle's byte for byte, and the combined verdicts agree (valid, corrupted proof, bad cell)"""
    tuples = []
    for seed in (61, 62):
        blob = rand_blob(seed)
        rc, cm = gpu_ctx.BlobToKZGCommitment(blob)
        rc2, cells, proofs = gpu_ctx.ComputeCellsAndKZGProofs(blob)
        assert rc == 0 and rc2 == 0
        for ci in (0, 5, 64, 127, 33, 2, 99):
            tuples.append((cm, ci, cells[ci], proofs[ci]))
    r32 = bytes(range(7, 39))
    cut = 9                                                   # ranks hold 9 and 5 cells; a commitment row spans the cut

    def records(eng, tups):
        recs = []
        for lo, hi in ((0, cut), (cut, len(tups))):
            sh = tups[lo:hi]
            rc, rec = eng.verify_cell_batch_partial([t[0] for t in sh], [t[1] for t in sh], [t[2] for t in sh], [t[3] for t in sh], lo, r32)
            recs.append((rc, rec))
        return recs

    g, o = records(gpu_ctx, tuples), records(oracle_ctx, tuples)
    assert [x[0] for x in g] == [0, 0] and g == o
    assert gpu_ctx.verify_cell_batch_finish([x[1] for x in g]) == 0
    assert oracle_ctx.verify_cell_batch_finish([x[1] for x in g]) == 0
    assert gpu_ctx.verify_cell_batch_finish([g[1][1]]) == 0            # a shard alone is a valid (r^offset-scaled) batch as well
    bad = list(tuples)
    bad[11] = (bad[11][0], bad[11][1], bad[11][2], bad[12][3])
    gb = records(gpu_ctx, bad)
    assert gb == records(oracle_ctx, bad)
    assert gpu_ctx.verify_cell_batch_finish([x[1] for x in gb]) == 1
    nc = list(tuples)
    nc[3] = (nc[3][0], nc[3][1], b"\xff" * 2048, nc[3][3])
    gn = records(gpu_ctx, nc)
    assert gn[0][0] == 3 and gpu_ctx.verify_cell_batch_finish([x[1] for x in gn]) == 3
    # an empty shard contributes the identity
    rc, rec = gpu_ctx.verify_cell_batch_partial([], [], [], [], 14, r32)
    assert rc == 0 and gpu_ctx.verify_cell_batch_finish([g[0][1], g[1][1], rec]) == 0


@pytest.mark.parametrize("verify_shape", [0, 1])
def test_sharded_cell_verifier_records_256(gpu_ctx, oracle_ctx, verify_shape):
    """all 256 cells of two blobs in two shards (200 + 56; more than 16 points per MSM, so the bucket kernels run, not the few-point one):
    records equal the oracle's byte for byte whichever shape the context is forced into; a swapped proof is rejected under both"""
    tuples = []
    for seed in (71, 72):
        blob = rand_blob(seed)
        rc, cm = gpu_ctx.BlobToKZGCommitment(blob)
        rc2, cells, proofs = gpu_ctx.ComputeCellsAndKZGProofs(blob)
        assert rc == 0 and rc2 == 0
        tuples += [(cm, ci, cells[ci], proofs[ci]) for ci in range(128)]
    r32 = bytes(range(11, 43))
    cut = 200

    def records(eng, tups):
        recs = []
        for lo, hi in ((0, cut), (cut, len(tups))):
            sh = tups[lo:hi]
            recs.append(eng.verify_cell_batch_partial([t[0] for t in sh], [t[1] for t in sh], [t[2] for t in sh], [t[3] for t in sh], lo, r32))
        return recs

    o = records(oracle_ctx, tuples)
    assert oracle_ctx.verify_cell_batch_finish([x[1] for x in o]) == 0
    prev = gpu_ctx.get_option("verify_shape")
    gpu_ctx.set_option("verify_shape", verify_shape)
    try:
        g = records(gpu_ctx, tuples)
        assert [x[0] for x in g] == [0, 0] and g == o
        assert gpu_ctx.verify_cell_batch_finish([x[1] for x in g]) == 0
        cms, idx, cells, prs = map(list, zip(*tuples))
        assert gpu_ctx.VerifyCellKZGProofBatch(cms, idx, cells, prs) == 0
        prs[40], prs[41] = prs[41], prs[40]
        assert gpu_ctx.VerifyCellKZGProofBatch(cms, idx, cells, prs) == 1
    finally:
        gpu_ctx.set_option("verify_shape", prev)


def test_recover_batch(gpu_ctx, oracle_ctx):
    """mkzg_recover_cells_and_kzg_proofs_batch: ragged jobs (different erasure patterns per blob, one invalid job in the middle) give the
    same bytes as the single-blob call and the oracle"""
    jobs, expect = [], []
    patterns = [list(range(0, 128, 2)), list(range(64, 128)), list(range(0, 100)), [i for i in range(128) if i % 5 != 0][:90], list(range(128))]
    for k, ids in enumerate(patterns):
        blob = rand_blob(80 + k)
        rc, cells, proofs = gpu_ctx.ComputeCellsAndKZGProofs(blob)
        assert rc == 0
        jobs.append((ids, [cells[i] for i in ids]))
        expect.append((cells, proofs))
    bad_job = (list(range(63)), [expect[0][0][i] for i in range(63)])            # too few cells
    jobs.insert(2, bad_job)
    expect.insert(2, None)
    rc, res = gpu_ctx.RecoverCellsAndComputeKZGProofsBatch(jobs)
    assert rc == 10                                                               # MKZG_ERR_NOT_ENOUGH_CELLS: first failing job
    for (st, oc, op), exp, job in zip(res, expect, jobs):
        if exp is None:
            assert st == 10 and oc is None
            continue
        assert st == 0 and oc == exp[0] and op == exp[1]
        rc1, oc1, op1 = gpu_ctx.RecoverCellsAndComputeKZGProofs(job[0], job[1])
        assert rc1 == 0 and oc1 == oc and op1 == op
    rco, oco, opo = oracle_ctx.recover_cells_and_kzg_proofs(jobs[1][0], jobs[1][1])
    assert rco == 0 and b"".join(res[1][1]) == oco and b"".join(res[1][2]) == opo
    # a non-canonical field element inside one job only fails that job
    poisoned = list(jobs[0][1])
    poisoned[3] = b"\xff" * 2048
    rc, res2 = gpu_ctx.RecoverCellsAndComputeKZGProofsBatch([(jobs[0][0], poisoned), jobs[1]])
    assert rc == 3 and res2[0][0] == 3 and res2[1][0] == 0 and res2[1][1] == res[1][1]


def test_recover_batch_device(gpu_ctx, oracle_ctx):
    """mkzg_recover_cells_and_kzg_proofs_batch_device: supplied cells concatenated in HBM, results left in HBM; ragged id sets, one blob with
    too few cells and one with unordered ids in the middle (their cells still occupy their place in the caller's buffer), cells-only form"""
    import torch
    patterns = [list(range(0, 128, 2)), list(range(64)), list(range(64, 128)), list(range(10, 90)), list(range(63)), [1, 0] + list(range(2, 70)), list(range(128)),
                sorted(set(range(0, 128, 3)) | set(range(1, 128, 3)))]
    blobs = [rand_blob(4400 + i) for i in range(len(patterns))]
    full = [oracle_ctx.compute_cells_and_kzg_proofs(b) for b in blobs]
    supplied = b"".join(b"".join(full[k][1][2048 * i:2048 * i + 2048] for i in ids) for k, ids in enumerate(patterns))
    d_in = torch.frombuffer(bytearray(supplied), dtype=torch.uint8).cuda()
    nb = len(patterns)
    d_oc = torch.zeros(nb * 262144, dtype=torch.uint8, device="cuda")
    d_op = torch.zeros(nb * 6144, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rc, sts = gpu_ctx.recover_batch_device(patterns, d_in.data_ptr(), d_oc.data_ptr(), d_op.data_ptr(), st)
    assert sts[4] == 10 and sts[5] == 8 and rc == 10                 # ErrNotEnoughCellsForReconstruction, ErrCellIDsNotOrdered
    oc, op = d_oc.cpu().numpy().tobytes(), d_op.cpu().numpy().tobytes()
    for k in range(nb):
        if k in (4, 5):
            continue
        assert sts[k] == 0
        assert oc[k * 262144:(k + 1) * 262144] == full[k][1] and op[k * 6144:(k + 1) * 6144] == full[k][2], k
    d_oc2 = torch.zeros(nb * 262144, dtype=torch.uint8, device="cuda")
    rc, sts2 = gpu_ctx.recover_batch_device(patterns, d_in.data_ptr(), d_oc2.data_ptr(), 0, st)
    assert sts2 == sts
    oc2 = d_oc2.cpu().numpy().tobytes()
    for k in range(nb):
        if k not in (4, 5):
            assert oc2[k * 262144:(k + 1) * 262144] == oc[k * 262144:(k + 1) * 262144]


def test_cells_and_proofs_async_on_two_streams(gpu_ctx, oracle_ctx):
    """mkzg_compute_cells_and_kzg_proofs_batch_device returns before the GPU is done and shares its workspaces between calls: two calls
    issued back to back on different streams must not clobber each other (the library orders them with an event)"""
    import torch
    blobs_a = [rand_blob(90 + i) for i in range(3)]
    blobs_b = [rand_blob(95 + i) for i in range(3)]
    da = torch.frombuffer(bytearray(b"".join(blobs_a)), dtype=torch.uint8).cuda()
    db = torch.frombuffer(bytearray(b"".join(blobs_b)), dtype=torch.uint8).cuda()
    outs = []
    streams = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)]
    torch.cuda.synchronize()
    for d, s in ((da, streams[0]), (db, streams[1])):
        cells = torch.zeros(3 * 262144, dtype=torch.uint8, device="cuda")
        proofs = torch.zeros(3 * 128 * 48, dtype=torch.uint8, device="cuda")
        st = torch.zeros(3, dtype=torch.int32, device="cuda")
        assert gpu_ctx.compute_cells_and_kzg_proofs_batch_device(d.data_ptr(), 3, cells.data_ptr(), proofs.data_ptr(), st.data_ptr(), s.cuda_stream) == 0
        outs.append((cells, proofs, st))
    torch.cuda.synchronize()
    for (cells, proofs, st), blobs in zip(outs, (blobs_a, blobs_b)):
        assert st.cpu().tolist() == [0, 0, 0]
        cb, pb = bytes(cells.cpu().tolist()), bytes(proofs.cpu().tolist())
        for i, b in enumerate(blobs):
            rc, oc, op = oracle_ctx.compute_cells_and_kzg_proofs(b)
            assert rc == 0 and cb[i * 262144:(i + 1) * 262144] == oc and pb[i * 6144:(i + 1) * 6144] == op


def test_structured_blobs_through_every_lane_shape(gpu_ctx, oracle_ctx):
    """Blobs whose FK20 intermediates collide -- the zero polynomial (every point the identity), constant polynomials (all but one
    coefficient zero: the addition networks meet P + P, P - P and identity operands), a blob with one non-zero evaluation, all r - 1, an
    arithmetic progression -- through ComputeCellsAndKZGProofs alone (quad of lanes per item), in a batch of 6 (quad), of 20 (pair of
    lanes in the products) and of 40 (one lane): the exceptional cases of g1_quad.hpp must give the oracle's bytes like the plain law."""
    import torch
    const7 = (7).to_bytes(32, "big") * 4096
    const_big = (R - 2).to_bytes(32, "big") * 4096
    one_hot = bytearray(131072)
    one_hot[32 * 1234 + 31] = 5
    ramp = b"".join(((i * 0x1000193 + 11) % R).to_bytes(32, "big") for i in range(4096))
    special = [bytes(131072), const7, const_big, bytes(one_hot), (R - 1).to_bytes(32, "big") * 4096, ramp]
    exp = []
    for b in special:
        rc, oc, op = oracle_ctx.compute_cells_and_kzg_proofs(b)
        assert rc == 0
        exp.append((oc, op))
        rc, cells, proofs = gpu_ctx.ComputeCellsAndKZGProofs(b)
        assert rc == 0 and b"".join(cells) == oc and b"".join(proofs) == op
    st = torch.cuda.current_stream().cuda_stream
    for n in (6, 20, 40):
        blobs = [special[i % 6] for i in range(n)]
        d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
        d_cells = torch.zeros(n * 262144, dtype=torch.uint8, device="cuda")
        d_proofs = torch.zeros(n * 128 * 48, dtype=torch.uint8, device="cuda")
        d_st = torch.ones(n, dtype=torch.int32, device="cuda")
        assert gpu_ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), n, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st) == 0
        torch.cuda.synchronize()
        assert d_st.cpu().tolist() == [0] * n
        cb, pb = d_cells.cpu().numpy().tobytes(), d_proofs.cpu().numpy().tobytes()
        for i in range(n):
            assert cb[i * 262144:(i + 1) * 262144] == exp[i % 6][0] and pb[i * 6144:(i + 1) * 6144] == exp[i % 6][1], (n, i)


def test_fk20_table_widths_agree(product, oracle_ctx):
    """fk20_bits 12 (default, 47 GB), 11, 13 (86 GB table, 20 instead of 24 additions per scalar) and 8 (4.3 GB, 32 additions: the width a context
    degrades to when the HBM is short) give the oracle's bytes, for one blob (few-blob lane layout) and for a batch (batch lane layout);
    switching the option rebuilds the table"""
    import torch
    torch.cuda.init()
    blobs = [rand_blob(900 + i) for i in range(12)] + [bytes(131072), (R - 1).to_bytes(32, "big") * 4096]
    exp = [oracle_ctx.compute_cells_and_kzg_proofs(b) for b in blobs[:2] + blobs[-2:]]
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    n = len(blobs)
    outs = {}
    ctx = product.Context()
    try:
        for bits in (13, 12, 11, 8, 13):
            assert ctx.set_option("fk20_bits", bits) == 0
            assert ctx.set_option("fk20_bits", 10) == 17
            rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(blobs[0])
            assert rc == 0 and ctx.get_option("fk20_bits_effective") == bits
            assert b"".join(cells) == exp[0][1] and b"".join(proofs) == exp[0][2]
            d_cells = torch.zeros(n * 262144, dtype=torch.uint8, device="cuda")
            d_proofs = torch.zeros(n * 128 * 48, dtype=torch.uint8, device="cuda")
            d_st = torch.ones(n, dtype=torch.int32, device="cuda")
            assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), n, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), 0) == 0
            torch.cuda.synchronize()
            assert int(d_st.abs().sum().item()) == 0
            pr = d_proofs.cpu().numpy().tobytes()
            for j, b in enumerate([0, 1, n - 2, n - 1]):
                assert pr[b * 6144:(b + 1) * 6144] == exp[j][2], (bits, b)
            outs[bits] = pr
        assert outs[11] == outs[13] == outs[8] == outs[12]
    finally:
        ctx.close()


@pytest.mark.parametrize("n_blobs", [2, 20])
def test_cell_verifier_rejects_points_outside_the_subgroup(gpu_ctx, n_blobs):
    """VerifyCellKZGProofBatch runs the r-torsion test of its proofs and commitments on a quad of lanes per point for calls of few cells and
    on ONE lane per point from 2048 points on (k_g1_validate modes 2 and 4): both must accept a good batch and answer a point of the
    curve outside G1 -- the order-3 point (0, 2), random curve points, anywhere in the batch, proof or commitment -- with the decoder's
    error, as gnark's SetBytes does under the reference (serialization.go:108-115, api_eip7594.go:190-213)."""
    import random
    from test_gpu_commit import rand_blob
    from test_mont29_host import P, R, ec_mul, compress
    NOT_IN_SUBGROUP = 13                                        # MKZG_ERR_G1_NOT_IN_SUBGROUP
    rng = random.Random(78)
    outside = [(0, 2)]
    while len(outside) < 4:
        x = rng.randrange(P)
        rhs = (x * x * x + 4) % P
        y = pow(rhs, (P + 1) // 4, P)
        if y * y % P == rhs and ec_mul(R, (x, y)) is not None:
            outside.append((x, y))
    blobs = [rand_blob(5000 + (i % 3)) for i in range(n_blobs)]
    per_blob = {}
    for b in set(blobs):
        rc, cm = gpu_ctx.BlobToKZGCommitment(b)
        assert rc == 0
        rc, cells, proofs = gpu_ctx.ComputeCellsAndKZGProofs(b)
        assert rc == 0
        per_blob[b] = (cm, cells, proofs)
    cms, idx, cells, proofs = [], [], [], []
    for b in blobs:
        cm, cs, ps = per_blob[b]
        cms += [cm] * 128
        idx += list(range(128))
        cells += cs
        proofs += ps
    n = len(cells)
    assert (n >= 2048) == (n_blobs == 20)
    assert gpu_ctx.VerifyCellKZGProofBatch(cms, idx, cells, proofs) == 0
    for k, at in enumerate((0, n // 2 + 1, n - 1)):
        bad = list(proofs)
        bad[at] = compress(outside[k % len(outside)])
        assert gpu_ctx.VerifyCellKZGProofBatch(cms, idx, cells, bad) == NOT_IN_SUBGROUP, ("proof", at)
    badc = list(cms)
    for j in range(128 * (n_blobs - 1), n):                      # the last blob's commitment
        badc[j] = compress(outside[1])
    assert gpu_ctx.VerifyCellKZGProofBatch(badc, idx, cells, proofs) == NOT_IN_SUBGROUP
    wrong = list(proofs)
    wrong[7], wrong[8] = wrong[8], wrong[7]                      # in G1, simply wrong
    assert gpu_ctx.VerifyCellKZGProofBatch(cms, idx, cells, wrong) == 1
