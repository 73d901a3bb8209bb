"""GPU parity tests of the EIP-4844 proof paths through the C ABI: ComputeKZGProof,
ComputeBlobKZGProof, VerifyKZGProof, VerifyBlobKZGProof, VerifyBlobKZGProofBatch.

Rules restated from the reference's consensus_specs_test.go:72-359: compute outputs byte-for-byte;
verifiers: output true <=> rc == 0, false <=> rc == MKZG_VERIFY_FAILED (1), null <=> input error.
"""
import hashlib

import pytest

from test_gpu_commit import rand_blob

pytestmark = pytest.mark.gpu
VERIFY_FAILED = 1
R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def _bad(*vals):
    from conftest import BadHex
    return any(isinstance(v, BadHex) for v in vals)


def _wrong_len(val, n):
    return not _bad(val) and len(val) != n


def _check_verifier(rc, expected, name):
    if expected is None:
        assert rc not in (0, VERIFY_FAILED), f"{name}: expected input error, got rc={rc}"
    elif expected is True:
        assert rc == 0, f"{name}: expected accept, got rc={rc}"
    else:
        assert rc == VERIFY_FAILED, f"{name}: expected reject, got rc={rc}"


def test_compute_kzg_proof(vectors, gpu_ctx):
    n_ok = 0
    for c in vectors.cases("compute_kzg_proof"):
        blob, z = vectors.val(c["input"]["blob"]), vectors.val(c["input"]["z"])
        exp = vectors.val(c["output"])
        if _bad(blob, z) or len(blob) != 131072 or len(z) != 32:
            assert exp is None
            continue
        rc, proof, y = gpu_ctx.ComputeKZGProof(blob, z)
        if exp is None:
            assert rc != 0, c["name"]
        else:
            assert rc == 0 and proof == exp[0] and y == exp[1], c["name"]
            n_ok += 1
    assert n_ok > 30


@pytest.fixture(params=[(-1, -1, 1), (0, 0, 1), (1, 1, 1), (0, 1, 1), (-1, -1, 0), (0, 0, 0)],
                ids=["defaults", "sha_device-latency_shape", "sha_host-throughput_shape", "sha_device-throughput_shape", "device_validation", "sha_device-latency_shape-device_validation"])
def hash_place(request, gpu_ctx):
    """Where the Fiat-Shamir transcript is hashed (option host_sha: by default the host pool takes small calls and the device large
    ones) and which shape the verifier takes (option verify_shape: inverse NTT + Horner and quad subgroup tests for a lone call,
    barycentric evaluation and one lane per point when several calls share the chip): the vectors run under each setting."""
    assert gpu_ctx.set_option("host_sha", request.param[0]) == 0
    assert gpu_ctx.set_option("verify_shape", request.param[1]) == 0
    # option host_validate (round 6, default 1): the single-proof entry points decode and subgroup-test their points on the calling thread
    # (host_validate.hpp); 0 = the device's kernels as for batches.  Same codes either way: the vectors' invalid points run under both
    assert gpu_ctx.set_option("host_validate", request.param[2]) == 0 and gpu_ctx.get_option("host_validate") == request.param[2]
    yield gpu_ctx
    assert gpu_ctx.set_option("host_sha", -1) == 0
    assert gpu_ctx.set_option("verify_shape", -1) == 0
    assert gpu_ctx.set_option("host_validate", 1) == 0


def test_compute_blob_kzg_proof(vectors, hash_place):
    gpu_ctx = hash_place
    n_ok = 0
    for c in vectors.cases("compute_blob_kzg_proof"):
        blob, cm = vectors.val(c["input"]["blob"]), vectors.val(c["input"]["commitment"])
        exp = vectors.val(c["output"])
        if _bad(blob, cm) or len(blob) != 131072 or len(cm) != 48:
            assert exp is None
            continue
        rc, proof = gpu_ctx.ComputeBlobKZGProof(blob, cm)
        if exp is None:
            assert rc != 0, c["name"]
        else:
            assert rc == 0 and proof == exp, c["name"]
            n_ok += 1
    assert n_ok == 7


def test_verify_kzg_proof(vectors, hash_place):
    gpu_ctx = hash_place
    n = 0
    for c in vectors.cases("verify_kzg_proof"):
        i = c["input"]
        cm, z, y, pr = (vectors.val(i[k]) for k in ("commitment", "z", "y", "proof"))
        exp = vectors.val(c["output"])
        if _bad(cm, z, y, pr) or len(cm) != 48 or len(pr) != 48 or len(z) != 32 or len(y) != 32:
            assert exp is None
            continue
        _check_verifier(gpu_ctx.VerifyKZGProof(cm, z, y, pr), exp, c["name"])
        n += 1
    assert n > 100


def test_points_outside_the_subgroup(gpu_ctx):
    """The decoder's r-torsion test (k_g1_validate: phi(P) == -[x^2]P without y, doublings on a lane quad per point) on curve points
    inside and outside G1 through VerifyKZGProof: the order-3 points (0, +-2) drive the chain through the identity, random curve
    points lie outside with probability 1 - 1/cofactor, h * Q lies inside for any Q.  Reference: serialization.go:108-115."""
    import random
    from test_mont29_host import P, R, G1X, G1Y, ec_mul, compress
    NOT_IN_SUBGROUP = 13                                        # MKZG_ERR_G1_NOT_IN_SUBGROUP
    rng = random.Random(77)
    outside = [(0, 2), (0, P - 2)]
    while len(outside) < 8:
        x = rng.randrange(P)
        rhs = (x * x * x + 4) % P
        y = pow(rhs, (P + 1) // 4, P)
        if y * y % P == rhs:
            outside.append((x, y))
    good = compress(ec_mul(5, (G1X, G1Y)))
    z, y32 = (3).to_bytes(32, "big"), (4).to_bytes(32, "big")
    for hv in (0, 1):                                           # the device's kernel, then the calling thread's host_validate.hpp
        assert gpu_ctx.set_option("host_validate", hv) == 0
        for pt in outside:
            assert ec_mul(R, pt) is not None
            assert gpu_ctx.VerifyKZGProof(compress(pt), z, y32, good) == NOT_IN_SUBGROUP, pt
            assert gpu_ctx.VerifyKZGProof(good, z, y32, compress(pt)) == NOT_IN_SUBGROUP, pt
        blob0 = rand_blob(6)
        assert gpu_ctx.ComputeBlobKZGProof(blob0, compress(outside[3]))[0] == NOT_IN_SUBGROUP
        assert gpu_ctx.VerifyBlobKZGProof(blob0, compress(outside[4]), good) == NOT_IN_SUBGROUP
        assert gpu_ctx.VerifyBlobKZGProof(blob0, good, compress(outside[5])) == NOT_IN_SUBGROUP
    H_COF = 0x396c8c005555e1568c00aaab0000aaab
    for pt in outside[2:5]:
        q = ec_mul(H_COF, pt)
        assert q is not None and ec_mul(R, q) is None
        assert gpu_ctx.VerifyKZGProof(compress(q), z, y32, good) == VERIFY_FAILED            # decodes, in G1; the proof is simply wrong
    # a batch that mixes them: the first offender's error wins (verify.go:102-123)
    blob = rand_blob(5)
    rc, cm = gpu_ctx.BlobToKZGCommitment(blob)
    rc, pr = gpu_ctx.ComputeBlobKZGProof(blob, cm)
    assert rc == 0
    n = 40
    cms, prs = [cm] * n, [pr] * n
    assert gpu_ctx.VerifyBlobKZGProofBatch([blob] * n, cms, prs) == 0
    for at in (0, 15, 16, 39):
        bad = list(prs)
        bad[at] = compress(outside[at % len(outside)])
        assert gpu_ctx.VerifyBlobKZGProofBatch([blob] * n, cms, bad) == NOT_IN_SUBGROUP, at


def test_verify_blob_kzg_proof(vectors, hash_place):
    gpu_ctx = hash_place
    for c in vectors.cases("verify_blob_kzg_proof"):
        i = c["input"]
        blob, cm, pr = (vectors.val(i[k]) for k in ("blob", "commitment", "proof"))
        exp = vectors.val(c["output"])
        if _bad(blob, cm, pr) or len(blob) != 131072 or len(cm) != 48 or len(pr) != 48:
            assert exp is None
            continue
        _check_verifier(gpu_ctx.VerifyBlobKZGProof(blob, cm, pr), exp, c["name"])


def test_verify_blob_kzg_proof_batch(vectors, hash_place):
    gpu_ctx = hash_place
    for c in vectors.cases("verify_blob_kzg_proof_batch"):
        i = c["input"]
        blobs, cms, prs = (vectors.val(i[k]) for k in ("blobs", "commitments", "proofs"))
        exp = vectors.val(c["output"])
        if _bad(*blobs, *cms, *prs) or any(len(b) != 131072 for b in blobs) or any(len(x) != 48 for x in cms) or any(len(x) != 48 for x in prs):
            assert exp is None
            continue
        _check_verifier(gpu_ctx.VerifyBlobKZGProofBatch(blobs, cms, prs), exp, c["name"])


def test_roundtrip_vs_oracle(gpu_ctx, oracle_ctx):
    """commit -> prove -> verify on GetRandBlob inputs (reference examples_test.go:13-53), each step
    bit-compared with the oracle; then a corrupted batch must be rejected."""
    blobs = [rand_blob(s) for s in range(5)]
    cms, prs = [], []
    for b in blobs:
        rc, cm = gpu_ctx.BlobToKZGCommitment(b)
        assert rc == 0 and cm == oracle_ctx.blob_to_kzg_commitment(b)[1]
        rc, pr = gpu_ctx.ComputeBlobKZGProof(b, cm)
        assert rc == 0 and pr == oracle_ctx.compute_blob_kzg_proof(b, cm)[1]
        cms.append(cm)
        prs.append(pr)
        assert gpu_ctx.VerifyBlobKZGProof(b, cm, pr) == 0
    assert gpu_ctx.VerifyBlobKZGProofBatch(blobs, cms, prs) == 0
    assert oracle_ctx.verify_blob_kzg_proof_batch(blobs, cms, prs) == 0
    swapped = [prs[1], prs[0]] + prs[2:]
    assert gpu_ctx.VerifyBlobKZGProofBatch(blobs, cms, swapped) == VERIFY_FAILED
    assert gpu_ctx.VerifyBlobKZGProofBatch(blobs, cms, prs[:4]) == 2            # ErrBatchLengthCheck
    assert gpu_ctx.VerifyBlobKZGProofBatch([], [], []) == 0
    # z inside the domain (compute_kzg_proof on-domain branch) vs oracle
    import oracle
    rc, ext = oracle.fr_fft((1).to_bytes(32, "big") + bytes(32 * 4095))         # all-ones: not roots; build a root explicitly
    root = 10238227357739495823651030575849232062558860180284477541189508159991286009131
    R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    w = pow(root, (1 << 32) // 4096, R)
    for k in (0, 1, 5, 4095):
        z = pow(w, k, R).to_bytes(32, "big")
        rc, pr, y = gpu_ctx.ComputeKZGProof(blobs[0], z)
        rco, pro, yo = oracle_ctx.compute_kzg_proof(blobs[0], z)
        assert rc == 0 and rco == 0 and pr == pro and y == yo
        assert gpu_ctx.VerifyKZGProof(cms[0], z, y, pr) == 0


def test_batch_64_device_generated(gpu_ctx, oracle_ctx):
    """larger batch: proofs produced on the device, verified on the device, spot-checked vs oracle"""
    blobs = [rand_blob(1000 + s) for s in range(24)]
    rc, cms, st = gpu_ctx.BlobToKZGCommitmentBatch(blobs)
    assert rc == 0
    prs = []
    for b, cm in zip(blobs, cms):
        rc, pr = gpu_ctx.ComputeBlobKZGProof(b, cm)
        assert rc == 0
        prs.append(pr)
    assert prs[7] == oracle_ctx.compute_blob_kzg_proof(blobs[7], cms[7])[1]
    assert gpu_ctx.VerifyBlobKZGProofBatch(blobs, cms, prs) == 0
    bad = list(prs)
    bad[13] = prs[12]
    assert gpu_ctx.VerifyBlobKZGProofBatch(blobs, cms, bad) == VERIFY_FAILED


@pytest.fixture(params=[0, 1], ids=["latency_shape", "throughput_shape"])
def verify_shape(request, gpu_ctx):
    """both shapes of the blob verifier (api_verify.hip): 0 = the call's own hash launch, quad-lane subgroup tests, side streams;
    1 = one stream per call, one lane per point, shared hash rounds -- forced, then put back; both evaluate with k_blob_eval_tree"""
    assert gpu_ctx.set_option("verify_shape", request.param) == 0
    yield request.param
    assert gpu_ctx.set_option("verify_shape", -1) == 0


def test_sharded_verify_records_match_oracle(gpu_ctx, oracle_ctx, product, verify_shape):
    """The partial-sum records of the sharded verifier are deterministic once r is fixed, so the GPU fold
    can be compared byte-for-byte with the oracle's -- a parity check INSIDE the batch verifier that the
    random r normally hides, under BOTH shapes of the verifier.  Also: GPU records finish on the oracle and vice versa."""
    import torch
    n = 7
    blobs = [rand_blob(500 + s) for s in range(n)]
    rc, cms, _ = gpu_ctx.BlobToKZGCommitmentBatch(blobs)
    prs = [gpu_ctx.ComputeBlobKZGProof(b, c)[1] for b, c in zip(blobs, cms)]
    r32 = bytes(range(7, 39))
    d_b = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_c = torch.frombuffer(bytearray(b"".join(cms)), dtype=torch.uint8).cuda()
    d_p = torch.frombuffer(bytearray(b"".join(prs)), dtype=torch.uint8).cuda()
    st = torch.cuda.current_stream().cuda_stream
    recs_gpu, recs_cpu = [], []
    for lo, hi in ((0, 4), (4, 7)):
        rc, rec = gpu_ctx.verify_blob_batch_partial_device(d_b.data_ptr() + lo * 131072, d_c.data_ptr() + lo * 48, d_p.data_ptr() + lo * 48, hi - lo, lo, r32, st)
        assert rc == 0
        rco, reco = oracle_ctx.verify_blob_batch_partial(blobs[lo:hi], cms[lo:hi], prs[lo:hi], lo, r32)
        assert rco == 0 and rec == reco
        recs_gpu.append(rec)
        recs_cpu.append(reco)
    assert gpu_ctx.verify_blob_batch_finish(recs_gpu) == 0
    assert oracle_ctx.verify_blob_batch_finish(recs_gpu) == 0
    # a shard folded with the wrong offset must break the batch
    rc, rec_bad = gpu_ctx.verify_blob_batch_partial_device(d_b.data_ptr() + 4 * 131072, d_c.data_ptr() + 4 * 48, d_p.data_ptr() + 4 * 48, 3, 5, r32, st)
    assert gpu_ctx.verify_blob_batch_finish([recs_gpu[0], rec_bad]) == 0   # offsets only relabel r^i: still a valid RLC
    # corrupted proof in shard 1
    d_p2 = d_p.clone()
    d_p2[5 * 48:6 * 48] = d_p[6 * 48:7 * 48]
    rc, rec2 = gpu_ctx.verify_blob_batch_partial_device(d_b.data_ptr() + 4 * 131072, d_c.data_ptr() + 4 * 48, d_p2.data_ptr() + 4 * 48, 3, 4, r32, st)
    assert rc == 0 and gpu_ctx.verify_blob_batch_finish([recs_gpu[0], rec2]) == VERIFY_FAILED
    # through the orchestration helper, single process
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("kzg_distributed", os.path.join(os.path.dirname(product.__file__), "distributed.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    eng = D.ProductEngine(gpu_ctx)
    assert D.verify_blob_kzg_proof_batch_sharded(eng, (d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, st), n) == 0


def test_proofs_with_direct_table(product, oracle_ctx):
    """with commit_direct_bits set, the proof MSMs (quotient polynomials as Montgomery Fr rows) go through the direct window
    table as well: ComputeKZGProof / ComputeBlobKZGProof / the device batch path (several max_batch pieces) equal the oracle"""
    import torch
    torch.cuda.init()
    from test_gpu_commit import rand_blob
    ctx = product.Context()
    assert ctx.set_option("max_batch", 2) == 0
    assert ctx.set_option("commit_direct_bits", 12) == 0
    blobs = [rand_blob(70 + i) for i in range(5)] + [bytes(131072)]
    cms = []
    for b in blobs:
        rc, cm = ctx.BlobToKZGCommitment(b)
        assert rc == 0 and cm == oracle_ctx.blob_to_kzg_commitment(b)[1]
        cms.append(cm)
    for b, cm in zip(blobs[:2] + blobs[5:], cms[:2] + cms[5:]):
        rc, pr = ctx.ComputeBlobKZGProof(b, cm)
        assert rc == 0 and pr == oracle_ctx.compute_blob_kzg_proof(b, cm)[1]
    z = (12345).to_bytes(32, "big")
    rc, pr, y = ctx.ComputeKZGProof(blobs[1], z)
    assert (rc, pr, y) == (0,) + tuple(oracle_ctx.compute_kzg_proof(blobs[1], z)[1:])
    n = len(blobs)
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cm = torch.frombuffer(bytearray(b"".join(cms)), dtype=torch.uint8).cuda()
    d_pr = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(n, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), n, d_pr.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert d_st.cpu().tolist() == [0] * n
    pr = bytes(d_pr.cpu().tolist())
    for i in range(n):
        assert pr[48 * i:48 * i + 48] == oracle_ctx.compute_blob_kzg_proof(blobs[i], cms[i])[1]
    assert ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), n, st) == 0
    ctx.close()
    # the lane shapes of the direct-table kernel on Fr rows: calls of few items (1, 2, 4 scalars per lane: 20, 40, 70 proofs) and the
    # one-scalar-per-lane shape with the serial pre-sum of 129..255 items (150 proofs)
    ctx = product.Context()
    assert ctx.set_option("commit_direct_bits", 12) == 0
    for m in (20, 40, 70, 150):
        d_b = torch.frombuffer(bytearray(b"".join(blobs[i % n] for i in range(m))), dtype=torch.uint8).cuda()
        d_c = torch.frombuffer(bytearray(b"".join(cms[i % n] for i in range(m))), dtype=torch.uint8).cuda()
        d_p = torch.zeros(m * 48, dtype=torch.uint8, device="cuda")
        d_s = torch.ones(m, dtype=torch.int32, device="cuda")
        assert ctx.compute_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), m, d_p.data_ptr(), d_s.data_ptr(), st) == 0
        torch.cuda.synchronize()
        assert d_s.cpu().tolist() == [0] * m
        got = bytes(d_p.cpu().tolist())
        assert all(got[48 * i:48 * i + 48] == pr[48 * (i % n):48 * (i % n) + 48] for i in range(m)), m
    ctx.close()


def test_shared_context_from_threads(gpu_ctx, oracle_ctx):
    """the reference's Context is shared by goroutines (verify.go:152-169, examples_test.go:11); here: eight host threads hammer one
    context with different calls (ctypes drops the GIL, so the calls really overlap: more callers than lanes, every kind of entry
    point, device-pointer and host-pointer forms mixed) and every result must equal the serial one"""
    import threading
    import torch
    from test_gpu_commit import rand_blob
    blobs = [rand_blob(300 + i) for i in range(4)]
    exp_cm = [oracle_ctx.blob_to_kzg_commitment(b)[1] for b in blobs]
    exp_pr = [oracle_ctx.compute_blob_kzg_proof(b, c)[1] for b, c in zip(blobs, exp_cm)]
    rc, cells0, proofs0 = gpu_ctx.ComputeCellsAndKZGProofs(blobs[0])
    assert rc == 0
    rco, oc, op = oracle_ctx.compute_cells_and_kzg_proofs(blobs[0])
    assert b"".join(cells0) == oc and b"".join(proofs0) == op
    # a device-resident batch of 256 for the throughput-shaped calls
    N = 256
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs[i % 4] for i in range(N))), dtype=torch.uint8).cuda()
    d_cm = torch.frombuffer(bytearray(b"".join(exp_cm[i % 4] for i in range(N))), dtype=torch.uint8).cuda()
    d_pr = torch.frombuffer(bytearray(b"".join(exp_pr[i % 4] for i in range(N))), dtype=torch.uint8).cuda()
    d_bad = d_pr.clone()
    d_bad[48 * 200:48 * 201] = d_pr[48 * 201:48 * 202]
    torch.cuda.synchronize()
    errors = []

    def worker(k):
        try:
            stream = torch.cuda.Stream()
            d_out = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
            d_st = torch.ones(N, dtype=torch.int32, device="cuda")
            torch.cuda.current_stream().synchronize()      # the fills above ran on this thread's default stream, the calls below use `stream`
            for it in range(8):
                i = (k + it) % 4
                which = (k + it) % 7
                if which == 0:
                    rc, cm = gpu_ctx.BlobToKZGCommitment(blobs[i])
                    assert rc == 0 and cm == exp_cm[i]
                elif which == 1:
                    rc, pr = gpu_ctx.ComputeBlobKZGProof(blobs[i], exp_cm[i])
                    assert rc == 0 and pr == exp_pr[i]
                elif which == 2:
                    assert gpu_ctx.VerifyBlobKZGProofBatch(blobs, exp_cm, exp_pr) == 0
                    bad = list(exp_pr)
                    bad[i] = exp_pr[(i + 1) % 4]
                    assert gpu_ctx.VerifyBlobKZGProofBatch(blobs, exp_cm, bad) == 1
                elif which == 3:
                    assert gpu_ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, stream.cuda_stream) == 0
                    assert gpu_ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_bad.data_ptr(), N, stream.cuda_stream) == 1
                elif which == 4:
                    assert gpu_ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), N, d_out.data_ptr(), d_st.data_ptr(), stream.cuda_stream) == 0
                    stream.synchronize()
                    assert int(d_st.abs().sum().item()) == 0 and bool((d_out == d_cm).all().item())
                elif which == 5:
                    ids = [0, 3, 64, 127, 1]
                    assert gpu_ctx.VerifyCellKZGProofBatch([exp_cm[0]] * 5, ids, [cells0[j] for j in ids], [proofs0[j] for j in ids]) == 0
                    assert gpu_ctx.VerifyCellKZGProofBatch([exp_cm[0]] * 5, ids, [cells0[j] for j in ids], [proofs0[j ^ 1] for j in ids]) == 1
                else:
                    half = list(range(i % 2, 128, 2))
                    rc, rcells, rproofs = gpu_ctx.RecoverCellsAndComputeKZGProofs(half, [cells0[j] for j in half])
                    assert rc == 0 and rcells == cells0 and rproofs == proofs0
        except Exception:      # noqa: BLE001
            import traceback
            errors.append(traceback.format_exc())

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=900)
    assert not errors, errors[0]


def test_hash_rounds_many_callers(product, oracle_ctx):
    """Twelve threads verify batches of different sizes on ONE context at once, so that the calls take the throughput shape and share
    hash launches (api_verify.hip hash_round_join: rounds of several jobs with different n, leaders and followers changing from round to
    round, two rounds in flight); every third batch carries a wrong proof, a few a non-canonical blob.  Every verdict must be the one the
    serial, forced-latency-shape call gives.  Reference: verify.go:88-169 (a Context shared by goroutines)."""
    import random
    import threading
    import torch
    torch.cuda.init()
    ctx = product.Context()
    NB = 48
    blobs = [rand_blob(7000 + i) for i in range(NB)]
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cm = torch.zeros(NB * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(NB * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.ones(NB, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), NB, d_cm.data_ptr(), d_st.data_ptr(), st) == 0
    assert ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), NB, d_pr.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    cm, pr = bytes(d_cm.cpu().tolist()), bytes(d_pr.cpu().tolist())
    for i in (0, 17, 47):
        assert cm[48 * i:48 * i + 48] == oracle_ctx.blob_to_kzg_commitment(blobs[i])[1]
    d_pr_bad = d_pr.clone()
    d_pr_bad[0:48 * (NB - 1)] = d_pr[48:48 * NB]                      # every proof but the last belongs to the next blob
    d_pr_bad[48 * (NB - 1):] = d_pr[0:48]
    d_blobs_bad = d_blobs.clone()
    for i in range(NB):
        d_blobs_bad[131072 * i + 32 * 5:131072 * i + 32 * 6] = 0xff   # a non-canonical field element in every blob
    torch.cuda.synchronize()
    # (first blob, count, kind) -> expected rc: serial calls in the latency shape
    rng = random.Random(5)
    jobs = []
    for k in range(12):
        mine = []
        for it in range(8):
            n = rng.choice([1, 2, 3, 5, 8, 13, 21, 34])
            lo = rng.randrange(0, NB - n + 1)
            mine.append((lo, n, (k + it) % 3 if (k + it) % 7 else 3))
        jobs.append(mine)

    def call(lo, n, kind, stream):
        b = (d_blobs_bad if kind == 3 else d_blobs).data_ptr() + 131072 * lo
        p = (d_pr_bad if kind == 1 else d_pr).data_ptr() + 48 * lo
        return ctx.verify_blob_kzg_proof_batch_device(b, d_cm.data_ptr() + 48 * lo, p, n, stream)

    assert ctx.set_option("verify_shape", 0) == 0
    want = {j: call(*j, st) for mine in jobs for j in mine}
    assert set(want.values()) == {0, 1, 3}
    assert all(want[j] == (1 if j[2] == 1 else 3 if j[2] == 3 else 0) for j in want)
    assert ctx.set_option("verify_shape", -1) == 0
    errors = []

    def worker(k):
        try:
            stream = torch.cuda.Stream(priority=-1 if k % 2 else 0)
            for j in jobs[k]:
                got = call(*j, stream.cuda_stream)
                assert got == want[j], (j, got, want[j])
        except Exception:      # noqa: BLE001
            import traceback
            errors.append(traceback.format_exc())

    # deep rounds (the default: the round also launches its calls' evaluations and MSMs, one launch each, and copies the verdicts back) and
    # plain ones; 28 callers, so that rounds of more than DEEP_MAX_JOBS = 16 calls can form (two launches per step)
    assert ctx.get_option("deep_rounds") == 1
    for deep, callers in ((1, 12), (1, 28), (0, 12)):
        assert ctx.set_option("deep_rounds", deep) == 0
        for rep in range(2):
            ts = [threading.Thread(target=worker, args=(k % 12,)) for k in range(callers)]
            for t in ts:
                t.start()
            for t in ts:
                t.join(timeout=600)
            assert not errors, errors[0]
    assert ctx.set_option("deep_rounds", 2) != 0
    ctx.close()


def test_first_calls_race_on_a_fresh_context(product, oracle_ctx):
    """every shared table of a context is built lazily by the first call that needs it; here eight threads make their FIRST calls on a
    brand-new context at the same moment -- commitments, proofs, cells with proofs, recovery, both batch verifiers -- so the lazy
    construction (twiddles, monomial SRS, commitment table, FK20 table, recovery tables) is raced from every side, and every result must
    still equal the oracle's"""
    import threading
    from test_gpu_commit import rand_blob
    import torch
    torch.cuda.init()
    blob = rand_blob(777)
    rc, cm = oracle_ctx.blob_to_kzg_commitment(blob)
    rc, pr = oracle_ctx.compute_blob_kzg_proof(blob, cm)
    rc, ocells, oproofs = oracle_ctx.compute_cells_and_kzg_proofs(blob)
    cells = [ocells[i * 2048:(i + 1) * 2048] for i in range(128)]
    proofs = [oproofs[i * 48:(i + 1) * 48] for i in range(128)]
    ctx = product.Context()
    errors = []
    gate = threading.Barrier(8)

    def worker(k):
        try:
            gate.wait()
            if k == 0:
                assert ctx.BlobToKZGCommitment(blob) == (0, cm)
            elif k == 1:
                assert ctx.ComputeBlobKZGProof(blob, cm) == (0, pr)
            elif k == 2:
                rc, c2, p2 = ctx.ComputeCellsAndKZGProofs(blob)
                assert rc == 0 and c2 == cells and p2 == proofs
            elif k == 3:
                ids = list(range(0, 128, 2))
                rc, c2, p2 = ctx.RecoverCellsAndComputeKZGProofs(ids, [cells[i] for i in ids])
                assert rc == 0 and c2 == cells and p2 == proofs
            elif k == 4:
                assert ctx.VerifyBlobKZGProofBatch([blob, blob], [cm, cm], [pr, pr]) == 0
            elif k == 5:
                ids = [0, 1, 64, 127]
                assert ctx.VerifyCellKZGProofBatch([cm] * 4, ids, [cells[i] for i in ids], [proofs[i] for i in ids]) == 0
            elif k == 6:
                rc, c2 = ctx.ComputeCells(blob)
                assert rc == 0 and c2 == cells
            else:
                ids = list(range(1, 128, 2))
                rc, c2 = ctx.RecoverCells(ids, [cells[i] for i in ids])
                assert rc == 0 and c2 == cells
        except Exception:      # noqa: BLE001
            import traceback
            errors.append(traceback.format_exc())

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=900)
    ctx.close()
    assert not errors, errors[0]


def test_throughput_shape_respects_stream_order(product, oracle_ctx):
    """Device-pointer entry points are stream-ordered: work the caller enqueued on its stream before the call (here: a long sleep
    kernel, then the copies that PRODUCE the blobs, commitments and proofs) is finished before the library reads them -- also in the
    throughput shape, whose shared hash launch runs on the context's own stream and skips the wait for the caller's stream only when
    that stream was idle at the call (api_verify.hip).  A hash over the stale (zero) buffers would reject the batch."""
    import torch
    torch.cuda.init()
    ctx = product.Context()
    NB = 8
    blobs = [rand_blob(700 + i) for i in range(NB)]
    h_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).pin_memory()
    d_src = h_blobs.cuda()
    d_cm = torch.zeros(NB * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(NB * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(NB, dtype=torch.int32, device="cuda")
    assert ctx.blob_to_kzg_commitment_batch_device(d_src.data_ptr(), NB, d_cm.data_ptr(), d_st.data_ptr(), 0) == 0
    assert ctx.compute_blob_kzg_proof_batch_device(d_src.data_ptr(), d_cm.data_ptr(), NB, d_pr.data_ptr(), d_st.data_ptr(), 0) == 0
    torch.cuda.synchronize()
    h_cm, h_pr = d_cm.cpu().pin_memory(), d_pr.cpu().pin_memory()
    assert ctx.set_option("verify_shape", 1) == 0
    for rep in range(3):
        stream = torch.cuda.Stream()
        d_b = torch.zeros_like(d_src)
        d_c = torch.zeros_like(d_cm)
        d_p = torch.zeros_like(d_pr)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            torch.cuda._sleep(20_000_000)                       # 10-200 ms of GPU time (clock dependent) ahead of the producers
            d_b.copy_(h_blobs, non_blocking=True)
            d_c.copy_(h_cm, non_blocking=True)
            d_p.copy_(h_pr, non_blocking=True)
        assert ctx.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), NB, stream.cuda_stream) == 0
        # and an idle stream (the other path)
        stream.synchronize()
        assert ctx.verify_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), NB, stream.cuda_stream) == 0
    ctx.close()


def test_sharded_verify_records_match_oracle_64(gpu_ctx, oracle_ctx, verify_shape):
    """the same byte-for-byte pin on 64 blobs (a whole workgroup of the challenge hash, 64 evaluation workgroups) under both shapes,
    with one non-canonical blob in a second run: the record then carries the error of the reference's precedence"""
    import torch
    n = 64
    blobs = [rand_blob(9000 + s) for s in range(n)]
    d_b = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_c = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
    d_s = torch.zeros(n, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert gpu_ctx.blob_to_kzg_commitment_batch_device(d_b.data_ptr(), n, d_c.data_ptr(), d_s.data_ptr(), st) == 0
    assert gpu_ctx.compute_blob_kzg_proof_batch_device(d_b.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_s.data_ptr(), st) == 0
    torch.cuda.synchronize()
    cms = [bytes(d_c[48 * i:48 * i + 48].cpu().numpy()) for i in range(n)]
    prs = [bytes(d_p[48 * i:48 * i + 48].cpu().numpy()) for i in range(n)]
    r32 = bytes(range(101, 133))
    rc, rec = gpu_ctx.verify_blob_batch_partial_device(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, 3, r32, st)
    rco, reco = oracle_ctx.verify_blob_batch_partial(blobs, cms, prs, 3, r32)
    assert rc == 0 and rco == 0 and rec == reco
    assert gpu_ctx.verify_blob_batch_finish([rec]) == 0
    # blob 41: scalar 17 := r (non-canonical)
    bad = bytearray(blobs[41])
    bad[17 * 32:18 * 32] = R_MOD.to_bytes(32, "big")
    d_b2 = d_b.clone()
    d_b2[41 * 131072:42 * 131072] = torch.frombuffer(bad, dtype=torch.uint8).cuda()
    rc, rec = gpu_ctx.verify_blob_batch_partial_device(d_b2.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, 3, r32, st)
    rco, reco = oracle_ctx.verify_blob_batch_partial(blobs[:41] + [bytes(bad)] + blobs[42:], cms, prs, 3, r32)
    assert rc == rco and rc != 0 and rec[:4] == reco[:4]


def _domain_roots_brp():
    """the 4096 evaluation points of a blob in the order its scalars are stored (api.go:127-132: bit-reversed powers of the 4096th root
    of unity derived from the 2^32-th root of internal/domain/domain.go:61)"""
    root32 = 10238227357739495823651030575849232062558860180284477541189508159991286009131
    w = pow(root32, 1 << (32 - 12), R_MOD)
    nat = [1] * 4096
    for i in range(1, 4096):
        nat[i] = nat[i - 1] * w % R_MOD
    return [nat[int(format(i, "012b")[::-1], 2)] for i in range(4096)]


def test_evaluation_kernels_agree_with_oracle(gpu_ctx, oracle_ctx):
    """Domain.EvaluateLagrangePolynomial (reference internal/domain/domain.go:163-235) through each of the library's three evaluation
    kernels -- inverse NTT + Horner (a cross-check only since round 4), the numerator tree of the barycentric formula that both verifier shapes run (k_blob_eval_tree), the proof
    path's per-lane batch inversion -- at caller-chosen points: random z, z = 0, z = r - 1, and POINTS OF THE DOMAIN (stored index 0 = the root
    1, an odd index, index 4095, a middle one), where the barycentric formula divides by zero and the value is the stored scalar itself.
    Each branch also with a non-canonical blob (status, no value) and a non-canonical z.  Byte-exact against the oracle."""
    import torch
    roots = _domain_roots_brp()
    blobs = [rand_blob(31000 + s) for s in range(6)] + [bytes(131072), (R_MOD - 1).to_bytes(32, "big") * 4096]
    bad_blob = bytearray(rand_blob(31100))
    bad_blob[4095 * 32:] = (R_MOD + 5).to_bytes(32, "big")               # the LAST scalar: the on-domain branch must still look at every scalar
    zs_off = [hashlib.sha256(b"z%d" % i).digest() for i in range(4)]
    zs_off = [(int.from_bytes(z, "big") % R_MOD).to_bytes(32, "big") for z in zs_off] + [bytes(32), (R_MOD - 1).to_bytes(32, "big")]
    on_idx = [0, 1, 2049, 4095, 1234]
    zs_on = [roots[i].to_bytes(32, "big") for i in on_idx]
    cases = []                                                             # (blob, z, expected rc)
    for b in blobs:
        for z in zs_off[:3] + zs_on[:4]:
            cases.append((b, z))
    for z in zs_off[3:] + zs_on[4:]:
        cases.append((blobs[1], z))
    for z in (zs_off[0], zs_on[0], zs_on[1], zs_on[3]):
        cases.append((bytes(bad_blob), z))                                 # non-canonical blob in the off-domain and every on-domain branch
    cases.append((blobs[2], R_MOD.to_bytes(32, "big")))                    # non-canonical z
    n = len(cases)
    d_b = torch.frombuffer(bytearray(b"".join(c[0] for c in cases)), dtype=torch.uint8).cuda()
    want = [oracle_ctx.evaluate_blob(b, z) for b, z in cases]
    assert sum(1 for rc, _ in want if rc) == 5
    # on-domain sanity of the expectation itself: the value IS the stored scalar
    for (b, z), (rc, y) in zip(cases, want):
        if rc == 0 and z in zs_on:
            i = roots.index(int.from_bytes(z, "big"))
            assert y == b[32 * i:32 * i + 32]
    for form in (0, 1, 2):
        rc, ys, st = gpu_ctx.evaluate_blobs_device(d_b.data_ptr(), [z for _, z in cases], form, torch.cuda.current_stream().cuda_stream)
        for i, ((rco, yo), y, sti) in enumerate(zip(want, ys, st)):
            if rco:
                assert sti != 0, (form, i)
            else:
                assert sti == 0 and y == yo, (form, i)
        first = next((s for s in st if s), 0)
        assert rc == first
