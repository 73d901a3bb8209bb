#!/usr/bin/env python3
"""tools/soak.py [seconds = 120] [threads = 24] [seed = 1] -- randomized soak of ONE shared context (the reference's usage: a Context used by
many goroutines, verify.go:152-169, examples_test.go:11): every thread loops over random entry points of the C ABI with random sizes,
random subsets, random corruptions and random error inputs, and every result -- bytes, verdicts and error codes -- is compared with what
the CPU oracle (oracle/, the checker) computed beforehand for a small pool of blobs.  Single-blob calls (which the call combiner groups),
host-pointer batches (pipelined gather / upload), device-pointer batches on per-thread streams, the hash rounds of the blob verifier, the
cell verifier, recovery and option changes under load all run against each other.  Prints one line of counts per kind of call and exits
non-zero on the first mismatch.  Not part of the test suite (it is as long as you ask); the record of a run goes to profiles/."""
import os
import random
import sys
import threading
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_product, rand_blob, R_MOD  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    import torch
    import oracle
    torch.cuda.init()
    kzg = load_product()
    octx = oracle.OracleContext(with_fk20=True)
    t0 = time.time()
    # ---- the pool and the oracle's answers for it
    NB = 6
    blobs = [rand_blob(7000 + 13 * seed + i) for i in range(NB - 2)]
    blobs.append(bytes(131072))                                                    # zero polynomial: identity commitment and proofs
    blobs.append((5).to_bytes(32, "big") * 4096)                                   # constant polynomial
    cms = [octx.blob_to_kzg_commitment(b)[1] for b in blobs]
    prs = [octx.compute_blob_kzg_proof(b, c)[1] for b, c in zip(blobs, cms)]
    rng0 = random.Random(seed)
    zs = [rng0.randrange(R_MOD).to_bytes(32, "big") for _ in range(NB)]
    zs[1] = blobs[1][0:0] + (1).to_bytes(32, "big")                                # z = 1 = omega^0: a point of the domain
    kp = [octx.compute_kzg_proof(b, z) for b, z in zip(blobs, zs)]                 # (rc, proof, y)
    NCELL = 3                                                                      # blobs whose cells and proofs the oracle computes (seconds each)
    cellsets = []
    for i in range(NCELL):
        rc, c, p = octx.compute_cells_and_kzg_proofs(blobs[i])
        assert rc == 0
        cellsets.append(([c[j * 2048:(j + 1) * 2048] for j in range(128)], [p[j * 48:(j + 1) * 48] for j in range(128)]))
    bad_blob = bytearray(blobs[0])
    bad_blob[32 * 77:32 * 78] = R_MOD.to_bytes(32, "big")                          # non-canonical scalar
    bad_blob = bytes(bad_blob)
    bad_point = bytes([0x80] + [0xff] * 47)                                        # x >= p: not a point encoding
    print(f"oracle answers for {NB} blobs ({NCELL} with cells) in {time.time() - t0:.1f} s", flush=True)

    ctx = kzg.Context(device=0)
    assert ctx.set_option("max_batch", 64) == 0
    ERR_NONCANON = ctx.BlobToKZGCommitment(bad_blob)[0]
    assert ERR_NONCANON != 0
    # device-resident material
    ND = 96
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs[i % NB] for i in range(ND))), dtype=torch.uint8).cuda()
    d_cms = torch.frombuffer(bytearray(b"".join(cms[i % NB] for i in range(ND))), dtype=torch.uint8).cuda()
    d_prs = torch.frombuffer(bytearray(b"".join(prs[i % NB] for i in range(ND))), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    counts, errors, lock = {}, [], threading.Lock()
    stop_at = time.time() + seconds

    def note(kind):
        with lock:
            counts[kind] = counts.get(kind, 0) + 1

    def worker(k):
        rng = random.Random(seed * 1000 + k)
        try:
            stream = torch.cuda.Stream()
            d_out = torch.zeros(ND * 48, dtype=torch.uint8, device="cuda")
            d_st = torch.zeros(ND, dtype=torch.int32, device="cuda")
            d_cells = torch.zeros(8 * 128 * 2048, dtype=torch.uint8, device="cuda")
            d_cproofs = torch.zeros(8 * 128 * 48, dtype=torch.uint8, device="cuda")
            torch.cuda.current_stream().synchronize()
            while time.time() < stop_at and not errors:
                kind = rng.choice(("commit1", "commitN", "commit_dev", "proof1", "kzgproof", "verify1", "verifyN", "verify_dev", "verifykzg", "cells1", "cellsonly", "cells_dev",
                                   "cellverify", "recover", "errors", "option"))
                i = rng.randrange(NB)
                if kind == "commit1":
                    rc, cm = ctx.BlobToKZGCommitment(blobs[i])
                    assert rc == 0 and cm == cms[i], (kind, i, rc)
                elif kind == "commitN":
                    n = rng.choice((1, 2, 3, 7, 16, 33, 40, 65))
                    ids = [rng.randrange(NB) for _ in range(n)]
                    poison = rng.random() < 0.2
                    bl = [blobs[j] for j in ids]
                    if poison:
                        bl[rng.randrange(n)] = bad_blob
                    rc, out, st = ctx.BlobToKZGCommitmentBatch(bl)
                    for q, j in enumerate(ids):
                        if bl[q] is bad_blob:
                            assert st[q] == ERR_NONCANON, (kind, st[q])
                        else:
                            assert st[q] == 0 and out[q] == cms[j], (kind, q, j, st[q])
                    assert (rc != 0) == poison, (kind, rc, poison)
                elif kind == "commit_dev":
                    n = rng.choice((1, 5, 32, 33, 64, 65, 96))
                    assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), n, d_out.data_ptr(), d_st.data_ptr(), stream.cuda_stream) == 0
                    stream.synchronize()
                    assert int(d_st[:n].abs().sum().item()) == 0 and bool((d_out[:n * 48] == d_cms[:n * 48]).all().item()), kind
                elif kind == "proof1":
                    rc, pr = ctx.ComputeBlobKZGProof(blobs[i], cms[i])
                    assert rc == 0 and pr == prs[i], (kind, i, rc)
                elif kind == "kzgproof":
                    rc, pr, y = ctx.ComputeKZGProof(blobs[i], zs[i])
                    assert rc == 0 and (0, pr, y) == tuple(kp[i]), (kind, i, rc)
                elif kind == "verify1":
                    good = rng.random() < 0.7
                    rc = ctx.VerifyBlobKZGProof(blobs[i], cms[i], prs[i] if good else prs[(i + 1) % (NB - 2)] if i < NB - 2 else prs[0])
                    assert (rc == 0) == good and rc in (0, 1), (kind, i, good, rc)
                elif kind == "verifyN":
                    n = rng.choice((1, 2, 6, 16, 17, 48, 64, 100))
                    ids = [rng.randrange(NB) for _ in range(n)]
                    pl = [prs[j] for j in ids]
                    want = 0
                    if rng.random() < 0.4:
                        q = rng.randrange(n)
                        alt = prs[(ids[q] + 1) % (NB - 2)]
                        if alt != pl[q]:
                            pl[q] = alt
                            want = 1
                    rc = ctx.VerifyBlobKZGProofBatch([blobs[j] for j in ids], [cms[j] for j in ids], pl)
                    assert rc == want, (kind, n, want, rc)
                elif kind == "verify_dev":
                    n = rng.choice((1, 6, 48, 64, 96))
                    assert ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cms.data_ptr(), d_prs.data_ptr(), n, stream.cuda_stream) == 0, kind
                    if n > NB:           # proofs shifted by one blob: wrong for at least one
                        assert ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cms.data_ptr(), d_prs.data_ptr() + 48, n - 1, stream.cuda_stream) == 1, kind
                elif kind == "verifykzg":
                    good = rng.random() < 0.7
                    _, pr, y = kp[i]
                    rc = ctx.VerifyKZGProof(cms[i], zs[i], y if good else zs[i], pr)
                    assert (rc == 0) == (good or y == zs[i]), (kind, i, good, rc)
                elif kind == "cells1":
                    j = rng.randrange(NCELL)
                    rc, c, p = ctx.ComputeCellsAndKZGProofs(blobs[j])
                    assert rc == 0 and c == cellsets[j][0] and p == cellsets[j][1], (kind, j, rc)
                elif kind == "cellsonly":
                    j = rng.randrange(NCELL)
                    rc, c = ctx.ComputeCells(blobs[j])
                    assert rc == 0 and c == cellsets[j][0], (kind, j, rc)
                elif kind == "cells_dev":
                    n = rng.choice((1, 3, 8))
                    assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), n, d_cells.data_ptr(), d_cproofs.data_ptr(), d_st.data_ptr(), stream.cuda_stream) == 0
                    stream.synchronize()
                    got_c, got_p = bytes(d_cells[:n * 128 * 2048].cpu().numpy()), bytes(d_cproofs[:n * 128 * 48].cpu().numpy())
                    for q in range(min(n, NCELL)):          # device blob q is pool blob q
                        assert got_c[q * 262144:(q + 1) * 262144] == b"".join(cellsets[q][0]) and got_p[q * 6144:(q + 1) * 6144] == b"".join(cellsets[q][1]), (kind, q)
                elif kind == "cellverify":
                    n = rng.choice((1, 2, 5, 64, 128, 200, 384))
                    picks = [(rng.randrange(NCELL), rng.randrange(128)) for _ in range(n)]
                    pr = [cellsets[j][1][c] for j, c in picks]
                    want = 0
                    if rng.random() < 0.4:
                        q = rng.randrange(n)
                        alt = cellsets[picks[q][0]][1][picks[q][1] ^ 1]
                        if alt != pr[q]:
                            pr[q] = alt
                            want = 1
                    rc = ctx.VerifyCellKZGProofBatch([cms[j] for j, _ in picks], [c for _, c in picks], [cellsets[j][0][c] for j, c in picks], pr)
                    assert rc == want, (kind, n, want, rc)
                elif kind == "recover":
                    j = rng.randrange(NCELL)
                    keep = sorted(rng.sample(range(128), rng.choice((64, 65, 100, 127, 128))))
                    rc, c, p = ctx.RecoverCellsAndComputeKZGProofs(keep, [cellsets[j][0][q] for q in keep])
                    assert rc == 0 and c == cellsets[j][0] and p == cellsets[j][1], (kind, j, len(keep), rc)
                elif kind == "errors":
                    which = rng.randrange(5)
                    if which == 0:
                        assert ctx.BlobToKZGCommitment(bad_blob)[0] == ERR_NONCANON
                    elif which == 1:
                        assert ctx.VerifyBlobKZGProof(blobs[i], bad_point, prs[i]) not in (0, 1)
                    elif which == 2:
                        assert ctx.VerifyBlobKZGProofBatch([blobs[0], blobs[1]], [cms[0]], [prs[0], prs[1]]) not in (0, 1)      # lengths differ
                    elif which == 3:
                        keep = list(range(0, 63))                                   # too few cells
                        assert ctx.RecoverCellsAndComputeKZGProofs(keep, [cellsets[0][0][q] for q in keep])[0] != 0
                    else:
                        assert ctx.VerifyCellKZGProofBatch([cms[0]], [128], [cellsets[0][0][0]], [cellsets[0][1][0]]) not in (0, 1)     # cell index out of range
                elif kind == "option" and k == 0 and rng.random() < 0.3:
                    # options that change routing, flipped while every other thread is in flight (same bytes and verdicts whatever the value)
                    key, vals = rng.choice((("combine_single_calls", (0, 1)), ("deep_rounds", (0, 1)), ("host_sha", (-1, 0, 1)), ("verify_shape", (-1, 0, 1)), ("host_zero_copy", (0, 1)),
                                            ("fk20_segments", (0, 1)), ("host_wait", (-1, 0)), ("host_fin", (0, 1)), ("host_validate", (0, 1))))
                    assert ctx.set_option(key, rng.choice(vals)) == 0, key
                else:
                    continue
                note(kind)
        except Exception:      # noqa: BLE001
            with lock:
                errors.append(f"thread {k}:\n" + traceback.format_exc())

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
    t1 = time.time()
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=seconds + 600)
    alive = [t for t in ts if t.is_alive()]
    total = sum(counts.values())
    print(f"soak: {n_threads} threads, {time.time() - t1:.1f} s, seed {seed}: {total} calls checked against the oracle's answers  " + " ".join(f"{k}={v}" for k, v in sorted(counts.items())), flush=True)
    if alive:
        print(f"HUNG: {len(alive)} threads did not come back", flush=True)
        os._exit(3)
    if errors:
        print("MISMATCH\n" + errors[0], flush=True)
        os._exit(1)
    print(f"combined_calls {ctx.get_option('combined_calls')}; soak ok", flush=True)
    ctx.close()
    octx.close()


if __name__ == "__main__":
    main()
