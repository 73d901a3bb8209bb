#!/usr/bin/env python3
"""tools/ab_host.py -- host-pointer paths at the C ABI for ONE library build (MI355KZG_LIB selects it): commitments of 256 blobs from
one and two callers, VerifyBlobKZGProofBatch of 1024, VerifyCellKZGProofBatch of 8192 cells; medians of several trials.  Run it once per
build on the same box to A/B host-side changes (pool, chunking).  GPU box."""
import ctypes
import json
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import rand_blob, load_product, BLOB  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context(device=0)
    ctx.set_option("commit_direct_bits", 15)
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        assert ctx.set_option(k, int(v)) == 0, kv
    n = 256
    blobs = [rand_blob(s % 16) for s in range(n)]
    keep = [ctypes.create_string_buffer(b, BLOB) for b in blobs]
    arr = (ctypes.c_char_p * n)(*[ctypes.cast(k, ctypes.c_char_p) for k in keep])
    L = kzg.lib()
    out = {"lib": os.path.basename(kzg.LIB_PATH)}

    def caller(reps, res):
        o = ctypes.create_string_buffer(48 * n)
        st = (ctypes.c_int * n)()
        for _ in range(reps):
            assert L.mkzg_blob_to_kzg_commitment_batch(ctx._h, arr, n, o, st) == 0
        res.append(o.raw)

    res = []
    caller(3, res)
    one, two = [], []
    for trial in range(5):
        reps = 8
        t0 = time.perf_counter()
        caller(reps, res)
        one.append((time.perf_counter() - t0) / reps)
        ts = [threading.Thread(target=caller, args=(reps, res)) for _ in range(2)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        two.append((time.perf_counter() - t0) / (2 * reps))
    if os.environ.get("AB_COMMIT_ONLY"):
        print(json.dumps({"args": sys.argv[1:], "one_ms": round(statistics.median(one) * 1e3, 3), "one_min_ms": round(min(one) * 1e3, 3), "two_blobs_s": round(n / statistics.median(two))}), flush=True)
        ctx.close()
        return
    out["commit256"] = {"one_ms": round(statistics.median(one) * 1e3, 3), "one_min_ms": round(min(one) * 1e3, 3), "one_blobs_s": round(n / statistics.median(one)),
                        "two_blobs_s": round(n / statistics.median(two)), "two_best": round(n / min(two))}
    # verifier, 1024 host blobs (16 distinct)
    N = 1024
    cm, pr = [], []
    for s in range(16):
        rc, c = ctx.BlobToKZGCommitment(blobs[s])
        rc2, p = ctx.ComputeBlobKZGProof(blobs[s], c)
        assert rc == 0 and rc2 == 0
        cm.append(c)
        pr.append(p)
    arrv = (ctypes.c_char_p * N)(*[ctypes.cast(keep[i % 16], ctypes.c_char_p) for i in range(N)])
    cmj = b"".join(cm[i % 16] for i in range(N))
    prj = b"".join(pr[i % 16] for i in range(N))
    assert L.mkzg_verify_blob_kzg_proof_batch(ctx._h, arrv, cmj, prj, N, N, N) == 0
    tv = []
    for trial in range(5):
        t0 = time.perf_counter()
        for _ in range(4):
            assert L.mkzg_verify_blob_kzg_proof_batch(ctx._h, arrv, cmj, prj, N, N, N) == 0
        tv.append((time.perf_counter() - t0) / 4)
    out["verify1024_host"] = {"ms": round(statistics.median(tv) * 1e3, 3), "min_ms": round(min(tv) * 1e3, 3)}
    for nb in (6, 48):
        a = (ctypes.c_char_p * nb)(*[ctypes.cast(keep[i % 16], ctypes.c_char_p) for i in range(nb)])
        assert L.mkzg_verify_blob_kzg_proof_batch(ctx._h, a, cmj, prj, nb, nb, nb) == 0
        t0 = time.perf_counter()
        for _ in range(10):
            assert L.mkzg_verify_blob_kzg_proof_batch(ctx._h, a, cmj, prj, nb, nb, nb) == 0
        out[f"verify{nb}_host_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
    # cell verifier, 8192 cells of 64 blobs (16 distinct)
    cells, proofs = [], []
    for s in range(16):
        rc, c, p = ctx.ComputeCellsAndKZGProofs(blobs[s])
        assert rc == 0
        cells.append([ctypes.create_string_buffer(x, 2048) for x in c])
        proofs.append(p)
    nc = 8192
    arrc = (ctypes.c_char_p * nc)(*[ctypes.cast(cells[(i // 128) % 16][i % 128], ctypes.c_char_p) for i in range(nc)])
    idx = (ctypes.c_uint64 * nc)(*[i % 128 for i in range(nc)])
    cj = b"".join(cm[(i // 128) % 16] for i in range(nc))
    pj = b"".join(proofs[(i // 128) % 16][i % 128] for i in range(nc))
    assert L.mkzg_verify_cell_kzg_proof_batch(ctx._h, cj, nc, idx, nc, arrc, nc, pj, nc) == 0
    tc = []
    for trial in range(5):
        t0 = time.perf_counter()
        for _ in range(4):
            assert L.mkzg_verify_cell_kzg_proof_batch(ctx._h, cj, nc, idx, nc, arrc, nc, pj, nc) == 0
        tc.append((time.perf_counter() - t0) / 4)
    out["cellverify8192"] = {"ms": round(statistics.median(tc) * 1e3, 3), "min_ms": round(min(tc) * 1e3, 3)}
    print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
