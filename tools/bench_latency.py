#!/usr/bin/env python3
"""tools/bench_latency.py -- single-call latency of the reference's per-blob API surface through the C ABI (host pointers in,
host bytes out, ctypes binding): what a drop-in caller that does NOT batch sees.  Run on the GPU box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import rand_blob, load_product  # noqa: E402


def timeit(fn, reps=10):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return round((time.perf_counter() - t0) / reps * 1e3, 3)


def main():
    import torch
    torch.cuda.init()
    kzg = load_product()
    out = {}
    for label, bits in (("bucket", 0), ("direct15", 15)):
        ctx = kzg.Context()
        if bits:
            ctx.set_option("commit_direct_bits", bits)
        blob = rand_blob(7)
        rc, cm = ctx.BlobToKZGCommitment(blob)
        rc, pr = ctx.ComputeBlobKZGProof(blob, cm)
        z = (99).to_bytes(32, "big")
        rc, pz, y = ctx.ComputeKZGProof(blob, z)
        r = {"BlobToKZGCommitment_ms": timeit(lambda: ctx.BlobToKZGCommitment(blob)),
             "ComputeBlobKZGProof_ms": timeit(lambda: ctx.ComputeBlobKZGProof(blob, cm)),
             "ComputeKZGProof_ms": timeit(lambda: ctx.ComputeKZGProof(blob, z)),
             "VerifyBlobKZGProof_ms": timeit(lambda: ctx.VerifyBlobKZGProof(blob, cm, pr)),
             "VerifyKZGProof_ms": timeit(lambda: ctx.VerifyKZGProof(cm, z, y, pz))}
        if not bits:
            rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(blob)
            ids = list(range(0, 128, 2))
            r["ComputeCells_ms"] = timeit(lambda: ctx.ComputeCells(blob))
            r["ComputeCellsAndKZGProofs_ms"] = timeit(lambda: ctx.ComputeCellsAndKZGProofs(blob), 5)
            r["RecoverCellsAndComputeKZGProofs_ms"] = timeit(lambda: ctx.RecoverCellsAndComputeKZGProofs(ids, [cells[i] for i in ids]), 5)
            r["VerifyCellKZGProofBatch_128_cells_ms"] = timeit(lambda: ctx.VerifyCellKZGProofBatch([cm] * 128, list(range(128)), cells, proofs), 5)
        out[label] = r
        ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
