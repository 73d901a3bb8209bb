#!/usr/bin/env python3
"""Each per-blob entry point of the reference's API three times in a row (host pointers, default options) for a rocprofv3 kernel +
memory-copy trace -- what bounds a caller that does not batch:
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace_single -- python3 tools/trace_single_calls.py
    python tools/timeline.py gpurun_out/trace_single"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import rand_blob, load_product  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context()
    blob = rand_blob(7)
    rc, cm = ctx.BlobToKZGCommitment(blob)
    rc, pr = ctx.ComputeBlobKZGProof(blob, cm)
    z = (99).to_bytes(32, "big")
    rc, pz, y = ctx.ComputeKZGProof(blob, z)
    rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(blob)
    ids = list(range(0, 128, 2))
    half = [cells[i] for i in ids]
    calls = [("BlobToKZGCommitment", lambda: ctx.BlobToKZGCommitment(blob)),
             ("ComputeBlobKZGProof", lambda: ctx.ComputeBlobKZGProof(blob, cm)),
             ("ComputeKZGProof", lambda: ctx.ComputeKZGProof(blob, z)),
             ("VerifyBlobKZGProof", lambda: ctx.VerifyBlobKZGProof(blob, cm, pr)),
             ("VerifyKZGProof", lambda: ctx.VerifyKZGProof(cm, z, y, pz)),
             ("ComputeCellsAndKZGProofs", lambda: ctx.ComputeCellsAndKZGProofs(blob)),
             ("RecoverCellsAndComputeKZGProofs", lambda: ctx.RecoverCellsAndComputeKZGProofs(ids, half)),
             ("VerifyCellKZGProofBatch128", lambda: ctx.VerifyCellKZGProofBatch([cm] * 128, list(range(128)), cells, proofs))]
    only = sys.argv[1:] or [c[0] for c in calls]
    for name, fn in calls:
        if name not in only:
            continue
        fn()
        torch.cuda.synchronize()
        time.sleep(0.02)                          # a visible gap in the trace between the groups
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            print(f"{name}: {1e3 * (time.perf_counter() - t0):.3f} ms host_finish {ctx.last_host_finish_us():.0f} us", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
