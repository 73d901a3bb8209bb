#!/usr/bin/env python3
"""tools/lat_hostval.py -- option host_validate 0 / 1 alternating in one process: latency of the entry points whose critical path is the
decoding + r-torsion test of a FEW points (VerifyKZGProof, VerifyBlobKZGProof, ComputeBlobKZGProof, host-pointer VerifyBlobKZGProofBatch of
1..9 blobs -- a block's blobs): 0.64 ms on a quad of GPU lanes however few the points, ~45 us per point on a host core.  Same verdicts."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rand_blob, load_product  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context()
    blobs = [rand_blob(50 + i) for i in range(48)]
    cms = [ctx.BlobToKZGCommitment(b)[1] for b in blobs]
    prs = [ctx.ComputeBlobKZGProof(b, c)[1] for b, c in zip(blobs, cms)]
    z = (99).to_bytes(32, "big")
    rc, pz, y = ctx.ComputeKZGProof(blobs[0], z)

    def timeit(fn, reps=100):
        for _ in range(5):
            fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps * 1e3

    for rep in range(2):
        for hv in (0, 1):
            assert ctx.set_option("host_validate", hv) == 0
            assert ctx.VerifyKZGProof(cms[0], z, y, pz) == 0 and ctx.VerifyBlobKZGProof(blobs[0], cms[0], prs[0]) == 0 and ctx.VerifyBlobKZGProof(blobs[0], cms[0], prs[1]) == 1
            r = {"VerifyKZGProof": timeit(lambda: ctx.VerifyKZGProof(cms[0], z, y, pz)),
                 "VerifyBlobKZGProof": timeit(lambda: ctx.VerifyBlobKZGProof(blobs[0], cms[0], prs[0])),
                 "ComputeBlobKZGProof": timeit(lambda: ctx.ComputeBlobKZGProof(blobs[0], cms[0]))}
            for n in (2, 6, 9, 16, 32, 33, 48):
                assert ctx.VerifyBlobKZGProofBatch(blobs[:n], cms[:n], prs[:n]) == 0
                bad = list(prs[:n])
                bad[n - 1] = prs[0] if n > 1 else prs[1]
                assert ctx.VerifyBlobKZGProofBatch(blobs[:n], cms[:n], bad) == 1
                r[f"VerifyBlobKZGProofBatch({n})"] = timeit(lambda: ctx.VerifyBlobKZGProofBatch(blobs[:n], cms[:n], prs[:n]), 50)
            print(f"host_validate={hv}: " + ", ".join(f"{k} {v:.3f}" for k, v in r.items()) + "  (ms)", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
