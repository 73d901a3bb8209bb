#!/usr/bin/env python3
"""tools/lat_hostfin.py -- option host_fin 0 / 1 alternating in one process: latency of the host-pointer entry points whose last step is
one point's inversion + compression (BlobToKZGCommitment, a batch of four host blobs, ComputeBlobKZGProof, ComputeKZGProof): on the
device a ~90 us one-wavefront chain the caller waits for, on the calling thread ~8 us.  Same bytes either way (asserted)."""
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
    blobs = [rand_blob(7 + i) for i in range(4)]
    z = (99).to_bytes(32, "big")
    rc, cm = ctx.BlobToKZGCommitment(blobs[0])
    rc, pr = ctx.ComputeBlobKZGProof(blobs[0], cm)
    rc, pz, y = ctx.ComputeKZGProof(blobs[0], z)

    def timeit(fn, reps=200):
        for _ in range(5):
            fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps * 1e3

    for rep in range(3):
        for hf in (0, 1):
            assert ctx.set_option("host_fin", hf) == 0
            assert ctx.BlobToKZGCommitment(blobs[0]) == (0, cm) and ctx.ComputeBlobKZGProof(blobs[0], cm) == (0, pr) and ctx.ComputeKZGProof(blobs[0], z) == (0, pz, y)
            t1 = timeit(lambda: ctx.BlobToKZGCommitment(blobs[0]))
            t4 = timeit(lambda: ctx.BlobToKZGCommitmentBatch(blobs), 100)
            tp = timeit(lambda: ctx.ComputeBlobKZGProof(blobs[0], cm), 100)
            tk = timeit(lambda: ctx.ComputeKZGProof(blobs[0], z), 100)
            print(f"host_fin={hf}: BlobToKZGCommitment {t1:.3f} ms, batch of 4 host blobs {t4:.3f} ms, ComputeBlobKZGProof {tp:.3f} ms, ComputeKZGProof {tk:.3f} ms", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
