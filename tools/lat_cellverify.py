#!/usr/bin/env python3
"""tools/lat_cellverify.py -- latency of host-pointer VerifyCellKZGProofBatch at the sizes a PeerDAS node sees (EIP-7594): one column
sidecar = one cell of every blob of the block (6..9 cells, as many commitments), a custody set of 8 columns (48..72 cells), one blob's
128 cells; option host_validate 0 / 1 alternating in one process.  Same verdicts (a swapped proof must be rejected)."""
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
    NBLOB = 9
    blobs = [rand_blob(80 + i) for i in range(NBLOB)]
    cms = [ctx.BlobToKZGCommitment(b)[1] for b in blobs]
    cp = [ctx.ComputeCellsAndKZGProofs(b) for b in blobs]
    assert all(x[0] == 0 for x in cp)

    def batch(n_blobs, cols):
        c, idx, cells, proofs = [], [], [], []
        for col in cols:
            for b in range(n_blobs):
                c.append(cms[b]); idx.append(col); cells.append(cp[b][1][col]); proofs.append(cp[b][2][col])
        return c, idx, cells, proofs

    shapes = {"1 column x 6 blobs": batch(6, [5]), "1 column x 9 blobs": batch(9, [77]), "8 columns x 6 blobs": batch(6, list(range(3, 128, 16))),
              "8 columns x 9 blobs": batch(9, list(range(1, 128, 16))), "one blob's 128 cells": batch(1, list(range(128)))}

    def timeit(fn, reps=60):
        for _ in range(5):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2] * 1e3

    for rep in range(2):
        for hv in (0, 1):
            assert ctx.set_option("host_validate", hv) == 0
            r = {}
            for name, (c, idx, cells, proofs) in shapes.items():
                assert ctx.VerifyCellKZGProofBatch(c, idx, cells, proofs) == 0
                if len(proofs) > 1:
                    assert ctx.VerifyCellKZGProofBatch(c, idx, cells, proofs[1:] + proofs[:1]) == 1
                r[f"{name} ({len(idx)} cells)"] = timeit(lambda: ctx.VerifyCellKZGProofBatch(c, idx, cells, proofs))
            print(f"host_validate={hv}: " + ", ".join(f"{k} {v:.3f}" for k, v in r.items()) + "  (ms, median)", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
