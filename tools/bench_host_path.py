#!/usr/bin/env python3
"""tools/bench_host_path.py -- PCIe-inclusive rate of the host-pointer entry point
mkzg_blob_to_kzg_commitment_batch (blobs in pageable host memory -> commitments in host memory), next to the
device-resident rate bench.py reports.  Run on the GPU box."""
import hashlib, json, os, struct, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_product
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def rand_blob(seed):
    out = bytearray()
    for i in range(0, 131072, 32):
        out += (int.from_bytes(hashlib.sha256(struct.pack(">q", seed + i)).digest(), "big") % R).to_bytes(32, "big")
    return bytes(out)


def main():
    import torch
    torch.cuda.init()
    kzg = load_product()
    res = {}
    for bits in (0, 15):
        ctx = kzg.Context()
        ctx.set_option("max_batch", 256)
        if bits:
            ctx.set_option("commit_direct_bits", bits)
        blobs = [rand_blob(s) for s in range(256)]
        rc, outs, st = ctx.BlobToKZGCommitmentBatch(blobs)          # warm-up (tables)
        assert rc == 0
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            rc, outs2, st = ctx.BlobToKZGCommitmentBatch(blobs)
        dt = (time.perf_counter() - t0) / reps
        assert outs2 == outs
        res["direct15" if bits else "bucket13"] = {"ms_per_256_blobs_host_to_host": round(dt * 1e3, 3), "blobs_per_s": round(256 / dt, 1)}
        ctx.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
