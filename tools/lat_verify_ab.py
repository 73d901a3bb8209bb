#!/usr/bin/env python3
"""tools/lat_verify_ab.py -- median latency of VerifyBlobKZGProof / VerifyKZGProof (host pointers, one caller) for A/B runs of host-side
settings passed in the environment, e.g.  for s in 1 2 3; do MKZG_MILLER_SEGMENTS=$s python3 tools/lat_verify_ab.py; done"""
import os, sys, time, statistics
sys.path.insert(0, os.getcwd())
from bench import rand_blob, load_product
import torch
torch.cuda.init()
kzg = load_product()
ctx = kzg.Context()
blob = rand_blob(7)
rc, cm = ctx.BlobToKZGCommitment(blob)
rc, pr = ctx.ComputeBlobKZGProof(blob, cm)
z = (99).to_bytes(32, "big")
rc, pz, y = ctx.ComputeKZGProof(blob, z)
def t(fn, reps=30):
    fn(); fn()
    xs = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); xs.append((time.perf_counter() - t0) * 1e3)
    return round(statistics.median(xs), 3)
print(os.environ.get("MKZG_MILLER_SEGMENTS"), "VerifyBlobKZGProof", t(lambda: ctx.VerifyBlobKZGProof(blob, cm, pr)), "VerifyKZGProof", t(lambda: ctx.VerifyKZGProof(cm, z, y, pz)), "host_finish_us", ctx.get_option("last_host_finish_us") if hasattr(ctx, "get_option") else None)
