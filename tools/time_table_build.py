#!/usr/bin/env python3
"""tools/time_table_build.py -- wall time of the first commitment call (= direct table construction) per commit_direct_bits."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import rand_blob, load_product  # noqa: E402
import torch
torch.cuda.init()
kzg = load_product()
blob = rand_blob(1)
for bits in (12, 13, 14, 15):
    ctx = kzg.Context()
    ctx.set_option("commit_direct_bits", bits)
    t0 = time.perf_counter()
    rc, cm = ctx.BlobToKZGCommitment(blob)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    rc2, cm2 = ctx.BlobToKZGCommitment(blob)
    print(f"bits={bits} first call {dt:.3f} s (table build), second {time.perf_counter() - t0:.4f} s, rc={rc} same={cm == cm2}")
    ctx.close()
