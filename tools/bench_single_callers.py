#!/usr/bin/env python3
"""tools/bench_single_callers.py -- K threads of single-blob calls on one context (bench.py's drop_in.single_blob_callers leg on its own):
    COMBINE=0|1 (option combine_single_calls), KINDS=blob_to_kzg_commitment,compute_cells_and_kzg_proofs,verify_blob_kzg_proof,
    CALLERS=8,32  WRONG_EVERY=m (every m-th verification brings a wrong proof)  SECONDS=0.35"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    kzg = bench.load_product()
    ctx = kzg.Context(device=0)
    if os.environ.get("COMBINE") is not None:
        assert ctx.set_option("combine_single_calls", int(os.environ["COMBINE"])) == 0
    kinds = os.environ["KINDS"].split(",") if os.environ.get("KINDS") else None
    callers = tuple(int(x) for x in os.environ.get("CALLERS", "8,32").split(","))
    out = bench.bench_single_blob_callers(ctx, kzg, torch, callers=callers, kinds=kinds, wrong_every=int(os.environ.get("WRONG_EVERY", "0")),
                                          seconds=float(os.environ.get("SECONDS", "0.35")))
    for kind, v in out.items():
        if isinstance(v, dict):
            print(kind, " ".join(f"{k}: {d['blobs_per_s']:.0f}/s ({d['host_cpus_busy']} CPUs)" for k, d in v.items()), flush=True)
    print(json.dumps({k: v for k, v in out.items() if not isinstance(v, dict)}))
    ctx.close()


if __name__ == "__main__":
    main()
