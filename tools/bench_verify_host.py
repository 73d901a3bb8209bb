#!/usr/bin/env python3
"""Host-pointer VerifyBlobKZGProofBatch (Go's []*Blob) at the C ABI: bench.py's `drop_in.verify_blob_kzg_proof_batch` leg on its own,
under the option sets given on the command line.  GPU box.
    python3 tools/bench_verify_host.py [N] [key=value ...] [-- key=value ...]      # option sets separated by --
e.g. python3 tools/bench_verify_host.py 1024 host_sha=0 -- host_sha=1"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    print(json.dumps({"numa": bench.bind_to_gpu_node(torch, 0)}), flush=True)      # as bench.py does (BENCH_NO_NUMA_BIND=1: leave the process alone)
    kzg = bench.load_product()
    ctx = kzg.Context(device=0)
    argv = sys.argv[1:]
    N = 1024
    if argv and argv[0].isdigit():
        N = int(argv.pop(0))
    sets, cur = [], {}
    for a in argv:
        if a == "--":
            sets.append(cur)
            cur = {}
        else:
            k, v = a.split("=")
            cur[k] = int(v)
    sets.append(cur)
    callers = tuple(int(x) for x in os.environ.get("VH_CALLERS", "4,8").split(",") if x)
    for opts in sets:
        for k, v in opts.items():
            assert ctx.set_option(k, v) == 0, (k, v)
        r = bench.bench_drop_in_verify(ctx, kzg, torch, N=N, callers=callers)
        print(json.dumps({"options": opts, **r}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
