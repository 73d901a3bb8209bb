#!/usr/bin/env python3
"""Host-pointer (drop-in) commitment path: mkzg_blob_to_kzg_commitment_batch on 256 scattered pageable blobs at the C ABI,
A/B over the pipeline's options (zero-copy reads of the pinned staging area, chunk size), one and two callers.  GPU box."""
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import rand_blob, load_product, BLOB  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    bits = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    kzg = load_product()
    ctx = kzg.Context(device=0)
    ctx.set_option("commit_direct_bits", bits)
    n = 256
    blobs = [rand_blob(s % 32) for s in range(n)]
    keep = [ctypes.create_string_buffer(b, BLOB) for b in blobs]
    arr = (ctypes.c_char_p * n)(*[ctypes.cast(k, ctypes.c_char_p) for k in keep])
    L = kzg.lib()
    ref = None
    out = {}

    def caller(reps, res):
        o = ctypes.create_string_buffer(48 * n)
        st = (ctypes.c_int * n)()
        for _ in range(reps):
            assert L.mkzg_blob_to_kzg_commitment_batch(ctx._h, arr, n, o, st) == 0
        res.append(o.raw)

    for zc in (0, 1):
        for chunk in (32, 64, 128, 256):
            ctx.set_option("host_zero_copy", zc)
            ctx.set_option("host_chunk", chunk)
            res = []
            caller(3, res)
            ref = ref or res[0]
            assert res[0] == ref
            reps = 10
            t0 = time.perf_counter()
            caller(reps, res)
            dt1 = (time.perf_counter() - t0) / reps
            ts = [threading.Thread(target=caller, args=(reps, res)) for _ in range(2)]
            t0 = time.perf_counter()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            dt2 = (time.perf_counter() - t0) / (2 * reps)
            assert all(r == ref for r in res)
            out[f"zc{zc}_chunk{chunk}"] = {"ms_one_caller": round(dt1 * 1e3, 3), "blobs_per_s_one": round(n / dt1), "blobs_per_s_two": round(n / dt2)}
            print(f"zc{zc}_chunk{chunk}", out[f"zc{zc}_chunk{chunk}"], flush=True)
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
