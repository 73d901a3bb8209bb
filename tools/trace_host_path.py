#!/usr/bin/env python3
"""A few host-pointer commitment calls for a rocprofv3 kernel + memory-copy trace (timeline of the chunked pipeline):
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace_host -- python3 tools/trace_host_path.py [bits] [chunk]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import rand_blob, load_product, BLOB  # noqa: E402


def main():
    bits = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    import torch
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context(device=0)
    ctx.set_option("commit_direct_bits", bits)
    ctx.set_option("host_chunk", chunk)
    n = 256
    blobs = [rand_blob(s % 16) for s in range(n)]
    keep = [ctypes.create_string_buffer(b, BLOB) for b in blobs]
    arr = (ctypes.c_char_p * n)(*[ctypes.cast(k, ctypes.c_char_p) for k in keep])
    out = ctypes.create_string_buffer(48 * n)
    st = (ctypes.c_int * n)()
    L = kzg.lib()
    for i in range(5):
        t0 = time.perf_counter()
        rc = L.mkzg_blob_to_kzg_commitment_batch(ctx._h, arr, n, out, st)
        print(f"call {i}: rc={rc} {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
