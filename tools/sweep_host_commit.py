#!/usr/bin/env python3
"""tools/sweep_host_commit.py -- host-pointer BlobToKZGCommitment batch of 256 (Go's []*Blob, reference prove.go:13-33) from ONE synchronous
caller over host_chunk x host_gather_tasks x host_zero_copy, beside the device-resident call on the same box (the ratio is what the
drop-in caller gets of the kernel rate).  GPU box.   python3 tools/sweep_host_commit.py [chunk,chunk,...] [tasks,tasks,...]"""
import ctypes
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import fast_blobs, load_product, BLOB  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context(device=0)
    n = 256
    hb = fast_blobs(n, 11)
    keep = [ctypes.create_string_buffer(hb[i].tobytes(), BLOB) for i in range(n)]
    arr = (ctypes.c_char_p * n)(*[ctypes.cast(k, ctypes.c_char_p) for k in keep])
    d_b = torch.from_numpy(hb.reshape(-1)).cuda()
    d_o = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
    d_s = torch.zeros(n, dtype=torch.int32, device="cuda")
    L = kzg.lib()
    st = torch.cuda.current_stream().cuda_stream
    out = ctypes.create_string_buffer(48 * n)
    sts = (ctypes.c_int * n)()

    def dev_ms():
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            assert ctx.blob_to_kzg_commitment_batch_device(d_b.data_ptr(), n, d_o.data_ptr(), d_s.data_ptr(), st) == 0
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return statistics.median(ts)

    def host_ms():
        ts = []
        for _ in range(12):
            t0 = time.perf_counter()
            assert L.mkzg_blob_to_kzg_commitment_batch(ctx._h, arr, n, out, sts) == 0
            ts.append((time.perf_counter() - t0) * 1e3)
        return statistics.median(ts[2:])

    dev_ms()
    d0 = dev_ms()
    expect = d_o.cpu().numpy().tobytes()
    print(f"device-resident, one synchronous caller: {d0:.3f} ms per 256 blobs = {n / d0:.1f} k blobs/s", flush=True)
    chunks = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32, 64, 128]
    tasks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 8, 16]
    for zc in (1, 0):
        assert ctx.set_option("host_zero_copy", zc) == 0
        for ch in chunks:
            for tk in tasks:
                assert ctx.set_option("host_chunk", ch) == 0 and ctx.set_option("host_gather_tasks", tk) == 0
                h = host_ms()
                assert out.raw == expect
                print(f"zero_copy={zc} host_chunk={ch:4d} gather_tasks={tk:3d}: {h:.3f} ms = {n / h:.1f} k blobs/s = {100 * d0 / h:.1f} % of device-resident", flush=True)
    d1 = dev_ms()
    print(f"device-resident again: {d1:.3f} ms", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
