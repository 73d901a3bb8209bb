#!/usr/bin/env python3
"""What the GPU box's host side offers (for sizing the host pool and the host-pointer pipelines): CPUs actually usable, SHA-256 and
memcpy throughput against thread count, pinned / pageable PCIe bandwidth, hipHostRegister cost.  Prints one JSON object."""
import hashlib
import json
import os
import subprocess
import threading
import time


def sh(cmd):
    try:
        return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception as e:
        return f"<{e}>"


def par(nthr, fn):
    ts = [threading.Thread(target=fn, args=(q,)) for q in range(nthr)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return time.perf_counter() - t0


def main():
    out = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "nproc": sh("nproc"),
           "cgroup_cpu_max": sh("cat /sys/fs/cgroup/cpu.max 2>/dev/null || cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us"),
           "model": sh("grep -m1 'model name' /proc/cpuinfo"), "sha_ni": "sha_ni" in sh("grep -m1 flags /proc/cpuinfo"),
           "sockets": sh("lscpu | grep -E 'Socket|Thread|Core|NUMA node\\(s\\)'"), "mem": sh("free -g | sed -n 2p"),
           "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
    import numpy as np
    blob = np.random.randint(0, 256, 131072, dtype=np.uint8).tobytes()
    # SHA-256 (OpenSSL: SHA-NI when the CPU has it) of 131 KB messages, GB/s against threads
    sha = {}
    for nthr in (1, 4, 8, 16, 32, 64, 128):
        reps = 64
        dt = par(nthr, lambda q: [hashlib.sha256(blob).digest() for _ in range(reps)])
        sha[nthr] = round(nthr * reps * 131072 / dt / 1e9, 2)
    out["sha256_GBps_by_threads"] = sha
    # gather: memcpy of scattered 128 KiB blocks into one buffer
    src = [np.random.randint(0, 256, 131072, dtype=np.uint8) for _ in range(256)]
    dst = np.empty(256 * 131072, dtype=np.uint8)
    mc = {}
    for nthr in (1, 2, 4, 8, 16, 32):
        def work(q):
            for i in range(q, 256, nthr):
                dst[i * 131072:(i + 1) * 131072] = src[i]
        par(nthr, work)
        dt = min(par(nthr, work) for _ in range(5))
        mc[nthr] = round(256 * 131072 / dt / 1e9, 2)
    out["gather_GBps_by_threads"] = mc
    import torch
    torch.cuda.init()
    out["gpu"] = torch.cuda.get_device_name(0)
    free, total = torch.cuda.mem_get_info()
    out["hbm_free_GB"] = round(free / 1e9, 1)
    n = 128 << 20
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    hp = torch.empty(n, dtype=torch.uint8).pin_memory()
    hq = torch.empty(n, dtype=torch.uint8)
    for name, h in (("pinned", hp), ("pageable", hq)):
        d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        out[f"h2d_{name}_GBps"] = round(5 * n / (time.perf_counter() - t0) / 1e9, 2)
    t0 = time.perf_counter()
    for _ in range(5):
        hp.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    out["d2h_pinned_GBps"] = round(5 * n / (time.perf_counter() - t0) / 1e9, 2)
    # small pinned copies: latency of a 8 MB and a 48 KB transfer
    for sz in (8 << 20, 48 << 10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            d[:sz].copy_(hp[:sz], non_blocking=True)
        torch.cuda.synchronize()
        out[f"h2d_pinned_{sz >> 10}KB_us"] = round((time.perf_counter() - t0) / 20 * 1e6, 1)
    # hipHostRegister of 33.5 MB of pageable memory (what pinning a caller's batch in place would cost)
    buf = torch.empty(256 * 131072, dtype=torch.uint8)
    rt = torch.cuda.cudart()
    t0 = time.perf_counter()
    rc = rt.cudaHostRegister(buf.data_ptr(), buf.numel(), 0)
    t1 = time.perf_counter()
    rt.cudaHostUnregister(buf.data_ptr())
    out["hostRegister_33MB_ms"] = [round((t1 - t0) * 1e3, 3), round((time.perf_counter() - t1) * 1e3, 3), int(rc)]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
