#!/usr/bin/env python3
"""tools/trace.py -- the workloads this project traces with rocprofv3, and the readers of the traces, in one place.

Workloads (run under the profiler on the GPU box; the program itself goes after `--`):
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/t -- python3 tools/trace.py single [CallName ...]
        every per-blob entry point of the reference's API three times in a row (host pointers, default options): what bounds a caller
        that does not batch
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/t -- python3 tools/trace.py verify_host
        four host-pointer VerifyBlobKZGProofBatch calls of 1024 blobs (gather / upload / kernels overlap)
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/t -- python3 tools/trace.py verify_inflight [K=8] [R=6]
        K caller threads x R device-resident VerifyBlobKZGProofBatch calls of 1024 on one context: the sustained leg (hash rounds)
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/t -- python3 tools/trace.py msm_paths
        ComputeCellsAndKZGProofs and the recovery batch alternated in one process (IDS=even|first|all picks the supplied cells): the
        same FK20 launches behind different predecessors (the sparse-wavefront finding of DESIGN.md section 7)
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/t -- python3 tools/trace.py cells_batch [N=128] [R=3]
        R device-resident ComputeCellsAndKZGProofs calls of N blobs, one caller: the launches of one piece and the gaps between them
    rocprofv3 --kernel-trace --stats -d gpurun_out/t -- python3 tools/trace.py lone [R=5]
        R lone device-resident blob-verifier calls under each shape + R lone cell-verifier calls (cells in HBM): isolated kernel durations
Readers (anywhere):
    python3 tools/trace.py timeline DIR [t_from_ms t_to_ms]   kernel + copy records merged into one time-ordered listing
    python3 tools/trace.py busy DIR [gap_ms]                  last burst of kernels: wall time, time with a kernel running, kernels in flight
    python3 tools/trace.py summary KERNEL_TRACE.csv [SUBSTR]  per-launch durations of one kernel by grid size and hardware queue
"""
import csv
import ctypes
import ctypes as C
import glob
import os
import sys
import threading
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _product():
    from bench import BLOB, load_product, rand_blob
    return BLOB, load_product, rand_blob


def wl_single(args):
    BLOB, load_product, rand_blob = _product()

    import torch
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context()
    blob = rand_blob(7)
    rc, cm = ctx.BlobToKZGCommitment(blob)
    rc, pr = ctx.ComputeBlobKZGProof(blob, cm)
    z = (99).to_bytes(32, "big")
    rc, pz, y = ctx.ComputeKZGProof(blob, z)
    rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(blob)
    ids = list(range(0, 128, 2))
    half = [cells[i] for i in ids]
    calls = [("BlobToKZGCommitment", lambda: ctx.BlobToKZGCommitment(blob)),
             ("ComputeBlobKZGProof", lambda: ctx.ComputeBlobKZGProof(blob, cm)),
             ("ComputeKZGProof", lambda: ctx.ComputeKZGProof(blob, z)),
             ("VerifyBlobKZGProof", lambda: ctx.VerifyBlobKZGProof(blob, cm, pr)),
             ("VerifyKZGProof", lambda: ctx.VerifyKZGProof(cm, z, y, pz)),
             ("ComputeCellsAndKZGProofs", lambda: ctx.ComputeCellsAndKZGProofs(blob)),
             ("RecoverCellsAndComputeKZGProofs", lambda: ctx.RecoverCellsAndComputeKZGProofs(ids, half)),
             ("VerifyCellKZGProofBatch128", lambda: ctx.VerifyCellKZGProofBatch([cm] * 128, list(range(128)), cells, proofs))]
    only = args or [c[0] for c in calls]
    for name, fn in calls:
        if name not in only:
            continue
        fn()
        torch.cuda.synchronize()
        time.sleep(0.02)                          # a visible gap in the trace between the groups
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            print(f"{name}: {1e3 * (time.perf_counter() - t0):.3f} ms host_finish {ctx.last_host_finish_us():.0f} us", flush=True)
    ctx.close()


def wl_verify_host(args):
    BLOB, load_product, rand_blob = _product()

    import torch
    import bench
    print("numa:", bench.bind_to_gpu_node(torch, 0), flush=True)      # as bench.py does: the caller's blobs then live on the GPU's NUMA node
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context(device=0)
    N = 1024
    base = [rand_blob(s) for s in range(32)]
    d_blobs = torch.frombuffer(bytearray(b"".join(base)), dtype=torch.uint8).cuda()
    d_cm = torch.zeros(32 * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(32 * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(32, dtype=torch.int32, device="cuda")
    assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), 32, d_cm.data_ptr(), d_st.data_ptr(), 0) == 0
    assert ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), 32, d_pr.data_ptr(), d_st.data_ptr(), 0) == 0
    torch.cuda.synchronize()
    cm, pr = d_cm.cpu().numpy().tobytes(), d_pr.cpu().numpy().tobytes()
    keep = [ctypes.create_string_buffer(base[i % 32], BLOB) for i in range(N)]
    arr = (ctypes.c_char_p * N)(*[ctypes.cast(k, ctypes.c_char_p) for k in keep])
    cms = b"".join(cm[48 * (i % 32):48 * (i % 32) + 48] for i in range(N))
    prs = b"".join(pr[48 * (i % 32):48 * (i % 32) + 48] for i in range(N))
    L = kzg.lib()
    for i in range(4):
        t0 = time.perf_counter()
        rc = L.mkzg_verify_blob_kzg_proof_batch(ctx._h, arr, cms, prs, N, N, N)
        print(f"call {i}: rc={rc} {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)
    ctx.close()


def wl_verify_inflight(args):
    BLOB, load_product, rand_blob = _product()

    import torch
    torch.cuda.init()
    K = int(args[0]) if len(args) > 0 else 8
    R = int(args[1]) if len(args) > 1 else 6
    kzg = load_product()
    ctx = kzg.Context(device=0)
    N = 1024
    base = [rand_blob(s) for s in range(32)]
    d_blobs = torch.frombuffer(bytearray(b"".join(base[i % 32] for i in range(N))), dtype=torch.uint8).cuda()
    d_cm = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(N, dtype=torch.int32, device="cuda")
    assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), N, d_cm.data_ptr(), d_st.data_ptr(), 0) == 0
    assert ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), N, d_pr.data_ptr(), d_st.data_ptr(), 0) == 0
    torch.cuda.synchronize()

    def caller(k, reps):
        s = torch.cuda.Stream(priority=-1 if k % 2 == 0 else 0)
        for _ in range(reps):
            assert ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, s.cuda_stream) == 0

    def task_cpu():
        """CPU seconds of every thread of the process so far, by thread id: (name, user + system)"""
        out = {}
        tick = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            try:
                with open(f"/proc/self/task/{tid}/stat") as fh:
                    st = fh.read()
                name = st[st.index("(") + 1:st.rindex(")")]
                f = st[st.rindex(")") + 2:].split()
                out[tid] = (name, (int(f[11]) + int(f[12])) / tick)
            except OSError:
                pass
        return out

    for trial, reps in enumerate((2, R)):
        ts = [threading.Thread(target=caller, args=(k, reps), name=f"caller{k}") for k in range(K)]
        torch.cuda.synchronize()
        time.sleep(0.05)
        t0, c0 = time.perf_counter(), time.process_time()
        before = task_cpu()
        for t in ts:
            t.start()
        tids = {}
        while any(t.is_alive() for t in ts):          # thread ids die with their threads: keep the last reading of each
            tids.update(task_cpu())
            time.sleep(0.02)
        for t in ts:
            t.join()
        dt, cpu = time.perf_counter() - t0, time.process_time() - c0
        if os.environ.get("TRACE_THREAD_CPU"):
            groups = defaultdict(lambda: [0, 0.0])
            for tid, (name, v) in tids.items():
                d = v - before.get(tid, (name, 0.0))[1]
                groups[name][0] += 1
                groups[name][1] += d
            print("  thread CPU: " + "; ".join(f"{k} x{n}: {c / dt:.2f} cores" for k, (n, c) in sorted(groups.items(), key=lambda kv: -kv[1][1]) if c > 0), flush=True)
        print(f"trial {trial}: {K} callers x {reps} calls: {K * reps * N / dt:.0f} blobs/s, {1e3 * dt / (K * reps):.3f} ms per call amortised; host CPUs busy {cpu / dt:.1f}", flush=True)
    ctx.close()


def wl_lone(args):
    """isolated kernel durations: R lone device-resident VerifyBlobKZGProofBatch(1024) calls under each verifier shape, then R lone
    VerifyCellKZGProofBatch(8192) calls with the cells in HBM -- read with `rocprofv3 --kernel-trace --stats`"""
    BLOB, load_product, rand_blob = _product()

    import torch
    torch.cuda.init()
    R = int(args[0]) if args else 5
    kzg = load_product()
    ctx = kzg.Context(device=0)
    N, NB = 1024, 64
    base = [rand_blob(s) for s in range(NB)]
    d_blobs = torch.frombuffer(bytearray(b"".join(base[i % NB] for i in range(N))), dtype=torch.uint8).cuda()
    d_cm = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(N, dtype=torch.int32, device="cuda")
    assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), N, d_cm.data_ptr(), d_st.data_ptr(), 0) == 0
    assert ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), N, d_pr.data_ptr(), d_st.data_ptr(), 0) == 0
    torch.cuda.synchronize()
    for shape in (0, 1):
        assert ctx.set_option("verify_shape", shape) == 0
        ts = []
        for i in range(R + 1):
            t0 = time.perf_counter()
            assert ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, 0) == 0
            ts.append(1e3 * (time.perf_counter() - t0))
        print(f"verify_shape {shape}: lone VerifyBlobKZGProofBatch(1024) ms " + " ".join(f"{t:.3f}" for t in ts[1:]), flush=True)
    assert ctx.set_option("verify_shape", -1) == 0
    d_cells = torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
    assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), 0) == 0
    torch.cuda.synchronize()
    cms, proofs = d_cm.cpu().numpy().tobytes(), d_proofs.cpu().numpy().tobytes()
    n = NB * 128
    idx = (ctypes.c_uint64 * n)(*[i % 128 for i in range(n)])
    cj = b"".join(cms[48 * (i // 128):48 * (i // 128) + 48] for i in range(n))
    L = kzg.lib()
    ts = []
    for i in range(R + 1):
        t0 = time.perf_counter()
        assert L.mkzg_verify_cell_kzg_proof_batch_device(ctx._h, cj, ctypes.cast(idx, ctypes.c_void_p), ctypes.c_void_p(d_cells.data_ptr()), proofs, n, ctypes.c_void_p(0)) == 0
        ts.append(1e3 * (time.perf_counter() - t0))
    print("lone VerifyCellKZGProofBatch(8192, cells in HBM) ms " + " ".join(f"{t:.3f}" for t in ts[1:]), flush=True)
    ctx.close()


def wl_msm_paths(args):
    BLOB, load_product, rand_blob = _product()

    import torch
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context(device=0)
    NB = 64
    blobs = [rand_blob(s) for s in range(NB)]
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(NB, dtype=torch.int32, device="cuda")

    def compute():
        assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), 0) == 0
        torch.cuda.synchronize()

    compute()
    cells, proofs = d_cells.cpu().numpy().tobytes(), d_proofs.cpu().numpy().tobytes()
    which = os.environ.get("IDS", "even")
    ids = list(range(0, 128, 2)) if which == "even" else (list(range(64)) if which == "first" else list(range(128)))
    m = len(ids)
    keep = [[C.create_string_buffer(cells[(b * 128 + i) * 2048:(b * 128 + i + 1) * 2048], 2048) for i in ids] for b in range(NB)]
    id_arrs = [(C.c_uint64 * m)(*ids) for _ in range(NB)]
    cell_arrs = [(C.c_char_p * m)(*[C.cast(k, C.c_char_p) for k in keep[b]]) for b in range(NB)]
    pid = (C.POINTER(C.c_uint64) * NB)(*[C.cast(a, C.POINTER(C.c_uint64)) for a in id_arrs])
    pcl = (C.POINTER(C.c_char_p) * NB)(*[C.cast(a, C.POINTER(C.c_char_p)) for a in cell_arrs])
    nid = (C.c_size_t * NB)(*[m] * NB)
    oc = C.create_string_buffer(NB * 262144)
    op = C.create_string_buffer(NB * 128 * 48)
    st = (C.c_int32 * NB)()
    L = kzg.lib()

    def recover():
        assert L.mkzg_recover_cells_and_kzg_proofs_batch(ctx._h, NB, pid, nid, pcl, nid, oc, op, st) == 0
        assert oc.raw == cells and op.raw == proofs

    for _ in range(2):
        compute()
    for _ in range(3):
        recover()
    for _ in range(2):
        compute()
    ctx.close()


def wl_cells_batch(args):
    BLOB, load_product, rand_blob = _product()
    import torch
    torch.cuda.init()
    kzg = load_product()
    n = int(args[0]) if args else 128
    reps = int(args[1]) if len(args) > 1 else 3
    ctx = kzg.Context()
    d_blobs = torch.frombuffer(bytearray(b"".join(rand_blob(s % 8) for s in range(n))), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(n * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(n * 128 * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(n, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for i in range(reps + 2):
        t0 = time.perf_counter()
        assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), n, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st) == 0
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"call {i}: returned after {1e3 * (t1 - t0):.3f} ms, finished after {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)
        time.sleep(0.02)
    ctx.close()


def rd_busy(args):
    d = args[0]
    gap = float(args[1]) * 1e6 if len(args) > 1 else 20e6
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kzg::", "")[:40]))
    rows.sort()
    bursts, cur, end = [], [rows[0]], rows[0][1]
    for r in rows[1:]:
        if r[0] - end > gap:
            bursts.append(cur)
            cur = []
        cur.append(r)
        end = max(end, r[1])
    bursts.append(cur)
    b = max(bursts[-3:], key=len)
    t0, t1 = b[0][0], max(r[1] for r in b)
    ev = sorted([(r[0], 1) for r in b] + [(r[1], -1) for r in b])
    busy = area = 0
    lvl, last = 0, t0
    for t, dlt in ev:
        if lvl > 0:
            busy += t - last
        area += lvl * (t - last)
        lvl += dlt
        last = t
    wall = t1 - t0
    print(f"burst: {len(b)} launches, wall {wall / 1e6:.3f} ms, some kernel running {100 * busy / wall:.1f} % of it, mean kernels in flight {area / wall:.2f}")
    agg = defaultdict(lambda: [0, 0])
    for s, e, n in b:
        agg[n][0] += 1
        agg[n][1] += e - s
    for n, (c, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:42s} {c:5d} launches  sum {tot / 1e6:9.3f} ms  avg {tot / c / 1e6:7.3f} ms  {100 * tot / wall:6.1f} % of wall")


def rd_timeline(args):
    d = args[0]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kzg::", "")[:40], r.get("Queue_Id", "?"), r.get("Grid_Size", "")))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "copy")[:40], "-", ""))
    rows.sort()
    if not rows:
        sys.exit("no trace rows")
    t0 = rows[0][0]
    lo = float(args[1]) if len(args) > 1 else 0.0
    hi = float(args[2]) if len(args) > 2 else 1e18
    for s, e, kind, name, q, g in rows:
        a = (s - t0) / 1e6
        if lo <= a <= hi:
            print(f"{a:10.3f} +{(e - s) / 1e6:8.3f} ms  {kind} q={q:>3} grid={g:>8}  {name}")


def rd_summary(args):
    path = args[0]
    sub = args[1] if len(args) > 1 else "k_commit_direct_acc"
    groups = defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if sub not in r["Kernel_Name"]:
                continue
            grid = (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
            groups[(grid, r["Queue_Id"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print(f"kernel *{sub}*: launches by (grid threads, hardware queue); durations in ms")
    for (grid, q), ds in sorted(groups.items()):
        print(f"  grid {grid} queue {q}: n={len(ds)} mean={sum(ds) / len(ds):.3f} min={min(ds):.3f} max={max(ds):.3f}   [{' '.join(f'{d:.2f}' for d in ds)}]")


COMMANDS = {"single": wl_single, "lone": wl_lone, "verify_host": wl_verify_host, "verify_inflight": wl_verify_inflight, "msm_paths": wl_msm_paths, "cells_batch": wl_cells_batch,
            "busy": rd_busy, "timeline": rd_timeline, "summary": rd_summary}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in COMMANDS:
        sys.exit(__doc__)
    COMMANDS[sys.argv[1]](sys.argv[2:])
