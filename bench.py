#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native KZG engine.

Metric (BASELINE.json): blobs/sec of BlobToKZGCommitment on batches of 256 synthetic
4096-scalar blobs per GPU (config 2), inputs resident in HBM, outputs written to HBM.
One "step" = one pass of the hot path (mkzg_blob_to_kzg_commitment_batch_device) over one batch.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: blobs are independent, so each rank commits its own batch (weak scaling, no data-path
collective); ranks only meet in the barrier/max-reduction that brackets the timed region.

The JSON line also carries
  roofline     : dominant kernel (k_commit_direct_acc; k_bucket_accumulate with --direct-bits 0) -- algorithmic bytes per launch
                 (131,120 B per blob) / its average duration measured with
                 hipEvents on the launch stream inside the timed region, against 8 TB/s HBM;
  valu         : the honest ceiling of this integer-VALU-bound path: field multiplications per
                 second against the measured v_mad_u64_u32 issue rate;
  cpu_baseline : the plain-C oracle (oracle/, "port") timed on the host cores on a bounded sample.
torch is used for device memory, streams and torch.distributed only.
"""
import argparse
import ctypes
import hashlib
import importlib.util
import json
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
BLOB = 131072
ALGO_BYTES_PER_BLOB = 131072 + 48
HBM_PEAK_GBS = 8000.0                      # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def load_product():
    spec = importlib.util.spec_from_file_location("goethkzg_amd", os.path.join(ROOT, "go-eth-kzg_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["goethkzg_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def rand_blob(seed):
    """GetRandBlob of the reference's benches (bench_test.go:17-46)."""
    out = bytearray()
    for i in range(0, BLOB, 32):
        v = int.from_bytes(hashlib.sha256(struct.pack(">q", seed + i)).digest(), "big") % R_MOD
        out += v.to_bytes(32, "big")
    return bytes(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="blobs per GPU per step (config 2: 256)")
    ap.add_argument("--window", type=int, default=0, help="MSM window bits override (11..13)")
    ap.add_argument("--direct-bits", type=int, default=15,
                    help="commit_direct_bits option: 0 = bucket method (0.5 GB table), 12..15 = direct window table (15: 146 GB of the 288 GB HBM)")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams the steps alternate over (1 = strictly serial steps)")
    ap.add_argument("--no-chain", action="store_true", help="chain_accumulate=0: consecutive calls' accumulate kernels are not serialised")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with hipEvents (roofline section is then empty)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify-batch", type=int, default=1024, help="config 3: proofs per VerifyBlobKZGProofBatch call (0 = skip)")
    ap.add_argument("--verify-steps", type=int, default=3)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test aids for a 1-GPU box (never set by the driver): run the N > 1 code path with all ranks on GPU 0 over gloo
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_SHARE_GPU"):
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    kzg = load_product()
    ctx = kzg.Context(device=local_rank)
    if args.window:
        assert ctx.set_option("msm_window", args.window) == 0
    assert ctx.set_option("max_batch", args.batch) == 0
    if args.no_chain:
        assert ctx.set_option("chain_accumulate", 0) == 0
    B = args.batch

    # synthetic inputs of the reference benches, distinct seeds per rank, resident in HBM
    host = b"".join(rand_blob(rank * B + s) for s in range(B))
    d_blobs = torch.frombuffer(bytearray(host), dtype=torch.uint8).cuda()
    hog = None
    if os.environ.get("BENCH_HOG_GB"):       # test aid: occupy HBM first, to exercise the table-size fallback below
        hog = torch.empty(int(float(os.environ["BENCH_HOG_GB"]) * (1 << 30)), dtype=torch.uint8, device="cuda")
    d_ref = None
    if args.direct_bits:
        # untimed cross-check material: the same batch through the bucket method (the two methods share no kernel)
        d_ref = torch.zeros(B * 48, dtype=torch.uint8, device="cuda")
        d_ref_st = torch.zeros(B, dtype=torch.int32, device="cuda")
        assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), B, d_ref.data_ptr(), d_ref_st.data_ptr(), 0) == 0
        torch.cuda.synchronize()
        # the direct table is sized for an otherwise empty 288 GB part; if this box cannot give 146 GB, step down
        for bits in [b for b in (args.direct_bits, 14, 13, 12) if b <= args.direct_bits] + [0]:
            assert ctx.set_option("commit_direct_bits", bits) == 0
            if bits == 0:
                break
            probe = torch.zeros(48, dtype=torch.uint8, device="cuda")
            probe_st = torch.zeros(1, dtype=torch.int32, device="cuda")
            rc = ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), 1, probe.data_ptr(), probe_st.data_ptr(), 0)      # builds the table
            torch.cuda.synchronize()
            if rc == 0:
                break
            print(f"bench.py: direct table with {bits} bits unavailable (rc={rc}), stepping down", file=sys.stderr)
        args.direct_bits = bits
    # consecutive steps alternate between two HIP streams (each with its own output buffers): the library keeps
    # two workspaces, so the latency-bound bucket reduction of step i overlaps the accumulation of step i+1
    NS = max(1, args.streams)
    # distinct priorities -> distinct hardware queues (two default-priority torch streams can share one queue)
    streams = [torch.cuda.Stream(priority=(-1 if i % 2 == 0 else 0)) for i in range(NS)]
    d_outs = [torch.zeros(B * 48, dtype=torch.uint8, device="cuda") for _ in range(NS)]
    d_stats = [torch.zeros(B, dtype=torch.int32, device="cuda") for _ in range(NS)]
    d_out, d_status = d_outs[0], d_stats[0]
    torch.cuda.synchronize()
    counter = [0]

    def step():
        i = counter[0] % NS
        counter[0] += 1
        rc = ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), B, d_outs[i].data_ptr(), d_stats[i].data_ptr(), streams[i].cuda_stream)
        if rc:
            raise RuntimeError(f"mkzg_blob_to_kzg_commitment_batch_device rc={rc}")

    # priming (always, untimed, not counted as warmup): one call per stream with the profiler on, so that lazily built workspaces,
    # events and first-launch costs never land in the timed region even with --warmup 0
    ctx.profile_enable(not args.no_profile)
    for _ in range(NS):
        step()
    torch.cuda.synchronize()
    counter[0] = 0
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.profile_read(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    assert all(int(t.abs().sum().item()) == 0 for t in d_stats), "some blobs were rejected"
    if NS > 1 and args.steps + args.warmup >= 2:
        assert bool((d_outs[0] == d_outs[1]).all().item()), "streams disagree"
    if d_ref is not None:
        assert bool((d_outs[0] == d_ref).all().item()), "direct-table and bucket method disagree"

    # untimed: the kernels' isolated durations (one stream, nothing overlapping) and this GPU's VALU calibration
    isolated, calib = None, None
    if rank == 0 and not args.no_profile:
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        for _ in range(3):
            rc = ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), B, d_outs[0].data_ptr(), d_stats[0].data_ptr(), streams[0].cuda_stream)
            assert rc == 0
            torch.cuda.synchronize()
        isolated = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        calib = ctx.valu_calibrate()

    verify = None
    if args.verify_batch > 0:
        verify = bench_verify(ctx, torch, dist, world, rank, args, d_blobs, B)

    if rank == 0:
        total_blobs = B * world * args.steps
        value = total_blobs / elapsed
        # roofline of the dominant kernel
        if args.no_profile:
            prof = {"(profiling off)": (elapsed * 1e3, args.steps)}
        dom, (dom_ms, dom_n) = max(prof.items(), key=lambda kv: kv[1][0])
        avg_ms = dom_ms / max(1, dom_n)
        achieved = ALGO_BYTES_PER_BLOB * B / (avg_ms * 1e-3) / 1e9
        pmc = pmc_profile(dom)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": pmc.get("traffic_bytes"),
                    "avg_kernel_ms": round(avg_ms, 4), "launches": dom_n,
                    "kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in prof.items() if v[1]}}
        if NS > 1:
            roofline["note"] = f"{NS} streams: a launch's duration includes the time it shares the GPU with the other stream's kernels"
        # what actually bounds this path: integer VALU issue.  Wave-instructions of the dominant kernel (rocprofv3 SQ_INSTS_VALU,
        # profiles/*_traffic.json) over its isolated duration, against this GPU's calibrated v_mad_u64_u32 issue rate.
        valu = None
        if isolated is not None and calib and pmc.get("valu_wave_insts"):
            iso_ms = isolated.get(dom, (0, 0))
            if iso_ms[1]:
                # the committed PMC pass is for batch 256 and (direct mode) c = 15: instructions scale with blobs and windows
                scale = B / 256.0
                if dom == "k_commit_direct_acc" and args.direct_bits not in (0, 15):
                    scale *= -(-255 // args.direct_bits) / 17.0
                insts = pmc["valu_wave_insts"] * scale
                rate = insts / (iso_ms[0] / iso_ms[1] * 1e-3)
                valu = {"kernel": dom, "isolated_kernel_ms": round(iso_ms[0] / iso_ms[1], 4), "wave_insts_per_launch": insts,
                        "achieved": round(rate / 1e9, 2), "peak_calibrated": round(calib / 1e9, 2), "unit": "G wave-instr/s",
                        "frac": round(rate / calib, 4)}
        out = {
            "metric": "blobs/sec BlobToKZGCommitment (batch=256 per GPU)", "value": round(value, 2), "unit": "blobs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "dtype_note": "381-bit Montgomery field elements as 14 x 29-bit limbs in u32, 64-bit column accumulators (v_mad_u64_u32)",
            "data": "synthetic (GetRandBlob seeds, reference bench_test.go:17-46)",
            "config": {"workload": "configs[1]: BlobToKZGCommitment batch=256 blobs x 4096 scalars per GPU, fixed-base G1 MSM",
                       "batch_per_gpu": B, "msm": (f"direct window table c={args.direct_bits}" if args.direct_bits else f"bucket method c={args.window or 13}"), "parallelism": f"blob-sharded x{world}, no collective", "streams": NS},
            "roofline": roofline, "valu": valu,
        }
        if verify is not None:
            out["verify_blob_kzg_proof_batch"] = verify
        if not args.no_cpu_baseline and world == 1:        # reported baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


def pmc_profile(kernel):
    """Per-launch PMC figures of `kernel` from the committed rocprofv3 passes (profiles/*_traffic.json: HBM-side bytes,
    VALU wave-instructions; the counters cannot be read from inside the process).  {} if no profile covers the kernel."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            with open(f) as fh:
                k = json.load(fh)["kernels"].get(kernel)
            if k:
                return k
        except Exception:
            pass
    return {}


def bench_verify(ctx, torch, dist, world, rank, args, d_blobs256, B):
    """config 3: VerifyBlobKZGProofBatch on N synthetic (blob, commitment, proof) triples per GPU.
    Commitments and proofs are produced on the device by the product itself; one corrupted run
    must be rejected."""
    N = args.verify_batch
    reps = (N + B - 1) // B
    # N distinct blobs: seeds beyond the commitment batch
    extra = [rand_blob(100000 + rank * N + s) for s in range(N - B)] if N > B else []
    if N > B:
        d_blobs = torch.cat([d_blobs256, torch.frombuffer(bytearray(b"".join(extra)), dtype=torch.uint8).cuda()])
    else:
        d_blobs = d_blobs256[: N * BLOB]
    d_cm = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(N, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), N, d_cm.data_ptr(), d_st.data_ptr(), st) == 0
    assert ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), N, d_pr.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    #  ComputeBlobKZGProof over the same N blobs, timed once more now that every table exists
    tp0 = time.perf_counter()
    assert ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), N, d_pr.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    proofs_per_s = N / (time.perf_counter() - tp0)
    rc = ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, st)     # warm-up + correctness
    assert rc == 0, f"valid batch rejected rc={rc}"
    d_bad = d_pr.clone()
    d_bad[0:48] = d_pr[48:96]
    assert ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_bad.data_ptr(), N, st) == 1, "corrupted batch accepted"
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_us = 0.0
    if world > 1:
        # one global batch of N * world proofs, sharded: broadcast r, fold locally, all-gather 512-byte records
        # over RCCL, finish with one pairing (go-eth-kzg_amd/distributed.py)
        spec = importlib.util.spec_from_file_location("kzg_distributed", os.path.join(ROOT, "go-eth-kzg_amd", "distributed.py"))
        D = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(D)
        eng = D.ProductEngine(ctx)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        for _ in range(args.verify_steps):
            rc = D.verify_blob_kzg_proof_batch_sharded(eng, (d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, st), N * world, dist, dev)
            assert rc == 0
            host_us += ctx.last_host_finish_us()
    else:
        for _ in range(args.verify_steps):
            assert ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, st) == 0
            host_us += ctx.last_host_finish_us()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    ms = dt / args.verify_steps * 1e3
    kern = {k: round(v[0] / args.verify_steps, 4) for k, v in prof.items() if v[1]}
    dom = max(kern.items(), key=lambda kv: kv[1])
    algo = (131072 + 48 + 48) * N          # bytes read per blob
    return {"metric": "blobs/sec VerifyBlobKZGProofBatch", "value": round(N * world / dt * args.verify_steps, 2), "unit": "blobs/s", "batch_per_gpu": N,
            "parallelism": "single call" if world == 1 else f"one global batch of {N * world} sharded x{world}; RCCL broadcast(r) + all_gather(512 B records)",
            "compute_blob_kzg_proof_batch_proofs_per_s": round(proofs_per_s, 1),
            "ms_per_call": round(ms, 3), "host_finish_ms": round(host_us / args.verify_steps / 1e3, 3), "kernels_ms_per_call": kern,
            "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": round(algo / (dom[1] * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(algo / (dom[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                         "traffic": (pmc_profile(dom[0]).get("traffic_bytes") if N == 1024 else None)}}


def cpu_baseline():
    """oracle/ (plain-C port of the same path, OpenMP over blobs) on the host cores, bounded sample."""
    sys.path.insert(0, ROOT)
    import oracle
    cores = os.cpu_count() or 1
    octx = oracle.OracleContext(with_fk20=False)
    n = max(8, min(512, 2 * cores))
    blobs = b"".join(oracle.rand_blob(s) for s in range(n))
    octx.blob_to_kzg_commitment_batch(blobs[:BLOB * 2], 2)   # warm
    passes, t0 = 0, time.perf_counter()
    while True:                                               # about 10 s of CPU work
        octx.blob_to_kzg_commitment_batch(blobs, n)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= 10.0 or passes >= 200:
            break
    octx.close()
    return {"value": round(n * passes / dt, 3), "unit": "blobs/s", "cores": cores, "kind": "port",
            "sample": f"{passes} passes over {n} GetRandBlob blobs, OpenMP over blobs ({dt:.2f} s)"}


if __name__ == "__main__":
    main()
