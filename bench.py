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
  roofline     : dominant kernel (k_bucket_accumulate) -- algorithmic bytes per launch
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
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="blobs per GPU per step (config 2: 256)")
    ap.add_argument("--window", type=int, default=0, help="MSM window bits override (11..13)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    kzg = load_product()
    ctx = kzg.Context(device=local_rank)
    if args.window:
        assert ctx.set_option("msm_window", args.window) == 0
    assert ctx.set_option("max_batch", args.batch) == 0
    B = args.batch

    # synthetic inputs of the reference benches, distinct seeds per rank, resident in HBM
    host = b"".join(rand_blob(rank * B + s) for s in range(B))
    d_blobs = torch.frombuffer(bytearray(host), dtype=torch.uint8).cuda()
    d_out = torch.zeros(B * 48, dtype=torch.uint8, device="cuda")
    d_status = torch.zeros(B, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream()

    def step():
        rc = ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), B, d_out.data_ptr(), d_status.data_ptr(), stream.cuda_stream)
        if rc:
            raise RuntimeError(f"mkzg_blob_to_kzg_commitment_batch_device rc={rc}")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.profile_enable(True)
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
    assert int(d_status.abs().sum().item()) == 0, "some blobs were rejected"

    if rank == 0:
        total_blobs = B * world * args.steps
        value = total_blobs / elapsed
        # roofline of the dominant kernel
        dom, (dom_ms, dom_n) = max(prof.items(), key=lambda kv: kv[1][0])
        avg_ms = dom_ms / max(1, dom_n)
        achieved = ALGO_BYTES_PER_BLOB * B / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                    "avg_kernel_ms": round(avg_ms, 4), "launches": dom_n,
                    "kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in prof.items()}}
        out = {
            "metric": "blobs/sec BlobToKZGCommitment (batch=256 per GPU)", "value": round(value, 2), "unit": "blobs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs (381-bit Montgomery, radix 2^29)",
            "data": "synthetic (GetRandBlob seeds, reference bench_test.go:17-46)",
            "config": {"workload": "configs[1]: BlobToKZGCommitment batch=256 blobs x 4096 scalars per GPU, fixed-base G1 MSM",
                       "batch_per_gpu": B, "parallelism": f"blob-sharded x{world}, no collective"},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


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
