#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native KZG engine.

Metric (BASELINE.json): blobs/sec of BlobToKZGCommitment on batches of 256 synthetic
4096-scalar blobs per GPU (configs[1]), inputs resident in HBM, outputs written to HBM.
One "step" = one pass of the hot path (mkzg_blob_to_kzg_commitment_batch_device) over one batch.

    python bench.py --gpus N --steps K --warmup W          # starts its own N ranks when WORLD_SIZE is not set
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: blobs are independent, so each rank commits its own batch (weak scaling, no data-path
collective); ranks only meet in the barrier/max-reduction that brackets the timed region.  The two batch
verifiers are the part of the path with an exchange step: one global batch sharded over the ranks, broadcast
of r, all_gather of one 512-byte record per rank (go-eth-kzg_amd/distributed.py).

The JSON line also carries
  repeats      : the timed region run five times in all (the first one is `value`): min / median / max of ms per step;
  multi_gpu    : (N > 1) ranks, the PCI bus ids of the GPUs they run on, the cost of the verifiers' exchange step alone, and a
                 STRONG-scaling figure for both batch verifiers (one fixed global batch of 8192 blobs / 65,536 cells cut over the ranks);
  roofline     : dominant kernel (k_commit_direct_acc; k_bucket_accumulate with --direct-bits 0) -- algorithmic bytes per launch
                 (131,120 B per blob) / its average duration measured with
                 hipEvents on the launch stream inside the timed region, against 8 TB/s HBM;
  valu         : the ceiling that actually binds this integer path: VALU wave-instructions per second of the dominant kernel
                 against the hardware issue limit (SIMDs x clock / 4) and against this GPU's calibrated v_mad_u64_u32 rate;
  drop_in      : what a caller of the reference API gets -- host pointers in, host bytes out (PCIe inclusive; never `value`): commitments,
                 and VerifyBlobKZGProofBatch of 1024 separate host allocations from one caller and from K callers with DISTINCT blobs
                 each, with a `pcie` object (GB/s over the link against 63 GB/s: the roofline that binds the host-pointer verifier);
  verify_blob_kzg_proof_batch : configs[2], single-call latency and sustained rate with several calls in flight;
  eip7594      : configs[3] and configs[4] (cells + proofs, cell batch verification, recovery);
                 every leg carries its own `roofline` (dominant kernel of the leg, algorithmic bytes of SURVEY.md 8d) and `valu`
                 (wave-instructions per second of each kernel against the hardware issue limit) objects;
  cpu_baseline : the plain-C oracle (oracle/, "port") timed on the host cores this process may use, bounded sample.
torch is used for device memory, streams and torch.distributed only.
"""
import argparse
import ctypes
import hashlib
import importlib.util
import json
import os
import socket
import struct
import subprocess
import sys
import threading
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


def load_distributed():
    spec = importlib.util.spec_from_file_location("kzg_distributed", os.path.join(ROOT, "go-eth-kzg_amd", "distributed.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    return D


def rand_blob(seed):
    """GetRandBlob of the reference's benches (bench_test.go:17-46)."""
    out = bytearray()
    for i in range(0, BLOB, 32):
        v = int.from_bytes(hashlib.sha256(struct.pack(">q", seed + i)).digest(), "big") % R_MOD
        out += v.to_bytes(32, "big")
    return bytes(out)


def usable_cpus():
    """CPUs this process may actually burn: the affinity mask capped by the cgroup's CPU quota (a box can show 256 CPUs and grant 16)"""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, p = fh.read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except Exception:
        pass
    return n


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="blobs per GPU per step (configs[1]: 256)")
    ap.add_argument("--window", type=int, default=0, help="MSM window bits override (11..13)")
    ap.add_argument("--direct-bits", type=int, default=-1,
                    help="commit_direct_bits option: -1 = the library's default (widest direct window table that leaves 40 GB of free HBM: 15 bits = 146 GB on an empty MI355X), "
                         "0 = bucket method (10.5 MB table), 12..15 = that width")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the first is `value`, all are reported (min / median / max)")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling legs of the two batch verifiers (8192 blobs / 65,536 cells in all)")
    ap.add_argument("--streams", type=int, default=4, help="HIP streams the steps alternate over (1 = strictly serial steps; 2 until the second session of round 4: 4 streams measure 3-4 % more on the same box, 6 and 8 no more than 4)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with hipEvents (roofline section is then empty)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-drop-in", action="store_true")
    ap.add_argument("--drop-in-callers", type=int, nargs="*", default=[4, 8], help="caller counts of the host-pointer VerifyBlobKZGProofBatch leg (each caller holds its own 1024 distinct blobs in host memory)")
    ap.add_argument("--verify-batch", type=int, default=1024, help="configs[2]: proofs per VerifyBlobKZGProofBatch call per GPU (0 = skip)")
    ap.add_argument("--verify-steps", type=int, default=10, help="single VerifyBlobKZGProofBatch calls timed one by one (min / median / max reported)")
    ap.add_argument("--verify-shape", type=int, default=-1, help="verify_shape option for the single calls: -1 library default (latency shape for a lone call), 1 forces the throughput shape (PMC passes of the sustained leg's kernels)")
    ap.add_argument("--ring", type=int, default=8, help="distinct input batches the timed steps cycle through (8 x 33.5 MB > the 256 MB Infinity Cache: the algorithmic bytes stream from HBM); the sustained verifier leg holds as many distinct 1024-blob sets")
    ap.add_argument("--no-big-cells", action="store_true", help="skip the ComputeCellsAndKZGProofs calls of 128 / 192 / 256 blobs (PMC passes: a kernel's largest launches must be the 64-blob leg's)")
    ap.add_argument("--verify-inflight", type=int, default=32, help="caller threads of the sustained-rate leg (one call in flight each; a context has 32 lanes)")
    ap.add_argument("--fk20-bits", type=int, default=12, help="window width of the FK20 fixed-base table: 12 (47 GB, the library default), 11 (25.8 GB), 8 (4.3 GB) or 13 (86 GB: fewer additions but TLB-bound gathers, measured no faster); narrower ones are the fallback when HBM is short")
    ap.add_argument("--cell-blobs", type=int, default=64, help="configs[3]/[4]: blobs per GPU for the EIP-7594 legs (64 x 128 = 8192 cells; 0 = skip)")
    ap.add_argument("--full-line", action="store_true", help="print the whole record (every leg with its roofline / valu objects, ~22 KB) as the stdout line instead of the compact one; "
                                                              "the whole record always goes to --detail-out and, prefixed BENCH_DETAIL, to stderr")
    ap.add_argument("--detail-out", default=None, help="file the whole record is written to (default: gpurun_out/bench_detail_n<N>.json under the repo root)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher: start the N ranks as children (this parent never touches a GPU), relay rank 0's line"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    sys.exit(next((rc for rc in rcs if rc), 0))


def main():
    args = parse_args()
    if os.environ.get("BENCH_WATCHDOG"):               # diagnosis aid: every thread's Python stack after that many seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_WATCHDOG"]), exit=True)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test aids for a 1-GPU box (never set by the driver): run the N > 1 code path with all ranks on GPU 0 over gloo (BENCH_SHARE_GPU,
    # BENCH_DIST_BACKEND); run the exchange steps on real RCCL with ONE rank (BENCH_FORCE_DIST: the process group is created although
    # the world is 1, and the batch verifiers take the sharded route -- broadcast and all_gather on device tensors over "nccl")
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    shared_gpu = bool(os.environ.get("BENCH_SHARE_GPU"))
    if shared_gpu:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU fallback")
    torch.cuda.set_device(local_rank)
    numa = bind_to_gpu_node(torch, local_rank)          # before any buffer is allocated and before the library starts its host pool
    dist_on = world > 1 or bool(os.environ.get("BENCH_FORCE_DIST"))
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("LOCAL_WORLD_SIZE", str(world))        # the library sizes its host pool by the ranks that share the node
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), world_size=world, rank=rank)
        else:
            dist.init_process_group(backend=backend, world_size=world, rank=rank)
    dist_dev = (torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")) if dist_on else None

    kzg = load_product()
    ctx = kzg.Context(device=local_rank)
    if args.window:
        assert ctx.set_option("msm_window", args.window) == 0
    assert ctx.set_option("max_batch", args.batch) == 0
    if shared_gpu and world > 2:
        # test aid only: eight ranks' tables do not fit ONE GPU's HBM -- bucket method and the smallest FK20 table for every rank
        args.direct_bits, args.fk20_bits = 0, 8
    B = args.batch

    # synthetic inputs of the reference benches, distinct seeds per rank, resident in HBM
    host_blobs = [rand_blob(rank * B + s) for s in range(B)]
    d_blobs = torch.frombuffer(bytearray(b"".join(host_blobs)), dtype=torch.uint8).cuda()
    hog = None
    if os.environ.get("BENCH_HOG_GB"):       # test aid: occupy HBM first, to exercise the library's table-size fallback
        hog = torch.empty(int(float(os.environ["BENCH_HOG_GB"]) * (1 << 30)), dtype=torch.uint8, device="cuda")
    # a ring of distinct input batches for the timed steps: ring[0] is the GetRandBlob batch every cross-check uses, the others are
    # uniform canonical scalars (fast_blobs) -- together more than the Infinity Cache holds, so that no step finds its blobs cached
    ring = [d_blobs] + [torch.from_numpy(fast_blobs(B, 900 + rank * 64 + i).reshape(-1)).cuda() for i in range(max(1, args.ring) - 1)]
    drop_in = {}
    d_ref = None
    if args.direct_bits:
        # untimed cross-check material: the same batch through the bucket method (the two methods share no kernel) -- and what a
        # host-pointer caller gets from it
        assert ctx.set_option("commit_direct_bits", 0) == 0
        d_ref = torch.zeros(B * 48, dtype=torch.uint8, device="cuda")
        d_ref_st = torch.zeros(B, dtype=torch.int32, device="cuda")
        assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), B, d_ref.data_ptr(), d_ref_st.data_ptr(), 0) == 0
        torch.cuda.synchronize()
        if rank == 0 and not args.no_drop_in:
            drop_in["bucket_method"] = bench_drop_in(ctx, kzg, host_blobs, d_ref)
    assert ctx.set_option("commit_direct_bits", args.direct_bits) == 0
    requested_bits = args.direct_bits
    if args.direct_bits:
        probe = torch.zeros(48, dtype=torch.uint8, device="cuda")
        probe_st = torch.zeros(1, dtype=torch.int32, device="cuda")
        rc = ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), 1, probe.data_ptr(), probe_st.data_ptr(), 0)      # builds the table (or degrades)
        torch.cuda.synchronize()
        assert rc == 0, rc
        eff = ctx.get_option("commit_direct_bits_effective")
        if args.direct_bits > 0 and eff != args.direct_bits:
            print(f"bench.py: direct table with {args.direct_bits} bits unavailable on this box, the library degraded to {eff or 'the bucket method'}", file=sys.stderr)
        args.direct_bits = eff
    # consecutive steps alternate between HIP streams (each with its own output buffers): every call works in its own lane of the
    # context, so the latency-bound finish of step i overlaps the accumulation of the steps behind it
    NS = max(1, args.streams)
    # distinct priorities -> distinct hardware queues (two default-priority torch streams can share one queue)
    streams = [torch.cuda.Stream(priority=(-1 if i % 2 == 0 else 0)) for i in range(NS)]
    d_outs = [torch.zeros(B * 48, dtype=torch.uint8, device="cuda") for _ in range(NS)]
    d_stats = [torch.zeros(B, dtype=torch.int32, device="cuda") for _ in range(NS)]
    torch.cuda.synchronize()
    counter = [0]
    ns_active = [NS]

    def step(batch=None):
        i = counter[0] % ns_active[0]
        src = ring[counter[0] % len(ring)] if batch is None else batch
        counter[0] += 1
        rc = ctx.blob_to_kzg_commitment_batch_device(src.data_ptr(), B, d_outs[i].data_ptr(), d_stats[i].data_ptr(), streams[i].cuda_stream)
        if rc:
            raise RuntimeError(f"mkzg_blob_to_kzg_commitment_batch_device rc={rc}")

    def timed_region():
        """EXACTLY --steps steps between barrier + synchronize on both sides; the MAX over ranks"""
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            if backend != "nccl":
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # priming (always, untimed, not counted as warmup): calls with the profiler on, so that lazily built workspaces of every lane,
    # events and first-launch costs never land in the timed region even with --warmup 0
    ctx.profile_enable(not args.no_profile)
    for _ in range(max(NS, ctx.get_option("lanes") or 4)):
        step()
    torch.cuda.synchronize()
    counter[0] = 0
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.profile_read(reset=True)
    elapsed = timed_region()                                  # THE timed region: `value` comes from this one
    prof = ctx.profile_read(reset=True)
    region_s = [elapsed] + [timed_region() for _ in range(max(0, args.repeats - 1))]
    prof_repeats = ctx.profile_read(reset=True)               # the same kernels over the repeated regions (what a profiler's average mixes in)
    ctx.profile_enable(False)
    # the same region over fewer streams, for the record (not `value`): how much of the figure is overlap between consecutive steps
    streams_sweep = {}
    for m in sorted({1, 2, NS}):
        if m < NS:
            ns_active[0] = m
            timed_region()
            streams_sweep[str(m)] = round(B * world * args.steps / timed_region(), 1)
    ns_active[0] = NS
    assert all(int(t.abs().sum().item()) == 0 for t in d_stats), "some blobs were rejected"
    # untimed check pass: the cross-check batch through every stream of the timed loop
    for _ in range(NS):
        step(d_blobs)
    torch.cuda.synchronize()
    assert all(int(t.abs().sum().item()) == 0 for t in d_stats), "some blobs were rejected"
    if NS > 1:
        assert bool((d_outs[0] == d_outs[1]).all().item()), "streams disagree"
    if d_ref is not None:
        assert bool((d_outs[0] == d_ref).all().item()), "direct-table and bucket method disagree"

    # untimed: the kernels' isolated durations (one stream, nothing overlapping) and this GPU's VALU calibration
    isolated, calib = None, None
    if rank == 0 and not args.no_profile:
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        for q in (1, 2, 0):                        # the last one leaves the cross-check batch's commitments in d_outs[0]
            rc = ctx.blob_to_kzg_commitment_batch_device(ring[q % len(ring)].data_ptr(), B, d_outs[0].data_ptr(), d_stats[0].data_ptr(), streams[0].cuda_stream)
            assert rc == 0
            torch.cuda.synchronize()
        isolated = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        calib = ctx.valu_calibrate()
    if rank == 0 and not args.no_drop_in:
        key = ("default_options" if requested_bits < 0 else "direct_table") + (f"_c{args.direct_bits}" if args.direct_bits else "_bucket_method")
        if args.direct_bits or not drop_in:
            drop_in[key] = bench_drop_in(ctx, kzg, host_blobs, d_outs[0])

    props = torch.cuda.get_device_properties(local_rank)
    hw = hw_limits(props)
    multi = multi_gpu_info(torch, dist, dist_on, world, rank, local_rank, backend, dist_dev, props, ctx) if dist_on else None
    verify = None
    if args.verify_batch > 0:
        verify = bench_verify(ctx, torch, dist, dist_on, dist_dev, world, rank, args, d_blobs, B, hw)
    if rank == 0 and world == 1 and not dist_on and not args.no_drop_in and args.verify_batch > 0:
        # the other half of the metric as the reference API sees it: host pointers in, PCIe-bound
        drop_in["verify_blob_kzg_proof_batch"] = bench_drop_in_verify(ctx, kzg, torch, N=args.verify_batch, callers=tuple(args.drop_in_callers))
    eip7594 = None
    if args.cell_blobs > 0:
        eip7594 = bench_eip7594(ctx, kzg, torch, dist, dist_on, dist_dev, world, rank, args, hw)

    single = None
    if rank == 0 and world == 1 and not dist_on and not args.no_drop_in and args.verify_batch > 0 and args.cell_blobs > 0:
        drop_in["single_calls_ms"] = bench_single_calls(ctx, host_blobs)
        single = bench_single_blob_callers(ctx, kzg, torch)
        assert ctx.set_option("combine_single_calls", 0) == 0            # for the record: every call on its own, as until round 5
        alone = bench_single_blob_callers(ctx, kzg, torch, callers=(32,))
        assert ctx.set_option("combine_single_calls", 1) == 0
        for kind in ("blob_to_kzg_commitment", "compute_cells_and_kzg_proofs", "verify_blob_kzg_proof"):
            single[kind]["callers_32_uncombined"] = alone[kind]["callers_32"]
        # ratios to the batch entry points of the same run: host-pointer commitments (one caller), cells with proofs (64 device-resident
        # blobs per call, eight callers if measured), host-pointer VerifyBlobKZGProofBatch (most callers measured)
        ref = {}
        dk = next((k for k in drop_in if k.startswith(("default_options", "direct_table"))), None)
        if dk:
            ref["blob_to_kzg_commitment"] = ("drop_in.%s.blobs_per_s_two_callers" % dk, drop_in[dk]["blobs_per_s_two_callers"])
        if eip7594:
            cc = eip7594.get("compute_cells_and_kzg_proofs", {})
            ref["compute_cells_and_kzg_proofs"] = (("eip7594.compute_cells_and_kzg_proofs.sustained.blobs_per_s", cc["sustained"]["blobs_per_s"]) if "sustained" in cc
                                                   else ("eip7594.compute_cells_and_kzg_proofs.blobs_per_s", cc.get("blobs_per_s")))
        hv = drop_in.get("verify_blob_kzg_proof_batch", {})
        ks = sorted(int(k.split("_")[1]) for k in hv if k.startswith("callers_"))
        if ks:
            ref["verify_blob_kzg_proof"] = ("drop_in.verify_blob_kzg_proof_batch.callers_%d.blobs_per_s" % ks[-1], hv[f"callers_{ks[-1]}"]["blobs_per_s"])
        for kind, (what, v) in ref.items():
            if v:
                best = max(single[kind].values(), key=lambda d: d["blobs_per_s"])["blobs_per_s"]
                single[kind]["batch_entry_point"] = {"what": what, "blobs_per_s": v}
                single[kind]["ratio_to_batch"] = round(best / v, 3)
        drop_in["single_blob_callers"] = single

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
        # `achieved` / `frac` / `avg_kernel_ms`: ONE launch alone on the GPU (three untimed single-stream steps of this run) -- the figure
        # that is comparable from round to round and across --streams settings; the same launch while it shares the chip with the other
        # streams' launches of the timed region is `*_overlapped` (it was the headline until round 4 and moved with the stream count:
        # 0.0013 over two streams, 0.00077 over four, same kernel)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": pmc.get("traffic_bytes"),
                    "avg_kernel_ms": round(avg_ms, 4), "launches": dom_n,
                    "kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in prof.items() if v[1]}}
        roofline["achieved_overlapped"], roofline["frac_overlapped"], roofline["avg_kernel_ms_overlapped"] = roofline["achieved"], roofline["frac"], roofline["avg_kernel_ms"]
        roofline["measured"] = "timed region (no isolated launches in this run)"
        if not args.no_profile and prof_repeats.get(dom, (0, 0))[1]:
            rep_ms, rep_n = prof_repeats[dom]
            roofline["avg_kernel_ms_all_regions"] = round((dom_ms + rep_ms) / (dom_n + rep_n), 4)
            roofline["launches_all_regions"] = dom_n + rep_n
        if pmc.get("stale"):
            roofline["traffic_note"] = pmc["stale"]
        if isolated is not None and isolated.get(dom, (0, 0))[1]:
            iso_ms_r = isolated[dom][0] / isolated[dom][1]
            roofline["avg_kernel_ms_one_stream"] = round(iso_ms_r, 4)
            roofline["achieved_one_stream"] = round(ALGO_BYTES_PER_BLOB * B / (iso_ms_r * 1e-3) / 1e9, 3)
            roofline["frac_one_stream"] = round(ALGO_BYTES_PER_BLOB * B / (iso_ms_r * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)
            roofline["achieved"], roofline["frac"], roofline["avg_kernel_ms"] = roofline["achieved_one_stream"], roofline["frac_one_stream"], roofline["avg_kernel_ms_one_stream"]
            roofline["measured"] = "one launch alone on the GPU, HIP events on its launch stream (three untimed single-stream steps); the timed region's overlapped launches: *_overlapped"
        roofline["input_ring"] = {"batches": len(ring), "bytes": len(ring) * B * BLOB, "note": "the timed steps cycle through this many distinct input batches resident in HBM"}
        step_ms = elapsed / args.steps * 1e3
        roofline["achieved_per_step"] = round(ALGO_BYTES_PER_BLOB * B / (step_ms * 1e-3) / 1e9, 3)
        roofline["frac_per_step"] = round(ALGO_BYTES_PER_BLOB * B / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)
        if NS > 1:
            roofline["note"] = (f"{NS} streams: *_overlapped divide a launch's algorithmic bytes by ITS duration while it shares the GPU with the launches of the other {NS - 1} "
                                "streams (that duration exceeds ms_per_step); achieved / frac / avg_kernel_ms: the same launch alone on the GPU; achieved_per_step / frac_per_step: "
                                "a launch's bytes over ms_per_step, what the overlapped pipeline streams per unit of wall time")
        # what actually bounds this path: integer VALU issue.  Wave-instructions of the dominant kernel (rocprofv3 SQ_INSTS_VALU,
        # profiles/*_traffic.json, valid only for the kernel sources they were collected on) over its duration, against the hardware
        # limit -- one VALU instruction per SIMD per 4 cycles at the device's maximum clock -- and against the rate this particular GPU
        # sustained on a pure v_mad_u64_u32 stream just now (a LOWER figure than the limit: it carries the clock the GPU really held)
        valu = None
        if isolated is not None and pmc.get("valu_wave_insts"):
            iso = isolated.get(dom, (0, 0))
            if iso[1]:
                # the committed PMC pass is for batch 256 and (direct mode) c = 15: instructions scale with blobs and windows
                scale = B / 256.0
                if dom == "k_commit_direct_acc" and args.direct_bits not in (0, 15):
                    scale *= -(-255 // args.direct_bits) / 17.0
                insts = pmc["valu_wave_insts"] * scale
                iso_ms = iso[0] / iso[1]
                peak_hw = hw["valu_peak"]
                rate_iso = insts / (iso_ms * 1e-3)
                rate_timed = insts / (elapsed / args.steps)
                valu = {"kernel": dom, "wave_insts_per_launch": insts, "unit": "G wave-instr/s",
                        "isolated_kernel_ms": round(iso_ms, 4), "achieved_isolated": round(rate_iso / 1e9, 2),
                        "achieved_timed_region": round(rate_timed / 1e9, 2),
                        "peak_hw": round(peak_hw / 1e9, 1), "peak_hw_note": hw["note"],
                        "frac_hw_isolated": round(rate_iso / peak_hw, 4), "frac_hw_timed_region": round(rate_timed / peak_hw, 4),
                        "calibrated_mix_rate": round(calib / 1e9, 2) if calib else None,
                        "calibration_note": "measured in this run: four wavefronts per SIMD issuing the multiplier's mix (eight v_mad_u64_u32, one v_lshl_add_u64, one v_and_b32 per ten instructions) on independent accumulators -- what this GPU's clock under load sustains for the kernels' instruction mix",
                        "pmc_source": pmc.get("source_file")}
        if valu is not None:
            # the ceiling that binds, inside the object the driver parses: wave-instructions per second over the hardware issue limit
            roofline["valu_frac"] = valu["frac_hw_timed_region"]
            roofline["valu_frac_one_stream"] = valu["frac_hw_isolated"]
            roofline["valu_note"] = "integer VALU issue is what bounds this kernel: SQ_INSTS_VALU per launch (committed PMC pass) x launches/s of the timed region (valu_frac) or over the isolated launch (valu_frac_one_stream), against 256 CUs x 4 SIMDs x f_max / 4"
        ms_steps = sorted(r / args.steps * 1e3 for r in region_s)
        out = {
            "metric": "blobs/sec BlobToKZGCommitment (batch=256 per GPU)", "value": round(value, 2), "unit": "blobs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "dtype_note": "381-bit Montgomery field elements as 14 x 29-bit limbs in u32, 64-bit column accumulators (v_mad_u64_u32)",
            "data": "synthetic (GetRandBlob seeds, reference bench_test.go:17-46)",
            "config": {"workload": "configs[1]: BlobToKZGCommitment batch=256 blobs x 4096 scalars per GPU, fixed-base G1 MSM",
                       "batch_per_gpu": B, "msm": (f"direct window table c={args.direct_bits}" + (" (library default: auto-sized)" if requested_bits < 0 else "") if args.direct_bits else f"bucket method c={args.window or 13}"),
                       "parallelism": f"blob-sharded x{world}, no collective", "streams": NS},
            "repeats": {"regions": len(region_s), "steps_each": args.steps, "ms_per_step_each": [round(r / args.steps * 1e3, 3) for r in region_s],
                        "ms_per_step_min": round(ms_steps[0], 3), "ms_per_step_median": round(ms_steps[len(ms_steps) // 2], 3), "ms_per_step_max": round(ms_steps[-1], 3),
                        "blobs_per_s_median": round(B * world / (ms_steps[len(ms_steps) // 2] * 1e-3), 1),
                        "note": "`value` is the first region (the contract's K steps); the others repeat it back to back"},
            "streams_sweep": {"blobs_per_s_by_streams": streams_sweep, "note": "one more timed region of the same K steps over fewer streams (after an untimed one), for the record: "
                              "what the overlap between consecutive steps is worth; `value` is the region over config.streams streams"},
            "roofline": roofline, "valu": valu, "numa": numa,
        }
        if multi is not None:
            out["multi_gpu"] = multi
        if drop_in:
            out["drop_in"] = drop_in
        if verify is not None:
            out["verify_blob_kzg_proof_batch"] = verify
            # the second half of BASELINE.json's metric at the top level too: whole-job blobs/s of VerifyBlobKZGProofBatch over all ranks
            shape = (f"{verify['sustained']['callers']} device-resident callers on one context" if "sustained" in verify else "single device-resident calls")
            out["verify_metric"] = {"metric": "blobs/sec VerifyBlobKZGProofBatch (batch=%d per GPU; %s)" % (args.verify_batch, shape), "value": verify["value"], "unit": "blobs/s", "n_gpus": world,
                                    "what": verify["value_is"]}
            hostv = drop_in.get("verify_blob_kzg_proof_batch", {}) if drop_in else {}
            ks = sorted(int(k.split("_")[1]) for k in hostv if k.startswith("callers_"))
            if ks:
                out["verify_metric"]["reference_shaped"] = {"metric": "blobs/sec VerifyBlobKZGProofBatch (batch=%d; host pointers as verify.go:88 hands them over, %d callers)" % (args.verify_batch, ks[-1]),
                                                            "value": hostv[f"callers_{ks[-1]}"].get("blobs_per_s"), "bound": "PCIe", "see": "drop_in.verify_blob_kzg_proof_batch"}
        if eip7594 is not None:
            out["eip7594"] = eip7594
        if not args.no_cpu_baseline and world == 1:        # reported baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline()
        line = json.dumps(out)
    else:
        line = None
    if dist_on:
        dist.barrier()
    ctx.close()
    # the OTHER multi-GPU form, one process over all the job's devices (api_multi.hip): after every rank has given its HBM back
    if dist_on:
        dist.barrier()
    if rank == 0 and not args.no_drop_in:
        n_dev = min(torch.cuda.device_count(), 1 if shared_gpu else world)
        if n_dev < 2:
            in_proc = bench_multi_in_process(kzg, torch, n_dev)
        else:
            # in a CHILD process with a time limit: a leg that has never met real multi-GPU hardware must not be able to cost the line
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--in-process-leg", str(n_dev)], capture_output=True, text=True, timeout=600)
                got = [ln for ln in r.stdout.splitlines() if ln.startswith("IN_PROCESS ")]
                in_proc = json.loads(got[-1][len("IN_PROCESS "):]) if got else {"error": (r.stderr or r.stdout)[-400:]}
            except Exception as e:      # noqa: BLE001
                in_proc = {"error": repr(e)[:300]}
        out.setdefault("multi_gpu", {})["in_process"] = in_proc
        line = json.dumps(out)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        # The whole record (every leg, ~22 KB) goes to a file and to stderr; stdout carries ONE line -- by default the COMPACT one (the
        # contract's keys, `roofline`, `cpu_baseline` and one figure per leg, < 4 KB): round 5's 21.6 KB line came back from the driver
        # with `parsed: null` where round 4's 18.2 KB line had parsed, so the line stays far below any capture limit.
        detail_path = args.detail_out or os.path.join(ROOT, "gpurun_out", f"bench_detail_n{world}.json")
        try:
            os.makedirs(os.path.dirname(detail_path) or ".", exist_ok=True)
            with open(detail_path, "w") as fh:
                fh.write(line + "\n")
        except OSError as e:
            detail_path = f"(not written: {e})"
        print("BENCH_DETAIL " + line, file=sys.stderr, flush=True)
        if not args.full_line:
            line = json.dumps(compact_line(out, os.path.relpath(detail_path, ROOT) if os.path.isabs(detail_path) else detail_path), separators=(",", ":"))
        # the JSON line is the LAST thing on stdout: RCCL and gloo write banners to the C-level stdout (RCCL's version block sits in the
        # stdio buffer until the process exits), so that buffer goes out first
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:      # noqa: BLE001
            pass
        print(line, flush=True)


def dig(d, path, default=None):
    """d["a"]["b"]["c"] for path "a.b.c", `default` where a key is missing"""
    for k in path.split("."):
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def compact_line(out, detail_path):
    """The stdout line: the contract's keys, the dominant kernel's `roofline` (with the VALU fractions: the ceiling that binds), `cpu_baseline`,
    and ONE figure per leg of the whole record (`legs`; names follow the record's paths).  Everything else is in the detail file."""
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in out}
    c["config"] = {k: v for k, v in out.get("config", {}).items()}
    rf = out.get("roofline") or {}
    c["roofline"] = {k: rf[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_kernel_ms", "launches", "frac_overlapped", "avg_kernel_ms_overlapped",
                                        "frac_per_step", "valu_frac", "valu_frac_one_stream") if k in rf}
    if "traffic" in c["roofline"] and c["roofline"]["traffic"]:
        c["roofline"]["traffic"] = int(c["roofline"]["traffic"])
    c["roofline"]["algorithmic_bytes_per_launch"] = ALGO_BYTES_PER_BLOB * dig(out, "config.batch_per_gpu", 256)
    c["roofline"]["note"] = ("achieved = algorithmic bytes per launch / the launch's duration alone on the GPU (HIP events on its stream); *_overlapped: the same launch inside "
                             "the timed region; valu_frac: SQ_INSTS_VALU per launch x launches/s over 256 CUs x 4 SIMDs x f/4 -- integer VALU issue is the ceiling that binds")
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "single_thread_blobs_per_s") if k in cb}
        c["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
    c["repeats_ms_per_step"] = dig(out, "repeats.ms_per_step_each")
    vm = out.get("verify_metric")
    if vm:
        c["verify_metric"] = {"metric": vm.get("metric"), "value": vm.get("value"), "unit": vm.get("unit"), "n_gpus": vm.get("n_gpus"),
                              "reference_shaped_host_pointers_blobs_per_s": dig(vm, "reference_shaped.value")}
    legs = {}
    for name, path in (
            ("commit.host_pointers_one_caller_blobs_per_s", None),
            ("verify_blob_batch_1024.single_call_ms", "verify_blob_kzg_proof_batch.single_call.ms_per_call_median"),
            ("verify_blob_batch_1024.sustained_blobs_per_s", "verify_blob_kzg_proof_batch.sustained.blobs_per_s"),
            ("verify_blob_batch_1024.sustained_callers", "verify_blob_kzg_proof_batch.sustained.callers"),
            ("verify_blob_batch_1024.sustained_valu_frac", "verify_blob_kzg_proof_batch.sustained.valu.frac_hw"),
            ("verify_blob_batch_1024.host_pointers_one_caller_ms", "drop_in.verify_blob_kzg_proof_batch.one_caller.ms_per_call_median"),
            ("compute_blob_kzg_proof_batch.proofs_per_s", "verify_blob_kzg_proof_batch.compute_blob_kzg_proof_batch_proofs_per_s"),
            ("cells_and_proofs_64.ms_per_call", "eip7594.compute_cells_and_kzg_proofs.ms_per_batch"),
            ("cells_and_proofs_256.blobs_per_s", "eip7594.compute_cells_and_kzg_proofs.blobs_per_s_256_per_call"),
            ("cells_and_proofs_64.sustained_blobs_per_s", "eip7594.compute_cells_and_kzg_proofs.sustained.blobs_per_s"),
            ("verify_cell_batch_8192.ms_per_call", "eip7594.verify_cell_kzg_proof_batch.ms_per_call_median"),
            ("verify_cell_batch_8192.device_cells_ms", "eip7594.verify_cell_kzg_proof_batch.device_cells.ms_per_call_median"),
            ("verify_cell_batch_8192.sustained_cells_per_s", "eip7594.verify_cell_kzg_proof_batch.sustained.cells_per_s"),
            ("recover_cells_and_proofs_64.ms_per_call", "eip7594.recover_cells_and_kzg_proofs.ms_per_batch"),
            ("single_call_ms.BlobToKZGCommitment", "drop_in.single_calls_ms.BlobToKZGCommitment"),
            ("single_call_ms.ComputeBlobKZGProof", "drop_in.single_calls_ms.ComputeBlobKZGProof"),
            ("single_call_ms.VerifyBlobKZGProof", "drop_in.single_calls_ms.VerifyBlobKZGProof"),
            ("single_call_ms.VerifyBlobKZGProofBatch_6", "drop_in.single_calls_ms.VerifyBlobKZGProofBatch_6"),
            ("single_call_ms.ComputeCellsAndKZGProofs", "drop_in.single_calls_ms.ComputeCellsAndKZGProofs"),
            ("single_blob_callers_32.commit_blobs_per_s", "drop_in.single_blob_callers.blob_to_kzg_commitment.callers_32.blobs_per_s"),
            ("single_blob_callers_32.cells_blobs_per_s", "drop_in.single_blob_callers.compute_cells_and_kzg_proofs.callers_32.blobs_per_s"),
            ("single_blob_callers_32.verify_blobs_per_s", "drop_in.single_blob_callers.verify_blob_kzg_proof.callers_32.blobs_per_s"),
            ("multi_gpu.verify_strong_8192_blobs_per_s", "verify_blob_kzg_proof_batch.strong_scaling.blobs_per_s"),
            ("multi_gpu.verify_cells_strong_65536_cells_per_s", "eip7594.verify_cell_kzg_proof_batch.strong_scaling.cells_per_s"),
            ("multi_gpu.exchange_us", "multi_gpu.exchange_us"),
            ("multi_gpu.in_process_commit_blobs_per_s", "multi_gpu.in_process.blob_to_kzg_commitment_batch.blobs_per_s")):
        if path is None:
            dk = next((k for k in out.get("drop_in", {}) if k.startswith(("default_options", "direct_table"))), None)
            v = dig(out, f"drop_in.{dk}.blobs_per_s_one_caller") if dk else None
        else:
            v = dig(out, path)
        if v is not None:
            legs[name] = v
    hv = dig(out, "drop_in.verify_blob_kzg_proof_batch", {})
    ks = sorted(int(k.split("_")[1]) for k in hv if k.startswith("callers_"))
    if ks:
        legs[f"verify_blob_batch_1024.host_pointers_{ks[-1]}_callers_blobs_per_s"] = hv[f"callers_{ks[-1]}"].get("blobs_per_s")
        legs[f"verify_blob_batch_1024.host_pointers_{ks[-1]}_callers_pcie_frac"] = dig(hv[f"callers_{ks[-1]}"], "pcie.frac")
    c["legs"] = legs
    c["detail"] = {"file": detail_path, "note": "the whole record (per-leg roofline / valu / pcie objects, spreads, kernel times) -- also on stderr behind BENCH_DETAIL; "
                                                "`python bench.py --full-line` prints it as the stdout line"}
    return c


def bind_to_gpu_node(torch, device):
    """What `numactl --cpunodebind` does for a rank of a multi-GPU job, done by the rank itself: bind this process (and every thread it
    starts later: the library's host pool, the caller threads of the sustained legs) to the CPUs of the NUMA node its GPU hangs off,
    so that the host buffers it allocates from here on are local to the pinned staging memory HIP places on that node.  *Measured* on the
    two-socket boxes of this pool (host-pointer VerifyBlobKZGProofBatch(1024), lone call): 3.92-3.96 ms bound, 4.0-5.1 ms floating --
    the figure then depends on which socket the process happened to allocate its blobs on.  BENCH_NO_NUMA_BIND=1 leaves the process alone.
    Returns what was done, for the JSON line."""
    info = {"bound": False}
    if os.environ.get("BENCH_NO_NUMA_BIND"):
        info["note"] = "BENCH_NO_NUMA_BIND"
        return info
    try:
        p = torch.cuda.get_device_properties(device)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{getattr(p, 'pci_device_id', 0):02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as fh:
            node = int(fh.read())
        if node < 0:
            return info
        with open(f"/sys/devices/system/node/node{node}/cpulist") as fh:
            cpus = set()
            for part in fh.read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        mine = os.sched_getaffinity(0)
        want = cpus & mine
        info.update({"gpu_pci": bdf, "gpu_numa_node": node, "node_cpus": len(cpus)})
        if want and want != mine and len(want) >= 4:
            os.sched_setaffinity(0, want)
            info["bound"] = True
    except Exception as e:      # noqa: BLE001
        info["note"] = repr(e)
    return info


def hw_limits(props):
    """the hardware's VALU issue limit: one wave64 instruction per SIMD per 4 cycles at the device's maximum clock"""
    clk_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3
    return {"valu_peak": props.multi_processor_count * 4 * clk_hz / 4.0,
            "note": f"{props.multi_processor_count} CUs x 4 SIMDs x {clk_hz / 1e9:.2f} GHz / 4 cycles per wave64 instruction"}


# profile kinds of mkzg_profile_read that bracket several kernels -> the kernels (x launches per call) whose PMC figures add up to them
KIND_KERNELS = {
    "k_fk20_ffts": {"k_fk20_pre": 3, "k_fk20_toom_mul": 1, "k_fk20_post": 3, "k_fk20_finish": 1},
    "k_fk20_msm": {"k_fk20_msm": 1, "k_fk20_msm_reduce": 1},
    "k_vmsm_window": {"k_vmsm_window": 1},                                           # combines its slices itself since round 5 (k_vmsm_combine: the quad path only)
    "k_cell_verify_scalars": {"k_cell_rpowers": 1, "k_cell_interp": 1},              # weights ride in k_cell_rpowers, the final sum in k_cell_interp (round 5)
    "k_g1_decompress": {"k_g1_validate": 1},                                         # the cell verifier's decoding launch (mode 1)
    "k_recover_*": {"k_recover_vanishing": 1, "k_recover_stage_a": 1, "k_recover_stage_b1": 1, "k_recover_stage_c": 1, "k_recover_stage_d": 1},
}


def leg_roofline(kern_ms, algo_bytes, hw, pmc_valid_for_size=True, leg="verify", kinds=None):
    """`roofline` + `valu` objects of one leg.  kern_ms: {profile kind: ms per call}; algo_bytes: SURVEY.md 8(d)'s algorithmic bytes of
    one call.  The dominant kernel is the kind with the largest duration.  PMC figures (HBM-side traffic, VALU wave-instructions per
    launch) come from the committed rocprofv3 passes and are used only while the kernel sources they were collected on are unchanged
    and the leg runs at the size they were collected at."""
    if not kern_ms:
        return None, None
    dom, dom_ms = max(kern_ms.items(), key=lambda kv: kv[1])
    achieved = algo_bytes / (dom_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
            "traffic": None, "avg_kernel_ms": round(dom_ms, 4), "algorithmic_bytes_per_call": algo_bytes}
    valu = {"unit": "G wave-instr/s", "peak_hw": round(hw["valu_peak"] / 1e9, 1), "peak_hw_note": hw["note"], "kernels": {},
            "note": "ms: hipEvent brackets of this run; wave-instructions and traffic: committed rocprofv3 PMC passes (per kernel the mean over its largest launches; "
                    "a kind that brackets several kernels sums them), used only while the kernel sources they were collected on are unchanged"}
    stale = None
    for kind, ms in kern_ms.items():
        parts = (kinds or {}).get(kind) or KIND_KERNELS.get(kind, {kind: 1})
        insts, traffic, src, ok = 0.0, 0.0, None, pmc_valid_for_size
        for kname, mult in parts.items():
            pm = pmc_profile(kname, leg)
            if pm.get("stale"):
                stale = pm["stale"]
            if not pm.get("valu_wave_insts"):
                ok = False
                break
            insts += pm["valu_wave_insts"] * mult
            traffic += pm.get("traffic_bytes", 0.0) * mult
            src = pm.get("source_file")
        if not ok or ms <= 0:
            continue
        rate = insts / (ms * 1e-3)
        valu["kernels"][kind] = {"wave_insts_per_call": round(insts), "ms": round(ms, 4), "achieved": round(rate / 1e9, 2), "frac_hw": round(rate / hw["valu_peak"], 4),
                                 "traffic_bytes": round(traffic), "pmc_source": src}
        if kind == dom:
            roof["traffic"] = round(traffic)
            roof["traffic_over_algorithmic"] = round(traffic / algo_bytes, 2)
    if not valu["kernels"]:
        valu["note"] = stale or ("no committed PMC pass covers this leg at this size" if not pmc_valid_for_size else "no committed PMC pass covers these kernels")
        roof["traffic_note"] = valu["note"]
    return roof, valu


# the kernels of ONE VerifyBlobKZGProofBatch call in the throughput shape (api_verify.hip): profile kind -> kernels x launches
# (the evaluation's z-only rows ride in the hash launch and the scalars of the linear combination in the evaluation kernel's tail)
TP_KINDS = {"k_g1_validate": {"k_g1_validate_lean": 1}, "k_challenge_sha256": {"k_challenge_sha256_jobs": 1}, "k_blob_eval_tree": {"k_blob_eval_tree": 1},
            "k_vmsm_window": {"k_vmsm_window": 1}}
# the cell verifier's kinds (EIP-7594 leg): the r-torsion test of 8192 proofs is the register-lean one-lane form
CELL_KINDS = dict(KIND_KERNELS, k_g1_validate={"k_g1_validate_lean": 1})


def sustained_roofline(kinds, leg, algo_bytes_per_call, calls_per_s, hw, pmc_valid_for_size=True):
    """`roofline` + `valu` of a leg in which many calls share the chip: no kernel's duration is its own there, so the figures are RATES --
    algorithmic bytes per second against HBM peak, and the wave-instructions of all the kernels of a call (committed PMC passes of the
    same kernels run alone, valid while their sources are unchanged) times calls per second against the hardware issue limit."""
    gbs = algo_bytes_per_call * calls_per_s / 1e9
    roof = {"bound": "hbm", "achieved": round(gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_call": algo_bytes_per_call,
            "calls_per_s": round(calls_per_s, 2), "traffic": None, "note": "rate over the whole leg: calls overlap, no kernel's duration is its own"}
    insts, traffic, per, stale, src = 0.0, 0.0, {}, None, None
    ok = pmc_valid_for_size
    for kind, parts in kinds.items():
        ki, kt = 0.0, 0.0
        for kname, mult in parts.items():
            pm = pmc_profile(kname, leg)
            if pm.get("stale"):
                stale = pm["stale"]
            if not pm.get("valu_wave_insts"):
                ok = False
                break
            ki += pm["valu_wave_insts"] * mult
            kt += pm.get("traffic_bytes", 0.0) * mult
            src = pm.get("source_file")
        if not ok:
            break
        per[kind] = round(ki)
        insts += ki
        traffic += kt
    if not ok:
        return roof, {"note": stale or "no committed PMC pass covers this leg's kernels at this size", "unit": "G wave-instr/s", "peak_hw": round(hw["valu_peak"] / 1e9, 1)}
    rate = insts * calls_per_s
    roof["traffic"] = round(traffic)
    roof["traffic_over_algorithmic"] = round(traffic / algo_bytes_per_call, 2)
    valu = {"unit": "G wave-instr/s", "wave_insts_per_call": round(insts), "wave_insts_per_call_by_kernel": per, "achieved": round(rate / 1e9, 2), "peak_hw": round(hw["valu_peak"] / 1e9, 1),
            "peak_hw_note": hw["note"], "frac_hw": round(rate / hw["valu_peak"], 4), "pmc_source": src,
            "note": "sum over the kernels of one call (PMC passes of the same kernels run alone) x calls per second of the sustained leg"}
    return roof, valu


def multi_gpu_info(torch, dist, dist_on, world, rank, local_rank, backend, dev, props, ctx):
    """who the ranks are (proof of N distinct GPUs), what the library's host pools add up to, and what the batch verifiers' exchange
    step costs on its own: all_gather of the shard sizes, broadcast of r, all_gather of one 512-byte record per rank
    (go-eth-kzg_amd/distributed.py) -- timed alone, between barriers, over 20 repetitions"""
    D = load_distributed()
    bus = getattr(props, "pci_bus_id", None)
    ident = {"rank": rank, "local_rank": local_rank, "device": props.name,
             "pci": (f"{getattr(props, 'pci_domain_id', 0):04x}:{bus:02x}:{getattr(props, 'pci_device_id', 0):02x}" if bus is not None else None),
             "uuid": str(getattr(props, "uuid", "")) or None, "host_threads": ctx.get_option("host_threads")}
    idents = [None] * world
    dist.all_gather_object(idents, ident)
    ex = D.measure_exchange(dist, dev, reps=20)
    t = torch.tensor([ex["exchange_us"], ex["broadcast_us"], ex["all_gather_sizes_us"], ex["all_gather_records_us"]], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    pcis = [i["pci"] or i["uuid"] for i in idents]
    return {"ranks": world, "backend": ("nccl (RCCL)" if backend == "nccl" else backend), "rank_gpus": idents, "distinct_gpus": len(set(pcis)),
            "host_threads_total": sum(i["host_threads"] or 0 for i in idents), "usable_cpus": usable_cpus(),
            "exchange_us": round(float(t[0]), 1), "exchange_breakdown_us": {"broadcast_r_32B": round(float(t[1]), 1), "all_gather_shard_sizes": round(float(t[2]), 1),
                                                                           "all_gather_records_512B": round(float(t[3]), 1)},
            "exchange_note": "max over ranks of the mean of 20 repetitions, tensors on " + ("the GPUs over RCCL" if backend == "nccl" else "the host over " + backend)}


def source_hash(files):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "go-eth-kzg_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


PMC_SOURCES = {"commit": ["msm_kernels.hip", "msm_device.hpp", "g1.hpp", "g1_quad.hpp", "mont29.hpp", "modinv30.hpp"],
               "verify": ["verify_kernels.hip", "cells_kernels.hip", "g1.hpp", "g1_quad.hpp", "mont29.hpp", "g1_decode.hpp", "pub_ticket.hpp"],
               "eip7594": ["fk20_kernels.hip", "recover_kernels.hip", "cells_kernels.hip", "verify_kernels.hip", "g1.hpp", "g1_quad.hpp", "mont29.hpp", "g1_decode.hpp", "pub_ticket.hpp"]}


def pmc_profile(kernel, leg="commit"):
    """Per-launch PMC figures of `kernel` from the committed rocprofv3 passes (profiles/*_traffic.json: HBM-side bytes,
    VALU wave-instructions; the counters cannot be read from inside the process), as collected for `leg` ("commit", "verify",
    "eip7594": one bench.py configuration each, so that a kernel's largest launches are that leg's).  A profile is only used for the
    kernel sources it was collected on (it records their hash); {} if no valid profile covers the kernel."""
    import glob
    fam = leg if leg in PMC_SOURCES else "verify"               # the source family whose hash guards the leg's figures
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            with open(f) as fh:
                j = json.load(fh)
            k = (j.get("legs", {}).get(leg) or j.get("kernels") or {}).get(kernel)
            if not k:
                continue
            want = (j.get("sources") or {}).get(fam)
            if want != source_hash(PMC_SOURCES[fam]):
                stale = f"{os.path.basename(f)} was collected on other kernel sources ({fam}: {want}); re-run tools/pmc_collect.sh"
                continue
            return dict(k, source_file=os.path.basename(f))
        except Exception:
            pass
    return {"stale": stale} if stale else {}


def bench_drop_in(ctx, kzg, host_blobs, d_expect):
    """What a caller of the reference API gets (prove.go:13, Go's []*Blob): 256 scattered pageable host blobs in, 256 x 48 host bytes out,
    through mkzg_blob_to_kzg_commitment_batch at the C ABI (arguments marshalled once, as a cgo caller would hold them); one synchronous
    caller, then two callers (two lanes: one batch uploads while the other computes)."""
    import torch
    n = len(host_blobs)
    L = kzg.lib()
    keep = [ctypes.create_string_buffer(b, BLOB) for b in host_blobs]                # separate allocations, like Go's []*Blob
    arr = (ctypes.c_char_p * n)(*[ctypes.cast(k, ctypes.c_char_p) for k in keep])
    expect = d_expect.cpu().numpy().tobytes()

    def caller(reps, res):
        out = ctypes.create_string_buffer(48 * n)
        st = (ctypes.c_int * n)()
        for _ in range(reps):
            rc = L.mkzg_blob_to_kzg_commitment_batch(ctx._h, arr, n, out, st)
            assert rc == 0
        res.append(out.raw)

    res = []
    caller(2, res)                                                                    # warm: staging buffers, lanes
    assert res[0] == expect, "host-pointer path disagrees with the device-resident one"
    reps = 6
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
    assert all(r == expect for r in res)
    # the same batch device-resident from one SYNCHRONOUS caller on this box, right now: what the host-pointer caller could get at best
    d_in = torch.frombuffer(bytearray(b"".join(host_blobs)), dtype=torch.uint8).cuda()
    d_o = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
    d_s = torch.zeros(n, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    dts = []
    for _ in range(8):
        t0 = time.perf_counter()
        assert ctx.blob_to_kzg_commitment_batch_device(d_in.data_ptr(), n, d_o.data_ptr(), d_s.data_ptr(), st) == 0
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dtd = sorted(dts[2:])[len(dts[2:]) // 2]
    return {"blobs_per_s_one_caller": round(n / dt1, 1), "ms_per_call": round(dt1 * 1e3, 3), "blobs_per_s_two_callers": round(n / dt2, 1), "batch": n,
            "device_resident_one_synchronous_caller": {"ms_per_call": round(dtd * 1e3, 3), "blobs_per_s": round(n / dtd, 1)},
            "one_caller_over_device_resident": round(dtd / dt1, 3), "two_callers_over_device_resident": round(dtd / dt2, 3),
            "note": "PCIe inclusive: pageable host blobs in (gathered into pinned staging chunk by chunk), host bytes out"}


def bench_single_calls(ctx, host_blobs, reps=30):
    """The reference's API as its tests and clients call it -- ONE item per call, host pointers in, host bytes out, one caller
    (prove.go:13-111, verify.go:12-143, api_eip7594.go:28-52,144-214) -- latency in ms, median of `reps` calls each; the block-sized
    VerifyBlobKZGProofBatch (6 and 9 blobs: EIP-4844 / EIP-7691) beside the single-proof forms.  Results checked against each other
    (the proofs verify, a swapped proof does not)."""
    b = host_blobs[:9]
    cms = [ctx.BlobToKZGCommitment(x)[1] for x in b]
    prs = [ctx.ComputeBlobKZGProof(x, c)[1] for x, c in zip(b, cms)]
    z = (0x5eed).to_bytes(32, "big")
    rc, pz, y = ctx.ComputeKZGProof(b[0], z)
    assert rc == 0 and ctx.VerifyKZGProof(cms[0], z, y, pz) == 0 and ctx.VerifyBlobKZGProof(b[0], cms[0], prs[0]) == 0 and ctx.VerifyBlobKZGProof(b[0], cms[0], prs[1]) == 1
    assert ctx.VerifyBlobKZGProofBatch(b, cms, prs) == 0 and ctx.VerifyBlobKZGProofBatch(b, cms, prs[1:] + prs[:1]) == 1
    rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(b[0])
    assert rc == 0
    half = list(range(0, 128, 2))
    calls = {"BlobToKZGCommitment": lambda: ctx.BlobToKZGCommitment(b[0]),
             "ComputeKZGProof": lambda: ctx.ComputeKZGProof(b[0], z),
             "ComputeBlobKZGProof": lambda: ctx.ComputeBlobKZGProof(b[0], cms[0]),
             "VerifyKZGProof": lambda: ctx.VerifyKZGProof(cms[0], z, y, pz),
             "VerifyBlobKZGProof": lambda: ctx.VerifyBlobKZGProof(b[0], cms[0], prs[0]),
             "VerifyBlobKZGProofBatch_6": lambda: ctx.VerifyBlobKZGProofBatch(b[:6], cms[:6], prs[:6]),
             "VerifyBlobKZGProofBatch_9": lambda: ctx.VerifyBlobKZGProofBatch(b, cms, prs),
             "ComputeCellsAndKZGProofs": lambda: ctx.ComputeCellsAndKZGProofs(b[0]),
             "VerifyCellKZGProofBatch_128": lambda: ctx.VerifyCellKZGProofBatch([cms[0]] * 128, list(range(128)), cells, proofs),
             "RecoverCellsAndComputeKZGProofs": lambda: ctx.RecoverCellsAndComputeKZGProofs(half, [cells[i] for i in half])}
    out = {}
    for name, fn in calls.items():
        fn()
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        out[name] = round(ts[len(ts) // 2] * 1e3, 3)
    out["note"] = ("ms per call, median of %d, one caller, host pointers through the ctypes binding (its marshalling included); options host_fin / host_validate "
                   "at their defaults: the one resulting point and the few input points of these calls are finished / validated on the calling thread" % reps)
    return out


def bench_single_blob_callers(ctx, kzg, torch, callers=(8, 32), per_thread=4, seconds=0.35, kinds=None, wrong_every=0):
    """What an UNMODIFIED caller of the reference gets: BlobToKZGCommitment, ComputeCellsAndKZGProofs and VerifyBlobKZGProof take ONE blob
    (prove.go:13-33, api_eip7594.go:28-52, verify.go:48-83) and clients call them from many goroutines on one shared context
    (verify.go:152-169, examples_test.go:11).  K threads on one context, each looping the single-blob host-pointer entry point of the C
    ABI over its own blobs (arguments marshalled once, as a cgo caller holds them); blobs/s over all threads, host CPUs busy; the caller
    of this function puts the ratio to the batch entry point of the same box beside it.  ctypes releases the GIL around every call."""
    L = kzg.lib()
    K_max = max(callers)
    n = K_max * per_thread
    hb = fast_blobs(n, 8800)
    d_b = torch.from_numpy(hb.reshape(-1)).cuda()
    d_cm = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(n, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert ctx.blob_to_kzg_commitment_batch_device(d_b.data_ptr(), n, d_cm.data_ptr(), d_st.data_ptr(), st) == 0
    assert ctx.compute_blob_kzg_proof_batch_device(d_b.data_ptr(), d_cm.data_ptr(), n, d_pr.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    cms, prs = d_cm.cpu().numpy().tobytes(), d_pr.cpu().numpy().tobytes()
    blobs = [ctypes.create_string_buffer(hb[i].tobytes(), BLOB) for i in range(n)]
    cm = [cms[48 * i:48 * i + 48] for i in range(n)]
    pr = [prs[48 * i:48 * i + 48] for i in range(n)]

    timed = [False]

    def run(kind, K):
        errs, counts = [], [0] * K
        stop = threading.Event()

        def caller(k):
            try:
                out48 = ctypes.create_string_buffer(48)
                cells = ctypes.create_string_buffer(128 * 2048)
                proofs = ctypes.create_string_buffer(128 * 48)
                i = 0
                while not stop.is_set():
                    j = k * per_thread + i % per_thread
                    if kind == "blob_to_kzg_commitment":
                        rc = L.mkzg_blob_to_kzg_commitment(ctx._h, blobs[j], 0, out48)
                        ok = rc == 0 and (i >= per_thread or out48.raw == cm[j])
                    elif kind == "compute_cells_and_kzg_proofs":
                        rc = L.mkzg_compute_cells_and_kzg_proofs(ctx._h, blobs[j], 0, cells, proofs)
                        ok = rc == 0
                    else:
                        # (wrong_every = m: every m-th call brings another blob's proof and must be rejected -- a group that holds one
                        # verifies member by member; the timed trials of bench.py run clean, a wrong proof per thread in the priming trial)
                        wrong = (wrong_every > 0 and i % wrong_every == wrong_every - 1) or (not timed[0] and i == 1)
                        rc = L.mkzg_verify_blob_kzg_proof(ctx._h, blobs[j], cm[j], pr[(j + 1) % n] if wrong else pr[j])
                        ok = rc == (1 if wrong else 0)
                    if not ok:
                        errs.append((kind, k, i, rc))
                        return
                    i += 1
                    counts[k] = i
            except Exception as e:      # noqa: BLE001
                errs.append(repr(e))

        rates, cpus = [], []
        for trial in range(3):                                                  # one priming trial, then two timed
            timed[0] = trial > 0
            stop.clear()
            counts[:] = [0] * K
            ts = [threading.Thread(target=caller, args=(k,)) for k in range(K)]
            t0, c0 = time.perf_counter(), time.process_time()
            for t in ts:
                t.start()
            time.sleep(seconds if trial else 0.1)
            stop.set()
            for t in ts:
                t.join()
            dt, cpu = time.perf_counter() - t0, time.process_time() - c0
            assert not errs, errs[0]
            if trial:
                rates.append(sum(counts) / dt)
                cpus.append(cpu / dt)
        return {"blobs_per_s": round(max(rates), 1), "blobs_per_s_each": [round(r, 1) for r in rates], "host_cpus_busy": round(max(cpus), 1), "us_per_call_per_thread": round(K / max(rates) * 1e6, 1)}

    out = {"blobs_per_thread": per_thread, "seconds_per_trial": seconds, "combine_single_calls": int(ctx.get_option("combine_single_calls") or 0)}
    before = ctx.get_option("combined_calls") or 0
    for kind in kinds or ("blob_to_kzg_commitment", "compute_cells_and_kzg_proofs", "verify_blob_kzg_proof"):
        out[kind] = {f"callers_{K}": run(kind, K) for K in callers}
    out["calls_served_in_groups"] = int((ctx.get_option("combined_calls") or 0) - before)
    return out


def bench_multi_in_process(kzg, torch, n_dev, B=256, N=1024, reps=4, devices=None):
    """The form ONE Go process on a multi-GPU node would use (verify.go:152-169: one shared Context, many goroutines): a single context
    over n_dev devices (mkzg_ctx_new with n_dev ids, api_multi.hip), host-pointer batches cut into contiguous shards inside the library,
    the batch verifiers' per-device sums added in process -- no RCCL, no ranks.  Runs on rank 0 after every rank's own context is closed
    (the tables of both forms do not fit one GPU's HBM together).  Host-pointer batches of n_dev x the per-GPU sizes of the headline legs."""
    if os.environ.get("BENCH_INPROC_DEVICES"):          # test aid for a 1-GPU box: e.g. "0,0" = two peers on the one GPU (tests/test_gpu_bench_multirank.py)
        devices = [int(x) for x in os.environ["BENCH_INPROC_DEVICES"].split(",")]
        n_dev = len(devices)
        if os.environ.get("BENCH_INPROC_SIZES"):
            B, N = (int(x) for x in os.environ["BENCH_INPROC_SIZES"].split(","))
    if n_dev < 2:
        return {"skipped": "one device (--gpus 1, or one device visible): the in-process multi-device context is the single-device context"}
    try:
        os.sched_setaffinity(0, range(os.cpu_count()))                    # this leg spans sockets: each peer's pool goes to its own GPU's node instead
    except OSError:
        pass
    os.environ["MKZG_POOL_AFFINITY"] = "1"
    L = kzg.lib()
    ctx = kzg.Context(devices=devices if devices is not None else list(range(n_dev)))
    out = {"devices": n_dev, "device_ids": devices if devices is not None else list(range(n_dev)), "host_threads": int(ctx.get_option("host_threads") or 0)}
    try:
        torch.cuda.set_device(0)
        nb, nv = B * n_dev, N * n_dev
        hb = fast_blobs(nv, 9900)
        keep = [ctypes.create_string_buffer(hb[i].tobytes(), BLOB) for i in range(nv)]
        arr = (ctypes.c_char_p * nv)(*[ctypes.cast(k, ctypes.c_char_p) for k in keep])
        # commitments + proofs of all nv blobs through the context itself (host pointers in: sharded over the devices)
        cms = ctypes.create_string_buffer(48 * nv)
        st = (ctypes.c_int * nv)()
        assert L.mkzg_blob_to_kzg_commitment_batch(ctx._h, arr, nv, cms, st) == 0
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            assert L.mkzg_blob_to_kzg_commitment_batch(ctx._h, arr, nb, cms, st) == 0
            ts.append(time.perf_counter() - t0)
        med = sorted(ts)[len(ts) // 2]
        out["blob_to_kzg_commitment_batch"] = {"blobs": nb, "ms_per_call": round(med * 1e3, 3), "blobs_per_s": round(nb / med, 1)}
        assert L.mkzg_blob_to_kzg_commitment_batch(ctx._h, arr, nv, cms, st) == 0
        d_b = torch.from_numpy(hb.reshape(-1)).cuda()
        d_cm = torch.frombuffer(bytearray(cms.raw), dtype=torch.uint8).cuda()
        d_pr = torch.zeros(nv * 48, dtype=torch.uint8, device="cuda")
        d_st = torch.zeros(nv, dtype=torch.int32, device="cuda")
        assert ctx.compute_blob_kzg_proof_batch_device(d_b.data_ptr(), d_cm.data_ptr(), nv, d_pr.data_ptr(), d_st.data_ptr(), 0) == 0      # runs where the pointers live: device 0
        torch.cuda.synchronize()
        prs = d_pr.cpu().numpy().tobytes()
        bad = prs[:48 * (nv - 1)] + prs[:48]                               # the LAST shard's last proof is another blob's
        assert L.mkzg_verify_blob_kzg_proof_batch(ctx._h, arr, cms.raw, prs, nv, nv, nv) == 0
        assert L.mkzg_verify_blob_kzg_proof_batch(ctx._h, arr, cms.raw, bad, nv, nv, nv) == 1
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            assert L.mkzg_verify_blob_kzg_proof_batch(ctx._h, arr, cms.raw, prs, nv, nv, nv) == 0
            ts.append(time.perf_counter() - t0)
        med = sorted(ts)[len(ts) // 2]
        out["verify_blob_kzg_proof_batch"] = {"blobs": nv, "ms_per_call": round(med * 1e3, 3), "blobs_per_s": round(nv / med, 1),
                                              "pcie_GBps_all_links": round(nv * VERIFY_BYTES_PER_BLOB / med / 1e9, 2)}
        out["tables_on_own_device"] = int(ctx.get_option("tables_on_own_device") or 0)
    finally:
        ctx.close()
    return out


PCIE_PEAK_GBS = 63.0                       # PCIe 5.0 x16, one direction: 32 GT/s x 16 lanes x 128/130 = 63.0 GB/s before packet overheads
VERIFY_BYTES_PER_BLOB = 131072 + 48 + 48   # what one proof brings across the link


def fast_blobs(n, seed):
    """n distinct canonical blobs, fast (numpy Philox instead of 4096 SHA-256 calls per blob): uniform scalars below 2^254 < r"""
    import numpy as np
    a = np.random.Generator(np.random.Philox(seed)).integers(0, 256, size=(n, 4096, 32), dtype=np.uint8)
    a[:, :, 0] &= 0x3F
    return a.reshape(n, BLOB)


def bench_drop_in_verify(ctx, kzg, torch, N=1024, callers=(4, 8), calls_one=12, reps=6, trials=3):
    """VerifyBlobKZGProofBatch as the reference API sees it (verify.go:88-143, bench_test.go:107-123): N separate pageable host
    allocations (Go's []*Blob) + N x 48 B commitments + N x 48 B proofs in host memory, the verdict back.  The binding roofline of this
    half of the metric is the PCIe link, not HBM: every blob crosses it once.  One synchronous caller (latency: min / median / max over
    `calls_one` calls), then K callers on the shared context, EVERY CALLER WITH ITS OWN N DISTINCT BLOBS (K x 134 MB of host memory: no
    cache holds them), each also submitting one corrupted batch that must be rejected; median of `trials` timed trials."""
    import numpy as np
    L = kzg.lib()
    K_max = max([1] + list(callers))
    st = torch.cuda.current_stream().cuda_stream
    sets = []
    for k in range(K_max):
        hb = fast_blobs(N, 7000 + k)
        d_b = torch.from_numpy(hb.reshape(-1)).cuda()
        d_cm = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
        d_pr = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
        d_st = torch.zeros(N, dtype=torch.int32, device="cuda")
        assert ctx.blob_to_kzg_commitment_batch_device(d_b.data_ptr(), N, d_cm.data_ptr(), d_st.data_ptr(), st) == 0
        assert ctx.compute_blob_kzg_proof_batch_device(d_b.data_ptr(), d_cm.data_ptr(), N, d_pr.data_ptr(), d_st.data_ptr(), st) == 0
        torch.cuda.synchronize()
        assert int(d_st.abs().sum().item()) == 0
        cms, prs = d_cm.cpu().numpy().tobytes(), d_pr.cpu().numpy().tobytes()
        keep = [ctypes.create_string_buffer(hb[i].tobytes(), BLOB) for i in range(N)]              # N separate allocations, like Go's []*Blob
        arr = (ctypes.c_char_p * N)(*[ctypes.cast(b, ctypes.c_char_p) for b in keep])
        v = min(700, N - 2)
        bad = prs[:48 * v] + prs[48 * (v + 1):48 * (v + 2)] + prs[48 * (v + 1):]                   # proof v replaced by its neighbour's
        sets.append((keep, arr, cms, prs, bad))
        del d_b, d_cm, d_pr, d_st, hb

    def call(k, proofs=None):
        _, arr, cms, prs, _ = sets[k]
        return L.mkzg_verify_blob_kzg_proof_batch(ctx._h, arr, cms, proofs if proofs is not None else prs, N, N, N)

    def pcie(blobs_per_s):
        gbs = blobs_per_s * VERIFY_BYTES_PER_BLOB / 1e9
        return {"achieved_GBps": round(gbs, 2), "peak": PCIE_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / PCIE_PEAK_GBS, 4), "bytes_per_blob": VERIFY_BYTES_PER_BLOB}

    assert call(0) == 0 and call(0, sets[0][4]) == 1                                                # warm: staging, lanes; correctness
    each = []
    for _ in range(calls_one):
        t0 = time.perf_counter()
        assert call(0) == 0
        each.append((time.perf_counter() - t0) * 1e3)
    srt = sorted(each)
    med = srt[len(srt) // 2]
    out = {"batch": N, "one_caller": {"ms_per_call_min": round(srt[0], 3), "ms_per_call_median": round(med, 3), "ms_per_call_max": round(srt[-1], 3), "calls": calls_one,
                                      "blobs_per_s": round(N / (med * 1e-3), 1), "pcie": pcie(N / (med * 1e-3))}}
    for K in callers:
        errs = []

        def caller(k):
            try:
                for _ in range(reps):
                    assert call(k) == 0
                assert call(k, sets[k][4]) == 1
            except Exception as e:      # noqa: BLE001
                errs.append(repr(e))

        dts, cpus = [], []
        for trial in range(trials + 1):                  # the first trial primes the lanes' staging areas
            ts = [threading.Thread(target=caller, args=(k,)) for k in range(K)]
            t0, c0 = time.perf_counter(), time.process_time()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            if trial:
                dts.append(time.perf_counter() - t0)
                cpus.append((time.process_time() - c0) / dts[-1])
        assert not errs, errs[0]
        order = sorted(range(len(dts)), key=lambda i: dts[i])
        mid = order[len(order) // 2]
        rate = K * (reps + 1) * N / dts[mid]
        out[f"callers_{K}"] = {"blobs_per_s": round(rate, 1), "blobs_per_s_each_trial": [round(K * (reps + 1) * N / d, 1) for d in dts], "ms_per_call_amortised": round(dts[mid] / (K * (reps + 1)) * 1e3, 3),
                               "calls": K * (reps + 1), "host_cpus_busy": round(cpus[mid], 1), "pcie": pcie(rate)}
    out["note"] = ("host pointers in, verdict out (PCIe inclusive): N separate pageable allocations per caller, distinct blobs per caller, gathered into pinned staging and uploaded inside "
                   "the call; median of the timed trials; every caller also submits a batch with one wrong proof, which must be rejected")
    return out


def bench_verify(ctx, torch, dist, dist_on, dev, world, rank, args, d_blobs256, B, hw):
    """configs[2]: VerifyBlobKZGProofBatch on N synthetic (blob, commitment, proof) triples per GPU.
    Commitments and proofs are produced on the device by the product itself; one corrupted run
    must be rejected."""
    N = args.verify_batch
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
    if args.verify_shape >= 0:
        assert ctx.set_option("verify_shape", args.verify_shape) == 0
    rc = ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, st)     # warm-up + correctness
    assert rc == 0, f"valid batch rejected rc={rc}"
    d_bad = d_pr.clone()
    d_bad[0:48] = d_pr[48:96]
    assert ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_bad.data_ptr(), N, st) == 1, "corrupted batch accepted"
    D = load_distributed() if dist_on else None
    eng = D.ProductEngine(ctx) if dist_on else None

    def maxed(dt):
        if not dist_on:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def one_call():
        if dist_on:
            # one global batch of N * world proofs, sharded: broadcast r, fold locally, all-gather 512-byte records
            # over RCCL, finish with one pairing (go-eth-kzg_amd/distributed.py)
            return D.verify_blob_kzg_proof_batch_sharded(eng, (d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, st), N * world, dist, dev)
        return ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, st)

    # per-kernel durations from one profiled call (hipEvent brackets); the timed calls run without them
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    assert one_call() == 0
    prof = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_us = 0.0
    each = []
    for _ in range(args.verify_steps):
        tc = time.perf_counter()
        assert one_call() == 0
        each.append(round((time.perf_counter() - tc) * 1e3, 3))
        host_us += ctx.last_host_finish_us()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = maxed(time.perf_counter() - t0)
    ms = dt / args.verify_steps * 1e3
    if args.verify_shape >= 0:
        assert ctx.set_option("verify_shape", -1) == 0
    srt_each = sorted(each)
    kern = {k: round(v[0], 4) for k, v in prof.items() if v[1]}
    algo = (131072 + 48 + 48) * N          # bytes read per blob
    # the latency shape brackets two launches under k_vmsm_window (two jobs beside the hash, the late one on quads of lanes)
    tp_forced = args.verify_shape == 1
    kinds = TP_KINDS if tp_forced else (None if dist_on else {"k_vmsm_window": {"k_vmsm_window": 1, "k_vmsm_window_quad": 1, "k_vmsm_combine": 1}})
    roof, valu = leg_roofline(kern, algo, hw, pmc_valid_for_size=(N == 1024), leg="verify_tp" if tp_forced else "verify", kinds=kinds)
    out = {"metric": "blobs/sec VerifyBlobKZGProofBatch", "batch_per_gpu": N, "unit": "blobs/s",
           "single_call": {"blobs_per_s": round(N * world / dt * args.verify_steps, 2), "ms_per_call": round(ms, 3), "host_finish_ms": round(host_us / args.verify_steps / 1e3, 3), "ms_each_call": each,
                           "ms_per_call_min": srt_each[0], "ms_per_call_median": srt_each[len(srt_each) // 2], "ms_per_call_max": srt_each[-1], "calls": len(each),
                           "shape": "throughput shape forced (verify_shape 1)" if tp_forced else "library default: latency shape for a lone call",
                           "parallelism": "one call at a time" if not dist_on else f"one global batch of {N * world} sharded x{world}; RCCL broadcast(r) + all_gather(512 B records)"},
           "compute_blob_kzg_proof_batch_proofs_per_s": round(proofs_per_s, 1),
           "kernels_ms_per_call": kern,
           "roofline": roof, "valu": valu}
    if dist_on and not args.no_strong:
        # STRONG scaling: one fixed global batch of 8192 proofs whatever the number of ranks -- rank r folds 8192 / world of them (its own
        # N distinct triples, repeated to fill its shard), one exchange, one pairing.  The driver sets the per-N values side by side.
        G = 8192
        lo, hi = D.shard_bounds(G, world, rank)
        m = hi - lo
        reps_t = -(-m // N)
        sb = d_blobs.view(N, BLOB).repeat(reps_t, 1)[:m].contiguous()
        sc = d_cm.view(N, 48).repeat(reps_t, 1)[:m].contiguous()
        sp = d_pr.view(N, 48).repeat(reps_t, 1)[:m].contiguous()
        spb = sp.clone()
        if rank == world - 1:
            spb[m - 1] = sc[m - 1]                       # the last proof of the batch replaced by a valid point that is not the proof
        torch.cuda.synchronize()

        def strong_call(proofs):
            return D.verify_blob_kzg_proof_batch_sharded(eng, (sb.data_ptr(), sc.data_ptr(), proofs.data_ptr(), m, st), G, dist, dev)

        assert strong_call(sp) == 0
        assert strong_call(spb) == 1, "a wrong proof in the last shard went unnoticed"
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.verify_steps):
            assert strong_call(sp) == 0
        dist.barrier()
        dts = maxed(time.perf_counter() - t0) / args.verify_steps
        out["strong_scaling"] = {"global_batch": G, "per_rank": m, "ranks": world, "ms_per_call": round(dts * 1e3, 3), "blobs_per_s": round(G / dts, 1),
                                 "note": f"one global batch of {G} proofs cut over {world} rank(s) ({N} distinct triples per rank, repeated to fill its shard); "
                                         "broadcast(r) + all_gather(sizes) + all_gather(512 B records), one pairing on every rank"}
        del sb, sc, sp, spb
    if dist_on and dev is not None and dev.type == "cuda":
        # the same sharded call with the exchange step INSIDE the library (csrc/api_exchange.hip: RCCL all-gather of the shard sizes,
        # broadcast of r, all-gather of the 512-byte records, issued by libmi355kzg.so itself -- what a Go host calls; torch.distributed
        # only carries the 128-byte communicator id here)
        try:
            Lc = sys.modules["goethkzg_amd"].lib()
            ident = [None]
            if rank == 0:
                buf = ctypes.create_string_buffer(128)
                assert Lc.mkzg_comm_unique_id(buf) == 0
                ident = [buf.raw]
            dist.broadcast_object_list(ident, src=0)
            hcomm = ctypes.c_void_p()
            rc = Lc.mkzg_comm_init(ctx._h, ident[0], rank, world, ctypes.byref(hcomm))
            assert rc == 0, f"mkzg_comm_init rc={rc}"

            def lib_call(proofs):
                return Lc.mkzg_verify_blob_kzg_proof_batch_exchange_device(hcomm, d_blobs.data_ptr(), d_cm.data_ptr(), proofs.data_ptr(), N, N * world, st)

            assert lib_call(d_pr) == 0
            assert lib_call(d_bad) == 1, "a wrong proof in rank shards went unnoticed by the library's exchange"
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ex_us = 0.0
            for _ in range(args.verify_steps):
                assert lib_call(d_pr) == 0
                ex_us += Lc.mkzg_comm_last_exchange_us(hcomm)
            dist.barrier()
            dtl = maxed(time.perf_counter() - t0) / args.verify_steps
            out["library_exchange"] = {"ms_per_call": round(dtl * 1e3, 3), "blobs_per_s": round(N * world / dtl, 1), "exchange_us_per_call_this_rank": round(ex_us / args.verify_steps, 1),
                                       "note": "mkzg_verify_blob_kzg_proof_batch_exchange_device on a communicator made by mkzg_comm_init: the three collectives are issued by the library (RCCL, dlopen'ed)"}
            Lc.mkzg_comm_free(hcomm)
        except Exception as e:      # noqa: BLE001
            out["library_exchange"] = {"error": repr(e)}
    if world == 1 and not dist_on:
        # the reference's own API shape at the sizes a block carries: host pointers in (Go's []*Blob), one synchronous caller
        hb = d_blobs[: min(N, 48) * BLOB].cpu().numpy().tobytes()
        hcm, hpr = d_cm.cpu().numpy().tobytes(), d_pr.cpu().numpy().tobytes()
        small = {}
        for nb in (1, 6, 16, 48):
            if nb > N:
                continue
            bl = [hb[i * BLOB:(i + 1) * BLOB] for i in range(nb)]
            cl = [hcm[48 * i:48 * i + 48] for i in range(nb)]
            pl = [hpr[48 * i:48 * i + 48] for i in range(nb)]
            assert ctx.VerifyBlobKZGProofBatch(bl, cl, pl) == 0
            t0 = time.perf_counter()
            for _ in range(5):
                assert ctx.VerifyBlobKZGProofBatch(bl, cl, pl) == 0
            small[str(nb)] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
        out["host_pointer_ms_per_call_by_blobs"] = small
    # sustained rate: K caller threads on one shared context, one call in flight each (reference verify.go:152-169: the Context is
    # shared by goroutines).  Each call works in its own lane; calls that find others in flight take the throughput shape (barycentric
    # evaluation, one stream per call) and share hash launches (api_verify.hip hash_round_join).
    K = args.verify_inflight
    if os.environ.get("BENCH_SHARE_GPU") and world > 1:
        K = min(K, 4)                                    # test aid: all ranks on ONE GPU
    if K > 1:
        # at N > 1 every rank runs this leg on ITS OWN batches at the same time (a node verifying many batches: no exchange -- the one
        # exchange step exists only where a single batch spans the GPUs, which is `single_call` / `strong_scaling`); the rates add up
        # long enough to leave the start-up convoy behind (all callers start together; a round is 5-7 ms): *measured* (round 5, one box, tools/trace.py
        # verify_inflight 32 N) 12 calls per caller 1.96-2.01 M blobs/s, 26: 2.21 M, 50: 2.22-2.26 M -- 30 + 1 calls per caller, ~0.45 s per trial
        reps = max(30, args.verify_steps)
        errs = []
        # DISTINCT inputs: `--ring` sets of N (blob, commitment, proof) triples resident in HBM (8 x 134 MB: four times the Infinity
        # Cache), caller k verifies set k mod ring -- the algorithmic bytes of this leg really stream from HBM
        sets = [(d_blobs, d_cm, d_pr, d_bad)]
        for q in range(1, max(1, args.ring)):
            sb = torch.from_numpy(fast_blobs(N, 5000 + 97 * rank + q).reshape(-1)).cuda()
            sc = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
            sp = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
            assert ctx.blob_to_kzg_commitment_batch_device(sb.data_ptr(), N, sc.data_ptr(), d_st.data_ptr(), st) == 0
            assert ctx.compute_blob_kzg_proof_batch_device(sb.data_ptr(), sc.data_ptr(), N, sp.data_ptr(), d_st.data_ptr(), st) == 0
            torch.cuda.synchronize()
            assert int(d_st.abs().sum().item()) == 0
            sbad = sp.clone()
            sbad[0:48] = sp[48:96]
            sets.append((sb, sc, sp, sbad))

        def caller(k):
            try:
                s = torch.cuda.Stream(priority=-1 if k % 2 == 0 else 0)
                b_, c_, p_, bad_ = sets[k % len(sets)]
                for _ in range(reps):
                    assert ctx.verify_blob_kzg_proof_batch_device(b_.data_ptr(), c_.data_ptr(), p_.data_ptr(), N, s.cuda_stream) == 0
                assert ctx.verify_blob_kzg_proof_batch_device(b_.data_ptr(), c_.data_ptr(), bad_.data_ptr(), N, s.cuda_stream) == 1
            except Exception as e:      # noqa: BLE001
                errs.append(repr(e))

        def trials(n_trials):
            """the timed trials after one priming trial (every lane's workspaces): (wall time, host CPUs kept busy) of each"""
            res = []
            for trial in range(n_trials + 1):
                ts = [threading.Thread(target=caller, args=(k,)) for k in range(K)]
                torch.cuda.synchronize()
                if dist_on:
                    dist.barrier()                       # the ranks' trials run at the same time
                t0, c0 = time.perf_counter(), time.process_time()
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
                dt, cpu = time.perf_counter() - t0, time.process_time() - c0
                if trial:
                    res.append((dt, cpu / dt))
            return sorted(res)

        def leg(res):
            med = res[len(res) // 2]
            rate = K * (reps + 1) * N / med[0]
            d = {"blobs_per_s": round(rate, 1), "blobs_per_s_each_trial": [round(K * (reps + 1) * N / r[0], 1) for r in res],
                 "ms_per_call_amortised": round(med[0] / (K * (reps + 1)) * 1e3, 3), "host_cpus_busy": round(med[1], 1)}
            if dist_on:                                  # whole job: the ranks' median rates added up
                t = torch.tensor([rate], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                d["blobs_per_s_this_rank"] = d["blobs_per_s"]
                d["blobs_per_s"] = round(float(t.item()), 1)
            return d

        default = trials(3)                              # library default: host_wait -1 (a crowded context asks and sleeps)
        assert ctx.set_option("host_wait", 0) == 0
        polling = trials(3)
        assert ctx.set_option("host_wait", -1) == 0
        assert not errs, errs[0]
        sus = leg(default)
        calls_per_s = sus.get("blobs_per_s_this_rank", sus["blobs_per_s"]) / N
        sroof, svalu = sustained_roofline(TP_KINDS, "verify_tp", algo, calls_per_s, hw, pmc_valid_for_size=(N == 1024))
        out["sustained"] = {"callers": K, "callers_per_rank": K, "ranks": world, "calls": K * (reps + 1) * world, **sus, "host_wait": int(ctx.get_option("host_wait")), "cpu_share": int(ctx.get_option("cpu_share")),
                            "distinct_input_sets": len(sets), "input_bytes_resident": len(sets) * N * VERIFY_BYTES_PER_BLOB,
                            "polling_waits": {**leg(polling), "note": "option host_wait 0: every waiting caller polls in hipStreamSynchronize / hipEventSynchronize"},
                            "roofline": sroof, "valu": svalu,
                            "shape": "throughput shape: shared hash rounds, barycentric evaluation, one stream per call (option verify_shape -1)",
                            "note": "verdicts identical to the serial calls (each caller also submits one corrupted batch, which must be rejected); caller k verifies input set k mod "
                                    f"{len(sets)}; one priming trial, then the MEDIAN of three timed trials"}
        del sets
    out["value"] = out.get("sustained", {}).get("blobs_per_s", out["single_call"]["blobs_per_s"])
    out["value_is"] = ((f"sustained: {K} callers on one context" + (f" per rank, {world} ranks on their own batches at the same time, rates added up" if dist_on else "")) if "sustained" in out else
                       ("one global batch sharded over the ranks, one call at a time" if dist_on else "single calls, one at a time"))
    return out


def bench_eip7594(ctx, kzg, torch, dist, dist_on, dev, world, rank, args, hw):
    """configs[3] (ComputeCellsAndKZGProofs, FK20) and configs[4] (VerifyCellKZGProofBatch over NB x 128 cells per GPU -- one global batch
    sharded over the ranks with one exchange step -- and RecoverCellsAndComputeKZGProofs of NB blobs per GPU from their even cells,
    sharded by blob with no collective).  SURVEY.md 8d: seeds 0..63, blob-major tuples, even-indexed cells."""
    NB = args.cell_blobs
    solo = world == 1 and not dist_on
    full_size = NB == 64                    # the committed PMC passes were collected at 64 blobs / 8192 cells
    assert ctx.set_option("fk20_bits", args.fk20_bits) == 0

    def maxed(dt):
        if not dist_on:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    blobs = [rand_blob(rank * NB + s) for s in range(NB)]
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
    d_cm = torch.zeros(NB * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(NB, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st) == 0      # builds the FK20 table
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    steps = 5
    out = {}
    ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st)        # a second warm call: the lane's workspaces at their size
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    each = []
    for _ in range(steps):                     # every call timed on its own: the leg's figure is the MEDIAN (one call in five that meets a hiccup used to move a mean of three by 20 %)
        t0 = time.perf_counter()
        ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st)
        torch.cuda.synchronize()
        each.append(time.perf_counter() - t0)
    dt = sorted(each)[steps // 2]
    prof = ctx.profile_read(reset=True)
    kern = {k: round(v[0] / steps, 3) for k, v in prof.items() if v[1]}
    roof, valu = leg_roofline(kern, 399360 * NB, hw, pmc_valid_for_size=full_size and ctx.get_option("fk20_bits_effective") == 12, leg="eip7594")
    out["compute_cells_and_kzg_proofs"] = {"blobs_per_gpu": NB, "fk20_table_bits": ctx.get_option("fk20_bits_effective"), "ms_per_batch": round(dt * 1e3, 3), "blobs_per_s_per_gpu": round(NB / dt, 1),
                                           "ms_each_call": [round(t * 1e3, 3) for t in each], "kernels_ms": kern, "algorithmic_GBps": round(399360 * NB / dt / 1e9, 3), "roofline": roof, "valu": valu}
    ctx.profile_enable(False)
    if solo:
        # the sizes a block carries (6 blobs today, up to 48 planned): one synchronous caller, ms per call
        by_batch = {}
        for nb in (1, 6, 9, 16, 32, 48):
            if nb > NB:
                continue
            ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), nb, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), nb, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st)
            torch.cuda.synchronize()
            by_batch[str(nb)] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st) == 0      # the outputs the later legs compare with
        torch.cuda.synchronize()
        if NB == 64 and not args.no_big_cells:
            # calls of more blobs than a block carries (a node catching up; pieces of up to 256 blobs, the products of 65..192 in segments):
            # the first NB blobs' proofs must equal the NB-blob call's
            big = 256
            d_b4 = d_blobs.repeat(big // NB)
            d_c4 = torch.zeros(big * 262144, dtype=torch.uint8, device="cuda")
            d_p4 = torch.zeros(big * 128 * 48, dtype=torch.uint8, device="cuda")
            d_s4 = torch.ones(big, dtype=torch.int32, device="cuda")
            for nb in (128, 192, 256):
                for _ in range(3):                      # back-to-back asynchronous calls take a lane each: prime the workspaces of all three
                    assert ctx.compute_cells_and_kzg_proofs_batch_device(d_b4.data_ptr(), nb, d_c4.data_ptr(), d_p4.data_ptr(), d_s4.data_ptr(), st) == 0
                torch.cuda.synchronize()
                assert int(d_s4[:nb].abs().sum().item()) == 0 and torch.equal(d_p4[:NB * 6144], d_proofs) and torch.equal(d_p4[(nb - NB) * 6144:nb * 6144], d_proofs)
                t0 = time.perf_counter()
                marks = []
                for _ in range(3):
                    ctx.compute_cells_and_kzg_proofs_batch_device(d_b4.data_ptr(), nb, d_c4.data_ptr(), d_p4.data_ptr(), d_s4.data_ptr(), st)
                    marks.append(round((time.perf_counter() - t0) * 1e3, 2))
                torch.cuda.synchronize()
                by_batch[str(nb)] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
                if os.environ.get("BENCH_DEBUG"):
                    print(f"[debug] cells nb={nb}: calls returned at {marks} ms, all done at {by_batch[str(nb)] * 3:.2f} ms, free HBM {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB", file=sys.stderr, flush=True)
            out["compute_cells_and_kzg_proofs"]["blobs_per_s_256_per_call"] = round(256 / (by_batch["256"] * 1e-3), 1)
            del d_b4, d_c4, d_p4, d_s4
        out["compute_cells_and_kzg_proofs"]["ms_per_call_by_blobs"] = by_batch
        # sustained: eight caller threads on the shared context (four until round 3: 17.2-17.7 k blobs/s against 18.6-18.8 k), each
        # pipelining calls of NB blobs on its own stream -- one call's latency-bound network of 686 scalar multiplications (two thirds
        # of the SIMDs, one wavefront each) runs beside another call's VALU-bound MSM
        KC, reps = 8, 10                                 # (trials longer than the start-up convoy: ~0.3 s each)
        # every caller its own NB distinct blobs (and the proofs a serial call gives for them, to compare with)
        csets = []
        for k in range(KC):
            cb = d_blobs if k == 0 else torch.from_numpy(fast_blobs(NB, 6000 + k).reshape(-1)).cuda()
            exp_p = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
            assert ctx.compute_cells_and_kzg_proofs_batch_device(cb.data_ptr(), NB, d_cells.data_ptr(), exp_p.data_ptr(), d_st.data_ptr(), st) == 0
            torch.cuda.synchronize()
            assert int(d_st.abs().sum().item()) == 0
            csets.append((cb, exp_p))
        assert torch.equal(csets[0][1], d_proofs)
        assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st) == 0      # d_cells back to set 0
        torch.cuda.synchronize()
        bufs = [(torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda"), torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda"),
                 torch.ones(NB, dtype=torch.int32, device="cuda")) for _ in range(KC)]
        torch.cuda.synchronize()
        errs = []

        def cells_caller(k):
            try:
                s = torch.cuda.Stream(priority=-1 if k % 2 == 0 else 0)
                cc, pp, ss = bufs[k]
                for _ in range(reps):
                    assert ctx.compute_cells_and_kzg_proofs_batch_device(csets[k][0].data_ptr(), NB, cc.data_ptr(), pp.data_ptr(), ss.data_ptr(), s.cuda_stream) == 0
                s.synchronize()
            except Exception as e:      # noqa: BLE001
                errs.append(repr(e))

        dts = []
        for trial in range(4):                           # the first trial primes the lanes' workspaces; the MEDIAN of the other three counts
            ts = [threading.Thread(target=cells_caller, args=(k,)) for k in range(KC)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            if trial:
                dts.append(time.perf_counter() - t0)
        dts.sort()
        dtk = dts[len(dts) // 2]
        assert not errs, errs[0]
        assert all(int(b[2].abs().sum().item()) == 0 and bool((b[1] == csets[k][1]).all().item()) for k, b in enumerate(bufs)), "concurrent calls disagree with the serial ones"
        ckinds = {kind: KIND_KERNELS.get(kind, {kind: 1}) for kind in kern}
        sroof, svalu = sustained_roofline(ckinds, "eip7594", 399360 * NB, KC * reps / dtk, hw, pmc_valid_for_size=full_size and ctx.get_option("fk20_bits_effective") == 12)
        out["compute_cells_and_kzg_proofs"]["sustained"] = {"callers": KC, "calls": KC * reps, "blobs_per_s": round(KC * reps * NB / dtk, 1), "blobs_per_s_each_trial": [round(KC * reps * NB / d, 1) for d in dts],
                                                            "ms_per_call_amortised": round(dtk / (KC * reps) * 1e3, 3), "distinct_blobs_per_caller": NB, "roofline": sroof, "valu": svalu,
                                                            "note": "every caller its own blobs; proofs equal to the serial calls'; one priming trial, then the median of three"}
        del bufs, csets
    ctx.profile_enable(True)
    assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), NB, d_cm.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    cms = d_cm.cpu().numpy().tobytes()
    cells = d_cells.cpu().numpy().tobytes()
    proofs = d_proofs.cpu().numpy().tobytes()
    n = NB * 128
    # C-ABI arguments marshalled once (what a cgo caller holds): cells as an array of pointers into separate allocations
    keep = [ctypes.create_string_buffer(cells[i * 2048:(i + 1) * 2048], 2048) for i in range(n)]
    arr = (ctypes.c_char_p * n)(*[ctypes.cast(k, ctypes.c_char_p) for k in keep])
    idx = (ctypes.c_uint64 * n)(*[i % 128 for i in range(n)])
    cj = b"".join(cms[48 * (i // 128):48 * (i // 128) + 48] for i in range(n))
    L = kzg.lib()
    D = load_distributed() if dist_on else None

    class RawCellEngine:
        """C-ABI calls on pre-marshalled arguments; `shard` = (commitments, indices, cell pointer array, proofs, count)"""

        def cell_partial(self, shard, offset, r32):
            c_, i_, a_, p_, m_ = shard
            rec = ctypes.create_string_buffer(512)
            rc = L.mkzg_verify_cell_kzg_proof_batch_partial(ctx._h, c_, m_, i_, m_, a_, m_, p_, m_, offset, bytes(r32), rec)
            return rc, rec.raw

        def cell_finish(self, records):
            return L.mkzg_verify_cell_kzg_proof_batch_finish(ctx._h, b"".join(records), len(records))

    if dist_on:
        def verify_once():
            return D.verify_cell_kzg_proof_batch_sharded(RawCellEngine(), (cj, idx, arr, proofs, n), n * world, dist, dev, n_local=n)
    else:
        def verify_once():
            return L.mkzg_verify_cell_kzg_proof_batch(ctx._h, cj, n, idx, n, arr, n, proofs, n)
    assert verify_once() == 0
    assert verify_once() == 0
    ctx.profile_read(reset=True)
    if dist_on:
        dist.barrier()
    # every call timed on its own (round 5: the mean of `steps` calls could not tell an outlier from a slow path -- the driver's box
    # once reported 3.0 ms where the same library measured 1.97 elsewhere): min / median / max, the host finish, the link's rate
    n_calls = max(steps, 12)
    each, fin = [], []
    t0 = time.perf_counter()
    for _ in range(n_calls):
        t1 = time.perf_counter()
        assert verify_once() == 0
        each.append((time.perf_counter() - t1) * 1e3)
        fin.append(ctx.last_host_finish_us() / 1e3)
    if dist_on:
        dist.barrier()
    dt_mean = maxed((time.perf_counter() - t0) / n_calls)
    prof = ctx.profile_read(reset=True)
    kern = {k: round(v[0] / n_calls, 3) for k, v in prof.items() if v[1]}
    roof, valu = leg_roofline(kern, 2152 * n, hw, pmc_valid_for_size=full_size, leg="eip7594", kinds=CELL_KINDS)
    srt = sorted(each)
    med = srt[len(srt) // 2]
    dt = dt_mean if dist_on else med * 1e-3                    # several ranks: the slowest rank's mean between barriers; one rank: the median call
    out["verify_cell_kzg_proof_batch"] = {"cells_per_gpu": n, "cells_total": n * world, "ms_per_call": round(dt * 1e3, 3), "cells_per_s": round(n * world / dt, 1),
                                          "calls": n_calls, "ms_per_call_min": round(srt[0], 3), "ms_per_call_median": round(med, 3), "ms_per_call_max": round(srt[-1], 3),
                                          "ms_per_call_mean": round(dt_mean * 1e3, 3), "ms_per_call_each": [round(x, 3) for x in each],
                                          "host_finish_ms": round(sorted(fin)[len(fin) // 2], 3),
                                          "pcie": {"bytes_per_call": 2152 * n, "achieved_GBps": round(2152 * n / (med * 1e-3) / 1e9, 2), "peak_GBps": PCIE_PEAK_GBS,
                                                   "frac": round(2152 * n / (med * 1e-3) / 1e9 / PCIE_PEAK_GBS, 4),
                                                   "note": "a lone call is a chain (gather + upload, interpolation, MSM, host finish), not a stream: the link is busy for a quarter of it"},
                                          "parallelism": "one call" if not dist_on else f"one global batch sharded x{world}; all_gather(sizes, r) + all_gather(512 B records)",
                                          "kernels_ms": kern, "roofline": roof, "valu": valu,
                                          "note": "host pointers in (16 MiB of cells per 8192 gathered and uploaded inside the call); ms_per_call = the median of `calls` calls timed one by one"}
    if solo:
        # the cells already in HBM (mkzg_verify_cell_kzg_proof_batch_device: commitments, indices and proofs stay host arrays): no gather, no 16 MiB upload
        assert L.mkzg_verify_cell_kzg_proof_batch_device(ctx._h, cj, idx, d_cells.data_ptr(), proofs, n, st) == 0
        each = []
        for _ in range(6):
            t0 = time.perf_counter()
            assert L.mkzg_verify_cell_kzg_proof_batch_device(ctx._h, cj, idx, d_cells.data_ptr(), proofs, n, st) == 0
            each.append((time.perf_counter() - t0) * 1e3)
        each.sort()
        out["verify_cell_kzg_proof_batch"]["device_cells"] = {"ms_per_call_min": round(each[0], 3), "ms_per_call_median": round(each[len(each) // 2], 3), "ms_per_call_max": round(each[-1], 3),
                                                              "cells_per_s": round(n / (each[len(each) // 2] * 1e-3), 1), "note": "cells contiguous in HBM, the three small arrays on the host"}
        ctx.profile_read(reset=True)
    if dist_on and not args.no_strong:
        # STRONG scaling: one fixed global batch of 65,536 cells (512 blobs' worth) whatever the number of ranks; a rank's shard is its
        # own NB x 128 tuples repeated to fill 65,536 / world (pointers into the same host cells: nothing is copied)
        G = 65536
        lo, hi = D.shard_bounds(G, world, rank)
        m = hi - lo
        reps_t = -(-m // n)
        arr_s = (ctypes.c_char_p * m)(*([ctypes.cast(k, ctypes.c_char_p) for k in keep] * reps_t)[:m])
        idx_s = (ctypes.c_uint64 * m)(*([i % 128 for i in range(n)] * reps_t)[:m])
        cj_s = (cj * reps_t)[:48 * m]
        pr_s = (proofs * reps_t)[:48 * m]
        pr_bad = pr_s[:48 * (m - 1)] + cj_s[48 * (m - 1):48 * m] if rank == world - 1 else pr_s      # last proof := a valid point that is not the proof

        def strong_call(pr):
            return D.verify_cell_kzg_proof_batch_sharded(RawCellEngine(), (cj_s, idx_s, arr_s, pr, m), G, dist, dev, n_local=m)

        assert strong_call(pr_s) == 0
        assert strong_call(pr_bad) == 1, "a wrong proof in the last shard went unnoticed"
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            assert strong_call(pr_s) == 0
        dist.barrier()
        dts = maxed((time.perf_counter() - t0) / steps)
        out["verify_cell_kzg_proof_batch"]["strong_scaling"] = {
            "global_batch": G, "per_rank": m, "ranks": world, "ms_per_call": round(dts * 1e3, 3), "cells_per_s": round(G / dts, 1),
            "note": f"one global batch of {G} cells cut over {world} rank(s) ({n} distinct tuples per rank, repeated to fill its shard); broadcast(r) + all_gather(sizes) + all_gather(512 B records)"}
        ctx.profile_read(reset=True)
    if solo:
        # sustained: twelve caller threads on the shared context, each also submitting one batch with a wrong proof that must be rejected
        ctx.profile_enable(False)
        badp = bytearray(proofs)
        badp[0:48] = proofs[48:96]
        badp = bytes(badp)
        # KV: measured (tools/bench_cellverify_inflight.py, one box, verdicts into pinned memory): 4 callers 8.5-9.5, 8: 7.4-9.8, 12: 8.9, 16: 8.3, 24: 8.4 M cells/s
        # reps: all callers start together and move in step for their first calls (twelve gathers, then twelve uploads, ...): *measured* (round 5, one box, distinct cells
        # per caller) 9 calls per caller 9.3-9.8 M cells/s with trials of 8.9-12.2, 30 calls 10.8-10.9 M (10.5-11.5), 60 calls 11.5 M (11.3-11.6) -- a SUSTAINED rate needs
        # trials longer than the start-up convoy: 36 + 1 calls per caller, ~0.35 s per trial
        KV, reps = 12, 36
        errs = []
        # every caller its own NB x 128 (commitment, index, cell, proof) tuples in host memory (12 x 17.6 MB)
        vsets = [(cj, arr, proofs, badp, keep)]
        for k in range(1, KV):
            vb = torch.from_numpy(fast_blobs(NB, 6500 + k).reshape(-1)).cuda()
            vc = torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda")
            vp = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
            vm = torch.zeros(NB * 48, dtype=torch.uint8, device="cuda")
            assert ctx.compute_cells_and_kzg_proofs_batch_device(vb.data_ptr(), NB, vc.data_ptr(), vp.data_ptr(), d_st.data_ptr(), st) == 0
            assert ctx.blob_to_kzg_commitment_batch_device(vb.data_ptr(), NB, vm.data_ptr(), d_st.data_ptr(), st) == 0
            torch.cuda.synchronize()
            vcms, vcells, vproofs = vm.cpu().numpy().tobytes(), vc.cpu().numpy().tobytes(), vp.cpu().numpy().tobytes()
            vkeep = [ctypes.create_string_buffer(vcells[i * 2048:(i + 1) * 2048], 2048) for i in range(n)]
            varr = (ctypes.c_char_p * n)(*[ctypes.cast(q, ctypes.c_char_p) for q in vkeep])
            vcj = b"".join(vcms[48 * (i // 128):48 * (i // 128) + 48] for i in range(n))
            vbad = vproofs[48:96] + vproofs[48:]
            vsets.append((vcj, varr, vproofs, vbad, vkeep))
            del vb, vc, vp, vm

        def cv_caller(k):
            try:
                cj_, arr_, pr_, bad_, _ = vsets[k]
                for _ in range(reps):
                    assert L.mkzg_verify_cell_kzg_proof_batch(ctx._h, cj_, n, idx, n, arr_, n, pr_, n) == 0
                assert L.mkzg_verify_cell_kzg_proof_batch(ctx._h, cj_, n, idx, n, arr_, n, bad_, n) == 1
            except Exception as e:      # noqa: BLE001
                errs.append(repr(e))

        dts, cpus = [], []
        for trial in range(4):                           # the first trial primes the lanes' workspaces; the MEDIAN of the other three counts
            ts = [threading.Thread(target=cv_caller, args=(k,)) for k in range(KV)]
            t0, c0 = time.perf_counter(), time.process_time()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            if trial:
                dts.append(time.perf_counter() - t0)
                cpus.append((time.process_time() - c0) / dts[-1])
        dts.sort()
        dtk = dts[len(dts) // 2]
        assert not errs, errs[0]
        vkinds = {kind: CELL_KINDS.get(kind, {kind: 1}) for kind in kern}
        sroof, svalu = sustained_roofline(vkinds, "eip7594", 2152 * n, KV * (reps + 1) / dtk, hw, pmc_valid_for_size=full_size)
        out["verify_cell_kzg_proof_batch"]["sustained"] = {"callers": KV, "calls": KV * (reps + 1), "cells_per_s": round(KV * (reps + 1) * n / dtk, 1),
                                                           "cells_per_s_each_trial": [round(KV * (reps + 1) * n / d, 1) for d in dts],
                                                           "ms_per_call_amortised": round(dtk / (KV * (reps + 1)) * 1e3, 3), "host_cpus_busy": round(sorted(cpus)[len(cpus) // 2], 1),
                                                           "roofline": sroof, "valu": svalu,
                                                           "pcie": {"achieved_GBps": round(KV * (reps + 1) * n * 2152 / dtk / 1e9, 2), "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                                                                    "frac": round(KV * (reps + 1) * n * 2152 / dtk / 1e9 / PCIE_PEAK_GBS, 4)},
                                                           "note": "host pointers in; every caller its own cells; one priming trial, then the median of three"}
        del vsets
        ctx.profile_enable(True)
    # recovery: every blob of this rank from its 64 even-indexed cells, one batch call (sharded by blob, no collective) at the C ABI
    # with the arguments marshalled ONCE, as a cgo caller holds them (api_eip7594.go:144: []uint64 + []*Cell per blob)
    ids = list(range(0, 128, 2))
    id_arr = (ctypes.c_uint64 * 64)(*ids)
    cell_arrs = [(ctypes.c_char_p * 64)(*[ctypes.cast(keep[b * 128 + i], ctypes.c_char_p) for i in ids]) for b in range(NB)]
    pid = (ctypes.POINTER(ctypes.c_uint64) * NB)(*[ctypes.cast(id_arr, ctypes.POINTER(ctypes.c_uint64)) for _ in range(NB)])
    pcl = (ctypes.POINTER(ctypes.c_char_p) * NB)(*[ctypes.cast(a, ctypes.POINTER(ctypes.c_char_p)) for a in cell_arrs])
    nid = (ctypes.c_size_t * NB)(*[64] * NB)
    oc = ctypes.create_string_buffer(NB * 262144)
    op = ctypes.create_string_buffer(NB * 6144)
    rst = (ctypes.c_int32 * NB)()

    def recover_once():
        return L.mkzg_recover_cells_and_kzg_proofs_batch(ctx._h, NB, pid, nid, pcl, nid, oc, op, rst)

    assert recover_once() == 0 and not any(rst)
    assert oc.raw == cells and op.raw == proofs, "recovered cells / proofs differ from the encoder's"
    ctx.profile_read(reset=True)
    if dist_on:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        assert recover_once() == 0
    if dist_on:
        dist.barrier()
    dt = maxed((time.perf_counter() - t0) / steps)
    prof = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    kern = {k: round(v[0] / steps, 3) for k, v in prof.items() if v[1]}
    roof, valu = leg_roofline(kern, 399872 * NB, hw, pmc_valid_for_size=full_size and ctx.get_option("fk20_bits_effective") == 12, leg="eip7594",
                              kinds={"k_cells_ntt": {"k_cells_first_half": 1, "k_cells_ntt": 1}})
    out["recover_cells_and_kzg_proofs"] = {"blobs_per_gpu": NB, "ms_per_batch": round(dt * 1e3, 3), "blobs_per_s": round(NB * world / dt, 1),
                                           "parallelism": f"blob-sharded x{world}, no collective", "kernels_ms": kern, "roofline": roof, "valu": valu,
                                           "note": "C ABI, arguments marshalled once: 64 scattered host cells in per blob, 128 cells + 128 proofs out to host memory (PCIe inclusive)"}
    return out


def cpu_baseline():
    """oracle/ (plain-C port of the same path, OpenMP over blobs) on the host cores this process may use, bounded sample.  The thread
    count is the affinity mask capped by the cgroup CPU quota (oversubscribing a 16-CPU quota with 256 threads is what made round 1's
    figure meaningless); the single-thread rate is reported beside it."""
    sys.path.insert(0, ROOT)
    import oracle
    threads = usable_cpus()
    octx = oracle.OracleContext(with_fk20=False)
    n = max(8, min(512, 4 * threads))
    blobs = b"".join(oracle.rand_blob(s) for s in range(n))
    octx.blob_to_kzg_commitment_batch(blobs[:BLOB * 2], 2, threads=threads)   # warm
    t0 = time.perf_counter()
    k1 = 0
    while time.perf_counter() - t0 < 3.0:                     # single thread, about 3 s
        octx.blob_to_kzg_commitment_batch(blobs[:BLOB * 4], 4, threads=1)
        k1 += 4
    rate1 = k1 / (time.perf_counter() - t0)
    passes, t0 = 0, time.perf_counter()
    while True:                                               # all usable threads, about 10 s of wall time
        octx.blob_to_kzg_commitment_batch(blobs, n, threads=threads)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= 10.0 or passes >= 200:
            break
    octx.close()
    v = n * passes / dt
    return {"value": round(v, 3), "unit": "blobs/s", "cores": threads, "kind": "port",
            "gnark_crypto_estimate": {"factor_over_this_port": [2.0, 4.0], "blobs_per_s_range_on_these_cores": [round(2 * v, 1), round(4 * v, 1)],
                                      "basis": "not measured (no Go toolchain, gnark-crypto v0.16.0 not vendored): the port multiplies in Fp with a portable 64-bit CIOS loop and adds "
                                               "into extended-Jacobian buckets; gnark's G1Affine.MultiExp uses hand-written ADX assembly for the field (about 2x per multiplication) and "
                                               "batch-affine bucket additions (about 1.3-1.5x fewer multiplications per addition) -- hold the GPU figures against this range, not against `value`"},
            "single_thread_blobs_per_s": round(rate1, 3), "scaling_efficiency": round(n * passes / dt / (rate1 * threads), 3),
            "cpus_visible": os.cpu_count(), "affinity_mask": len(os.sched_getaffinity(0)),
            "sample": f"{passes} passes over {n} GetRandBlob blobs, OpenMP over blobs with {threads} threads ({dt:.2f} s); 1 thread: {k1} blobs in 3 s"}


def in_process_leg_main(n_dev):
    """child of bench.py (rank 0, after every rank has closed its context): the one-process-over-all-devices form on its own"""
    import torch
    torch.cuda.init()
    kzg = load_product()
    try:
        res = bench_multi_in_process(kzg, torch, n_dev)
    except Exception as e:      # noqa: BLE001
        res = {"error": repr(e)[:300]}
    print("IN_PROCESS " + json.dumps(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--in-process-leg":
        in_process_leg_main(int(sys.argv[2]))
    else:
        main()
