"""The N > 1 path of bench.py on a box with ONE GPU: `python bench.py --gpus 2` (no launcher: bench.py starts its own ranks, the parent
never touches the GPU) with both ranks on GPU 0 talking over gloo (bench.py's BENCH_SHARE_GPU / BENCH_DIST_BACKEND test aids).  Checked:
the run completes, rank 0 prints ONE JSON line with n_gpus = 2 and every section -- weak-scaled commitments, the sharded blob batch
verifier, the sharded cell batch verifier (one global batch, broadcast + all_gather), the blob-sharded recovery, who the ranks are, what
the exchange step costs and the strong-scaling legs.  Also: EIGHT ranks on the one GPU (the shape of the driver's 8-GPU run: the host
pools must add up to the CPUs the box grants, not to eight times them), and ONE rank over the "nccl" backend, so that the broadcast /
all_gather calls of go-eth-kzg_amd/distributed.py run on device tensors over real RCCL at least once before an 8-GPU node sees them."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, launcher, gpus=2, sizes=("32", "32", "3")):
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    if env.get("BENCH_DIST_BACKEND") == "":
        del env["BENCH_DIST_BACKEND"]
    if env.get("BENCH_SHARE_GPU") == "":
        del env["BENCH_SHARE_GPU"]
    args = ["--gpus", str(gpus), "--steps", "3", "--warmup", "1", "--repeats", "2", "--batch", sizes[0], "--direct-bits", "12", "--verify-batch", sizes[1], "--verify-steps", "1",
            "--cell-blobs", sizes[2], "--no-cpu-baseline"]
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, "detail.json")
        cmd = launcher + [os.path.join(ROOT, "bench.py")] + args + ["--detail-out", detail]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        assert r.stdout.strip().splitlines()[-1] == lines[0], "the JSON line must be the last line on stdout (RCCL / gloo print banners there too)"
        # the stdout line is the COMPACT one (round 5's 21.6 KB line came back from the driver unparsed): the contract's keys, roofline, one
        # figure per leg; the whole record is in the detail file and on stderr
        c = json.loads(lines[0])
        assert len(lines[0]) < 6000, len(lines[0])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "legs", "detail"):
            assert k in c, k
        assert c["roofline"]["bound"] == "hbm" and c["roofline"]["peak"] == 8000.0 and "frac" in c["roofline"] and "workload" in c["config"]
        with open(detail) as fh:
            d = json.loads(fh.read())
        assert d["value"] == c["value"] and d["n_gpus"] == c["n_gpus"] and d["ms_per_step"] == c["ms_per_step"]
        assert c["legs"].get("verify_blob_batch_1024.sustained_blobs_per_s") == d["verify_blob_kzg_proof_batch"]["sustained"]["blobs_per_s"]
        on_err = [ln for ln in r.stderr.splitlines() if ln.startswith("BENCH_DETAIL ")]
        assert len(on_err) == 1 and json.loads(on_err[0][len("BENCH_DETAIL "):]) == d
        return d


def _check(d, n=2, cell_blobs=3):
    assert d["n_gpus"] == n and d["value"] > 0 and d["scaling"] == "weak"
    assert d["repeats"]["regions"] == 2 and d["repeats"]["ms_per_step_min"] <= d["repeats"]["ms_per_step_max"]
    m = d["multi_gpu"]
    assert m["ranks"] == n and len(m["rank_gpus"]) == n and sorted(g["rank"] for g in m["rank_gpus"]) == list(range(n))
    assert m["exchange_us"] > 0 and set(m["exchange_breakdown_us"]) == {"broadcast_r_32B", "all_gather_shard_sizes", "all_gather_records_512B"}
    v = d["verify_blob_kzg_proof_batch"]
    assert v["value"] > 0 and f"sharded x{n}" in v["single_call"]["parallelism"]
    # the sustained leg runs on every rank at once on its own batches; the whole-job rate is the sum and is the leg's (and the line's) value
    su = v["sustained"]
    assert su["ranks"] == n and su["blobs_per_s"] >= su["blobs_per_s_this_rank"] > 0 and v["value"] == su["blobs_per_s"] == d["verify_metric"]["value"]
    assert v["strong_scaling"]["global_batch"] == 8192 and v["strong_scaling"]["per_rank"] == 8192 // n and v["strong_scaling"]["blobs_per_s"] > 0
    e = d["eip7594"]
    assert e["compute_cells_and_kzg_proofs"]["blobs_per_s_per_gpu"] > 0
    cv = e["verify_cell_kzg_proof_batch"]
    assert cv["cells_total"] == n * cell_blobs * 128 and cv["cells_per_s"] > 0 and f"sharded x{n}" in cv["parallelism"]
    assert cv["strong_scaling"]["global_batch"] == 65536 and cv["strong_scaling"]["cells_per_s"] > 0
    assert e["recover_cells_and_kzg_proofs"]["blobs_per_s"] > 0 and f"x{n}" in e["recover_cells_and_kzg_proofs"]["parallelism"]
    assert "cpu_baseline" not in d                      # reported at N = 1 only
    return d


def test_bench_starts_its_own_ranks():
    _check(_run({}, [sys.executable]))


def test_bench_under_torch_distributed_run():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    _check(_run({}, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)]))


def test_eight_ranks_share_the_cpus():
    """`bench.py --gpus 8` as the driver runs it on an 8-GPU node, here with all eight ranks on the one GPU over gloo: completes, and the
    ranks' host pools together hold no more threads than the CPUs the box grants (two per rank at least: caller + one worker)"""
    d = _check(_run({}, [sys.executable], gpus=8, sizes=("16", "16", "2")), n=8, cell_blobs=2)
    m = d["multi_gpu"]
    assert m["distinct_gpus"] == 1                                      # they do share one GPU here, and the line says so
    assert m["host_threads_total"] <= max(16, m["usable_cpus"]), m
    assert all(g["host_threads"] >= 2 for g in m["rank_gpus"])


def test_exchange_on_rccl_with_one_rank():
    """BENCH_FORCE_DIST: world size 1, process group on the "nccl" backend (= RCCL): the sharded verifiers' broadcast / all_gather run on
    device tensors through RCCL"""
    d = _run({"BENCH_DIST_BACKEND": "", "BENCH_SHARE_GPU": "", "BENCH_FORCE_DIST": "1"}, [sys.executable], gpus=1)
    assert d["n_gpus"] == 1 and d["value"] > 0
    m = d["multi_gpu"]
    assert m["backend"].startswith("nccl") and m["ranks"] == 1 and m["distinct_gpus"] == 1 and m["exchange_us"] > 0
    v = d["verify_blob_kzg_proof_batch"]
    assert "sharded x1" in v["single_call"]["parallelism"] and v["strong_scaling"]["per_rank"] == 8192
    assert d["eip7594"]["verify_cell_kzg_proof_batch"]["strong_scaling"]["per_rank"] == 65536
    # ... and the same sharded call with the collectives issued by the LIBRARY (mkzg_comm_init + mkzg_verify_blob_kzg_proof_batch_exchange_device)
    le = v["library_exchange"]
    assert "error" not in le and le["blobs_per_s"] > 0 and le["exchange_us_per_call_this_rank"] > 0, le
    assert d["verify_metric"]["value"] == v["value"] and d["verify_metric"]["n_gpus"] == 1


def test_in_process_multi_device_leg_on_one_gpu():
    """bench.py's multi_gpu.in_process leg (ONE context over all the job's devices, the form a single Go process uses) has never met a
    node with two GPUs: run its child process with devices = [0, 0] -- two peers on the one GPU -- at small sizes, so that the day an
    8-GPU node appears the leg is not running for the first time"""
    env = dict(os.environ, BENCH_INPROC_DEVICES="0,0", BENCH_INPROC_SIZES="32,64", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--in-process-leg", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    got = [ln for ln in r.stdout.splitlines() if ln.startswith("IN_PROCESS ")]
    assert got, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(got[-1][len("IN_PROCESS "):])
    assert "error" not in d, d
    assert d["devices"] == 2 and d["blob_to_kzg_commitment_batch"]["blobs"] == 64 and d["blob_to_kzg_commitment_batch"]["blobs_per_s"] > 0
    assert d["verify_blob_kzg_proof_batch"]["blobs"] == 128 and d["verify_blob_kzg_proof_batch"]["blobs_per_s"] > 0 and d["tables_on_own_device"] == 1
