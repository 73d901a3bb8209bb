"""The N > 1 path of bench.py on a box with ONE GPU: `python bench.py --gpus 2` (no launcher: bench.py starts its own ranks, the parent
never touches the GPU) with both ranks on GPU 0 talking over gloo (bench.py's BENCH_SHARE_GPU / BENCH_DIST_BACKEND test aids).  Checked:
the run completes, rank 0 prints ONE JSON line with n_gpus = 2 and every section -- weak-scaled commitments, the sharded blob batch
verifier, the sharded cell batch verifier (one global batch, broadcast + all_gather) and the blob-sharded recovery."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, launcher):
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    env.pop("WORLD_SIZE", None)
    args = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "32", "--direct-bits", "12", "--verify-batch", "32", "--verify-steps", "1", "--cell-blobs", "3"]
    cmd = launcher + [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _check(d):
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    v = d["verify_blob_kzg_proof_batch"]
    assert v["value"] > 0 and "sharded x2" in v["single_call"]["parallelism"]
    e = d["eip7594"]
    assert e["compute_cells_and_kzg_proofs"]["blobs_per_s_per_gpu"] > 0
    cv = e["verify_cell_kzg_proof_batch"]
    assert cv["cells_total"] == 2 * 3 * 128 and cv["cells_per_s"] > 0 and "sharded x2" in cv["parallelism"]
    assert e["recover_cells_and_kzg_proofs"]["blobs_per_s"] > 0 and "x2" in e["recover_cells_and_kzg_proofs"]["parallelism"]
    assert "cpu_baseline" not in d                      # reported at N = 1 only


def test_bench_starts_its_own_ranks():
    _check(_run({}, [sys.executable]))


def test_bench_under_torch_distributed_run():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    _check(_run({}, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)]))
