"""The N > 1 path of bench.py (weak-scaled commitments + the sharded batch verifier with broadcast / all_gather) on a box with ONE
GPU: two ranks share GPU 0 and talk over gloo (bench.py's BENCH_SHARE_GPU / BENCH_DIST_BACKEND test aids).  What is checked: the
run completes, rank 0 prints one JSON line with n_gpus = 2 and both sections, and the sharded verifier accepted its batch."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "32", "--direct-bits", "12", "--verify-batch", "32", "--verify-steps", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    v = d["verify_blob_kzg_proof_batch"]
    assert v["value"] > 0 and "sharded x2" in v["parallelism"]
    assert "cpu_baseline" not in d                      # reported at N = 1 only
