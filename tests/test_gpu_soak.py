"""A short run of tools/soak.py as a test: 16 threads hammer ONE shared context for 20 s with random entry points, sizes, corruptions, error
inputs and option flips, and every byte, verdict and error code must equal what the oracle computed beforehand (the reference's Context
is shared by goroutines: verify.go:152-169, examples_test.go:11).  The long runs of the round are in profiles/r06_soak.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_soak_20_seconds():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "20", "16", "7"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "soak ok" in r.stdout, r.stdout[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("soak:")][0]
    counts = dict(kv.split("=") for kv in line.split("  ")[-1].split())
    # every kind of call was exercised
    for kind in ("commit1", "commitN", "commit_dev", "proof1", "kzgproof", "verify1", "verifyN", "verify_dev", "verifykzg", "cells1", "cellsonly", "cells_dev", "cellverify", "recover", "errors"):
        assert int(counts.get(kind, 0)) > 0, (kind, line)
