"""bench.py's stdout line (CPU only): the compact line built from a whole record -- here the committed record of a real run -- carries the
contract's keys, the `roofline` and `cpu_baseline` objects and one figure per leg, and stays far below the size at which round 5's line
(21.6 KB) came back from the driver unparsed (round 4's 18.2 KB line had parsed)."""
import glob
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def test_compact_line_of_committed_records():
    m = _bench()
    records = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_default.json")))
    assert records
    for path in records:
        with open(path) as fh:
            txt = fh.read().strip().splitlines()[-1]
        rec = json.loads(txt)
        if "legs" in rec:           # a compact line committed as such
            rec_line = txt
        else:
            rec_line = json.dumps(m.compact_line(rec, "gpurun_out/bench_detail_n1.json"), separators=(",", ":"))
        assert len(rec_line) < 6000, (path, len(rec_line))
        c = json.loads(rec_line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in c, (path, k)
        assert c["higher_is_better"] is True and c["scaling"] == "weak" and c["vs_baseline"] is None and "workload" in c["config"]
        rf = c["roofline"]
        assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5
        if c["n_gpus"] == 1 and "cpu_baseline" in rec:
            cb = c["cpu_baseline"]
            assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]


def test_dig():
    m = _bench()
    assert m.dig({"a": {"b": {"c": 3}}}, "a.b.c") == 3 and m.dig({"a": 1}, "a.b") is None and m.dig({}, "x", 7) == 7
