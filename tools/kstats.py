#!/usr/bin/env python3
"""tools/kstats.py DIR [SUBSTR ...] -- per-kernel launch count and average / min / max duration (ms) from the kernel trace rocprofv3 left
under DIR (`--kernel-trace --output-format csv`), largest total first; SUBSTR filters kernel names."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, subs = sys.argv[1], sys.argv[2:]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit(f"no *kernel_trace.csv under {d}")
    agg = defaultdict(list)
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kzg::", "")
                if subs and not any(s in name for s in subs):
                    continue
                agg[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for name, ds in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{name[:48]:48s} n={len(ds):5d} avg {sum(ds) / len(ds):8.3f} min {min(ds):8.3f} max {max(ds):8.3f} ms")


if __name__ == "__main__":
    main()
