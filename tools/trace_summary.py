#!/usr/bin/env python3
"""tools/trace_summary.py KERNEL_TRACE_CSV [KERNEL_SUBSTR] -- per-launch durations of one kernel from a rocprofv3 --kernel-trace CSV, grouped
by grid size and hardware queue (rocprofv3's --stats mixes every launch of a kernel: probe launches on one blob, the overlapped
two-stream launches of the timed region and the isolated ones)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_commit_direct_acc"
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
