#!/usr/bin/env python3
"""tools/timeline.py TRACE_DIR [t_from_ms t_to_ms] -- merge rocprofv3 kernel_trace / memory_copy_trace CSVs into one time-ordered listing
(start and duration in ms relative to the first record; queue / stream ids) to read a pipeline's overlap off the page."""
import csv
import glob
import os
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kzg::", "")[:40], r.get("Queue_Id", "?"), r.get("Grid_Size", "")))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "copy")[:40], "-", ""))
rows.sort()
if not rows:
    sys.exit("no trace rows")
t0 = rows[0][0]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
for s, e, kind, name, q, g in rows:
    a = (s - t0) / 1e6
    if lo <= a <= hi:
        print(f"{a:10.3f} +{(e - s) / 1e6:8.3f} ms  {kind} q={q:>3} grid={g:>8}  {name}")
