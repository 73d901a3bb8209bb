#!/usr/bin/env python3
"""tools/trace_busy.py TRACE_DIR [gap_ms] -- for the LAST burst of kernels in a rocprofv3 kernel trace (bursts are separated by gaps of
more than gap_ms, default 20): wall time, time with at least one kernel running, mean number of kernels running, and per kernel name the
launches, the summed duration and its share of the wall time."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 20e6
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kzg::", "")[:40]))
rows.sort()
bursts, cur, end = [], [rows[0]], rows[0][1]
for r in rows[1:]:
    if r[0] - end > gap:
        bursts.append(cur)
        cur = []
    cur.append(r)
    end = max(end, r[1])
bursts.append(cur)
b = max(bursts[-3:], key=len)
t0, t1 = b[0][0], max(r[1] for r in b)
ev = sorted([(r[0], 1) for r in b] + [(r[1], -1) for r in b])
busy = area = 0
lvl, last = 0, t0
for t, dlt in ev:
    if lvl > 0:
        busy += t - last
    area += lvl * (t - last)
    lvl += dlt
    last = t
wall = t1 - t0
print(f"burst: {len(b)} launches, wall {wall / 1e6:.3f} ms, some kernel running {100 * busy / wall:.1f} % of it, mean kernels in flight {area / wall:.2f}")
agg = defaultdict(lambda: [0, 0])
for s, e, n in b:
    agg[n][0] += 1
    agg[n][1] += e - s
for n, (c, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:42s} {c:5d} launches  sum {tot / 1e6:9.3f} ms  avg {tot / c / 1e6:7.3f} ms  {100 * tot / wall:6.1f} % of wall")
