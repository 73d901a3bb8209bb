#!/usr/bin/env python3
"""tools/round_gaps.py DIR -- the deep rounds of a rocprofv3 kernel trace (tools/trace.py verify_inflight under --kernel-trace --output-format csv):
per hardware queue the sequence hash -> evaluation -> window sums, with the gaps between them and between a round's end and the queue's next hash."""
import csv
import glob
import os
import statistics
import sys


def main():
    d = sys.argv[1]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kzg::", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
    rows.sort()
    byq = {}
    for s, e, n, q in rows:
        if n.startswith(("k_challenge_sha256_jobs", "k_blob_eval_tree_jobs", "k_vmsm_window_calls")):
            byq.setdefault(q, []).append((s, e, n))
    g1, g2, g3, dh, de, dv = [], [], [], [], [], []
    for q, seq in byq.items():
        for i in range(len(seq) - 2):
            a, b, c = seq[i], seq[i + 1], seq[i + 2]
            if a[2].startswith("k_challenge") and b[2].startswith("k_blob_eval") and c[2].startswith("k_vmsm"):
                g1.append((b[0] - a[1]) / 1e6)
                g2.append((c[0] - b[1]) / 1e6)
                dh.append((a[1] - a[0]) / 1e6)
                de.append((b[1] - b[0]) / 1e6)
                dv.append((c[1] - c[0]) / 1e6)
                if i + 3 < len(seq) and seq[i + 3][2].startswith("k_challenge"):
                    g3.append((seq[i + 3][0] - c[1]) / 1e6)
    med = lambda v: (statistics.median(v) if v else float("nan"))
    print(f"queues with rounds: {len(byq)}; rounds: {len(dh)}")
    print(f"hash {med(dh):.3f} ms | gap {med(g1):.3f} | evaluation {med(de):.3f} | gap {med(g2):.3f} | window sums {med(dv):.3f} | gap to the queue's next hash {med(g3):.3f}   (medians)")
    if g3:
        g3.sort()
        print("gap to the next hash: p10 %.3f p50 %.3f p90 %.3f ms" % (g3[len(g3) // 10], g3[len(g3) // 2], g3[9 * len(g3) // 10]))


if __name__ == "__main__":
    main()
