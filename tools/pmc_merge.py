#!/usr/bin/env python3
"""tools/pmc_merge.py OUT.json [@EARLIER_MERGED.json] commit=A_traffic.json verify=B_traffic.json eip7594=C_traffic.json verify_tp=D_traffic.json
-- fold the per-leg PMC summaries
(tools/pmc_collect.sh + pmc_summarize.py, one bench.py configuration per leg so that "the largest launches of a kernel" are that leg's)
into the one file bench.py reads: {"sources": hashes of the kernel sources, "legs": {leg: {kernel: figures}}}."""
import json
import sys

out = {"legs": {}, "sources": {}}
carried = {}                                                  # leg -> (family, hash) of legs taken over from an earlier merged file
fresh = {}                                                    # family -> hash of the legs supplied now
for arg in sys.argv[2:]:
    if arg.startswith("@"):                                   # carry the legs of an earlier merged file over (kept only while their family's kernel sources are unchanged)
        j = json.load(open(arg[1:]))
        for leg, kern in j["legs"].items():
            fam = leg if leg in j["sources"] else "verify"
            out["legs"][leg] = kern
            carried[leg] = (fam, j["sources"][fam])
        out.setdefault("correction", j["correction"])
        continue
    leg, path = arg.split("=")
    j = json.load(open(path))
    fam = leg if leg in j["sources"] else "verify"           # the source family whose hash guards this leg's figures in bench.py
    assert fresh.setdefault(fam, j["sources"][fam]) == j["sources"][fam], "legs of one family were collected on different kernel sources"
    out.setdefault("correction", j["correction"])
    out["legs"][leg] = {k: v for k, v in j["kernels"].items() if k.startswith("k_")}
    carried.pop(leg, None)
for leg, (fam, h) in carried.items():
    if fam in fresh and fresh[fam] != h:
        print(f"dropping carried leg {leg}: its family {fam} was re-collected on other kernel sources")
        del out["legs"][leg]
    else:
        fresh.setdefault(fam, h)
out["sources"] = fresh
out["source"] = ("rocprofv3 --pmc passes (tools/pmc_collect.sh; one counter group per pass, never combined with tracing) over short bench.py runs, one per leg: "
                 "commit = batch 256 with the library's defaults (15-bit direct table) + the bucket-method cross-check; verify = --verify-batch 1024 alone; "
                 "eip7594 = --cell-blobs 64 alone; verify_tp = --verify-batch 1024 --verify-shape 1 (the throughput shape's kernels, one call at a time).  "
                 "Per kernel: mean over its launches with the largest grid (PMC_GRIDS overrides)")
json.dump(out, open(sys.argv[1], "w"), indent=1)
print({leg: len(k) for leg, k in out["legs"].items()})
