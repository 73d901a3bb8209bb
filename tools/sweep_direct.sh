#!/bin/bash
# tools/sweep_direct.sh -- bench.py over commit_direct_bits / stream counts (GPU box)
for cfg in ${SWEEP:-15:2 15:1 14:2 13:2}; do
  set -- ${cfg/:/ }
  t0=$(date +%s.%N)
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --verify-batch 0 --direct-bits $1 --streams $2 $EXTRA 2>/tmp/sweep_err.txt | tail -1 > /tmp/sweep_line.json
  t1=$(date +%s.%N)
  python - "$1" "$2" "$t0" "$t1" <<'PY'
import json, sys
d = json.load(open('/tmp/sweep_line.json'))
print("direct_bits=%s streams=%s  %.0f blobs/s  %.3f ms/step  kernels=%s  wall=%.1fs" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"],
      {k: v for k, v in d["roofline"]["kernels_ms_per_step"].items() if v}, float(sys.argv[4]) - float(sys.argv[3])))
PY
done
