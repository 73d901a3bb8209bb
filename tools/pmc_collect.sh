#!/bin/bash
# tools/pmc_collect.sh ROUND_TAG [bench args...]  -- run on the GPU box (gpurun).
# Separate rocprofv3 --pmc passes (never combined with tracing) over a short bench.py run, then
# tools/pmc_summarize.py folds them into gpurun_out/<tag>_pmc_summary.csv and gpurun_out/<tag>_traffic.json.
# The three legs of profiles/r03_traffic.json (each leg's kernels must see that leg's launches as their largest, so the sustained and
# big-batch sections stay out):
#   bash tools/pmc_collect.sh r03c                                                          (commit: batch 256, defaults + bucket-method cross-check)
#   bash tools/pmc_collect.sh r03v --verify-batch 1024 --verify-inflight 1 --no-strong     (verify: single calls only)
#   bash tools/pmc_collect.sh r03e --cell-blobs 64                                          (eip7594; --no-big-cells is always passed)
#   PMC_GRIDS="k_g1_validate=4096" bash tools/pmc_collect.sh r04t --verify-batch 1024 --verify-inflight 1 --no-strong --verify-shape 1
#                                                                                           (verify_tp: the throughput shape's kernels, one call at a time)
# then, on the build host: python3 tools/pmc_merge.py profiles/r03_traffic.json commit=gpurun_out/r03c_traffic.json verify=... eip7594=...
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# one small counter group per pass: FETCH_SIZE + WRITE_SIZE in one pass exceed the hardware's counters (error 38, and
# rocprofv3 then hangs), hence also the per-pass timeout
PASSES=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAVES" "VALUBusy" "VALUUtilization")
if [ -n "${PMC_PASSES:-}" ]; then IFS=";" read -r -a PASSES <<< "$PMC_PASSES"; fi
i=0
for p in "${PASSES[@]}"; do
  timeout -k 5 ${PASS_TIMEOUT:-120} rocprofv3 --pmc $p -d $OUT/pass$i --output-format csv -- python3 bench.py --full-line --steps 2 --warmup 1 --no-cpu-baseline --no-drop-in --cell-blobs 0 --verify-batch 0 --streams 1 --no-big-cells "$@" > $OUT/pass$i.log 2>&1 || { echo "pass $i ($p) failed"; grep -m2 -i "error\|exceeds" $OUT/pass$i.log; }
  i=$((i+1))
done
python3 tools/pmc_summarize.py $OUT gpurun_out/${TAG}
