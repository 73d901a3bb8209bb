#!/bin/bash
# tools/pmc_collect.sh ROUND_TAG [bench args...]  -- run on the GPU box (gpurun).
# Separate rocprofv3 --pmc passes (never combined with tracing) over a short bench.py run, then
# tools/pmc_summarize.py folds them into gpurun_out/<tag>_pmc_summary.csv and gpurun_out/<tag>_traffic.json.
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
  timeout -k 5 ${PASS_TIMEOUT:-120} rocprofv3 --pmc $p -d $OUT/pass$i --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-drop-in --cell-blobs 0 --verify-batch 0 --streams 1 --no-big-cells "$@" > $OUT/pass$i.log 2>&1 || { echo "pass $i ($p) failed"; grep -m2 -i "error\|exceeds" $OUT/pass$i.log; }
  i=$((i+1))
done
python3 tools/pmc_summarize.py $OUT gpurun_out/${TAG}
