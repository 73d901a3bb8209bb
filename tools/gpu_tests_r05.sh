#!/bin/bash
# full GPU suite on the final build; the summary line is kept whatever else the run prints
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider < /dev/null > gpurun_out/t17_full.txt 2>&1
echo "exit $?" >> gpurun_out/t17_full.txt
grep -E "passed|failed|error" gpurun_out/t17_full.txt | tail -5
