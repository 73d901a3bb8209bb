#!/bin/bash
# tools/ab.sh "bench args" lib1.so lib2.so ... -- A/B of library builds on the GPU box (MI355KZG_LIB override)
ARGS=$1; shift
for lib in "$@"; do
  for rep in 1 2; do
    MI355KZG_LIB=$PWD/$lib python bench.py --no-cpu-baseline --verify-batch 0 $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.load(sys.stdin); print('$lib', d['value'], d['ms_per_step'], {k:v for k,v in d['roofline']['kernels_ms_per_step'].items() if v})"
  done
done
