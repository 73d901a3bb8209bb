#!/bin/bash
# tools/ab_verify.sh lib1.so lib2.so ... -- A/B of library builds on the batch verifier (config 3) and the cell verifier (config 5a)
for lib in "$@"; do
  for rep in 1 2; do
    MI355KZG_LIB=$PWD/$lib python bench.py --no-cpu-baseline --steps 3 --verify-steps 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.load(sys.stdin)['verify_blob_kzg_proof_batch']; print('$lib', d['value'], d['ms_per_call'], d['kernels_ms_per_call'])"
  done
done
