# 7-bit windows with the top window's buckets on eight lanes each: parity with the width forced, isolated durations, A/B of the sustained legs
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
MKZG_VMSM_BITS=7 timeout 900 python -m pytest tests/test_gpu_eip4844.py tests/test_gpu_eip7594.py tests/test_gpu_ragged_batches.py -m gpu -q -x -p no:cacheprovider < /dev/null > $O/t18_parity7.txt 2>&1
tail -3 $O/t18_parity7.txt
cd /tmp
for v in 8 7; do
  export MKZG_VMSM_BITS=$v
  rm -rf /tmp/tr$v
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$v -- python3 $GRAFT_REPO_ROOT/tools/trace.py lone 5 > $GRAFT_REPO_ROOT/$O/lone18_$v.txt 2>&1 < /dev/null
  python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/tr$v vmsm >> $GRAFT_REPO_ROOT/$O/lone18_$v.txt 2>&1
  grep -E "^verify_shape|^lone|^k_" $GRAFT_REPO_ROOT/$O/lone18_$v.txt
done
unset MKZG_VMSM_BITS
cd $GRAFT_REPO_ROOT
for v in 8 0 8 0 8 0; do
  echo "== MKZG_VMSM_BITS=$v" >> $O/ab_bits18.txt
  MKZG_VMSM_BITS=$v timeout 300 python3 tools/ab_verify.py --one >> $O/ab_bits18.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_bits18.txt | cut -c1-250
for v in 8 0 8 0; do
  echo "== MKZG_VMSM_BITS=$v" >> $O/ab_bits18c.txt
  MKZG_VMSM_BITS=$v CV_CALLERS=1,12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_bits18c.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_bits18c.txt | cut -c1-250
