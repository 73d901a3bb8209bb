# 7-bit Booth windows (two windows per workgroup, second wavefront exits after the scan) against 8-bit windows in the throughput shapes:
# parity tests first, then alternating runs on one box (MKZG_VMSM_BITS=8 forces the 8-bit form everywhere, 7 the 7-bit form)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_eip4844.py tests/test_gpu_eip7594.py tests/test_gpu_fullsize.py tests/test_gpu_ragged_batches.py -m gpu -q -x -p no:cacheprovider < /dev/null > $O/t16_parity.txt 2>&1
tail -3 $O/t16_parity.txt
MKZG_VMSM_BITS=7 timeout 900 python -m pytest tests/test_gpu_eip4844.py tests/test_gpu_eip7594.py tests/test_gpu_ragged_batches.py -m gpu -q -x -p no:cacheprovider < /dev/null > $O/t16_parity7.txt 2>&1
tail -3 $O/t16_parity7.txt
for v in 8 0 8 0 8 0; do
  echo "== MKZG_VMSM_BITS=$v" >> $O/ab_bits16.txt
  MKZG_VMSM_BITS=$v timeout 300 python3 tools/ab_verify.py --one >> $O/ab_bits16.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_bits16.txt | cut -c1-250
for v in 8 0 8 0; do
  echo "== MKZG_VMSM_BITS=$v" >> $O/ab_bits16c.txt
  MKZG_VMSM_BITS=$v CV_CALLERS=1,12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_bits16c.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_bits16c.txt | cut -c1-250
