# hardware queues of the HIP runtime (GPU_MAX_HW_QUEUES; default 4) with the round-5 kernels that reserve their SIMDs: both sustained verifier legs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
for q in 4 8 16 4 2 8 24; do
  echo "== GPU_MAX_HW_QUEUES=$q" >> $O/ab_queues21.txt
  GPU_MAX_HW_QUEUES=$q timeout 300 python3 tools/ab_verify.py --one >> $O/ab_queues21.txt 2>&1 < /dev/null
  GPU_MAX_HW_QUEUES=$q CV_CALLERS=12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_queues21.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_queues21.txt | cut -c1-250
