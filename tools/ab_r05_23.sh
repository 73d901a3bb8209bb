# hardware queues per priority level: 3 / 4 / 5 / 6 (GPU_MAX_HW_QUEUES; the library's streams use three priority levels = three pools of queues)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
for q in 4 5 6 3 4 5 6; do
  echo "== GPU_MAX_HW_QUEUES=$q" >> $O/ab_queues23.txt
  GPU_MAX_HW_QUEUES=$q timeout 300 python3 tools/ab_verify.py --one >> $O/ab_queues23.txt 2>&1 < /dev/null
  GPU_MAX_HW_QUEUES=$q CV_CALLERS=12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_queues23.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_queues23.txt | cut -c1-250
