# slices of the cell verifier's 8192-point window sums when calls share the GPU: 2048 (four slices, as alone) / 2731 (three) / 4096 (two) / 8192 (one) points each
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
for v in 2048 4096 2731 8192 2048 4096 2731 8192; do
  echo "== MKZG_CV_SLICE_POINTS=$v" >> $O/ab_slices22.txt
  MKZG_CV_SLICE_POINTS=$v CV_CALLERS=4,12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_slices22.txt 2>&1 < /dev/null
  MKZG_CV_SLICE_POINTS=$v CV_DEVICE=1 CV_CALLERS=12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_slices22.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_slices22.txt | cut -c1-250
