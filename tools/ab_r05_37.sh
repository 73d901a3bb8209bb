# cell verifier with host cells: the interpolation chunk by chunk (MKZG_CV_INTERP_CHUNKS=1; second attempt: uploads on a copy stream of their own, kernels behind events) against one launch behind the last upload (0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
MKZG_CV_INTERP_CHUNKS=1 timeout 900 python -m pytest tests/test_gpu_eip7594.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "cell or recover or 7594" < /dev/null > $O/t37_parity.txt 2>&1
grep -E "passed|failed|error" $O/t37_parity.txt | tail -3
for v in 0 1 0 1 0 1; do
  echo "== MKZG_CV_INTERP_CHUNKS=$v" >> $O/ab_chunks37.txt
  MKZG_CV_INTERP_CHUNKS=$v CV_REPS=30 CV_CALLERS=1,4,12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_chunks37.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_chunks37.txt | cut -c1-250
