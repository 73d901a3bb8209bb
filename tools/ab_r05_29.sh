# cell verifier, host cells: upload (MKZG_CV_ZERO_COPY=0) against the interpolation kernel reading the pinned staging area itself (1); every caller its own cells
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
MKZG_CV_ZERO_COPY=1 timeout 900 python -m pytest tests/test_gpu_eip7594.py -m gpu -q -x -p no:cacheprovider < /dev/null > $O/t29_parity.txt 2>&1
grep -E "passed|failed|error" $O/t29_parity.txt | tail -3
for v in 0 1 0 1 0 1; do
  echo "== MKZG_CV_ZERO_COPY=$v" >> $O/ab_zc29.txt
  MKZG_CV_ZERO_COPY=$v CV_CALLERS=1,4,12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_zc29.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_zc29.txt | cut -c1-250
