set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_eip4844.py tests/test_gpu_eip7594.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r05/t1.txt
timeout 900 python3 tools/ab_verify.py build_ab/libmi355kzg_r04.so go-eth-kzg_amd/libmi355kzg.so 2 > gpurun_out/r05/ab_verify1.txt 2>&1
for rep in 1 2; do for lib in build_ab/libmi355kzg_r04.so go-eth-kzg_amd/libmi355kzg.so; do
  echo "== $lib" >> gpurun_out/r05/ab_cells1.txt
  MI355KZG_LIB=$PWD/$lib CV_CALLERS=1,4,12 timeout 600 python3 tools/bench_cellverify_inflight.py >> gpurun_out/r05/ab_cells1.txt 2>&1
done; done
cat gpurun_out/r05/t1.txt gpurun_out/r05/ab_verify1.txt gpurun_out/r05/ab_cells1.txt
