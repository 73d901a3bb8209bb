set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
R04=$PWD/build_ab/libmi355kzg_r04.so; NEW=$PWD/go-eth-kzg_amd/libmi355kzg.so
timeout 900 python -m pytest tests/test_gpu_eip7594.py tests/test_gpu_eip4844.py tests/test_gpu_fullsize.py tests/test_gpu_exchange.py -m gpu -x -q 2>&1 | tail -6 > $O/t11.txt; cat $O/t11.txt
MI355KZG_LIB=$NEW timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/lone11 -o lone -- python3 tools/trace.py lone 5 > $O/lone11.txt 2>&1 < /dev/null
python3 tools/kstats.py $O/lone11 k_g1_validate k_vmsm k_cell k_blob_eval >> $O/lone11.txt 2>&1 < /dev/null; rm -rf $O/lone11
grep -v "rocprofv3\|SQLite\|amdgpu.ids\|generateRocpd\|Opened result\|HSA version" $O/lone11.txt
for v in "2" "3" "4" "2" "3"; do
  echo "== hash in flight $v" >> $O/ab_verify11.txt
  MI355KZG_LIB=$NEW MKZG_HASH_IN_FLIGHT=$v timeout 300 python3 tools/ab_verify.py --one >> $O/ab_verify11.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_verify11.txt | cut -c1-250
for v in "r04 $R04" "new $NEW" "new $NEW"; do set -- $v
  echo "== $1" >> $O/ab_cells11.txt
  MI355KZG_LIB=$2 CV_CALLERS=1,12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_cells11.txt 2>&1 < /dev/null
  MI355KZG_LIB=$2 CV_DEVICE=1 CV_CALLERS=1,12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_cells11.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_cells11.txt | cut -c1-250
