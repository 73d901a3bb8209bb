set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 120 ./tools/ubench_placement 129 > $O/placement.txt 2>&1 < /dev/null
timeout 600 python -m pytest tests/test_gpu_combine.py tests/test_gpu_eip4844.py tests/test_gpu_eip7594.py -m gpu -x -q 2>&1 | tail -8 > $O/t5.txt
R04=$PWD/build_ab/libmi355kzg_r04.so; NEW=$PWD/go-eth-kzg_amd/libmi355kzg.so; VM=$PWD/build_ab/libmi355kzg_vm386.so
for v in "new $NEW"; do set -- $v
  MI355KZG_LIB=$2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/lone5_$1 -o lone -- python3 tools/trace.py lone 5 > $O/lone5_$1.txt 2>&1 < /dev/null
  python3 tools/kstats.py $O/lone5_$1 k_g1_validate k_vmsm k_cell k_challenge k_blob_eval >> $O/lone5_$1.txt 2>&1 < /dev/null
  rm -rf $O/lone5_$1
done
for v in "r04 $R04" "new $NEW" "vm386 $VM" "r04 $R04" "new $NEW" "vm386 $VM"; do set -- $v
  echo "== $1" >> $O/ab_verify5.txt
  MI355KZG_LIB=$2 timeout 300 python3 tools/ab_verify.py --one >> $O/ab_verify5.txt 2>&1 < /dev/null
done
for v in "r04 $R04" "new $NEW" "r04 $R04" "new $NEW"; do set -- $v
  echo "== $1" >> $O/ab_cells5.txt
  MI355KZG_LIB=$2 CV_CALLERS=1,12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_cells5.txt 2>&1 < /dev/null
  MI355KZG_LIB=$2 CV_DEVICE=1 CV_CALLERS=1,12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_cells5.txt 2>&1 < /dev/null
done
cat $O/placement.txt $O/t5.txt; grep -v "rocprofv3\|SQLite\|amdgpu.ids\|generateRocpd\|Opened result\|HSA version" $O/lone5_*.txt; grep -v amdgpu.ids $O/ab_verify5.txt $O/ab_cells5.txt
