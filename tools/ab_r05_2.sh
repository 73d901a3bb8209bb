set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_eip4844.py tests/test_gpu_eip7594.py tests/test_gpu_commit.py -m gpu -x -q 2>&1 | tail -5 > $O/t2.txt
R04=$PWD/build_ab/libmi355kzg_r04.so; NEW=$PWD/go-eth-kzg_amd/libmi355kzg.so
for v in "r04 $R04 0" "new0 $NEW 0" "new40 $NEW 40000"; do set -- $v
  MI355KZG_LIB=$2 MKZG_VAL_LDS_PAD=$3 timeout 600 rocprofv3 --kernel-trace --stats -d $O/lone_$1 -o lone -- python3 tools/trace.py lone 5 > $O/lone_$1.txt 2>&1
  f=$(find $O/lone_$1 -name '*kernel_stats.csv' | head -1); head -16 $f | cut -d, -f1-5 | cut -c1-150 >> $O/lone_$1.txt
done
for v in "r04 $R04 0 4" "new0 $NEW 0 4" "new40 $NEW 40000 4" "r04q8 $R04 0 8" "new0q8 $NEW 0 8"; do set -- $v
  echo "== $1" >> $O/ab_verify2.txt
  MI355KZG_LIB=$2 MKZG_VAL_LDS_PAD=$3 GPU_MAX_HW_QUEUES=$4 timeout 600 python3 tools/ab_verify.py --one >> $O/ab_verify2.txt 2>&1
done
for v in "r04 $R04 0" "new0 $NEW 0" "new40 $NEW 40000"; do set -- $v
  echo "== $1" >> $O/ab_cells2.txt
  MI355KZG_LIB=$2 MKZG_VAL_LDS_PAD=$3 CV_CALLERS=1,12 timeout 600 python3 tools/bench_cellverify_inflight.py >> $O/ab_cells2.txt 2>&1
done
rm -rf $O/lone_*/
cat $O/t2.txt $O/lone_*.txt $O/ab_verify2.txt $O/ab_cells2.txt | grep -v amdgpu.ids
