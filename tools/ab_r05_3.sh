set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/t3.txt
R04=$PWD/build_ab/libmi355kzg_r04.so; NEW=$PWD/go-eth-kzg_amd/libmi355kzg.so
for v in "new0 $NEW 0" "new40 $NEW 40000"; do set -- $v
  MI355KZG_LIB=$2 MKZG_VAL_LDS_PAD=$3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/lone_$1 -o lone -- python3 tools/trace.py lone 5 > $O/lone_$1.txt 2>&1 < /dev/null
  python3 tools/kstats.py $O/lone_$1 k_g1_validate k_vmsm k_cell k_challenge k_blob_eval >> $O/lone_$1.txt 2>&1 < /dev/null
  rm -rf $O/lone_$1
done
for v in "r04 $R04 0 4" "new0 $NEW 0 4" "new40 $NEW 40000 4" "new0q8 $NEW 0 8" "r04 $R04 0 4" "new0 $NEW 0 4"; do set -- $v
  echo "== $1" >> $O/ab_verify3.txt
  MI355KZG_LIB=$2 MKZG_VAL_LDS_PAD=$3 GPU_MAX_HW_QUEUES=$4 timeout 300 python3 tools/ab_verify.py --one >> $O/ab_verify3.txt 2>&1 < /dev/null
done
for v in "r04 $R04 0" "new0 $NEW 0" "new40 $NEW 40000" "r04 $R04 0" "new0 $NEW 0"; do set -- $v
  echo "== $1" >> $O/ab_cells3.txt
  MI355KZG_LIB=$2 MKZG_VAL_LDS_PAD=$3 CV_CALLERS=1,12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_cells3.txt 2>&1 < /dev/null
done
cat $O/t3.txt; grep -v "rocprofv3\|SQLite\|amdgpu.ids" $O/lone_new*.txt; grep -v amdgpu.ids $O/ab_verify3.txt $O/ab_cells3.txt
