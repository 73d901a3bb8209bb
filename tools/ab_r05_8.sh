set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/t8.txt; cat $O/t8.txt
R04=$PWD/build_ab/libmi355kzg_r04.so; NEW=$PWD/go-eth-kzg_amd/libmi355kzg.so
for v in "r04 $R04" "new $NEW" "r04 $R04" "new $NEW"; do set -- $v
  echo "== $1" >> $O/ab_verify8.txt
  MI355KZG_LIB=$2 timeout 300 python3 tools/ab_verify.py --one >> $O/ab_verify8.txt 2>&1 < /dev/null
  echo "== $1" >> $O/ab_cells8.txt
  MI355KZG_LIB=$2 CV_CALLERS=1,12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_cells8.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_verify8.txt $O/ab_cells8.txt | cut -c1-250
