set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
NEW=$PWD/go-eth-kzg_amd/libmi355kzg.so
timeout 300 python -m pytest tests/test_gpu_combine.py -m gpu -x -q 2>&1 | tail -4 > $O/t7.txt; cat $O/t7.txt
for seg in 1 0; do echo "== fk20_segments $seg" >> $O/ab_fk20_7.txt
  FK20_SEGMENTS=$seg SWEEP_NB=192 SWEEP_SIZES=65,96,128,160,192 timeout 300 python3 tools/sweep_cells.py >> $O/ab_fk20_7.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_fk20_7.txt | cut -c1-200
for v in "0 2" "1 2" "1 3" "1 4" "1 6"; do set -- $v
  echo "== combine $1 running $2" >> $O/ab_single7.txt
  COMBINE=$1 MKZG_COMBINE_RUNNING=$2 CALLERS=8,32,64 timeout 300 python3 tools/bench_single_callers.py >> $O/ab_single7.txt 2>&1 < /dev/null
done
echo "== combine 1 running 3 wrong every 50" >> $O/ab_single7.txt
COMBINE=1 MKZG_COMBINE_RUNNING=3 WRONG_EVERY=50 KINDS=verify_blob_kzg_proof CALLERS=32 timeout 300 python3 tools/bench_single_callers.py >> $O/ab_single7.txt 2>&1 < /dev/null
grep -v amdgpu.ids $O/ab_single7.txt | cut -c1-300
