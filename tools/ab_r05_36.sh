# lone host-pointer VerifyCellKZGProofBatch(8192) on the experiment build: window width forced (MKZG_VMSM_BITS=7 / 8), alternating; and with the cells in HBM
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
 for b in 8 7; do
  echo "== MKZG_VMSM_BITS=$b" >> $O/ab_lone36.txt
  MI355KZG_LIB=$GRAFT_REPO_ROOT/build_ab/libmi355kzg_vmsm7.so MKZG_VMSM_BITS=$b CV_REPS=30 CV_CALLERS=1 timeout 300 python3 tools/bench_cellverify_inflight.py 2>&1 | grep callers >> $O/ab_lone36.txt
  MI355KZG_LIB=$GRAFT_REPO_ROOT/build_ab/libmi355kzg_vmsm7.so MKZG_VMSM_BITS=$b CV_DEVICE=1 CV_REPS=30 CV_CALLERS=1 timeout 300 python3 tools/bench_cellverify_inflight.py 2>&1 | grep callers >> $O/ab_lone36.txt
 done
done
cat $O/ab_lone36.txt
