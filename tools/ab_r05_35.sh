# steady-state re-test (31 calls per caller) of section 11's rejections on the pre-deep-rounds sources: build_ab/libmi355kzg_vmsm7.so = commit 7c40660 + the 7-bit
# windows / early-exit experiment (MKZG_VMSM_BITS=8: 8-bit windows with the second wavefront's early exit; 0: 7-bit windows in the throughput shapes);
# baseline: the final library with plain rounds, three in flight (the same kernels as 7c40660)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
run() { echo "== $1" >> $O/ab_steady35.txt; env $2 timeout 300 python3 tools/ab_verify.py --one 2>&1 | grep AB | cut -c1-250 >> $O/ab_steady35.txt; }
for rep in 1 2 3; do
  run "final library, plain rounds (3 in flight)" "MKZG_DEEP_ROUNDS=0 MKZG_HASH_IN_FLIGHT=3"
  run "experiment build, 8-bit windows + early exit" "MI355KZG_LIB=$GRAFT_REPO_ROOT/build_ab/libmi355kzg_vmsm7.so MKZG_VMSM_BITS=8"
  run "experiment build, 7-bit windows" "MI355KZG_LIB=$GRAFT_REPO_ROOT/build_ab/libmi355kzg_vmsm7.so MKZG_VMSM_BITS=0"
  run "final library, deep rounds" "MKZG_DEEP_ROUNDS=1"
done
cat $O/ab_steady35.txt
for rep in 1 2; do
 for cfg in "final MI355KZG_LIB=$GRAFT_REPO_ROOT/go-eth-kzg_amd/libmi355kzg.so" "exp8 MI355KZG_LIB=$GRAFT_REPO_ROOT/build_ab/libmi355kzg_vmsm7.so MKZG_VMSM_BITS=8" "exp7 MI355KZG_LIB=$GRAFT_REPO_ROOT/build_ab/libmi355kzg_vmsm7.so MKZG_VMSM_BITS=0"; do
  set -- $cfg; name=$1; shift
  echo "== $name" >> $O/ab_steady35c.txt
  env "$@" CV_REPS=30 CV_CALLERS=1,12 timeout 300 python3 tools/bench_cellverify_inflight.py 2>&1 | grep callers >> $O/ab_steady35c.txt
 done
done
cat $O/ab_steady35c.txt
