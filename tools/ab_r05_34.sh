# the round's final comparison on ONE box, alternating: the round-4 library (build_ab/libmi355kzg_r04.so) against the final build; sustained legs with
# trials past the start-up convoy (31 / 31 calls per caller), distinct inputs per caller
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python3 tools/ab_verify.py build_ab/libmi355kzg_r04.so go-eth-kzg_amd/libmi355kzg.so 3 > $O/ab_final34.txt 2>&1 < /dev/null
grep AB $O/ab_final34.txt | cut -c1-250
for rep in 1 2 3; do
 for lib in build_ab/libmi355kzg_r04.so go-eth-kzg_amd/libmi355kzg.so; do
  echo "== $lib" >> $O/ab_final34c.txt
  MI355KZG_LIB=$GRAFT_REPO_ROOT/$lib CV_REPS=30 CV_CALLERS=1,12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_final34c.txt 2>&1 < /dev/null
 done
done
grep -v amdgpu.ids $O/ab_final34c.txt | cut -c1-250
