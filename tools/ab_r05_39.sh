# cell verifier, host cells: the gather started at the top of the call (before the commitments' deduplication and the points' launches) against the committed build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_eip7594.py tests/test_gpu_fullsize.py tests/test_gpu_exchange.py -m gpu -q -x -p no:cacheprovider -k "cell or recover or 7594 or exchange" < /dev/null > $O/t39_parity.txt 2>&1
grep -E "passed|failed|error" $O/t39_parity.txt | tail -3
for rep in 1 2 3; do
 for lib in build_ab/libmi355kzg_head.so go-eth-kzg_amd/libmi355kzg.so; do
  echo "== $lib" >> $O/ab_gather39.txt
  MI355KZG_LIB=$GRAFT_REPO_ROOT/$lib CV_REPS=30 CV_CALLERS=1,12 timeout 300 python3 tools/bench_cellverify_inflight.py 2>&1 | grep callers >> $O/ab_gather39.txt
 done
done
cat $O/ab_gather39.txt
