# timeline of lone host-pointer VerifyCellKZGProofBatch(8192) calls: kernels and copies (rocprofv3 --kernel-trace --memory-copy-trace), the last call's rows
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/trl
CV_CALLERS=1 CV_REPS=4 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trl -- python3 $GRAFT_REPO_ROOT/tools/bench_cellverify_inflight.py > $O/lone38.txt 2>&1 < /dev/null
python3 $GRAFT_REPO_ROOT/tools/trace.py timeline /tmp/trl > $O/lone38_timeline.txt 2>&1
grep callers $O/lone38.txt
tail -60 $O/lone38_timeline.txt
