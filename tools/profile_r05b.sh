# tools/profile_r05b.sh -- the round's final lines and traces on the final build (PMC legs: tools/profile_r05.sh)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_bench_multirank.py -m gpu -x -q 2>&1 | tail -5 > $O/t16.txt; cat $O/t16.txt
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_combine.py -m gpu -x -q 2>&1 | tail -1; done > $O/t16_combine.txt; cat $O/t16_combine.txt
# THE line (default options), unprofiled
timeout 900 python3 bench.py --full-line > $O/bench_default.json 2> $O/bench_default.err < /dev/null
tail -c 400 $O/bench_default.err
# kernel stats: the headline command on one stream (the launch alone: what roofline.frac is quoted on), then the default line without the host-thread legs
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_commit -o ks -- python3 bench.py --full-line --steps 20 --warmup 3 --streams 1 --no-cpu-baseline --no-drop-in --cell-blobs 0 --verify-batch 0 > $O/bench_commit_only.json 2> $O/bench_commit_only.err < /dev/null
cp $(find $O/ks_commit -name '*kernel_stats.csv' | head -1) $O/bench_commit_only_kernel_stats.csv; rm -rf $O/ks_commit
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_default -o ks -- python3 bench.py --full-line --no-drop-in --no-cpu-baseline > $O/bench_default_profiled.json 2> $O/bench_default_profiled.err < /dev/null
f=$(find $O/ks_default -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/bench_default_kernel_stats.csv; rm -rf $O/ks_default
# busy summaries of the two sustained verifier legs
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr_v -o tr -- python3 tools/trace.py verify_inflight 32 10 > $O/verify_sustained_trace.txt 2>&1 < /dev/null
python3 tools/trace.py busy $O/tr_v > $O/verify_sustained_busy.txt 2>&1; rm -rf $O/tr_v
CV_CALLERS=12 CV_REPS=8 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr_c -o tr -- python3 tools/bench_cellverify_inflight.py > $O/cellverify_sustained_trace.txt 2>&1 < /dev/null
python3 tools/trace.py busy $O/tr_c > $O/cellverify_sustained_busy.txt 2>&1; rm -rf $O/tr_c
cat $O/verify_sustained_busy.txt $O/cellverify_sustained_busy.txt
ls -la $O
