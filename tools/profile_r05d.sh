# kernel statistics of the default bench command on the final build (rocprofv3 --kernel-trace --stats); the whole line first, without the drop-in legs if the profiler
# dies inside them (it did once in this round: SIGSEGV in its own frames under the single-blob callers' threads)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05q; mkdir -p $O
cd /tmp
rm -rf /tmp/ks_default
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_default -o ks -- python3 $GRAFT_REPO_ROOT/bench.py --full-line > $O/bench_default_profiled.json 2> $O/bench_default_profiled.err < /dev/null
rc=$?
echo "full line under the profiler: rc=$rc" | tee $O/profiled_rc.txt
if [ $rc -ne 0 ] || [ -z "$(find /tmp/ks_default -name '*kernel_stats.csv' | head -1)" ]; then
  rm -rf /tmp/ks_default
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_default -o ks -- python3 $GRAFT_REPO_ROOT/bench.py --full-line --no-drop-in > $O/bench_default_profiled.json 2> $O/bench_default_profiled.err < /dev/null
  echo "without the drop-in legs: rc=$?" | tee -a $O/profiled_rc.txt
fi
cp $(find /tmp/ks_default -name '*kernel_stats.csv' | head -1) $O/bench_default_kernel_stats.csv
head -12 $O/bench_default_kernel_stats.csv | cut -c1-160
tail -3 $O/bench_default_profiled.err | cut -c1-300
