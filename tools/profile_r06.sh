# tools/profile_r06.sh -- the round's records on the final build, everything under gpurun_out/r06p/ (copied to profiles/r06_* on the build host):
# the full GPU suite, THE default bench line (compact stdout line + the whole record), kernel statistics of the headline command on one stream
# (the launch alone: what roofline.frac is quoted on) and of the whole default command, busy summaries of the two sustained verifier legs.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06p; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider < /dev/null > $O/tests_full.txt 2>&1
grep -E "passed|failed|error" $O/tests_full.txt | tail -3
# PMC passes of every leg on the final kernel sources (one bench configuration per leg: tools/pmc_collect.sh), merged into profiles/r06_traffic.json
PASS_TIMEOUT=150 bash tools/pmc_collect.sh r06c > $O/pmc_c.log 2>&1 < /dev/null
PASS_TIMEOUT=150 bash tools/pmc_collect.sh r06v --verify-batch 1024 --verify-inflight 1 --no-strong > $O/pmc_v.log 2>&1 < /dev/null
PMC_GRIDS="k_g1_validate=4096" PASS_TIMEOUT=150 bash tools/pmc_collect.sh r06t --verify-batch 1024 --verify-inflight 1 --no-strong --verify-shape 1 > $O/pmc_t.log 2>&1 < /dev/null
PASS_TIMEOUT=200 bash tools/pmc_collect.sh r06e --cell-blobs 64 > $O/pmc_e.log 2>&1 < /dev/null
rm -rf gpurun_out/pmc_r06c gpurun_out/pmc_r06v gpurun_out/pmc_r06t gpurun_out/pmc_r06e
python3 tools/pmc_merge.py profiles/r06_traffic.json commit=gpurun_out/r06c_traffic.json verify=gpurun_out/r06v_traffic.json verify_tp=gpurun_out/r06t_traffic.json eip7594=gpurun_out/r06e_traffic.json
cp profiles/r06_traffic.json $O/r06_traffic.json
cp gpurun_out/r06c_pmc_summary.csv gpurun_out/r06v_pmc_summary.csv gpurun_out/r06t_pmc_summary.csv gpurun_out/r06e_pmc_summary.csv $O/ 2>/dev/null
timeout 900 python3 bench.py --steps 20 --warmup 5 --detail-out $O/bench_default_detail.json > $O/bench_default.json 2> $O/bench_default.err < /dev/null
wc -c $O/bench_default.json; tail -c 300 $O/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_commit -o ks -- python3 bench.py --full-line --steps 20 --warmup 3 --streams 1 --no-cpu-baseline --no-drop-in --cell-blobs 0 --verify-batch 0 --detail-out /tmp/d1.json > $O/bench_commit_only.json 2> $O/bench_commit_only.err < /dev/null
cp $(find /tmp/ks_commit -name '*kernel_stats.csv' | head -1) $O/bench_commit_only_kernel_stats.csv; rm -rf /tmp/ks_commit
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_default -o ks -- python3 bench.py --full-line --detail-out /tmp/d2.json > $O/bench_default_profiled.json 2> $O/bench_default_profiled.err < /dev/null
rc=$?; echo "full line under the profiler: rc=$rc" | tee $O/profiled_rc.txt
if [ $rc -ne 0 ] || [ -z "$(find /tmp/ks_default -name '*kernel_stats.csv' | head -1)" ]; then
  rm -rf /tmp/ks_default
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_default -o ks -- python3 bench.py --full-line --no-drop-in --detail-out /tmp/d2.json > $O/bench_default_profiled.json 2> $O/bench_default_profiled.err < /dev/null
  echo "without the drop-in legs: rc=$?" | tee -a $O/profiled_rc.txt
fi
f=$(find /tmp/ks_default -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/bench_default_kernel_stats.csv; rm -rf /tmp/ks_default
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_v -o tr -- python3 tools/trace.py verify_inflight 32 10 > $O/verify_sustained_trace.txt 2>&1 < /dev/null
python3 tools/trace.py busy /tmp/tr_v > $O/verify_sustained_busy.txt 2>&1; rm -rf /tmp/tr_v
CV_CALLERS=12 CV_REPS=8 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr_c -o tr -- python3 tools/bench_cellverify_inflight.py > $O/cellverify_sustained_trace.txt 2>&1 < /dev/null
python3 tools/trace.py busy /tmp/tr_c > $O/cellverify_sustained_busy.txt 2>&1; rm -rf /tmp/tr_c
cat $O/verify_sustained_busy.txt $O/cellverify_sustained_busy.txt
head -8 $O/bench_commit_only_kernel_stats.csv | cut -c1-160
ls -la $O
