# tools/profile_r05.sh -- the round's measurement artefacts, on the GPU box (gpurun): the HIP-graph A/B, the PMC legs whose kernel sources
# changed since the commit leg was collected, kernel stats, busy summaries, the bench lines.  Everything lands in gpurun_out/r05p/.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05p; mkdir -p $O
NEW=$PWD/go-eth-kzg_amd/libmi355kzg.so
# 1. the bounded HIP-graph experiment (MKZG_TP_GRAPH): parity under the forced throughput shape, then the sustained A/B
MKZG_TP_GRAPH=1 timeout 600 python -m pytest tests/test_gpu_eip4844.py -m gpu -x -q 2>&1 | tail -4 > $O/graph_tests.txt; cat $O/graph_tests.txt
for v in 0 1 0 1; do
  echo "== MKZG_TP_GRAPH=$v" >> $O/ab_graph.txt
  MKZG_TP_GRAPH=$v timeout 300 python3 tools/ab_verify.py --one >> $O/ab_graph.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_graph.txt | cut -c1-250
# 2. PMC legs of the verify / eip7594 source families (the commit leg of profiles/r05_traffic.json stands: its sources are unchanged)
PASS_TIMEOUT=150 bash tools/pmc_collect.sh r05v --verify-batch 1024 --verify-inflight 1 --no-strong > $O/pmc_v.log 2>&1 < /dev/null
PASS_TIMEOUT=150 bash tools/pmc_collect.sh r05t --verify-batch 1024 --verify-inflight 1 --no-strong --verify-shape 1 > $O/pmc_t.log 2>&1 < /dev/null
PASS_TIMEOUT=200 bash tools/pmc_collect.sh r05e --cell-blobs 64 > $O/pmc_e.log 2>&1 < /dev/null
rm -rf gpurun_out/pmc_r05v gpurun_out/pmc_r05t gpurun_out/pmc_r05e
python3 tools/pmc_merge.py profiles/r05_traffic.json @profiles/r05_traffic.json verify=gpurun_out/r05v_traffic.json verify_tp=gpurun_out/r05t_traffic.json eip7594=gpurun_out/r05e_traffic.json
cp profiles/r05_traffic.json $O/r05_traffic.json
# 3. kernel stats of the headline command (one stream: the launch alone, what roofline.frac is quoted on) and of the default line
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_commit -o ks -- python3 bench.py --full-line --steps 20 --warmup 3 --streams 1 --no-cpu-baseline --no-drop-in --cell-blobs 0 --verify-batch 0 > $O/bench_commit_only.json 2> $O/bench_commit_only.err < /dev/null
cp $(find $O/ks_commit -name '*kernel_stats.csv' | head -1) $O/bench_commit_only_kernel_stats.csv; rm -rf $O/ks_commit
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_default -o ks -- python3 bench.py --full-line > $O/bench_default_profiled.json 2> $O/bench_default_profiled.err < /dev/null
cp $(find $O/ks_default -name '*kernel_stats.csv' | head -1) $O/bench_default_kernel_stats.csv; rm -rf $O/ks_default
# 4. the default line, unprofiled: THE line
timeout 900 python3 bench.py --full-line > $O/bench_default.json 2> $O/bench_default.err < /dev/null
tail -c 600 $O/bench_default.err
# 5. busy summaries of the two sustained verifier legs
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr_v -o tr -- python3 tools/trace.py verify_inflight 32 10 > $O/verify_sustained_trace.txt 2>&1 < /dev/null
python3 tools/trace.py busy $O/tr_v > $O/verify_sustained_busy.txt 2>&1; rm -rf $O/tr_v
CV_CALLERS=12 CV_REPS=8 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr_c -o tr -- python3 tools/bench_cellverify_inflight.py > $O/cellverify_sustained_trace.txt 2>&1 < /dev/null
python3 tools/trace.py busy $O/tr_c > $O/cellverify_sustained_busy.txt 2>&1; rm -rf $O/tr_c
cat $O/verify_sustained_busy.txt $O/cellverify_sustained_busy.txt
ls -la $O
