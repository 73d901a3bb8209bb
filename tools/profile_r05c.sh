# tools/profile_r05c.sh -- after the deep rounds: the full GPU suite, the PMC legs of the verify / eip7594 source families, THE default bench line
# (unprofiled) and the kernel trace + busy summary of the sustained blob verifier.  Everything lands in gpurun_out/r05q/.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05q; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider < /dev/null > $O/tests_full.txt 2>&1
grep -E "passed|failed|error" $O/tests_full.txt | tail -3
PASS_TIMEOUT=150 bash tools/pmc_collect.sh r05v --verify-batch 1024 --verify-inflight 1 --no-strong > $O/pmc_v.log 2>&1 < /dev/null
PASS_TIMEOUT=150 bash tools/pmc_collect.sh r05t --verify-batch 1024 --verify-inflight 1 --no-strong --verify-shape 1 > $O/pmc_t.log 2>&1 < /dev/null
PASS_TIMEOUT=200 bash tools/pmc_collect.sh r05e --cell-blobs 64 > $O/pmc_e.log 2>&1 < /dev/null
rm -rf gpurun_out/pmc_r05v gpurun_out/pmc_r05t gpurun_out/pmc_r05e
python3 tools/pmc_merge.py profiles/r05_traffic.json @profiles/r05_traffic.json verify=gpurun_out/r05v_traffic.json verify_tp=gpurun_out/r05t_traffic.json eip7594=gpurun_out/r05e_traffic.json
cp profiles/r05_traffic.json $O/r05_traffic.json
cp gpurun_out/r05v_pmc_summary.csv gpurun_out/r05t_pmc_summary.csv gpurun_out/r05e_pmc_summary.csv $O/ 2>/dev/null
timeout 900 python3 bench.py --full-line > $O/bench_default.json 2> $O/bench_default.err < /dev/null
tail -c 400 $O/bench_default.err
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr_v -o tr -- python3 tools/trace.py verify_inflight 32 10 > $O/verify_sustained_trace.txt 2>&1 < /dev/null
python3 tools/trace.py busy $O/tr_v > $O/verify_sustained_busy.txt 2>&1; rm -rf $O/tr_v
cat $O/verify_sustained_busy.txt
ls -la $O
