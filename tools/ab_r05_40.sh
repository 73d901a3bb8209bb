# deep rounds: the round takes ANY hash stream whose last round has finished (new) against the streams in turn (build_ab/libmi355kzg_head.so), alternating; 31 calls per caller
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_eip4844.py -m gpu -q -x -p no:cacheprovider < /dev/null > $O/t40_parity.txt 2>&1
grep -E "passed|failed|error" $O/t40_parity.txt | tail -2
timeout 1200 python3 tools/ab_verify.py build_ab/libmi355kzg_head.so go-eth-kzg_amd/libmi355kzg.so 4 > $O/ab_anystream40.txt 2>&1 < /dev/null
grep AB $O/ab_anystream40.txt | cut -c1-250
cd /tmp; rm -rf /tmp/trg; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trg -- python3 $GRAFT_REPO_ROOT/tools/trace.py verify_inflight 32 30 2>&1 | grep "trial 1"; python3 $GRAFT_REPO_ROOT/tools/round_gaps.py /tmp/trg
