# kernel trace of the sustained blob verifier with deep rounds on: launches per round, durations, kernels in flight
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for v in 1 0; do
  rm -rf /tmp/trd$v
  MKZG_DEEP_ROUNDS=$v timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trd$v -- python3 $GRAFT_REPO_ROOT/tools/trace.py verify_inflight 32 10 > $O/deep25_$v.txt 2>&1 < /dev/null
  python3 $GRAFT_REPO_ROOT/tools/trace.py busy /tmp/trd$v >> $O/deep25_$v.txt 2>&1
  echo "== deep=$v"; grep -v amdgpu.ids $O/deep25_$v.txt | grep -E "burst|k_|copy|blobs/s|M blobs" | head -20
done
