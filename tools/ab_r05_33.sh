# deep rounds: 4 rounds in flight (lowest-priority streams) against 6 / 8 (streams 4..7 on the middle priority level: a second pool of hardware queues)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
for r in 4 6 8 4 6 8; do
  echo "== MKZG_HASH_IN_FLIGHT=$r" >> $O/ab_rounds33.txt
  MKZG_HASH_IN_FLIGHT=$r timeout 300 python3 tools/trace.py verify_inflight 32 40 2>&1 | grep "trial 1" >> $O/ab_rounds33.txt
done
cat $O/ab_rounds33.txt
