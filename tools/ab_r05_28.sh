# deep rounds: tail on the round's stream (MKZG_DEEP_TAIL=0) / on the leader's call stream once the hash is done (1), 3 / 4 hash rounds in flight, against plain rounds; parity first
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_eip4844.py tests/test_gpu_fullsize.py tests/test_gpu_exchange.py tests/test_gpu_soak.py -m gpu -q -x -p no:cacheprovider < /dev/null > $O/t28_parity.txt 2>&1
grep -E "passed|failed|error" $O/t28_parity.txt | tail -3
for cfg in "0 0 3" "1 1 3" "1 1 4" "1 0 4" "0 0 3" "1 1 3" "1 1 4" "1 0 4"; do
  set -- $cfg
  echo "== MKZG_DEEP_ROUNDS=$1 MKZG_DEEP_TAIL=$2 MKZG_HASH_IN_FLIGHT=$3" >> $O/ab_deep28.txt
  MKZG_DEEP_ROUNDS=$1 MKZG_DEEP_TAIL=$2 MKZG_HASH_IN_FLIGHT=$3 timeout 300 python3 tools/ab_verify.py --one >> $O/ab_deep28.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_deep28.txt | cut -c1-250
