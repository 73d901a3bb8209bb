# deep rounds (the hash round also launches its calls' evaluations and MSMs, one launch each): parity, then MKZG_DEEP_ROUNDS=0 / 1 alternating on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_eip4844.py tests/test_gpu_fullsize.py tests/test_gpu_ragged_batches.py tests/test_gpu_exchange.py tests/test_gpu_soak.py tests/test_gpu_defaults.py -m gpu -q -x -p no:cacheprovider < /dev/null > $O/t24_parity.txt 2>&1
tail -3 $O/t24_parity.txt
for v in 0 1 0 1 0 1; do
  echo "== MKZG_DEEP_ROUNDS=$v" >> $O/ab_deep24.txt
  MKZG_DEEP_ROUNDS=$v timeout 300 python3 tools/ab_verify.py --one >> $O/ab_deep24.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_deep24.txt | cut -c1-250
