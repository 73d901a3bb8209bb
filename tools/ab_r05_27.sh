# deep rounds with 3 / 4 rounds in flight against the plain rounds (3 in flight), alternating on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
for cfg in "0 3" "1 3" "1 4" "0 4" "0 3" "1 3" "1 4" "0 4"; do
  set -- $cfg
  echo "== MKZG_DEEP_ROUNDS=$1 MKZG_HASH_IN_FLIGHT=$2" >> $O/ab_deep27.txt
  MKZG_DEEP_ROUNDS=$1 MKZG_HASH_IN_FLIGHT=$2 timeout 300 python3 tools/ab_verify.py --one >> $O/ab_deep27.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_deep27.txt | cut -c1-250
