set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -12 > $O/box15.txt; cat /proc/cpuinfo | grep -m1 "model name" >> $O/box15.txt; cat $O/box15.txt
for v in 0 1 0 1 0 1; do
  echo "== MKZG_TP_GRAPH=$v" >> $O/ab_graph15.txt
  MKZG_TP_GRAPH=$v timeout 300 python3 tools/ab_verify.py --one >> $O/ab_graph15.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_graph15.txt | cut -c1-250
