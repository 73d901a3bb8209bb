set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/t14.txt; cat $O/t14.txt
for v in 0 1 0 1; do
  echo "== MKZG_TP_GRAPH=$v" >> $O/ab_graph14.txt
  MKZG_TP_GRAPH=$v timeout 300 python3 tools/ab_verify.py --one >> $O/ab_graph14.txt 2>&1 < /dev/null
  MKZG_TP_GRAPH=$v timeout 300 python3 bench.py --full-line --steps 20 --warmup 3 --no-cpu-baseline --no-drop-in --cell-blobs 0 --verify-batch 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HEADLINE', d['value'], d['ms_per_step'], d['repeats']['blobs_per_s_median'], d['streams_sweep']['blobs_per_s_by_streams'])" >> $O/ab_graph14.txt 2>&1
done
grep -v amdgpu.ids $O/ab_graph14.txt | cut -c1-250
