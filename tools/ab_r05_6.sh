set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
R04=$PWD/build_ab/libmi355kzg_r04.so; NEW=$PWD/go-eth-kzg_amd/libmi355kzg.so
timeout 600 python -m pytest tests/test_gpu_eip7594.py tests/test_fk20_segments.py tests/test_gpu_ragged_batches.py -m gpu -x -q 2>&1 | tail -6 > $O/t6a.txt
cat $O/t6a.txt
for v in "r04 $R04" "new $NEW" "r04 $R04" "new $NEW"; do set -- $v
  echo "== $1" >> $O/ab_fk20_6.txt
  MI355KZG_LIB=$2 SWEEP_NB=256 SWEEP_SIZES=1,16,33,48,64,65,128,193,256 timeout 300 python3 tools/sweep_cells.py >> $O/ab_fk20_6.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_fk20_6.txt
timeout 900 python bench.py --full-line > $O/bench6.json 2> $O/bench6.err < /dev/null
tail -c 2500 $O/bench6.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench6.json').read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('metric','value','ms_per_step','roofline','valu')},indent=0)[:3000])
print(json.dumps(d.get('drop_in',{}).get('single_blob_callers'),indent=0)[:4000])
print(json.dumps(d.get('multi_gpu'),indent=0)[:600])
print(json.dumps(d.get('verify_metric'),indent=0)[:900])
e=d['eip7594']['verify_cell_kzg_proof_batch']; print({k:e[k] for k in e if k.startswith('ms_per_call') or k in ('host_finish_ms','pcie','cells_per_s')})
print(d['eip7594']['verify_cell_kzg_proof_batch'].get('sustained',{}).get('cells_per_s'), d['verify_blob_kzg_proof_batch']['sustained']['blobs_per_s'])
PY
timeout 900 python -m pytest tests/test_gpu_bench_multirank.py tests/test_gpu_exchange.py -m gpu -x -q 2>&1 | tail -15 > $O/t6.txt
cat $O/t6.txt
