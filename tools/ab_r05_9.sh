set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
R04=$PWD/build_ab/libmi355kzg_r04.so; NEW=$PWD/go-eth-kzg_amd/libmi355kzg.so
timeout 900 python -m pytest tests/test_gpu_eip7594.py tests/test_fk20_segments.py tests/test_gpu_ragged_batches.py tests/test_gpu_fullsize.py tests/test_gpu_combine.py -m gpu -x -q 2>&1 | tail -6 > $O/t9.txt; cat $O/t9.txt
for v in "r04 $R04" "new $NEW"; do set -- $v
  echo "== $1" >> $O/ab_fk20_9.txt
  MI355KZG_LIB=$2 SWEEP_NB=256 SWEEP_SIZES=64,65,96,128,160,192,256 timeout 300 python3 tools/sweep_cells.py >> $O/ab_fk20_9.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_fk20_9.txt | cut -c1-200
timeout 900 python bench.py --full-line > $O/bench9.json 2> $O/bench9.err < /dev/null
tail -c 1500 $O/bench9.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench9.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_overlapped'))
s=d['drop_in']['single_blob_callers']
for k in ('blob_to_kzg_commitment','compute_cells_and_kzg_proofs','verify_blob_kzg_proof'):
    print(k, {c:v.get('blobs_per_s') for c,v in s[k].items() if isinstance(v,dict) and 'blobs_per_s' in v}, s[k].get('ratio_to_batch'))
e=d['eip7594']['verify_cell_kzg_proof_batch']; print({k:e[k] for k in e if k.startswith('ms_per_call') and k!='ms_per_call_each'}, e.get('sustained',{}).get('cells_per_s'))
print(d['verify_blob_kzg_proof_batch']['single_call']['ms_per_call'], d['verify_blob_kzg_proof_batch']['sustained']['blobs_per_s'])
c=d['eip7594']['compute_cells_and_kzg_proofs']; print({k:c[k] for k in c if not isinstance(c[k],dict)}, c.get('sustained',{}).get('blobs_per_s'))
PY
