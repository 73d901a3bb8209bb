# why does bench.py's twelve-caller cell-verifier leg read 9.0-9.2 M cells/s inside the default line and 11.3 M when the EIP-7594 section runs alone?
# A: section alone; B: the default line; C: section alone again right behind it (a hot GPU?); D: the default line without the drop-in legs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
show() { python3 - "$1" "$2" <<PY
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
v=d["eip7594"]["verify_cell_kzg_proof_batch"]
print(sys.argv[2], "headline", d["value"], "cell verifier lone", v["ms_per_call_median"], "sustained", v["sustained"]["cells_per_s"], v["sustained"]["cells_per_s_each_trial"], "cpus", v["sustained"]["host_cpus_busy"])
PY
}
timeout 600 python3 bench.py --full-line --steps 2 --warmup 1 --no-cpu-baseline --no-drop-in --verify-batch 0 --cell-blobs 64 > $O/b31_A.json 2>/dev/null < /dev/null; show $O/b31_A.json A
timeout 900 python3 bench.py --full-line --no-cpu-baseline > $O/b31_B.json 2>/dev/null < /dev/null; show $O/b31_B.json B
timeout 600 python3 bench.py --full-line --steps 2 --warmup 1 --no-cpu-baseline --no-drop-in --verify-batch 0 --cell-blobs 64 > $O/b31_C.json 2>/dev/null < /dev/null; show $O/b31_C.json C
timeout 900 python3 bench.py --full-line --no-cpu-baseline --no-drop-in > $O/b31_D.json 2>/dev/null < /dev/null; show $O/b31_D.json D
timeout 900 python3 bench.py --full-line --no-cpu-baseline --verify-batch 0 > $O/b31_E.json 2>/dev/null < /dev/null; show $O/b31_E.json E
