# does the single-blob-callers leg crash under rocprofv3 (it did inside the full bench line's profile)?  combiner on / off, per kind
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for comb in 1 0; do
 for kind in blob_to_kzg_commitment verify_blob_kzg_proof compute_cells_and_kzg_proofs; do
  export COMBINE=$comb KINDS=$kind CALLERS=32 SECONDS=0.35
  rm -rf /tmp/trc
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trc -- python3 $GRAFT_REPO_ROOT/tools/bench_single_callers.py > $O/prof20_${comb}_$kind.txt 2>&1 < /dev/null
  echo "combine=$comb $kind rc=$?" | tee -a $O/prof20.txt
  grep -E "^[a-z_]+ callers|Segmentation|SIGSEGV|Error|error" $O/prof20_${comb}_$kind.txt | head -5 | tee -a $O/prof20.txt
 done
done
