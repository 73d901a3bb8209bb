# early exit of k_vmsm_window's second wavefront (8-bit windows forced) against the committed build (build_ab/libmi355kzg_head.so), alternating on one box
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
export MKZG_VMSM_BITS=8
for rep in 1 2 3; do
 for lib in build_ab/libmi355kzg_head.so go-eth-kzg_amd/libmi355kzg.so; do
  echo "== $lib" >> $O/ab_exit19c.txt
  MI355KZG_LIB=$GRAFT_REPO_ROOT/$lib CV_CALLERS=1,12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_exit19c.txt 2>&1 < /dev/null
  echo "== $lib (cells in HBM)" >> $O/ab_exit19c.txt
  MI355KZG_LIB=$GRAFT_REPO_ROOT/$lib CV_DEVICE=1 CV_CALLERS=1,12 timeout 300 python3 tools/bench_cellverify_inflight.py >> $O/ab_exit19c.txt 2>&1 < /dev/null
 done
done
grep -v amdgpu.ids $O/ab_exit19c.txt | cut -c1-250
timeout 900 python3 tools/ab_verify.py build_ab/libmi355kzg_head.so go-eth-kzg_amd/libmi355kzg.so 3 > $O/ab_exit19.txt 2>&1 < /dev/null
grep AB $O/ab_exit19.txt | cut -c1-250
