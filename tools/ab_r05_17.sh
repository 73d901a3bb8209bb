# isolated durations of k_vmsm_window in its 8-bit and 7-bit forms (lone calls; verify_shape 1 = three 1024-point jobs in one launch; the cell verifier's four jobs)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for v in 8 7; do
  export MKZG_VMSM_BITS=$v
  rm -rf /tmp/tr$v
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$v -- python3 $GRAFT_REPO_ROOT/tools/trace.py lone 5 > $GRAFT_REPO_ROOT/$O/lone17_$v.txt 2>&1 < /dev/null
  python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/tr$v vmsm validate cell_ eval sha >> $GRAFT_REPO_ROOT/$O/lone17_$v.txt 2>&1
  grep -v amdgpu.ids $GRAFT_REPO_ROOT/$O/lone17_$v.txt | tail -20
done
