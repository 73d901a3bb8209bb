set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
NEW=$PWD/go-eth-kzg_amd/libmi355kzg.so; S176=$PWD/build_ab/libmi355kzg_sha176.so; S256=$PWD/build_ab/libmi355kzg_sha256.so
for v in "new $NEW" "sha176 $S176" "sha256 $S256" "new $NEW" "sha176 $S176" "sha256 $S256"; do set -- $v
  echo "== $1" >> $O/ab_verify12.txt
  MI355KZG_LIB=$2 timeout 300 python3 tools/ab_verify.py --one >> $O/ab_verify12.txt 2>&1 < /dev/null
done
grep -v amdgpu.ids $O/ab_verify12.txt | cut -c1-250
