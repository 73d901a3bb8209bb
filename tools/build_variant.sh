#!/bin/bash
# tools/build_variant.sh -- an A/B or diagnostic build of the library beside the product (some sources recompiled with extra flags, e.g.
#   tools/build_variant.sh cpuprof "-DMKZG_CPU_PROF" api_verify.hip; MI355KZG_LIB=build_ab/libmi355kzg_cpuprof.so python3 tools/trace.py verify_inflight 32 26)
# usage: build_variant.sh NAME "-DFLAG=..." "file.hip [file2.hip ...]" [extra/source.hip ...]
#   -> build_ab/libmi355kzg_NAME.so: the listed product sources recompiled with the flags, extra sources (csrc-relative, e.g. the rejected
#      experiments/commit_pairs_kernels.hip) compiled with the flags and linked in, every other object reused from the product build
set -e
cd /root/repo/go-eth-kzg_amd/csrc
name=$1; flags=$2; srcs=$3; shift 3 || true
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result"
pids=""
for src in $srcs "$@"; do
  o=/tmp/variant_${name}_$(basename $src .hip).o
  $CC $flags -c $src -o $o 2>/dev/null &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
objs=""
for f in api api_verify api_cells api_multi api_exchange msm_kernels verify_kernels cells_kernels fk20_kernels recover_kernels; do
  if echo " $srcs " | grep -q " $f.hip "; then objs="$objs /tmp/variant_${name}_$f.o"; else objs="$objs $f.o"; fi
done
for src in "$@"; do objs="$objs /tmp/variant_${name}_$(basename $src .hip).o"; done
mkdir -p /root/repo/build_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build_ab/libmi355kzg_$name.so $objs -ldl
echo built $name
