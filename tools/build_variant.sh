#!/bin/bash
# tools/build_variant.sh -- an A/B or diagnostic build of the library beside the product (one source recompiled with extra flags, e.g.
#   tools/build_variant.sh cpuprof "-DMKZG_CPU_PROF" api_verify.hip; MI355KZG_LIB=build_ab/libmi355kzg_cpuprof.so python3 tools/trace.py verify_inflight 32 26)
# usage: build_variant.sh NAME "-DFLAG=..." file.hip   -> build_ab/libmi355kzg_NAME.so (other objects reused)
set -e
cd /root/repo/go-eth-kzg_amd/csrc
name=$1; flags=$2; src=$3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result $flags -c $src -o /tmp/variant_$name.o 2>/dev/null
objs=""
for f in api api_verify api_cells api_multi api_exchange msm_kernels verify_kernels cells_kernels fk20_kernels recover_kernels; do
  if [ "$f.hip" == "$src" ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build_ab/libmi355kzg_$name.so $objs -ldl
echo built $name
