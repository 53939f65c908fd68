#!/bin/bash
# One source recompiled with extra flags and linked with the objects of the regular build into opendwm_amd/variants/libdwm_hip_<tag>.so
# (A/B timing builds: load with DWM_HIP_LIB=<path>).  usage: build_variant.sh <tag> <source.hip>[,<source2.hip>...] <flags...>
set -e
cd "$(dirname "$0")/../.."
tag=$1; srcs=$2; shift 2
C=opendwm_amd/csrc
mkdir -p opendwm_amd/variants $C/build/var
objs=$(ls $C/build/*.o)
new=""
for src in ${srcs//,/ }; do
  vg="-mllvm -amdgpu-mfma-vgpr-form"; ff=""
  case $src in gemm_bf16_4w.hip) vg="";; attention_stream.hip) vg=""; ff="-fno-slp-vectorize";; esac
  o=$C/build/var/${src%.hip}_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Iinclude -I$C $vg $ff "$@" -c $C/$src -o $o 2> $C/build/var/${tag}_${src%.hip}.log
  objs=$(echo "$objs" | grep -v "/${src%.hip}.o")
  new="$new $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o opendwm_amd/variants/libdwm_hip_$tag.so $objs $new
echo opendwm_amd/variants/libdwm_hip_$tag.so
