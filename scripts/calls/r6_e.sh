#!/bin/bash
export TAG=${1:-r6e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for lib in "" conddma condq; do
  if [ -z "$lib" ]; then timeout 200 python scripts/experiments/attn_stream_debug.py 2>&1 | grep -v "Warn\|warn\|amdgpu.ids" | tee -a $OUT/debug.txt
  else DWM_HIP_LIB=$GRAFT_REPO_ROOT/opendwm_amd/variants/libdwm_hip_$lib.so timeout 200 python scripts/experiments/attn_stream_debug.py 2>&1 | grep -v "Warn\|warn\|amdgpu.ids" | tee -a $OUT/debug.txt; fi
done
timeout 200 python scripts/microbench.py attnr4 2>&1 | grep "602\|dual" | cut -c1-150 | tee $OUT/microbench.txt
