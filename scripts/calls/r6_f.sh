#!/bin/bash
export TAG=${1:-r6f}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
DWM_HIP_LIB=$GRAFT_REPO_ROOT/opendwm_amd/variants/libdwm_hip_xnofb.so timeout 200 python scripts/experiments/attn_stream_debug2.py 2>&1 | grep -v "Warn\|warn\|amdgpu.ids" | cut -c1-300 | tee $OUT/debug2.txt
