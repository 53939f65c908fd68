#!/bin/bash
export TAG=${1:-r6f}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 200 python scripts/experiments/attn_stream_debug5.py 2>&1 | grep -v "Warn\|warn\|amdgpu.ids" | cut -c1-300 | tee $OUT/debug5.txt
