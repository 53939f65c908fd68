#!/bin/bash
# round 6, call C: timeline of attn_stream_kernel (trace build)
export TAG=${1:-r6c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
DWM_HIP_LIB=$GRAFT_REPO_ROOT/opendwm_amd/variants/libdwm_hip_trace.so timeout 120 python scripts/experiments/attn_trace_stream.py 154 > $OUT/trace_L602.txt 2>&1; tail -14 $OUT/trace_L602.txt | cut -c1-400
DWM_HIP_LIB=$GRAFT_REPO_ROOT/opendwm_amd/variants/libdwm_hip_trace.so timeout 120 python scripts/experiments/attn_trace_stream.py 0 > $OUT/trace_L448.txt 2>&1; tail -5 $OUT/trace_L448.txt | cut -c1-400
