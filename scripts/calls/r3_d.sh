#!/bin/bash
# gpurun call D: trace build of the attention kernel + timeline
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3d; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
cp opendwm_amd/libdwm_hip.so /tmp/libdwm_hip.so.keep
DWM_EXTRA_FLAGS=-DDWM_ATTN_TRACE python -m opendwm_amd.build --force > $OUT/build.log 2>&1; echo "build exit $?"
python scripts/experiments/attn_trace.py 154 > $OUT/trace_joint.txt 2>&1; echo "exit $?"; cat $OUT/trace_joint.txt | cut -c1-330
python scripts/experiments/attn_trace.py 0 > $OUT/trace_dual.txt 2>&1; tail -12 $OUT/trace_dual.txt | cut -c1-330
cp /tmp/libdwm_hip.so.keep opendwm_amd/libdwm_hip.so
