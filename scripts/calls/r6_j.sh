#!/bin/bash
# round 6, call J: is the store time at the head seam of attn_stream_kernel a per-CU limit or a chip-wide one?  Timeline with 256 / 128 / 32
# workgroups (same work per workgroup-head), and with workgroup classes started apart (stagger); isolated timing of the stagger builds
export TAG=${1:-r6j}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/opendwm_amd/variants
for lib in trace trace_nblk128 trace_nblk32 trace_stag5x4k trace_stag5x11k; do
  echo "-- $lib"
  DWM_HIP_LIB=$V/libdwm_hip_$lib.so timeout 200 python scripts/experiments/attn_trace_stream.py 154 0x8000 > $OUT/${lib}_L602.txt 2>&1; grep "wave [0123] mean" $OUT/${lib}_L602.txt | head -4 | cut -c1-300
done
echo "-- timing: default build, then stagger builds (variants 0 = stream, 32768 = stream + prescaled q)"
timeout 200 python scripts/microbench.py attnr4 2>&1 | grep "attn" | grep "variant\": 0\|variant\": 32768" | cut -c1-150 | tee $OUT/microbench_default.log
for lib in stag5x4k stag5x11k stag8x2k stag4x6k; do
  echo "-- $lib"
  DWM_HIP_LIB=$V/libdwm_hip_$lib.so timeout 200 python scripts/microbench.py attnr4 2>&1 | grep "attn" | grep "variant\": 0\|variant\": 32768" | cut -c1-150 | tee $OUT/microbench_$lib.log
done
echo "-- back-to-back launches against synchronised ones (stream kernel, then the 12-wave kernel)"
timeout 300 python scripts/experiments/attn_stream_debug6.py 0 > $OUT/debug6_stream.txt 2>&1; cut -c1-400 $OUT/debug6_stream.txt | tail -40
timeout 300 python scripts/experiments/attn_stream_debug6.py 0x2000 > $OUT/debug6_res12.txt 2>&1; cut -c1-400 $OUT/debug6_res12.txt | tail -12
