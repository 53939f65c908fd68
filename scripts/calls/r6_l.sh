#!/bin/bash
# round 6, call L: what do the output stores cost UNDER the key-step loop?  Timing builds (wrong results): no stores at the head seam,
# 1 / 2 / 4 one-KiB stores per key step instead (19 / 38 / 76 per head at L = 602; the real number is 20 per 5-tile wave)
export TAG=${1:-r6l}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/opendwm_amd/variants
for lib in trace trace_loopstore; do
  echo "-- $lib"
  DWM_HIP_LIB=$V/libdwm_hip_$lib.so timeout 200 python scripts/experiments/attn_trace_stream.py 154 0x8000 > $OUT/${lib}_L602.txt 2>&1; grep "wave [0123] mean" $OUT/${lib}_L602.txt | head -4 | cut -c1-300
  DWM_HIP_LIB=$V/libdwm_hip_$lib.so timeout 200 python scripts/experiments/attn_trace_stream.py 0 0x8000 > $OUT/${lib}_L448.txt 2>&1; grep "wave [0123] mean" $OUT/${lib}_L448.txt | head -4 | cut -c1-300
done
echo "-- timing: default build"
timeout 200 python scripts/microbench.py attnr4 2>&1 | grep "attn" | grep "variant\": 0\|variant\": 32768" | cut -c1-150 | tee $OUT/microbench_default.log
for lib in loopstore1 loopstore2 loopstore4; do
  echo "-- $lib"
  DWM_HIP_LIB=$V/libdwm_hip_$lib.so timeout 200 python scripts/microbench.py attnr4 2>&1 | grep "attn" | grep "variant\": 0\|variant\": 32768" | cut -c1-150 | tee $OUT/microbench_$lib.log
done
echo "-- deferred stores (real): attention tests on the variant library, timeline, timing"
DWM_HIP_LIB=$V/libdwm_hip_def.so timeout 300 python -m pytest tests/test_round5_kernels_gpu.py -q -m gpu -k "attention" -p no:cacheprovider 2>&1 | tail -4 | cut -c1-250
DWM_HIP_LIB=$V/libdwm_hip_def.so timeout 600 python -m pytest tests/test_hip_gpu.py -q -m gpu -k "attention or attn" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-250
for lib in trace_def; do
  DWM_HIP_LIB=$V/libdwm_hip_$lib.so timeout 200 python scripts/experiments/attn_trace_stream.py 154 0x8000 > $OUT/${lib}_L602.txt 2>&1; grep "wave [0123] mean" $OUT/${lib}_L602.txt | head -4 | cut -c1-300
  DWM_HIP_LIB=$V/libdwm_hip_$lib.so timeout 200 python scripts/experiments/attn_trace_stream.py 0 0x8000 > $OUT/${lib}_L448.txt 2>&1; grep "wave [0123] mean" $OUT/${lib}_L448.txt | head -4 | cut -c1-300
done
DWM_HIP_LIB=$V/libdwm_hip_def.so timeout 200 python scripts/microbench.py attnr4 2>&1 | grep "attn" | grep "variant\": 0\|variant\": 32768" | cut -c1-150 | tee $OUT/microbench_def.log
