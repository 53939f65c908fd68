#!/bin/bash
# round 6, call M: 64-byte-run stores for every tile count (the 5-tile form's spills fixed) and 64-byte-run Q loads: tests, timeline, timing
export TAG=${1:-r6m}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/opendwm_amd/variants
for lib in s64all q64; do
  echo "-- tests on $lib"
  DWM_HIP_LIB=$V/libdwm_hip_$lib.so timeout 300 python -m pytest tests/test_round5_kernels_gpu.py -q -m gpu -k "attention" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-250
  DWM_HIP_LIB=$V/libdwm_hip_$lib.so timeout 600 python -m pytest tests/test_hip_gpu.py -q -m gpu -k "attention or attn" -p no:cacheprovider 2>&1 | tail -2 | cut -c1-250
done
for lib in trace trace_s64all trace_q64; do
  echo "-- $lib"
  DWM_HIP_LIB=$V/libdwm_hip_$lib.so timeout 200 python scripts/experiments/attn_trace_stream.py 154 0x8000 > $OUT/${lib}_L602.txt 2>&1; grep "wave [0123] mean" $OUT/${lib}_L602.txt | head -4 | cut -c1-300
done
echo "-- timing: default, s64all, q64, default, s64all, q64"
for rep in 1 2; do
timeout 200 python scripts/microbench.py attnr4 2>&1 | grep "attn" | grep "variant\": 32768" | cut -c1-150 | tee -a $OUT/microbench_default.log
for lib in s64all q64; do
  echo "-- $lib"
  DWM_HIP_LIB=$V/libdwm_hip_$lib.so timeout 200 python scripts/microbench.py attnr4 2>&1 | grep "attn" | grep "variant\": 32768" | cut -c1-150 | tee -a $OUT/microbench_$lib.log
done
done
