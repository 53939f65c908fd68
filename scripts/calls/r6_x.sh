#!/bin/bash
# round 6, call X: the V requests of a key step issued by the waves with fewer query tiles (19 tiles: the 4-tile wave all four pieces; 14 tiles: the 3-tile waves two each): tests, timeline, timing
export TAG=${1:-r6x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/opendwm_amd/variants
DWM_HIP_LIB=$V/libdwm_hip_reb.so timeout 300 python -m pytest tests/test_round5_kernels_gpu.py -q -m gpu -k "attention" -p no:cacheprovider 2>&1 | tail -2 | cut -c1-250
DWM_HIP_LIB=$V/libdwm_hip_reb.so timeout 600 python -m pytest tests/test_hip_gpu.py -q -m gpu -k "attention or attn" -p no:cacheprovider 2>&1 | tail -2 | cut -c1-250
for lib in trace trace_reb; do
  echo "-- $lib"
  for Lc in 154 0; do
  DWM_HIP_LIB=$V/libdwm_hip_$lib.so timeout 200 python scripts/experiments/attn_trace_stream.py $Lc 0x8000 > $OUT/${lib}_Lc$Lc.txt 2>&1; grep "wave [0123] mean" $OUT/${lib}_Lc$Lc.txt | head -4 | cut -c1-300
  done
done
for rep in 1 2; do
echo "-- default"; timeout 200 python scripts/microbench.py attnr4 2>&1 | grep "attn" | grep "variant\": 32768" | cut -c1-150 | tee -a $OUT/microbench_default.log
echo "-- reb"; DWM_HIP_LIB=$V/libdwm_hip_reb.so timeout 200 python scripts/microbench.py attnr4 2>&1 | grep "attn" | grep "variant\": 32768" | cut -c1-150 | tee -a $OUT/microbench_reb.log
done
