#!/bin/bash
# round 6, call P: the V request of a key step issued BEHIND the step's K waits (ST_DMA_LATE) - per tile count: timeline L = 602 / 448, timing
export TAG=${1:-r6p}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/opendwm_amd/variants
for lib in trace trace_dmalate; do
  echo "-- $lib"
  for Lc in 154 0; do
  DWM_HIP_LIB=$V/libdwm_hip_$lib.so timeout 200 python scripts/experiments/attn_trace_stream.py $Lc 0x8000 > $OUT/${lib}_Lc$Lc.txt 2>&1; grep "wave [0123] mean" $OUT/${lib}_Lc$Lc.txt | head -4 | cut -c1-300
  done
done
for rep in 1 2; do
echo "-- default"; timeout 200 python scripts/microbench.py attnr4 2>&1 | grep "attn" | grep "variant\": 32768" | cut -c1-150 | tee -a $OUT/microbench_default.log
echo "-- dmalate"; DWM_HIP_LIB=$V/libdwm_hip_dmalate.so timeout 200 python scripts/microbench.py attnr4 2>&1 | grep "attn" | grep "variant\": 32768" | cut -c1-150 | tee -a $OUT/microbench_dmalate.log
done
