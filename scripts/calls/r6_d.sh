#!/bin/bash
# round 6, call D: attn_stream_kernel - tests, isolated timing, timeline (+ timing builds without the K loads / the V requests)
export TAG=${1:-r6d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== 0. debug script"; date
timeout 200 python scripts/experiments/attn_stream_debug.py 2>&1 | grep -v "Warn\|warn\|amdgpu.ids" | cut -c1-220 | tee $OUT/debug.txt | grep -v "n_bad 0" | head -20
echo "== 1. tests"; date
timeout 900 python -m pytest tests/test_round5_kernels_gpu.py -q -m gpu -k "attention" 2>&1 | tail -8 | cut -c1-300 | tee $OUT/pytest_attention.log
echo "== 2. isolated timing"; date
timeout 300 python scripts/microbench.py attnr4 > $OUT/microbench_attention.log 2>&1; cut -c1-200 $OUT/microbench_attention.log | tail -24
echo "== 3. timeline"; date
for lib in trace xnodma xnok xnone; do
  echo "-- $lib"
  DWM_HIP_LIB=$GRAFT_REPO_ROOT/opendwm_amd/variants/libdwm_hip_$lib.so timeout 120 python scripts/experiments/attn_trace_stream.py 154 0x9000 > $OUT/${lib}_L602.txt 2>&1; tail -4 $OUT/${lib}_L602.txt | cut -c1-400
  DWM_HIP_LIB=$GRAFT_REPO_ROOT/opendwm_amd/variants/libdwm_hip_$lib.so timeout 120 python scripts/experiments/attn_trace_stream.py 0 0x9000 > $OUT/${lib}_L448.txt 2>&1; tail -4 $OUT/${lib}_L448.txt | cut -c1-400
done
date
