#!/bin/bash
# round 6, call G: attn_stream_kernel as the default - attention tests of the suite, isolated timing (default / 12-wave), timeline, in-bench
export TAG=${1:-r6g}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 300 python -m pytest tests/test_round5_kernels_gpu.py -q -m gpu -k "attention" 2>&1 | tail -3 | cut -c1-200; done
timeout 600 python -m pytest tests/test_hip_gpu.py -q -m gpu -k "attention or attn" 2>&1 | tail -3 | cut -c1-200
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
grep "item_seams" $OUT/gpu_parity.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d.get('I'),d.get('N'),d.get('Lc'),d.get('hs'),'two launches',d.get('rel_between_two_launches'))" | sort | uniq -c
echo "-- timing (variants: 8192 = 12-wave, 0 = default (stream), 32768 = stream + prescaled q, 40960 = 12-wave + prescaled)"
timeout 200 python scripts/microbench.py attnr4 2>&1 | grep "attn" | cut -c1-150 | tee $OUT/microbench_attention.log
for lib in trace; do
  echo "-- $lib"
  DWM_HIP_LIB=$GRAFT_REPO_ROOT/opendwm_amd/variants/libdwm_hip_$lib.so timeout 120 python scripts/experiments/attn_trace_stream.py 154 0x8000 > $OUT/${lib}_L602.txt 2>&1; tail -4 $OUT/${lib}_L602.txt | grep "wave [03]" | cut -c1-400
  DWM_HIP_LIB=$GRAFT_REPO_ROOT/opendwm_amd/variants/libdwm_hip_$lib.so timeout 120 python scripts/experiments/attn_trace_stream.py 0 0x8000 > $OUT/${lib}_L448.txt 2>&1; tail -4 $OUT/${lib}_L448.txt | grep "wave [03]" | cut -c1-400
done
echo "-- store32 variant (four stores of 32 rows x 32 bytes per tile instead of 16 rows x 64 bytes)"
DWM_HIP_LIB=$GRAFT_REPO_ROOT/opendwm_amd/variants/libdwm_hip_store32.so timeout 200 python scripts/microbench.py attnr4 2>&1 | grep "attn" | grep "variant\": 0\|variant\": 32768" | cut -c1-150 | tee $OUT/microbench_attention_store32.log
bash scripts/calls/r6_h.sh ${TAG}_bench 2>&1 | grep -v "^exit\|UTC"
