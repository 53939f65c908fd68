#!/bin/bash
# gpurun call J of round 2: shared-LDS group attention (row-wise cross-view): unit test of the three forms, microbench, model tests that use it
TAG=${1:-r2j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/gpu_parity.log
timeout 300 python -m pytest tests/test_hip_gpu.py -q --tb=short -p no:cacheprovider -k "group_forms or rowmaps" > $OUT/pytest_group.log 2>&1
echo "exit $?"; tail -3 $OUT/pytest_group.log; grep -E "^E |^FAILED" $OUT/pytest_group.log | head
grep attention_group_forms gpurun_out/gpu_parity.log | cut -c1-220
echo "== microbench cv"
timeout 200 python scripts/microbench.py cv > $OUT/microbench.log 2>&1; grep crossview $OUT/microbench.log
echo "== full-size property tests (V = 6)"
timeout 600 python -m pytest tests/test_fullsize_gpu.py tests/test_fulldepth_gpu.py -q --tb=short -p no:cacheprovider -k "full_depth or fullsize or full_size" > $OUT/pytest_full.log 2>&1
echo "exit $?"; tail -3 $OUT/pytest_full.log; grep -E "^E |^FAILED" $OUT/pytest_full.log | head
grep full_depth gpurun_out/gpu_parity.log | cut -c1-200
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
