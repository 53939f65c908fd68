#!/bin/bash
# gpurun call C of round 2: fp32-path tests, the kernel-variant experiments (attention tile body, GEMM schedule),
# the UNet config-1 block-by-block bisect, rerun of the test that failed in call A
TAG=${1:-r2c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/gpu_parity.log
echo "== variant check"
timeout 300 python scripts/experiments/variant_check.py > $OUT/variant_check.log 2>&1; echo "exit $?"; tail -40 $OUT/variant_check.log
echo "== microbench attnx gemmx"
timeout 400 python scripts/microbench.py attnx gemmx > $OUT/microbench.log 2>&1; echo "exit $?"; cat $OUT/microbench.log | cut -c1-400
echo "== fp32 tests"
timeout 900 python -m pytest tests/test_fp32_gpu.py -q -rA --tb=short -p no:cacheprovider -x > $OUT/pytest_fp32.log 2>&1
echo "exit $?"; grep -E "passed|failed|error" $OUT/pytest_fp32.log | tail -3; grep -E "^E |^FAILED|Error" $OUT/pytest_fp32.log | head -30
echo "== unet bisect"
for args in "6 32 56" "6 32 56 32"; do
  echo "-- unet_bisect $args"
  timeout 500 python scripts/unet_bisect.py $args > "$OUT/bisect_${args// /_}.log" 2>&1
  grep -E "<<<<|final" "$OUT/bisect_${args// /_}.log" | head -12
  tail -3 "$OUT/bisect_${args// /_}.log"
done
echo "== rerun of the failed test"
timeout 600 python -m pytest tests/test_fulldepth_gpu.py -q -rA --tb=short -p no:cacheprovider -k "train_gradients" > $OUT/pytest_rerun.log 2>&1
grep -E "passed|failed" $OUT/pytest_rerun.log | tail -2
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
