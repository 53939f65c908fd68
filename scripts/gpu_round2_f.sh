#!/bin/bash
# gpurun call F of round 2: UNet training branch tests
TAG=${1:-r2f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/gpu_parity.log
timeout 900 python -m pytest tests/test_unet_train_gpu.py -q -rA --tb=short -p no:cacheprovider > $OUT/pytest_unet_train.log 2>&1
echo "exit $?"; grep -E "passed|failed|error" $OUT/pytest_unet_train.log | tail -3
grep -E "^E  |^FAILED|^ERROR|Error|error:" $OUT/pytest_unet_train.log | head -40
grep -B2 -A25 "^____" $OUT/pytest_unet_train.log | head -150
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
cat gpurun_out/gpu_parity.log 2>/dev/null | cut -c1-600
