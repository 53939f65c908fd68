#!/bin/bash
# gpurun call P of round 2: UNet inference + training tests after the last host-side edits (TextContext.from_rows, protocol methods)
TAG=${1:-r2p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/gpu_parity.log
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_unet_train_gpu.py -x -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "exit $?"; tail -3 $OUT/pytest.log; grep -E "^E  |^FAILED|^ERROR" $OUT/pytest.log | head
