#!/bin/bash
# gpurun call R of round 2: every training test after the explicit-perspective change to forward_train / vt_block_backward
TAG=${1:-r2r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 280 python -m pytest tests/test_train_gpu.py tests/test_fulldepth_gpu.py -x -q --tb=short -p no:cacheprovider -k "not full_depth and not forty and not unet_full" > $OUT/pytest.log 2>&1
echo "exit $?"; tail -3 $OUT/pytest.log; grep -E "^E  |^FAILED|^ERROR" $OUT/pytest.log | head
