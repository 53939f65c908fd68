#!/bin/bash
# gpurun call M of round 2: layout adapter backward generalised to pooling at any block (SD 2.1 UNet training with the adapter)
TAG=${1:-r2m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/gpu_parity.log
timeout 900 python -m pytest tests/test_unet_train_gpu.py tests/test_train_gpu.py -q --tb=short -p no:cacheprovider -k "adapter or unet_gradients or trainer" > $OUT/pytest.log 2>&1
echo "exit $?"; tail -3 $OUT/pytest.log; grep -E "^E  |^FAILED|^ERROR" $OUT/pytest.log | head -20
grep -E "adapter" gpurun_out/gpu_parity.log | cut -c1-500
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
