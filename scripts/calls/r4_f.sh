#!/bin/bash
# gpurun call F of round 4: the fp32 mode with explicit perspective modelling and with frame sharding (new), the ray-feature kernel
# in both precisions, the UNet gradient test with the conditioning-scaled bound
TAG=${1:-r4f}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
timeout 900 python -m pytest tests/test_fp32_gpu.py tests/test_unet_train_gpu.py "tests/test_hip_gpu.py::test_explicit_perspective_forward_vs_oracle_and_reference_fixture" \
  "tests/test_train_gpu.py::test_model_gradients_explicit_perspective_vs_oracle" -q -p no:cacheprovider --durations=8 -k "explicit or frame_shard or unet_gradients or model_forward_fp32" > $OUT/pytest.log 2>&1; echo "exit $?"; tail -25 $OUT/pytest.log | cut -c1-400
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
grep -i "explicit\|frame_shard" $OUT/gpu_parity.log | cut -c1-400
