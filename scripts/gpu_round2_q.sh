#!/bin/bash
# gpurun call Q of round 2: training with explicit perspective modelling (RayEncoder gradient)
TAG=${1:-r2q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/gpu_parity.log
timeout 300 python -m pytest tests/test_train_gpu.py -q --tb=short -p no:cacheprovider -k "explicit or model_gradients_vs_oracle" > $OUT/pytest.log 2>&1
echo "exit $?"; tail -3 $OUT/pytest.log; grep -E "^E  |^FAILED|^ERROR" $OUT/pytest.log | head
grep explicit gpurun_out/gpu_parity.log | cut -c1-500
