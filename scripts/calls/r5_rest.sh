#!/bin/bash
# the test files behind the first failure of the suite run r5s1 (-x had stopped there)
TAG=${1:-r5rest}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
date
timeout 1200 python -m pytest tests/test_gemm4w_gpu.py tests/test_hip_gpu.py tests/test_rccl_gpu.py tests/test_round5_kernels_gpu.py tests/test_stream32_gpu.py tests/test_train_gpu.py tests/test_unet_gpu.py tests/test_unet_train_gpu.py -q -m gpu -p no:cacheprovider --durations=10 -rs > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log
tail -40 $OUT/pytest.log | cut -c1-250
date
