#!/bin/bash
# gpurun call M of round 4: the remaining full-depth cases on the final code (everything of tests/test_fulldepth_gpu.py that calls K and
# L did not run, except the 5-minute temporal-VAE window test, whose parts were re-run in calls G and L)
TAG=${1:-r4m}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
date
timeout 640 python -m pytest tests/test_fulldepth_gpu.py -x -q -p no:cacheprovider --durations=8 -k "not tvae and not seed1_4f and not unet_full_width_config1 and not full_width_train and not full_depth_full_size_forward and not forty_step_denoise_vs_oracle_loop" > $OUT/pytest.log 2>&1; echo "exit $?"; tail -14 $OUT/pytest.log | cut -c1-300
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
date
