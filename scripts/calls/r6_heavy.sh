#!/bin/bash
# the extended parity cases (DWM_HEAVY_TESTS=1: more seeds / frames / views, all 40 steps of the tVAE window, the UNet's CFG batch) on
# the final code of round 6 - for the record under profiles/, not part of the driver's suite
TAG=${1:-r6heavy}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
date
DWM_HEAVY_TESTS=1 timeout 2000 python -m pytest tests/test_fulldepth_gpu.py -q -m gpu -p no:cacheprovider --durations=15 -k "forty_step_denoise_full_depth or tvae or unet_full_width" > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log
tail -25 $OUT/pytest.log | cut -c1-220
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
date
