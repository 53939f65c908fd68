#!/bin/bash
# gpurun call I of round 4: cross-view kernel with pipelined K-fragment reads + opaque copy requests (tests, microbench), and the
# experiment "opaque LDS-DMA requests in the tiled / resident forward kernels too" (library B) against the default library
TAG=${1:-r4i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
echo "== attention tests, default library"; date
timeout 900 python -m pytest tests/test_hip_gpu.py tests/test_unet_gpu.py -q -p no:cacheprovider -k "attention or group or crossview or cross_view" > $OUT/pytest_a.log 2>&1; echo "exit $?"; tail -4 $OUT/pytest_a.log | cut -c1-300
echo "== microbench, default library"; date
timeout 300 python scripts/microbench.py cv attnr attnx > $OUT/micro_a.log 2>&1; grep '^{' $OUT/micro_a.log | cut -c1-200
echo "== microbench, library B"; date
DWM_HIP_LIB=$GRAFT_REPO_ROOT/opendwm_amd/libdwm_hip_b.so timeout 300 python scripts/microbench.py attnr attnx > $OUT/micro_b.log 2>&1; grep '^{' $OUT/micro_b.log | cut -c1-200
echo "== attention tests, library B"; date
DWM_HIP_LIB=$GRAFT_REPO_ROOT/opendwm_amd/libdwm_hip_b.so timeout 900 python -m pytest tests/test_hip_gpu.py tests/test_unet_gpu.py tests/test_train_gpu.py -q -p no:cacheprovider -k "attention" > $OUT/pytest_b.log 2>&1; echo "exit $?"; tail -4 $OUT/pytest_b.log | cut -c1-300
date
