#!/bin/bash
# round 6, call T: the FAR instantiation of attn_stream_kernel (segments further apart than the folded offsets reach): its test, the
# attention tests, isolated timing of the default path (unchanged code expected), then the whole suite
export TAG=${1:-r6t}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_round6_gpu.py -q -m gpu -p no:cacheprovider -k "far_apart" 2>&1 | tail -5 | cut -c1-250
for i in 1 2; do timeout 300 python -m pytest tests/test_round5_kernels_gpu.py -q -m gpu -k "attention" -p no:cacheprovider 2>&1 | tail -1 | cut -c1-200; done
timeout 600 python -m pytest tests/test_hip_gpu.py -q -m gpu -k "attention or attn" -p no:cacheprovider 2>&1 | tail -1 | cut -c1-200
timeout 200 python scripts/microbench.py attnr4 2>&1 | grep "attn" | grep "variant\": 32768\|variant\": 0" | cut -c1-150 | tee $OUT/microbench.log
bash scripts/calls/r6_suite.sh ${TAG}_suite --maxfail=5 2>&1 | tail -12
