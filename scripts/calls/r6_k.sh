#!/bin/bash
# round 6, call K: where do two full-size forwards differ (checksums per ops call); the seam test's launch-to-launch difference located
export TAG=${1:-r6k}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
timeout 600 python scripts/experiments/determinism_trace.py > $OUT/determinism_stream.txt 2>&1; tail -12 $OUT/determinism_stream.txt | cut -c1-400
timeout 600 python scripts/experiments/determinism_trace.py 0x2000 > $OUT/determinism_res12.txt 2>&1; tail -12 $OUT/determinism_res12.txt | cut -c1-400
for i in 1 2; do timeout 300 python -m pytest tests/test_round5_kernels_gpu.py -q -m gpu -k "attention" -p no:cacheprovider 2>&1 | tail -2 | cut -c1-200; done
grep "two_launches_differ" gpurun_out/gpu_parity.log | cut -c1-1200
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log
