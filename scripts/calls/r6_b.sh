#!/bin/bash
# round 6, call B: attn_stream_kernel - correctness (the one-wave-per-SIMD tests, now on the streaming kernel) and isolated timing
export TAG=${1:-r6b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== 1. tests"; date
timeout 900 python -m pytest tests/test_round5_kernels_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -15 | tee $OUT/pytest_attention.log
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
echo "== 2. isolated timing"; date
timeout 300 python scripts/microbench.py attnr4 > $OUT/microbench_attention.log 2>&1; cut -c1-200 $OUT/microbench_attention.log | tail -30
date
