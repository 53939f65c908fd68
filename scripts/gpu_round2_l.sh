#!/bin/bash
# gpurun call L of round 2: attention with register-staged V (variant bit 6) against the LDS-DMA form: equality, microbench
TAG=${1:-r2l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python scripts/experiments/variant_check.py > $OUT/variant_check.log 2>&1; echo "exit $?"; grep -v amdgpu $OUT/variant_check.log | cut -c1-200
echo "== microbench attnx"
timeout 300 python scripts/microbench.py attnx > $OUT/microbench.log 2>&1; grep '"attn"' $OUT/microbench.log | grep -v "pointwise\|crossview"
