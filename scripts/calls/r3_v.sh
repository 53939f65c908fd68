#!/bin/bash
# gpurun call of round 3: TN GEMM K-range rule - tests, microbench, train benches
TAG=${1:-r3v}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_train_gpu.py tests/test_unet_train_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "exit $?"; tail -3 $OUT/pytest.log | cut -c1-200
timeout 300 python scripts/microbench.py gemmtn 2>&1 | grep gemm_tn | tee $OUT/gemm_tn.log
for i in 1 2; do
  timeout 400 python bench.py --train --steps 3 --warmup 1 --no-cpu-baseline 2>> $OUT/bench.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('train', d['ms_per_step'], d['value'])
" | tee -a $OUT/bench_ab.log
  timeout 400 python bench.py --train --unet --steps 3 --warmup 1 --no-cpu-baseline 2>> $OUT/bench.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('train unet', d['ms_per_step'], d['value'])
" | tee -a $OUT/bench_ab.log
done
