#!/bin/bash
# gpurun call Q of round 3: the TN weight-gradient GEMM - parity, training suites, A/B of the train steps
TAG=${1:-r3q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== pytest gemm_tn"; date
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "gemm_tn or conv_wgrad or linear_backward" -p no:cacheprovider > $OUT/pytest_tn.log 2>&1; echo "exit $?"; tail -12 $OUT/pytest_tn.log | cut -c1-300
echo "== pytest training suites"; date
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_unet_train_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_train.log 2>&1; echo "exit $?"; tail -8 $OUT/pytest_train.log | cut -c1-300
echo "== train benches: TN / transposed path"; date
for tn in 1 0 1 0; do
  DWM_WGRAD_TN=$tn timeout 400 python bench.py --train --steps 3 --warmup 1 --no-cpu-baseline 2>> $OUT/bench.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('train tn=$tn', d['ms_per_step'], d['value'])
" | tee -a $OUT/bench_ab.log
  DWM_WGRAD_TN=$tn timeout 400 python bench.py --train --unet --steps 3 --warmup 1 --no-cpu-baseline 2>> $OUT/bench.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('train unet tn=$tn', d['ms_per_step'], d['value'])
" | tee -a $OUT/bench_ab.log
done
date
