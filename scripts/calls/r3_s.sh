#!/bin/bash
# gpurun call of round 3: multi-tensor AdamW - optimizer / trainer tests + train benches
TAG=${1:-r3s}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_unet_train_gpu.py -m gpu -q -x -k "adamw or train_step or descent or checkpoint or trainer or ddp or repacked" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "exit $?"; tail -4 $OUT/pytest.log | cut -c1-300
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
