#!/bin/bash
# gpurun call Z of round 3: final secondary bench lines (UNet, train steps) on the final code + single-rank preflight
TAG=${1:-r3z}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --unet --no-cpu-baseline > $OUT/bench_unet.log 2>> $OUT/bench.err; echo "exit $?"; grep '^{' $OUT/bench_unet.log | cut -c1-200
timeout 400 python bench.py --train --no-cpu-baseline > $OUT/bench_train.log 2>> $OUT/bench.err; echo "exit $?"; grep '^{' $OUT/bench_train.log | cut -c1-200
timeout 300 python bench.py --train --unet --no-cpu-baseline > $OUT/bench_train_unet.log 2>> $OUT/bench.err; echo "exit $?"; grep '^{' $OUT/bench_train_unet.log | cut -c1-200
timeout 300 python bench.py --preflight --steps 1 --warmup 1 --no-cpu-baseline --no-text-only-leg > $OUT/bench_preflight.log 2>> $OUT/bench.err; echo "exit $?"; grep -o '"preflight": {[^}]*}' $OUT/bench_preflight.log
