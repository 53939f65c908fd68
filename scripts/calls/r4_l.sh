#!/bin/bash
# gpurun call L of round 4: the heavy full-depth cases on the final code (one 40-step full-depth seed, the full-width UNet config),
# a kernel trace of the UNet step, the UNet train step and the configs[4] line
TAG=${1:-r4l}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
echo "== tests"; date
timeout 900 python -m pytest tests/test_fulldepth_gpu.py -q -p no:cacheprovider --durations=5 -k "seed1_4f or unet_full_width_config1 or full_width_train" > $OUT/pytest.log 2>&1; echo "exit $?"; tail -8 $OUT/pytest.log | cut -c1-300
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
echo "== benches"; date
timeout 300 python bench.py --train --unet --steps 4 --warmup 2 > $OUT/bench_train_unet.json 2> $OUT/bench.err; echo "exit $?"; cut -c1-260 $OUT/bench_train_unet.json
timeout 400 python bench.py --tvae-ar > $OUT/bench_tvae_ar.json 2>> $OUT/bench.err; echo "exit $?"; cut -c1-330 $OUT/bench_tvae_ar.json
echo "== UNet trace"; date
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_u -o p -- python $GRAFT_REPO_ROOT/bench.py --unet --steps 3 --warmup 1 > $OUT/unet_under_rocprofv3.log 2>> $OUT/bench.err
echo "exit $?"; f=$(find /tmp/prof_u -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/unet_kernel_stats.csv; head -16 $OUT/unet_kernel_stats.csv | cut -c1-170
date
