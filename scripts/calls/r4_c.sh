#!/bin/bash
# gpurun call C of round 4: fp32 path of the UNet / 2-D VAE / adapter glue, cached fp32 adapter residuals (40-step parity),
# revised RCCL / full-size property tests, secondary bench lines
TAG=${1:-r4c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
echo "== fp32 tests + rccl + stream32 + full-size adapter property"; date
timeout 1200 python -m pytest tests/test_fp32_gpu.py tests/test_rccl_gpu.py tests/test_stream32_gpu.py tests/test_fullsize_gpu.py tests/test_unet_gpu.py -q -p no:cacheprovider --durations=8 > $OUT/pytest_a.log 2>&1; echo "exit $?"; tail -30 $OUT/pytest_a.log | cut -c1-400
echo "== 40-step: cached fp32 adapter, 8-layer, forward"; date
timeout 900 python -m pytest tests/test_fulldepth_gpu.py -q -k "cached or test_forty_step_denoise_vs_oracle or full_depth_full_size_forward" -p no:cacheprovider --durations=8 > $OUT/pytest_b.log 2>&1; echo "exit $?"; tail -12 $OUT/pytest_b.log | cut -c1-300
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null; grep -i "denoise_40\|fp32\|full_depth_forward\|vae" $OUT/gpu_parity.log | cut -c1-600
echo "== secondary bench lines"; date
timeout 300 python bench.py --unet --steps 10 --warmup 3 > $OUT/bench_unet.json 2> $OUT/bench2.err; echo "exit $?"; cut -c1-500 $OUT/bench_unet.json
timeout 400 python bench.py --train --steps 4 --warmup 2 > $OUT/bench_train.json 2>> $OUT/bench2.err; echo "exit $?"; cut -c1-400 $OUT/bench_train.json
timeout 300 python bench.py --train --unet --steps 4 --warmup 2 > $OUT/bench_train_unet.json 2>> $OUT/bench2.err; echo "exit $?"; cut -c1-400 $OUT/bench_train_unet.json
timeout 300 python bench.py --adapter-cache --steps 10 --warmup 3 --no-cpu-baseline --no-text-only-leg > $OUT/bench_adapter_cache.json 2>> $OUT/bench2.err; echo "exit $?"; cut -c1-300 $OUT/bench_adapter_cache.json
tail -3 $OUT/bench2.err
date
