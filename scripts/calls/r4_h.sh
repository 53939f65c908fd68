#!/bin/bash
# gpurun call H of round 4: attention backward with opaque LDS-DMA requests and software-pipelined fragment reads (training tests,
# train bench, kernel trace of the train step), the fp32 adapter without zero convolutions, the DDP-over-RCCL test
TAG=${1:-r4h}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_unet_train_gpu.py tests/test_rccl_gpu.py "tests/test_fp32_gpu.py::test_model_with_layout_adapter_fp32_vs_cpu_oracle" -q -p no:cacheprovider --durations=5 > $OUT/pytest.log 2>&1; echo "exit $?"; tail -12 $OUT/pytest.log | cut -c1-300
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
echo "== bench --train"; date
timeout 400 python bench.py --train --steps 4 --warmup 2 > $OUT/bench_train.json 2> $OUT/bench.err; echo "exit $?"; cut -c1-330 $OUT/bench_train.json
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o p -- python $GRAFT_REPO_ROOT/bench.py --train --steps 3 --warmup 1 > $OUT/train_under_rocprofv3.log 2>> $OUT/bench.err
echo "exit $?"; f=$(find /tmp/prof_t -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/train_kernel_stats.csv; grep -i "attn\|layernorm_bwd" $OUT/train_kernel_stats.csv | cut -c1-200
date
