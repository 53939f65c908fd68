#!/bin/bash
# gpurun call A of round 4: fp32 residual streams + attention refill points - kernel / block / model tests, micro-benchmarks
# (A/B by variant / stream), the bench in both stream modes, the 40-step parity cases
TAG=${1:-r4a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
echo "== tests (everything but the full-depth file)"; date
timeout 1200 python -m pytest tests -m gpu -q -x --ignore=tests/test_fulldepth_gpu.py -p no:cacheprovider --durations=8 > $OUT/pytest_main.log 2>&1; echo "exit $?"; tail -25 $OUT/pytest_main.log | cut -c1-300
echo "== microbench"; date
timeout 400 python scripts/microbench.py attnr s32 > $OUT/microbench.log 2>&1; echo "exit $?"; cut -c1-220 $OUT/microbench.log
echo "== bench, fp32 streams (default)"; date
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_fp32_stream.json 2> $OUT/bench.err; echo "exit $?"; cut -c1-3000 $OUT/bench_fp32_stream.json; tail -3 $OUT/bench.err
echo "== bench, bf16 streams"; date
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --residual-bf16 > $OUT/bench_bf16_stream.json 2>> $OUT/bench.err; echo "exit $?"; cut -c1-600 $OUT/bench_bf16_stream.json
echo "== bench, fp32 streams again (order effects)"; date
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-text-only-leg > $OUT/bench_fp32_stream_2.json 2>> $OUT/bench.err; echo "exit $?"; cut -c1-400 $OUT/bench_fp32_stream_2.json
echo "== 40-step parity"; date
timeout 1500 python -m pytest tests/test_fulldepth_gpu.py -q -k "forty_step and (layout or heavy)" -p no:cacheprovider --durations=8 > $OUT/pytest_fulldepth.log 2>&1; echo "exit $?"; tail -14 $OUT/pytest_fulldepth.log | cut -c1-300
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null; grep "denoise_40\|stream" $OUT/gpu_parity.log | cut -c1-600
date
