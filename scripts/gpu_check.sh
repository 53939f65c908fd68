#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench (+ rocprofv3 kernel stats). Logs under gpurun_out/.
# usage: scripts/gpu_check.sh [tag]
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/gpu_parity.log
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/device.txt
nproc >> $OUT/device.txt
echo "== pytest -m gpu" 
timeout 1500 python -m pytest tests -m gpu -q -n 1 -rA --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest.log
tail -60 $OUT/pytest.log
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
echo "== smoke"
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -5 $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py > $OUT/bench.log 2>&1; echo "bench exit $?" | tee -a $OUT/bench.log; tail -5 $OUT/bench.log
timeout 600 python bench.py --no-cpu-baseline --no-text-only-leg --adapter-cache > $OUT/bench_adapter_cache.log 2>&1; tail -1 $OUT/bench_adapter_cache.log | cut -c1-300
echo "== rocprof"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-text-only-leg > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.log 2>&1)
find /tmp/prof_$TAG -name "*stats*" | head
for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv"); do cp $f $OUT/; done
for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1); do head -12 $f | cut -c1-160; done
echo "== train bench (BASELINE config 4, one GPU)"
timeout 900 python bench.py --train --steps 3 --warmup 1 > $OUT/bench_train.log 2>&1; echo "train bench exit $?" | tee -a $OUT/bench_train.log; tail -2 $OUT/bench_train.log | cut -c1-400
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft_$TAG -o train -- python $GRAFT_REPO_ROOT/bench.py --train --steps 1 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/rocprof_train.log 2>&1)
for f in $(find /tmp/proft_$TAG -name "*kernel_stats*.csv"); do cp $f $OUT/train_kernel_stats.csv; done
head -12 $OUT/train_kernel_stats.csv | cut -c1-200
echo "== unet bench (BASELINE config 2, one GPU)"
timeout 600 python bench.py --unet > $OUT/bench_unet.log 2>&1; echo "unet bench exit $?" | tee -a $OUT/bench_unet.log; tail -2 $OUT/bench_unet.log | cut -c1-400
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profu_$TAG -o unet -- python $GRAFT_REPO_ROOT/bench.py --unet --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof_unet.log 2>&1)
for f in $(find /tmp/profu_$TAG -name "*kernel_stats*.csv"); do cp $f $OUT/unet_kernel_stats.csv; done
head -25 $OUT/unet_kernel_stats.csv | cut -c1-200
