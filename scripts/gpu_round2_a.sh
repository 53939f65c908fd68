#!/bin/bash
# gpurun call A of round 2: CPU full-size oracle step in the background, GPU parity tests, smoke, bench (+ rocprofv3), train / unet bench.
TAG=${1:-r2a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/gpu_parity.log
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/device.txt
nproc >> $OUT/device.txt; free -g | head -2 >> $OUT/device.txt
NT=$(( $(nproc) > 16 ? $(nproc) - 8 : $(nproc) ))
(timeout 2400 python scripts/cpu_full_step.py --threads $NT > $OUT/cpu_full_step.json 2> $OUT/cpu_full_step.err) &
CPUJOB=$!
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -q -n 1 -rA --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -5
grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
echo "== smoke"
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
echo "== train bench"
timeout 900 python bench.py --train --steps 3 --warmup 1 > $OUT/bench_train.log 2>&1; echo "train bench exit $?"; tail -1 $OUT/bench_train.log | cut -c1-600
echo "== unet bench"
timeout 600 python bench.py --unet --no-cpu-baseline > $OUT/bench_unet.log 2>&1; echo "unet bench exit $?"; tail -1 $OUT/bench_unet.log | cut -c1-600
echo "== microbench"
timeout 600 python scripts/microbench.py > $OUT/micro.log 2>&1; tail -30 $OUT/micro.log
echo "== wait for the CPU job"
wait $CPUJOB; cat $OUT/cpu_full_step.json | cut -c1-600; tail -3 $OUT/cpu_full_step.err
echo "== bench"
timeout 900 python bench.py > $OUT/bench.log 2>&1; echo "bench exit $?" | tee -a $OUT/bench.log; tail -1 $OUT/bench.log | cut -c1-1500
echo "== rocprof"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-text-only-leg > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.log 2>&1)
for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv"); do cp $f $OUT/; done
for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1); do head -14 $f | cut -c1-200; done
