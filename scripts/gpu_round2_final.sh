#!/bin/bash
# gpurun final call of round 2: the whole GPU suite, smoke, the default bench command, rocprofv3 kernel stats of the same command
TAG=${1:-r2final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/gpu_parity.log
echo "== full GPU suite"
timeout 1500 python -m pytest tests -x -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log; grep -E "^E  |^FAILED|^ERROR" $OUT/pytest.log | head -20
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
echo "== bench (default command)"
timeout 900 python bench.py > $OUT/bench.log 2>&1; echo "bench exit $?"; tail -1 $OUT/bench.log | cut -c1-200
echo "== rocprof"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-text-only-leg > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.log 2>&1)
for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/bench_kernel_stats.csv; head -14 $f | cut -c1-150; done
