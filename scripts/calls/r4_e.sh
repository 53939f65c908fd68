#!/bin/bash
# gpurun call E of round 4: the LayerNorm-backward DPP reductions (committed after call Z's snapshot): training tests, train benches,
# kernel traces of the two train steps (launch mix), clock / power samples over a whole bench run
TAG=${1:-r4e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
echo "== training tests"; date
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_unet_train_gpu.py -q -p no:cacheprovider --durations=5 > $OUT/pytest_train.log 2>&1; echo "exit $?"; tail -12 $OUT/pytest_train.log | cut -c1-300
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
echo "== bench --train"; date
timeout 400 python bench.py --train --steps 4 --warmup 2 > $OUT/bench_train.json 2> $OUT/bench.err; echo "exit $?"; cut -c1-330 $OUT/bench_train.json
echo "== bench --train --unet"; date
timeout 400 python bench.py --train --unet --steps 4 --warmup 2 > $OUT/bench_train_unet.json 2>> $OUT/bench.err; echo "exit $?"; cut -c1-330 $OUT/bench_train_unet.json
echo "== clock / power over a bench run"; date
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-text-only-leg > $OUT/bench_clk.log 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power" | tr -s ' ' | tr '\n' ';' >> $OUT/clock_samples.txt; echo >> $OUT/clock_samples.txt
  sleep 1
done
wait $BP
grep '^{' $OUT/bench_clk.log | cut -c1-200; wc -l $OUT/clock_samples.txt
echo "== kernel traces of the train steps"; date
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o p -- python $GRAFT_REPO_ROOT/bench.py --train --steps 3 --warmup 1 > $OUT/train_under_rocprofv3.log 2>> $OUT/bench.err
echo "exit $?"; f=$(find /tmp/prof_t -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/train_kernel_stats.csv; head -14 $OUT/train_kernel_stats.csv | cut -c1-180
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_u -o p -- python $GRAFT_REPO_ROOT/bench.py --train --unet --steps 3 --warmup 1 > $OUT/train_unet_under_rocprofv3.log 2>> $OUT/bench.err
echo "exit $?"; f=$(find /tmp/prof_u -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/train_unet_kernel_stats.csv; head -14 $OUT/train_unet_kernel_stats.csv | cut -c1-180
date
