#!/bin/bash
# gpurun call Z of round 4: the whole GPU suite, smoke, the default bench, and the measurement set the docs cite (rocprofv3 kernel
# trace of the bench command, PMC traffic passes, clock / power samples, per-shape GEMM table, attention microbench)
TAG=${1:-r4z}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
echo "== pytest -m gpu"; date
timeout 2700 python -m pytest tests -m gpu -q -rf --tb=short -p no:cacheprovider --durations=20 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log
tail -40 $OUT/pytest.log | cut -c1-250
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
echo "== smoke"; date
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
echo "== bench (default flags)"; date
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-3000 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== clock / power during the bench"; date
rocm-smi --showclocks --showpower --showtemp > $OUT/clock_idle.txt 2>&1
python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-text-only-leg > $OUT/bench_clk.log 2>/dev/null &
BP=$!
sleep 14
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power" | tr -s ' ' | tr '\n' ';' >> $OUT/clock_samples.txt; echo >> $OUT/clock_samples.txt
  sleep 0.5
done
wait $BP
grep '^{' $OUT/bench_clk.log | cut -c1-200; head -4 $OUT/clock_samples.txt | cut -c1-300
echo "== rocprofv3 --kernel-trace --stats of the bench command"; date
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprofv3.log 2>> $OUT/bench.err
echo "exit $?"; grep '^{' $OUT/bench_under_rocprofv3.log | cut -c1-300
f=$(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/bench_kernel_stats.csv; head -12 $OUT/bench_kernel_stats.csv | cut -c1-200
echo "== PMC passes"; date
bash $GRAFT_REPO_ROOT/scripts/pmc_traffic.sh $TAG > $OUT/pmc.log 2>&1; tail -30 $OUT/pmc.log | cut -c1-200
cd $GRAFT_REPO_ROOT
echo "== per-shape GEMM table"; date
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-text-only-leg --gemm-shapes > $OUT/bench_shapes.log 2> $OUT/gemm_shapes.err; grep '^{"M"' $OUT/gemm_shapes.err > $OUT/gemm_shapes.jsonl; wc -l $OUT/gemm_shapes.jsonl
echo "== attention microbench"; date
timeout 200 python scripts/microbench.py attnr > $OUT/microbench_attn.log 2>&1; cut -c1-200 $OUT/microbench_attn.log
date
