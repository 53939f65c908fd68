#!/bin/bash
# gpurun call of round 3: GPU clock / power while the headline step runs (rocm-smi samples next to bench.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3clk
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocm-smi --showclocks --showpower --showtemp > $OUT/idle.txt 2>&1
python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-text-only-leg > $OUT/bench.log 2>/dev/null &
BP=$!
sleep 9
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power" | tr -s ' ' | tr '\n' ';' >> $OUT/samples.txt; echo >> $OUT/samples.txt
  sleep 0.5
done
wait $BP
grep '^{' $OUT/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['roofline']['achieved'])"
head -30 $OUT/idle.txt | cut -c1-150
cat $OUT/samples.txt | cut -c1-300
