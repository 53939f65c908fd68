#!/bin/bash
# gpurun call N of round 3: kernel traces of the secondary workloads (UNet inference, SD 3.5 train step, UNet train step)
TAG=${1:-r3n}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for w in "train --train" "train_unet --train --unet"; do
  set -- $w; name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$name -o p -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_$name.log 2> $OUT/bench_$name.err
  echo "$name exit $?"; grep '^{' $OUT/bench_$name.log | cut -c1-200
  f=$(find /tmp/prof_${TAG}_$name -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/${name}_kernel_stats.csv
  python - "$OUT/${name}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total_ms", tot / 1e6)
for r in rows[:22]:
    print(f'{r["Name"][:84]:84s} calls {int(r["Calls"]):6d} ms {float(r["TotalDurationNs"])/1e6:9.2f} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {float(r["Percentage"]):5.1f}')
PY
done
