#!/bin/bash
# usage: scripts/prof_cmd.sh <tag> <python args...>  -> gpurun_out/<tag>_kernel_stats.txt (top kernels, names trimmed)
TAG=$1; shift
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python "$@" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_run.log 2>&1
f=$(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1)
python - "$f" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total_ms", tot / 1e6)
for r in rows[:28]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {float(r["Percentage"]):5.1f}')
PY
