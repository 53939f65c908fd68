#!/bin/bash
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp
ARGS="$@"
pass() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc2_${TAG}_$n -o p -- python $GRAFT_REPO_ROOT/scripts/one_kernel.py $ARGS > $OUT/pass$n.log 2>&1; f=$(find /tmp/pmc2_${TAG}_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY' | tee -a $OUT/summary2.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:50]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "attn" in k or "gemm" in k:
        print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
}
pass 1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass 2 FETCH_SIZE
pass 3 WRITE_SIZE TCC_EA0_RDREQ_sum
pass 4 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
