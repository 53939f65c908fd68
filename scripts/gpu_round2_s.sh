#!/bin/bash
# gpurun call S of round 2: L2 <-> CU request counters of the attention kernels (one rocprofv3 --pmc pass over the attention microbench)
TAG=${1:-r2s}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmc_$TAG -o p -- python $GRAFT_REPO_ROOT/scripts/microbench.py attn > $OUT/run.log 2>&1
echo "exit $?"; tail -3 $OUT/run.log | cut -c1-200
python - "$OUT" /tmp/pmc_$TAG <<'PY'
import csv, glob, json, sys, collections
out, d = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "attn_fwd_kernel" if "attn_fwd_kernel" in k else "attn_small_kernel" if "attn_small" in k else "attn_group_lds_kernel" if "attn_group_lds" in k else None
        if k is None: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_HIT_sum": calls[k] += 1
res = {k: dict(launches=calls[k], **{c: v / max(calls[k], 1) for c, v in d.items()}) for k, d in agg.items()}
json.dump(res, open(out + "/attn_l2_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
