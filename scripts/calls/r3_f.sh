#!/bin/bash
# gpurun call F of round 3: the measurement set the docs cite - rocprofv3 kernel trace of the bench command, PMC traffic
# passes, L2 request counters of the attention kernels, VAE roofline line, UNet / train benches
TAG=${1:-r3f}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
echo "== rocprofv3 --kernel-trace --stats of the bench command"; date
timeout 900 python $GRAFT_REPO_ROOT/bench.py > $OUT/bench_default.log 2> $OUT/bench.err; grep '^{' $OUT/bench_default.log | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/bench.log 2>> $OUT/bench.err
echo "exit $?"; grep '^{' $OUT/bench.log | cut -c1-400
f=$(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/bench_kernel_stats.csv
echo "== PMC passes"; date
bash $GRAFT_REPO_ROOT/scripts/pmc_traffic.sh $TAG > $OUT/pmc.log 2>&1; tail -5 $OUT/pmc.log
echo "== attention L2 request counters"; date
cd /tmp
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmca_$TAG -o p -- python $GRAFT_REPO_ROOT/scripts/microbench.py attnx > $OUT/attn_counters_run.log 2>&1
echo "exit $?"; tail -12 $OUT/attn_counters_run.log | cut -c1-200
python - "$OUT" /tmp/pmca_$TAG <<'PY'
import csv, glob, json, sys, collections
out, d = sys.argv[1], sys.argv[2]
# per (kernel, grid, LDS) group = one microbench case; requests are 128 B
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "attn_res_kernel" if "attn_res_kernel" in k else "attn_fwd_kernel" if "attn_fwd_kernel" in k else None
        if k is None: continue
        key = (k, r.get("Grid_Size", "?"), r.get("LDS_Block_Size", "?"))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_HIT_sum": calls[key] += 1
res = [dict(kernel=k[0], grid_size=k[1], lds_block_size=k[2], launches=calls[k], **{c: v / max(calls[k], 1) for c, v in d.items()},
            read_bytes_per_launch=d.get("TCP_TCC_READ_REQ_sum", 0) / max(calls[k], 1) * 128) for k, d in sorted(agg.items())]
json.dump(res, open(out + "/attn_l2_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cd $GRAFT_REPO_ROOT
echo "== VAE"; date
timeout 300 python scripts/vae_bench.py > $OUT/vae_bench.log 2>&1; echo "exit $?"; tail -4 $OUT/vae_bench.log | cut -c1-600
echo "== per-shape GEMM table"; date
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-text-only-leg --gemm-shapes > $OUT/bench_shapes.log 2> $OUT/gemm_shapes.err; grep '^{"M"' $OUT/gemm_shapes.err > $OUT/gemm_shapes.jsonl; wc -l $OUT/gemm_shapes.jsonl
echo "== UNet bench"; date
timeout 300 python bench.py --unet --no-cpu-baseline > $OUT/bench_unet.log 2>> $OUT/bench.err; echo "exit $?"; grep '^{' $OUT/bench_unet.log | cut -c1-300
echo "== train bench"; date
timeout 400 python bench.py --train --no-cpu-baseline > $OUT/bench_train.log 2>> $OUT/bench.err; echo "exit $?"; grep '^{' $OUT/bench_train.log | cut -c1-300
timeout 300 python bench.py --train --unet --no-cpu-baseline > $OUT/bench_train_unet.log 2>> $OUT/bench.err; echo "exit $?"; grep '^{' $OUT/bench_train_unet.log | cut -c1-300
date
