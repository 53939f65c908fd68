#!/bin/bash
# evidence call on the final code of round 6: the headline bench (as the driver runs it), the same command under rocprofv3 (kernel stats)
# and under the PMC passes (HBM traffic + MFMA busy per kernel), the other bench lines (train, UNet, tVAE window job), smoke()
export TAG=${1:-r6z}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
summ() { python - "$1" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("  ms/step", round(d["ms_per_step"], 2), "gemm", round(d["roofline"]["achieved"] or 0, 1), "TFLOP/s", "attn_res frac", round(d["roofline_attention"]["frac"], 4),
              "by_kernel", {k: (round(v["tflops"], 1), round(v["avg_us"], 1), v["launches"]) for k, v in (d["roofline"].get("by_kernel") or {}).items()})
PY
}
echo "== 1. headline bench, default flags"; date
timeout 900 python bench.py --gemm-shapes > $OUT/bench.json 2> $OUT/gemm_shapes.jsonl; echo "exit $?"; summ $OUT/bench.json
echo "== 2. rocprofv3 kernel stats of the headline leg"; date
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-text-only-leg > $OUT/bench_under_rocprofv3.json 2> $OUT/bench_under_rocprofv3.err )
f=$(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/rocprofv3_kernel_stats.csv 2>/dev/null
head -10 $OUT/rocprofv3_kernel_stats.csv | cut -c1-200; summ $OUT/bench_under_rocprofv3.json
echo "== 3. PMC: HBM traffic + MFMA busy per kernel"; date
timeout 800 bash scripts/pmc_traffic.sh $TAG > $OUT/pmc_traffic.log 2>&1; python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", os.environ.get("TAG", "r6z"), "pmc_traffic.json")))
for k, v in d.items(): print(k, v["launches"], round(v["hbm_bytes_per_launch"] / 1e9, 3), "GB L2->fabric;", "DRAM share r/w", (v.get("dram") or {}).get("read_request_fraction_to_dram"), (v.get("dram") or {}).get("write_request_fraction_to_dram"), "mfma busy", v.get("mfma", {}).get("mfma_pipe_utilisation"))
PY
echo "== 4. other lines"; date
timeout 400 python bench.py --train --steps 4 --warmup 2 > $OUT/bench_train.json 2> $OUT/bench_train.err; grep '^{' $OUT/bench_train.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  train', d['value'], d['unit'], round(d['ms_per_step'],2), 'ms', d.get('approx_mfma_frac'))"
timeout 400 python bench.py --train --preflight --steps 2 --warmup 1 > $OUT/bench_train_ddp1.json 2> $OUT/bench_train_ddp1.err; grep '^{' $OUT/bench_train_ddp1.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); t=d['config']['ddp']['bucket_timeline']; print('  ddp timeline', {k: t[k] for k in t if k.startswith(('fp32','bf16','backward_ms','buckets'))})"
timeout 300 python bench.py --unet --steps 10 --warmup 3 > $OUT/bench_unet.json 2> $OUT/bench_unet.err; grep '^{' $OUT/bench_unet.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  unet', d['value'], d['unit'], round(d['ms_per_step'],2), 'ms')"
timeout 300 python bench.py --unet --graph --steps 10 --warmup 3 > $OUT/bench_unet_graph.json 2> $OUT/bench_unet_graph.err; grep '^{' $OUT/bench_unet_graph.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  unet (HIP graph)', d['value'], d['unit'], round(d['ms_per_step'],2), 'ms')"
timeout 300 python bench.py --unet --train --steps 4 --warmup 2 > $OUT/bench_unet_train.json 2> $OUT/bench_unet_train.err; grep '^{' $OUT/bench_unet_train.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  unet train', d['value'], d['unit'], round(d['ms_per_step'],2), 'ms')"
timeout 400 python bench.py --tvae-ar > $OUT/bench_tvae_ar.json 2> $OUT/bench_tvae_ar.err; grep '^{' $OUT/bench_tvae_ar.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  tvae-ar', d['value'], d['unit'], round(d['ms_per_step'],2), 'ms')"
timeout 200 python scripts/microbench.py attnr4 attnfull attnunet cv pw > $OUT/microbench_attention.log 2>&1; cut -c1-160 $OUT/microbench_attention.log | tail -22
echo "== 4b. full-depth training gradients against the oracle's autograd (DWM_HEAVY_TESTS case)"; date
DWM_HEAVY_TESTS=1 timeout 900 python -m pytest tests/test_fulldepth_gpu.py -q -m gpu -p no:cacheprovider -k "train_gradients" > $OUT/full_depth_train_gradients.log 2>&1; tail -3 $OUT/full_depth_train_gradients.log | cut -c1-200
grep full_width_train_gradients gpurun_out/gpu_parity.log | cut -c1-600 | tee -a $OUT/full_depth_train_gradients.log
echo "== 5. smoke"; date
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
date
