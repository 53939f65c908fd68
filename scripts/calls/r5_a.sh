#!/bin/bash
# gpurun call A of round 5 (verdict item 1): make the driver line reproducible from profiles/ and settle the code written without a GPU.
#   1. rocprofv3 --kernel-trace --stats of the headline bench (gemm4w_kernel rows) + the bench's own by_kernel averages of the same run
#   2. PMC passes of the same command: HBM traffic + MFMA busy per kernel (gemm4w_kernel, gemm_bf16_kernel, attention, layernorm)
#   3. the gated tests of the kernels written without a GPU (tests/test_unvalidated_gpu.py)
#   4. attention microbench: resident kernel against its paired form; "full" temporal attention at L = 8512
#   5. bench A/B on this box: default | DWM_ATTN_RES2=1 | DWM_GEMM4W=2 | --stack-modulation | default
#   6. other models on the 4-wave kernels (train step, UNet)
# (no full suite here: the driver runs it)
# usage: gpurun --timeout 1500 -- 'bash scripts/calls/r5_a.sh'
TAG=${1:-r5a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
summ() { python - "$1" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("  ms/step", round(d["ms_per_step"], 2), "gemm", round(d["roofline"]["achieved"] or 0, 1), "TFLOP/s", "attn_res frac", round(d["roofline_attention"]["frac"], 4),
              "by_kernel", {k: (round(v["tflops"], 1), round(v["avg_us"], 1), v["launches"]) for k, v in (d["roofline"].get("by_kernel") or {}).items()})
PY
}
echo "== 1. rocprofv3 kernel stats of the headline bench"; date
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-text-only-leg > $OUT/bench_under_rocprofv3.json 2> $OUT/bench_under_rocprofv3.err )
f=$(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/rocprofv3_kernel_stats.csv 2>/dev/null
head -12 $OUT/rocprofv3_kernel_stats.csv | cut -c1-220; summ $OUT/bench_under_rocprofv3.json
echo "== 2. PMC: HBM traffic + MFMA busy per kernel"; date
timeout 500 bash scripts/pmc_traffic.sh $TAG > $OUT/pmc_traffic.log 2>&1; tail -60 $OUT/pmc_traffic.log | cut -c1-200
echo "== 3. gated tests (DWM_TEST_UNVALIDATED=1)"; date
DWM_TEST_UNVALIDATED=1 timeout 600 python -m pytest tests/test_unvalidated_gpu.py -m gpu -q -rf --tb=short -p no:cacheprovider > $OUT/pytest_unvalidated.log 2>&1
echo "exit $?" | tee -a $OUT/pytest_unvalidated.log; tail -30 $OUT/pytest_unvalidated.log | cut -c1-300
cp gpurun_out/gpu_parity.log $OUT/gpu_parity_unvalidated.log 2>/dev/null
echo "== 4. attention microbench: variant 0 (attn_res_kernel<12>) against 64 (attn_res2_kernel); full temporal L=8512"; date
timeout 200 python scripts/microbench.py attnr2 > $OUT/microbench_attn_res2.log 2>&1; cut -c1-200 $OUT/microbench_attn_res2.log
timeout 200 python scripts/microbench.py attnfull > $OUT/microbench_attn_full.log 2>&1; cut -c1-200 $OUT/microbench_attn_full.log
echo "== SQ counters of the joint attention: resident kernel (variant 0) against its paired form (64)"; date
for v in 0 64; do timeout 200 bash scripts/pmc.sh ${TAG}_attn_joint_v$v attn_joint $v > $OUT/pmc_attn_joint_v$v.log 2>&1; cp gpurun_out/pmc_${TAG}_attn_joint_v$v/summary.txt $OUT/pmc_attn_joint_v$v.txt 2>/dev/null; cut -c1-400 $OUT/pmc_attn_joint_v$v.txt; done
echo "== 5. bench A/B"; date
for cfg in "default:" "res2:DWM_ATTN_RES2=1" "gemm4wgen:DWM_GEMM4W=2" "default2:"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-text-only-leg > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name exit $?"; summ $OUT/bench_$name.json
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-text-only-leg --stack-modulation > $OUT/bench_stackmod.json 2> $OUT/bench_stackmod.err
echo "stack-modulation exit $?"; summ $OUT/bench_stackmod.json
echo "== per-shape GEMM table inside the bench, default mix"; date
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-text-only-leg --gemm-shapes > $OUT/bench_shapes_mix.log 2> $OUT/gemm_shapes_mix.err
grep '^{"M"' $OUT/gemm_shapes_mix.err > $OUT/gemm_shapes_mix.jsonl; echo "mix: $(wc -l < $OUT/gemm_shapes_mix.jsonl) shapes"; head -30 $OUT/gemm_shapes_mix.jsonl | cut -c1-160
echo "== 6. other models on the 4-wave kernels (DWM_GEMM4W=1 / 2 force them for every covered launch): train step, UNet"; date
for cfg in "train_8w:" "train_4w:DWM_GEMM4W=1" "unet_8w:" "unet_4wgen:DWM_GEMM4W=2"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  flags="--train --steps 4 --warmup 2"; case $name in unet*) flags="--unet --steps 10 --warmup 3";; esac
  env $envs timeout 300 python bench.py $flags > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name exit $?"; grep '^{' $OUT/bench_$name.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  ', d['value'], d['unit'], round(d['ms_per_step'],2), 'ms')"
done
date
