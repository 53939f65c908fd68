#!/bin/bash
# gpurun call B of round 5: the one-wave-per-SIMD resident attention (attention_res4.hip) - tests, microbench against the 12-wave kernel,
# SQ counters, bench A/B; the promotions of call A (general 4-wave GEMM form, stacked modulation, training on the 4-wave kernels)
# usage: gpurun --timeout 900 -- 'bash scripts/calls/r5_b.sh [tag]'
TAG=${1:-r5b}
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
echo "== 1. tests of the round-5 kernels"; date
timeout 600 python -m pytest tests/test_round5_kernels_gpu.py -m gpu -q -x -rf --tb=short -p no:cacheprovider > $OUT/pytest_round5.log 2>&1
echo "exit $?" | tee -a $OUT/pytest_round5.log; tail -30 $OUT/pytest_round5.log | cut -c1-300
cp gpurun_out/gpu_parity.log $OUT/gpu_parity_round5.log 2>/dev/null
echo "== 2. attention microbench: 12-wave kernel (4096) against the one-wave-per-SIMD form (0 / 8192 / 16384 / 24576)"; date
timeout 200 python scripts/microbench.py attnr4 > $OUT/microbench_attn_res4.log 2>&1; cut -c1-200 $OUT/microbench_attn_res4.log
echo "== 3. SQ counters of the joint attention"; date
for v in 4096 0; do timeout 200 bash scripts/pmc.sh ${TAG}_attn_joint_v$v attn_joint $v > $OUT/pmc_attn_joint_v$v.log 2>&1; cp gpurun_out/pmc_${TAG}_attn_joint_v$v/summary.txt $OUT/pmc_attn_joint_v$v.txt 2>/dev/null; cut -c1-400 $OUT/pmc_attn_joint_v$v.txt; done
echo "== 4. bench A/B"; date
for cfg in "default:" "res12:DWM_ATTN_RES4=0" "res4ilv:DWM_ATTN_RES4=2" "default2:"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-text-only-leg > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name exit $?"; summ $OUT/bench_$name.json
done
echo "== 5. train step on the 4-wave kernels (default now) against 8-wave"; date
for cfg in "train:" "train_8w:DWM_GEMM4W=0" "train_fast_only:DWM_GEMM4W=f"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py --train --steps 4 --warmup 2 > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name exit $?"; grep '^{' $OUT/bench_$name.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  ', d['value'], d['unit'], round(d['ms_per_step'],2), 'ms')"
done
echo "== 6. quick regression: attention + gemm4w + stream tests"; date
timeout 600 python -m pytest tests/test_hip_gpu.py tests/test_gemm4w_gpu.py -m gpu -q -x -k "attention or gemm4w or four_wave" -p no:cacheprovider > $OUT/pytest_quick.log 2>&1
echo "exit $?"; tail -5 $OUT/pytest_quick.log | cut -c1-300
date
