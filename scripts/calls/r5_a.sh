#!/bin/bash
# gpurun call A of round 5 (prepared at the end of round 4, whose GPU budget was spent before these kernels were written):
#   1. the gated tests of the kernels written without a GPU (tests/test_unvalidated_gpu.py): attn_res2_kernel, the general 4-wave GEMM
#   2. attention microbench: resident kernel against its paired form, alternating in one process
#   3. bench A/B on this box: default | DWM_ATTN_RES2=1 | DWM_GEMM4W=2 | both
#   4. the default GPU suite with durations (the driver's limit for it is 1200 s; tests/conftest.py)
# usage: gpurun --timeout 2700 -- 'bash scripts/calls/r5_a.sh'
TAG=${1:-r5a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
echo "== gated tests (DWM_TEST_UNVALIDATED=1)"; date
DWM_TEST_UNVALIDATED=1 timeout 900 python -m pytest tests/test_unvalidated_gpu.py -m gpu -q -rf --tb=short -p no:cacheprovider > $OUT/pytest_unvalidated.log 2>&1
echo "exit $?" | tee -a $OUT/pytest_unvalidated.log; tail -30 $OUT/pytest_unvalidated.log | cut -c1-300
cp gpurun_out/gpu_parity.log $OUT/gpu_parity_unvalidated.log 2>/dev/null
echo "== attention microbench: variant 0 (attn_res_kernel<12>) against 64 (attn_res2_kernel)"; date
timeout 300 python scripts/microbench.py attnr2 > $OUT/microbench_attn_res2.log 2>&1; cut -c1-200 $OUT/microbench_attn_res2.log
echo "== SQ counters of the joint attention: resident kernel (variant 0) against its paired form (64)"; date
for v in 0 64; do timeout 240 bash scripts/pmc.sh ${TAG}_attn_joint_v$v attn_joint $v > $OUT/pmc_attn_joint_v$v.log 2>&1; cp gpurun_out/pmc_${TAG}_attn_joint_v$v/summary.txt $OUT/pmc_attn_joint_v$v.txt 2>/dev/null; cut -c1-400 $OUT/pmc_attn_joint_v$v.txt; done
echo "== bench A/B"; date
for cfg in "default:" "res2:DWM_ATTN_RES2=1" "gemm4wgen:DWM_GEMM4W=2" "both:DWM_ATTN_RES2=1 DWM_GEMM4W=2" "default2:"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-text-only-leg > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name exit $?"; python - "$OUT/bench_$name.json" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("  ms/step", round(d["ms_per_step"], 2), "gemm", round(d["roofline"]["achieved"] or 0, 1), "TFLOP/s", "attn_res frac", round(d["roofline_attention"]["frac"], 4),
              "by_kernel", {k: round(v["tflops"], 1) for k, v in (d["roofline"].get("by_kernel") or {}).items()})
PY
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-text-only-leg --stack-modulation > $OUT/bench_stackmod.json 2> $OUT/bench_stackmod.err
echo "stack-modulation exit $?"; grep '^{' $OUT/bench_stackmod.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  ms/step', round(d['ms_per_step'],2), 'finite', d['config'].get('finite'))"
echo "== per-shape GEMM table inside the bench: 8-wave only against the default mix (which epilogues keep the 4-wave gain)"; date
for cfg in "8w:DWM_GEMM4W=0" "mix:"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-text-only-leg --gemm-shapes > $OUT/bench_shapes_$name.log 2> $OUT/gemm_shapes_$name.err
  grep '^{"M"' $OUT/gemm_shapes_$name.err > $OUT/gemm_shapes_$name.jsonl; echo "$name: $(wc -l < $OUT/gemm_shapes_$name.jsonl) shapes"; head -8 $OUT/gemm_shapes_$name.jsonl | cut -c1-160
done
echo "== other models on the 4-wave kernels (DWM_GEMM4W=1 / 2 force them for every covered launch): train step, UNet"; date
for cfg in "train_8w:" "train_4w:DWM_GEMM4W=1" "unet_8w:" "unet_4wgen:DWM_GEMM4W=2"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  flags="--train --steps 4 --warmup 2"; case $name in unet*) flags="--unet --steps 10 --warmup 3";; esac
  env $envs timeout 600 python bench.py $flags > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name exit $?"; grep '^{' $OUT/bench_$name.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  ', d['value'], d['unit'], round(d['ms_per_step'],2), 'ms')"
done
echo "== default GPU suite (as the driver runs it)"; date
rm -f gpurun_out/gpu_parity.log
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=25 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log
tail -45 $OUT/pytest.log | cut -c1-250
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
date
