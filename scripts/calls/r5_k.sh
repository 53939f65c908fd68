#!/bin/bash
# call K: 4-wave GEMM epilogue with the residual rows requested two passes ahead - tests, then bench A/B on one box against the library
# with the previous epilogue (opendwm_amd/libdwm_hip_prev4w.so, DWM_HIP_LIB)
TAG=${1:-r5k}
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
timeout 600 python -m pytest tests/test_gemm4w_gpu.py tests/test_stream32_gpu.py tests/test_round5_kernels_gpu.py -m gpu -q -x -k "not one_wave" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest.log | cut -c1-250
timeout 600 python -m pytest tests/test_hip_gpu.py -m gpu -q -x -k "gemm or block or stack or forward" -p no:cacheprovider > $OUT/pytest2.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest2.log | cut -c1-250
for cfg in "new:" "prev:DWM_HIP_LIB=$GRAFT_REPO_ROOT/opendwm_amd/libdwm_hip_prev4w.so" "new2:" "prev2:DWM_HIP_LIB=$GRAFT_REPO_ROOT/opendwm_amd/libdwm_hip_prev4w.so"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-text-only-leg > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name exit $?"; summ $OUT/bench_$name.json; tail -2 $OUT/bench_$name.err | cut -c1-200
done
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-text-only-leg --gemm-shapes > $OUT/bench_shapes.log 2> $OUT/gemm_shapes.err
grep '^{"M"' $OUT/gemm_shapes.err > $OUT/gemm_shapes.jsonl; head -12 $OUT/gemm_shapes.jsonl | cut -c1-160
for cfg in "train:" "train_prev:DWM_HIP_LIB=$GRAFT_REPO_ROOT/opendwm_amd/libdwm_hip_prev4w.so"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py --train --steps 4 --warmup 2 > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name exit $?"; grep '^{' $OUT/bench_$name.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  ', d['value'], d['unit'], round(d['ms_per_step'],2), 'ms')"
done
date
