#!/bin/bash
# gpurun call H of round 2: in-step A/B of the GEMM epilogue changes (old GEMM object vs the tree), then the whole GPU suite
TAG=${1:-r2h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/gpu_parity.log
echo "== bench A/B (old = gemm_bf16.hip of commit 5d9c774)"
for v in old new old new; do
  if [ $v = old ]; then export DWM_HIP_LIB=$PWD/scripts/experiments/libdwm_hip_old_gemm.so; else unset DWM_HIP_LIB; fi
  timeout 300 python bench.py --steps 5 --warmup 2 > $OUT/bench_$v.json 2> $OUT/bench.err; echo "$v exit $?"
  python - "$OUT/bench_$v.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: round(d[k], 3) for k in ("value", "ms_per_step")}, "gemm", round(d["roofline"]["achieved"], 1), "attn", round(d["roofline_attention"]["achieved"], 1), "text_only", round(d["text_only"]["ms_per_step"], 2))
PY
done
unset DWM_HIP_LIB
echo "== full GPU suite"
timeout 1700 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_all.log 2>&1
echo "exit $?"; tail -5 $OUT/pytest_all.log; grep -E "^E  |^FAILED|^ERROR" $OUT/pytest_all.log | head -30
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
