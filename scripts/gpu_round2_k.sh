#!/bin/bash
# gpurun call K of round 2: shared-LDS group attention after the copy / output overlap: tests, microbench, bench
TAG=${1:-r2o}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/gpu_parity.log
timeout 300 python -m pytest tests/test_hip_gpu.py -q --tb=short -p no:cacheprovider -k "group_forms or rowmaps" > $OUT/pytest_group.log 2>&1
echo "exit $?"; tail -2 $OUT/pytest_group.log; grep -E "^E |^FAILED" $OUT/pytest_group.log | head
echo "== microbench cv"
timeout 200 python scripts/microbench.py cv > $OUT/microbench.log 2>&1; grep crossview $OUT/microbench.log
echo "== bench"
timeout 400 python bench.py > $OUT/bench.log 2>&1; echo "exit $?"
python - "$OUT/bench.log" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k: round(d[k], 3) for k in ("value", "ms_per_step")}, "gemm", round(d["roofline"]["achieved"], 1), "attn", round(d["roofline_attention"]["achieved"], 1),
      "crossview", {k: d["roofline_attention_crossview"][k] for k in ("kernel", "achieved", "frac", "avg_launch_us")}, "text_only", round(d["text_only"]["ms_per_step"], 2))
PY
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
