#!/bin/bash
# gpurun call G of round 2: GEMM epilogue changes (batched bias / norm-weight loads, FAST residual form): tests, microbench, bench
TAG=${1:-r2g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/gpu_parity.log
echo "== gemm / block tests"
timeout 600 python -m pytest tests/test_hip_gpu.py -q --tb=short -p no:cacheprovider -k "gemm or linear or conv or block or forward or adapter" > $OUT/pytest_gemm.log 2>&1
echo "exit $?"; tail -3 $OUT/pytest_gemm.log; grep -E "^E |^FAILED" $OUT/pytest_gemm.log | head
echo "== microbench gemmx"
timeout 300 python scripts/microbench.py gemmx > $OUT/microbench.log 2>&1; grep gemm $OUT/microbench.log | cut -c1-300
echo "== bench"
timeout 300 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: round(d[k], 3) for k in ("value", "ms_per_step")}, "gemm", round(d["roofline"]["achieved"], 1), "attn", round(d["roofline_attention"]["achieved"], 1), "text_only", round(d["text_only"]["ms_per_step"], 2))
PY
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
