#!/bin/bash
# gpurun call U of round 4 (last one): the MMDiT inference forward now asks for the 4-wave GEMM kernels by default (tile = 3) - the
# full-size tests that had not run on them, and the bench
TAG=${1:-r4u}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
date
timeout 100 python bench.py --no-cpu-baseline --no-text-only-leg --steps 8 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("ms", d["ms_per_step"], "gemm", d["roofline"]["achieved"], "attn", d["roofline_attention"]["achieved"], "finite", d["config"].get("finite"))
PY
date
timeout 200 python -m pytest tests/test_fullsize_gpu.py tests/test_gemm4w_gpu.py -x -q -p no:cacheprovider --durations=4 > $OUT/pytest.log 2>&1; echo "tests exit $?"; tail -9 $OUT/pytest.log | cut -c1-300
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
date
