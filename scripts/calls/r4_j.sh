#!/bin/bash
# gpurun call J of round 4: compile-time operand forms (RS) of the RESID epilogue - GEMM / stream / UNet tests, the stream microbench,
# the UNet bench and the default bench
TAG=${1:-r4j}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
timeout 1200 python -m pytest tests/test_hip_gpu.py tests/test_stream32_gpu.py tests/test_unet_gpu.py -q -p no:cacheprovider --durations=5 > $OUT/pytest.log 2>&1; echo "exit $?"; tail -10 $OUT/pytest.log | cut -c1-300
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
echo "== microbench s32"; date
timeout 300 python scripts/microbench.py s32 > $OUT/micro_s32.log 2>&1; grep '^{' $OUT/micro_s32.log | cut -c1-200
echo "== bench --unet"; date
timeout 400 python bench.py --unet > $OUT/bench_unet.json 2> $OUT/bench.err; echo "exit $?"; cut -c1-300 $OUT/bench_unet.json
echo "== default bench"; date
timeout 900 python bench.py > $OUT/bench.json 2>> $OUT/bench.err; echo "exit $?"; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("ms", d["ms_per_step"], "gemm", d["roofline"]["achieved"], "attn", d["roofline_attention"]["achieved"], "text_only", d.get("text_only",{}).get("ms_per_step"), "cached", d.get("adapter_cached",{}).get("ms_per_step"))
print({k: (v.get("achieved"), v.get("frac")) for k, v in d.items() if k.startswith("roofline")})
PY
date
