#!/bin/bash
# gpurun call D of round 4: LayerNorm backward change (training tests, train bench), the default bench with its three legs
TAG=${1:-r4d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
echo "== training tests"; date
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_unet_train_gpu.py -q -p no:cacheprovider --durations=5 > $OUT/pytest_train.log 2>&1; echo "exit $?"; tail -12 $OUT/pytest_train.log | cut -c1-300
echo "== bench --train"; date
timeout 400 python bench.py --train --steps 4 --warmup 2 > $OUT/bench_train.json 2> $OUT/bench.err; echo "exit $?"; cut -c1-330 $OUT/bench_train.json
echo "== default bench (three legs + CPU baseline)"; date
/usr/bin/time -v timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "exit $?"; python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
print("ms", d["ms_per_step"], "gemm", d["roofline"]["achieved"], "attn", d["roofline_attention"]["achieved"], "text_only", d.get("text_only",{}).get("ms_per_step"), "cached", d.get("adapter_cached",{}).get("ms_per_step"), "cpu", d.get("cpu_baseline",{}).get("value"))
PY
grep "Elapsed\|Maximum resident" $OUT/bench_default.err
date
