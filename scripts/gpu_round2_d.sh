#!/bin/bash
# gpurun call D of round 2: TransformerModel sub-block bisect at the UNet config-1 geometry; bench with the GEMM schedule 2
TAG=${1:-r2d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for lvl in 0 1 2; do
  echo "-- unet_bisect2 level $lvl"
  timeout 300 python scripts/unet_bisect2.py $lvl > $OUT/bisect2_$lvl.log 2>&1; echo "exit $?"
  grep -v amdgpu.ids $OUT/bisect2_$lvl.log | tail -12
done
echo "== bench"
timeout 600 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2d/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["achieved"], d["roofline_attention"]["achieved"], d["text_only"]["ms_per_step"])
PY
