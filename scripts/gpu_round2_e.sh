#!/bin/bash
# gpurun call E of round 2: UNet config-1 parity after the aliasing fix, VAE tests, bench A/B of the GEMM schedule, attention barrier ablation
TAG=${1:-r2e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/gpu_parity.log
echo "== unet config 1 + vae"
timeout 600 python -m pytest tests/test_fulldepth_gpu.py tests/test_hip_gpu.py tests/test_unet_gpu.py -q --tb=short -p no:cacheprovider -k "config1 or vae or unet" > $OUT/pytest_a.log 2>&1
echo "exit $?"; tail -3 $OUT/pytest_a.log; grep -E "^E |^FAILED" $OUT/pytest_a.log | head
grep -E "unet_full_width" gpurun_out/gpu_parity.log | cut -c1-200
echo "== bench A/B"
for s in 0 2 0 2; do
  DWM_GEMM_SCHED=$s timeout 300 python bench.py --steps 5 --warmup 2 > $OUT/bench_s$s.json 2> $OUT/bench.err; echo "sched $s exit $?"
  python - "$OUT/bench_s$s.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: round(d[k], 3) for k in ("value", "ms_per_step")}, "gemm", round(d["roofline"]["achieved"], 1), "attn", round(d["roofline_attention"]["achieved"], 1), "text_only", round(d["text_only"]["ms_per_step"], 2))
PY
done
echo "== attention barrier ablation"
timeout 200 python - > $OUT/attn_nobarrier.log 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
sys.argv = ["microbench"]
import scripts.microbench as mb
mb.bench_attn([1, 1 | (1 << 12), 1 | 64 | (1 << 12)])
PY
grep -v "pointwise\|crossview" $OUT/attn_nobarrier.log | cut -c1-200
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
