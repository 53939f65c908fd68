#!/bin/bash
# diagnostic call: which shapes of attn_res4_kernel fail, and timing of the forced fallback
TAG=${1:-r5c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
timeout 600 python -m pytest tests/test_round5_kernels_gpu.py -m gpu -q -rf --tb=line -k "one_wave" -p no:cacheprovider > $OUT/pytest_round5.log 2>&1
echo "exit $?"; grep -c PASS $OUT/pytest_round5.log; tail -60 $OUT/pytest_round5.log | cut -c1-250
cp gpurun_out/gpu_parity.log $OUT/gpu_parity_round5.log 2>/dev/null
python - <<'PY'
import json
for ln in open("gpurun_out/gpu_parity.log"):
    d = json.loads(ln)
    if d.get("test") == "attention_one_wave_per_simd_forms":
        print(d["scale"], d["N"], d["Lc"], {k: round(v, 4) for k, v in d.items() if k in ("old", "0", "16", "8192", "16384")})
PY
timeout 200 python scripts/microbench.py attnr4x > $OUT/microbench.log 2>&1; cut -c1-200 $OUT/microbench.log
