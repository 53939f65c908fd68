#!/bin/bash
# call D: attn_res4_kernel (asm S MFMAs + builtin AGPR PV): tests, microbench, timeline trace
TAG=${1:-r5d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
timeout 600 python -m pytest tests/test_round5_kernels_gpu.py -m gpu -q -rf --tb=line -k "one_wave" -p no:cacheprovider > $OUT/pytest_round5.log 2>&1
echo "exit $?"; tail -12 $OUT/pytest_round5.log | cut -c1-250
cp gpurun_out/gpu_parity.log $OUT/gpu_parity_round5.log 2>/dev/null
python - <<'PY'
import json
for ln in open("gpurun_out/gpu_parity.log"):
    d = json.loads(ln)
    if d.get("test") == "attention_one_wave_per_simd_forms" and d["scale"] == 1.0:
        print(d["N"], d["Lc"], {k: round(v, 4) for k, v in d.items() if k in ("old", "0", "16", "8192")})
PY
timeout 200 python scripts/microbench.py attnr4 > $OUT/microbench.log 2>&1; cut -c1-200 $OUT/microbench.log
echo "== trace build"; date
cp opendwm_amd/libdwm_hip.so /tmp/libdwm_hip.so.keep
DWM_EXTRA_FLAGS=-DDWM_ATTN_TRACE timeout 600 python -m opendwm_amd.build > $OUT/build_trace.log 2>&1; tail -2 $OUT/build_trace.log
for m in 1; do
DWM_ATTN_RES4=$m timeout 120 python scripts/experiments/attn_trace4.py 154 > $OUT/trace4_L602_mode$m.txt 2>&1; tail -12 $OUT/trace4_L602_mode$m.txt | cut -c1-330
DWM_ATTN_RES4=$m timeout 120 python scripts/experiments/attn_trace4.py 0 > $OUT/trace4_L448_mode$m.txt 2>&1; tail -5 $OUT/trace4_L448_mode$m.txt | cut -c1-330
done
date
