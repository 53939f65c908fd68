#!/bin/bash
# gpurun call B of round 4: RCCL paths on one GPU, the tVAE autoregressive bench line + its one-window full-size parity test,
# the remaining 40-step cases (text-only at full size; stress case in both stream modes), the default bench with the measured CPU baseline
TAG=${1:-r4b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
echo "== RCCL tests + item seams"; date
timeout 900 python -m pytest tests/test_rccl_gpu.py tests/test_hip_gpu.py -q -k "rccl or item_seams" -p no:cacheprovider --durations=6 > $OUT/pytest_rccl.log 2>&1; echo "exit $?"; tail -22 $OUT/pytest_rccl.log | cut -c1-400
echo "== bench --tvae-ar"; date
timeout 600 python bench.py --tvae-ar > $OUT/bench_tvae_ar.json 2> $OUT/bench_tvae.err; echo "exit $?"; cut -c1-2500 $OUT/bench_tvae_ar.json; tail -5 $OUT/bench_tvae.err
echo "== bench --tvae-ar --graph"; date
timeout 600 python bench.py --tvae-ar --graph > $OUT/bench_tvae_ar_graph.json 2>> $OUT/bench_tvae.err; echo "exit $?"; cut -c1-700 $OUT/bench_tvae_ar_graph.json
echo "== parity: tVAE AR window, text-only 40 steps, stress case"; date
timeout 1500 python -m pytest tests/test_fulldepth_gpu.py -q -k "tvae or heavy or (forty_step_denoise_full_depth and text_only)" -p no:cacheprovider --durations=8 > $OUT/pytest_fulldepth.log 2>&1; echo "exit $?"; tail -16 $OUT/pytest_fulldepth.log | cut -c1-300
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null; grep "denoise_40\|tvae\|rccl\|seams" $OUT/gpu_parity.log | cut -c1-700
echo "== default bench (with the measured CPU baseline)"; date
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "exit $?"; cut -c1-400 $OUT/bench_default.json; python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
print(json.dumps(d.get("cpu_baseline"))[:1500])
print("ms", d["ms_per_step"], "gemm", d["roofline"]["achieved"], "attn", d["roofline_attention"]["achieved"], "text_only", d.get("text_only",{}).get("ms_per_step"))
PY
tail -3 $OUT/bench_default.err
date
