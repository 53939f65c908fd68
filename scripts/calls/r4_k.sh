#!/bin/bash
# gpurun call K of round 4 (final code): the test files not re-run since call Z's snapshot changed under them (drivers, full size,
# the cheaper full-depth cases, the touched fp32 / stream tests), smoke, the bench under rocprofv3 and the default bench
TAG=${1:-r4k}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
echo "== tests"; date
timeout 1500 python -m pytest tests/test_drivers_gpu.py tests/test_fullsize_gpu.py tests/test_stream32_gpu.py tests/test_rccl_gpu.py \
  "tests/test_fp32_gpu.py::test_model_with_layout_adapter_fp32_vs_cpu_oracle" \
  "tests/test_fulldepth_gpu.py::test_full_depth_full_size_forward_vs_oracle_on_device" \
  "tests/test_fulldepth_gpu.py::test_forty_step_denoise_vs_oracle_loop_on_device" \
  -q -p no:cacheprovider --durations=8 > $OUT/pytest.log 2>&1; echo "exit $?"; tail -14 $OUT/pytest.log | cut -c1-300
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
echo "== smoke"; date
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
echo "== default bench"; date
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?"; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("ms", d["ms_per_step"], "gemm", d["roofline"]["achieved"], "attn", d["roofline_attention"]["achieved"], "text_only", d.get("text_only",{}).get("ms_per_step"), "cached", d.get("adapter_cached",{}).get("ms_per_step"), "cpu", d.get("cpu_baseline",{}).get("value"))
PY
echo "== bench under rocprofv3"; date
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprofv3.log 2>> $OUT/bench.err
echo "exit $?"; grep '^{' $OUT/bench_under_rocprofv3.log | cut -c1-200
f=$(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/bench_kernel_stats.csv; head -10 $OUT/bench_kernel_stats.csv | cut -c1-190
date
