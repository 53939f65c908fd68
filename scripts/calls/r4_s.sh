#!/bin/bash
# gpurun call S of round 4 (last GPU minutes): the opt-in 4-wave GEMM kernels - their own battery, then the GEMM / full-width / stream
# tests, the full-depth full-size forward and the bench with DWM_GEMM4W=1
TAG=${1:-r4s}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
date
timeout 200 python -m pytest tests/test_gemm4w_gpu.py -q -p no:cacheprovider > $OUT/pytest_battery.log 2>&1; rc=$?; echo "battery exit $rc"; tail -15 $OUT/pytest_battery.log | cut -c1-600
grep gemm4w gpurun_out/gpu_parity.log | cut -c1-1500
if [ $rc -ne 0 ]; then exit 0; fi
export DWM_GEMM4W=1
date
timeout 300 python -m pytest tests/test_hip_gpu.py tests/test_stream32_gpu.py -q -p no:cacheprovider -k "gemm or full_width or stream or block" > $OUT/pytest_4w.log 2>&1; echo "tests exit $?"; tail -6 $OUT/pytest_4w.log | cut -c1-400
date
timeout 200 python -m pytest "tests/test_fulldepth_gpu.py::test_full_depth_full_size_forward_vs_oracle_on_device" -q -p no:cacheprovider > $OUT/pytest_fulldepth_4w.log 2>&1; echo "fulldepth exit $?"; tail -3 $OUT/pytest_fulldepth_4w.log | cut -c1-300
grep full_depth gpurun_out/gpu_parity.log | cut -c1-300
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
date
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_4w.json 2> $OUT/bench.err; echo "bench exit $?"; python - <<PY
import json
d=json.load(open("$OUT/bench_4w.json"))
print("ms", d["ms_per_step"], "gemm", d["roofline"]["achieved"], "attn", d["roofline_attention"]["achieved"], "text_only", d.get("text_only",{}).get("ms_per_step"), "cached", d.get("adapter_cached",{}).get("ms_per_step"), "finite", d["config"].get("finite"))
PY
tail -3 $OUT/bench.err | cut -c1-300
date
