#!/bin/bash
# gpurun call I of round 2 (evidence): GEMM tests for the FAST epilogue forms, smoke, bench (default command), rocprofv3 kernel
# stats of the same command, PMC traffic, UNet bench, UNet / MMDiT train-step bench
TAG=${1:-r2i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/gpu_parity.log
echo "== gemm tests"
timeout 400 python -m pytest tests/test_hip_gpu.py tests/test_unet_gpu.py -q --tb=short -p no:cacheprovider -k "gemm or linear or conv or block or forward" > $OUT/pytest_gemm.log 2>&1
echo "exit $?"; tail -2 $OUT/pytest_gemm.log; grep -E "^E |^FAILED" $OUT/pytest_gemm.log | head
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
echo "== bench (default command)"
timeout 900 python bench.py > $OUT/bench.log 2>&1; echo "bench exit $?"; tail -1 $OUT/bench.log | cut -c1-300
echo "== rocprof (kernel stats of bench.py --steps 3 --warmup 1)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-text-only-leg > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.log 2>&1)
for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/bench_kernel_stats.csv; head -12 $f | cut -c1-160; done
tail -1 $OUT/rocprof_bench.log | cut -c1-200
echo "== unet bench"
timeout 400 python bench.py --unet --no-cpu-baseline > $OUT/bench_unet.log 2>&1; echo "exit $?"; tail -1 $OUT/bench_unet.log | cut -c1-260
echo "== unet train bench"
timeout 600 python bench.py --train --unet --steps 3 --warmup 1 > $OUT/bench_train_unet.log 2>&1; echo "exit $?"; tail -1 $OUT/bench_train_unet.log | cut -c1-900
echo "== mmdit train bench"
timeout 600 python bench.py --train --steps 3 --warmup 1 > $OUT/bench_train.log 2>&1; echo "exit $?"; tail -1 $OUT/bench_train.log | cut -c1-400
echo "== pmc traffic"
timeout 600 bash scripts/pmc_traffic.sh $TAG > $OUT/pmc.log 2>&1; echo "exit $?"; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r2i/pmc_traffic.json"))
    for k, v in d.items():
        print(k, {kk: (round(vv / 1e9, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if "per_launch" in kk}, v.get("mfma", {}).get("mfma_pipe_utilisation"))
except Exception as e:
    print("pmc:", e)
PY
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
