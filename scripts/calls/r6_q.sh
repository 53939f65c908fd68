#!/bin/bash
# round 6, call Q: the late V request as the default: attention tests, in-bench A/B against the early form (variant library), then the whole suite
export TAG=${1:-r6q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/opendwm_amd/variants
for i in 1 2; do timeout 300 python -m pytest tests/test_round5_kernels_gpu.py -q -m gpu -k "attention" -p no:cacheprovider 2>&1 | tail -1 | cut -c1-200; done
timeout 600 python -m pytest tests/test_hip_gpu.py -q -m gpu -k "attention or attn" -p no:cacheprovider 2>&1 | tail -1 | cut -c1-200
summ() { python - "$1" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        a = d["roofline_attention"]
        print("  ms/step", round(d["ms_per_step"], 2), "gemm", round(d["roofline"]["achieved"] or 0, 1), "attn frac", round(a["frac"], 4), "avg us", round(a["avg_launch_us"], 1))
PY
}
for rep in 1 2; do
  echo "-- default (late V request)"; timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-text-only-leg > $OUT/bench_late_$rep.json 2> $OUT/bench_late_$rep.err; summ $OUT/bench_late_$rep.json
  echo "-- early V request"; DWM_HIP_LIB=$V/libdwm_hip_dmaearly.so timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-text-only-leg > $OUT/bench_early_$rep.json 2> $OUT/bench_early_$rep.err; summ $OUT/bench_early_$rep.json
done
bash scripts/calls/r6_suite.sh ${TAG}_suite --maxfail=5 2>&1 | tail -12
