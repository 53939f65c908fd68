#!/bin/bash
# round 6, call O: nontemporal stores of layernorm_kernel's outputs, in-bench A/B on one box (the library under DWM_HIP_LIB differs in norm.hip only)
export TAG=${1:-r6o}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/opendwm_amd/variants
summ() { python - "$1" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("  ms/step", round(d["ms_per_step"], 2), "gemm", round(d["roofline"]["achieved"] or 0, 1), "attn frac", round(d["roofline_attention"]["frac"], 4))
PY
}
for rep in 1 2; do
  echo "-- default"; timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-text-only-leg > $OUT/bench_default_$rep.json 2> $OUT/bench_default_$rep.err; summ $OUT/bench_default_$rep.json
  echo "-- ln_nt1"; DWM_HIP_LIB=$V/libdwm_hip_ln_nt1.so timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-text-only-leg > $OUT/bench_ln_nt1_$rep.json 2> $OUT/bench_ln_nt1_$rep.err; summ $OUT/bench_ln_nt1_$rep.json
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- env DWM_HIP_LIB=$V/libdwm_hip_ln_nt1.so python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-text-only-leg > /dev/null 2>&1
f=$(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/rocprofv3_kernel_stats_ln_nt1.csv 2>/dev/null; grep -i 'layernorm' $OUT/rocprofv3_kernel_stats_ln_nt1.csv | cut -c1-200
