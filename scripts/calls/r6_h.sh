#!/bin/bash
# round 6, call H: headline bench with attn_stream_kernel as the default + pre-scaled Q (A/B: DWM_ATTN_VARIANT=0x2000 keeps the 12-wave kernel)
export TAG=${1:-r6h}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
summ() { python - "$1" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("  ms/step", round(d["ms_per_step"], 2), "gemm", round(d["roofline"]["achieved"] or 0, 1), "attn", round(d["roofline_attention"]["achieved"], 1), "frac", round(d["roofline_attention"]["frac"], 4),
              "avg us", round(d["roofline_attention"]["avg_launch_us"], 1), "clock", (d.get("clock_power") or {}).get("sclk_mhz"))
PY
}
for cfg in "stream:" "res12:DWM_ATTN_VARIANT=0x2000" "stream2:" "res12b:DWM_ATTN_VARIANT=0x2000"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "== $name ($envs)"; date
  env $envs timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-text-only-leg > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "exit $?"; summ $OUT/bench_$name.json
done
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
