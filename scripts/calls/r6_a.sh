#!/bin/bash
# round 6, call A (evidence before any tuning): the headline bench with clock / power sampling in the line, the library comparison on the
# SHIPPED 4-wave kernels with their epilogues, the counters this rocprofv3 offers for MALL / DRAM traffic, isolated attention + RESID numbers
export TAG=${1:-r6a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== 1. headline bench (default flags) + per-shape GEMM table"; date
timeout 900 python bench.py --gemm-shapes > $OUT/bench.json 2> $OUT/gemm_shapes.jsonl; echo "exit $?"
python - <<'PY'
import json, os
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", os.environ["TAG"], "bench.json")
for ln in open(p):
    if ln.startswith("{"):
        d = json.loads(ln)
        print(" ms/step", round(d["ms_per_step"], 2), "gemm", round(d["roofline"]["achieved"], 1), "attn frac", round(d["roofline_attention"]["frac"], 4),
              "clock_power", d.get("clock_power"), "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
PY
grep '^{' $OUT/gemm_shapes.jsonl | head -12
echo "== 2. library comparison, shipped kernels with epilogues"; date
timeout 400 python scripts/blaslt_compare.py > $OUT/library_gemm_comparison.log 2>&1; cut -c1-400 $OUT/library_gemm_comparison.log
echo "== 3. counters for MALL / DRAM"; date
( cd /tmp && timeout 120 rocprofv3 -L 2>/dev/null | grep -i -E "mall|dram|hbm|EA0?_RD|EA0?_WR|TCC_EA|TCC_BUBBLE|TCC_REQ|TCC_HIT|TCC_MISS" | cut -c1-200 | sort -u | head -80 ) > $OUT/counters_memory_side.txt 2>&1; wc -l $OUT/counters_memory_side.txt; head -60 $OUT/counters_memory_side.txt
echo "== 4. isolated kernels"; date
timeout 300 python scripts/microbench.py attnr s32 cv pw > $OUT/microbench.log 2>&1; cut -c1-200 $OUT/microbench.log | tail -40
date
