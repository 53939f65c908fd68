#!/bin/bash
# round 6, call U: the library rebuilt after comment-only header edits (new source hash): attention tests, FAR test, smoke, default bench
export TAG=${1:-r6u}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_round6_gpu.py tests/test_round5_kernels_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2 | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; python - <<'PY'
import json, os
d = json.loads([l for l in open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", os.environ["TAG"], "bench.json")) if l.startswith("{")][0])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline_attention"]["frac"], d["roofline_attention"].get("mfma_pipe_busy"), d["roofline"]["by_kernel"]["gemm4w_kernel"].get("traffic_source"))
PY
