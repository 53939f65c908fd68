#!/bin/bash
# round 6, call V: the other modes of the headline bench on the final library: one-rank RCCL preflight (the --gpus N line's self-diagnosis
# fields), whole-step HIP graph, cached adapter, one-rank frame-shard / CFG-split plumbing
export TAG=${1:-r6v}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
line() { python - "$1" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        pre = d.get("preflight") or {}
        print("  ", d["value"], d["unit"], round(d["ms_per_step"], 2), "ms;", {k: pre[k] for k in ("rccl_version", "ranks_seen", "distinct_devices", "allreduce_ms") if k in pre}, (pre.get("ranks") or [None])[0])
PY
}
echo "-- preflight (one-rank RCCL group)"; timeout 400 python bench.py --gpus 1 --preflight --steps 3 --warmup 1 --no-cpu-baseline --no-text-only-leg > $OUT/bench_preflight.json 2> $OUT/bench_preflight.err; line $OUT/bench_preflight.json
echo "-- whole step as one HIP graph"; timeout 400 python bench.py --graph --steps 5 --warmup 2 --no-cpu-baseline --no-text-only-leg > $OUT/bench_graph.json 2> $OUT/bench_graph.err; line $OUT/bench_graph.json
echo "-- adapter residuals cached"; timeout 400 python bench.py --adapter-cache --steps 5 --warmup 2 --no-cpu-baseline --no-text-only-leg > $OUT/bench_adapter_cache.json 2> $OUT/bench_adapter_cache.err; line $OUT/bench_adapter_cache.json
echo "-- frame shard, one rank"; timeout 400 python bench.py --gpus 1 --frame-shard --steps 3 --warmup 1 --no-cpu-baseline --no-text-only-leg > $OUT/bench_frame_shard1.json 2> $OUT/bench_frame_shard1.err; line $OUT/bench_frame_shard1.json
