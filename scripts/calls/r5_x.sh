#!/bin/bash
# timing experiments on attn_res4_kernel's tile loop (wrong results by construction): what is the loop bound by?
#   DWM_R4X=1 no row-sum adds, 2 no exponentials, 3 neither, 4 no fragment re-reads
# (the hooks are not in the product source: `patch -p1 < scripts/experiments/attn_res4_loop_hooks.patch` first; results of round 5:
#  profiles/r5x_attn_res4_loop_experiments.txt)
TAG=${1:-r5x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for x in 0 1 2 3 4; do
  if [ $x = 0 ]; then fl="-DDWM_ATTN_TRACE"; else fl="-DDWM_ATTN_TRACE -DDWM_R4X=$x"; fi
  DWM_EXTRA_FLAGS="$fl" timeout 600 python -m opendwm_amd.build > $OUT/build_$x.log 2>&1
  for m in 1 2; do
    echo "== DWM_R4X=$x mode $m"
    DWM_ATTN_RES4=$m timeout 120 python scripts/experiments/attn_trace4.py 154 > $OUT/trace_x${x}_m$m.txt 2>&1; grep "wave [03] mean" $OUT/trace_x${x}_m$m.txt | head -2 | cut -c1-300
  done
done
date
