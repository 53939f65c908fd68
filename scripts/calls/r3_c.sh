#!/bin/bash
# gpurun call C of round 3: the software-pipelined resident attention kernel - tests, microbench, SQ counters
TAG=${1:-r3c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
echo "== attention tests"; date
timeout 900 python -m pytest tests/test_hip_gpu.py -q -k "attention" -p no:cacheprovider > $OUT/pytest_attn.log 2>&1; echo "exit $?"; tail -12 $OUT/pytest_attn.log | cut -c1-300
echo "== microbench attnx"; date
timeout 600 python scripts/microbench.py attnx > $OUT/microbench_attn.log 2>&1; echo "exit $?"; grep -v "crossview\|pointwise" $OUT/microbench_attn.log | cut -c1-200
echo "== pmc"; date
bash scripts/pmc.sh ${TAG}_attn_dual attn_dual 2>&1 | grep "attn_" | cut -c1-400
bash scripts/pmc.sh ${TAG}_attn_joint attn_joint 2>&1 | grep "attn_" | cut -c1-400
date
