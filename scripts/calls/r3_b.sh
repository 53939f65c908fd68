#!/bin/bash
# gpurun call B of round 3: the resident attention kernel (tests, microbench, SQ counters), the fp32 skip path of the ImageAdapter
# (tests, drift bisect, full-depth 40-step parity of the text+layout model), a short bench
TAG=${1:-r3b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
echo "== attention + adapter tests"; date
timeout 900 python -m pytest tests/test_hip_gpu.py -q -x -k "attention or adapter or layout" -p no:cacheprovider > $OUT/pytest_attn.log 2>&1; echo "exit $?"; tail -8 $OUT/pytest_attn.log
echo "== microbench attnx"; date
timeout 600 python scripts/microbench.py attnx > $OUT/microbench_attn.log 2>&1; echo "exit $?"; cat $OUT/microbench_attn.log | cut -c1-200
echo "== drift bisect (text+layout / pointwise)"; date
DRIFT_ONLY=text+layout/pointwise timeout 600 python scripts/drift_bisect.py $OUT/drift_bisect.json > $OUT/drift_bisect.log 2>&1; echo "exit $?"; cut -c1-420 $OUT/drift_bisect.log
echo "== bench"; date
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?"; cut -c1-1500 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== full-depth 40 steps (text+layout)"; date
timeout 900 python -m pytest tests/test_fulldepth_gpu.py -q -x -k "forty_step_denoise_full_depth and layout" -p no:cacheprovider > $OUT/pytest_fulldepth40.log 2>&1; echo "exit $?"; tail -5 $OUT/pytest_fulldepth40.log
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null; grep denoise_40 $OUT/gpu_parity.log
echo "== pmc"; date
bash scripts/pmc.sh ${TAG}_attn_dual attn_dual 2>&1 | grep "attn_" | cut -c1-400
bash scripts/pmc.sh ${TAG}_attn_joint attn_joint 2>&1 | grep "attn_" | cut -c1-400
date
