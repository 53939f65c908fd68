#!/bin/bash
# gpurun call A of round 3: (1) drift bisect of the 40-step loop, (2) the full-depth 40-step parity test, (3) per-shape GEMM table,
# (4) SQ wait / LDS counters of the attention and GEMM kernels, (5) issue-rate probes
TAG=${1:-r3a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
echo "== drift bisect"; date
timeout 900 python scripts/drift_bisect.py $OUT/drift_bisect.json > $OUT/drift_bisect.log 2>&1; echo "exit $?"; tail -12 $OUT/drift_bisect.log | cut -c1-600
echo "== full-depth 40 steps"; date
timeout 1500 python -m pytest tests/test_fulldepth_gpu.py -q -x -k "forty_step_denoise_full_depth" -p no:cacheprovider > $OUT/pytest_fulldepth40.log 2>&1; echo "exit $?"; tail -5 $OUT/pytest_fulldepth40.log
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null; cat $OUT/gpu_parity.log
echo "== gemm shapes"; date
timeout 600 python bench.py --gemm-shapes --no-cpu-baseline --no-text-only-leg --steps 3 --warmup 2 > $OUT/bench_shapes.json 2> $OUT/gemm_shapes.jsonl; echo "exit $?"; cut -c1-200 $OUT/bench_shapes.json; cat $OUT/gemm_shapes.jsonl
echo "== pmc"; date
rocprofv3 -L > $OUT/counters_list.txt 2>&1
bash scripts/pmc.sh ${TAG}_attn_dual attn_dual 2>&1 | tail -12
bash scripts/pmc.sh ${TAG}_attn_joint attn_joint 2>&1 | tail -12
bash scripts/pmc.sh ${TAG}_gemm_out gemm_out 2>&1 | tail -12
bash scripts/pmc.sh ${TAG}_gemm_geglu gemm_geglu 2>&1 | tail -12
echo "== probes"; date
cd /tmp && hipcc --offload-arch=gfx950 -O3 -o mfma_valu_probe $GRAFT_REPO_ROOT/scripts/probes/mfma_valu_probe.hip && timeout 120 ./mfma_valu_probe > $OUT/mfma_valu_probe.txt 2>&1; cat $OUT/mfma_valu_probe.txt
date
