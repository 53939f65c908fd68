#!/bin/bash
# gpurun call G of round 3: the 256 x 128 GEMM tile (two workgroups per CU) - parity, per-shape A/B, whole-step A/B
TAG=${1:-r3g}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== pytest gemm / conv / adapter / vae"; date
timeout 900 python -m pytest tests/test_hip_gpu.py tests/test_fp32_gpu.py -m gpu -q -x -k "gemm or conv or adapter or vae or split" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -8 $OUT/pytest.log | cut -c1-300
echo "== microbench tiles"; date
timeout 600 python scripts/microbench.py gemmt > $OUT/gemm_tiles.log 2>&1; echo "exit $?"; grep gemm_tiles $OUT/gemm_tiles.log | cut -c1-250
echo "== dev hooks: occupancy / no epilogue"; date
DWM_HIP_LIB=$GRAFT_REPO_ROOT/opendwm_amd/libdwm_hip_dev.so timeout 600 python scripts/microbench.py gemmd > $OUT/gemm_tiles_dev.log 2>&1; echo "exit $?"; grep gemm_tiles $OUT/gemm_tiles_dev.log | cut -c1-400
echo "== bench"; date
for t in 0 0; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-text-only-leg 2>> $OUT/bench.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('bench', d['ms_per_step'], d['roofline']['achieved'], d['config']['finite'])
" | tee -a $OUT/bench_ab.log
done
date
