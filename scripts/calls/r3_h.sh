#!/bin/bash
# gpurun call H of round 3: A/B of the GEMM main-loop change (counted LDS waits) against the previous library, same box
TAG=${1:-r3h}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/opendwm_amd/libdwm_hip_old.so

echo "== pytest gemm / conv"; date
timeout 900 python -m pytest tests/test_hip_gpu.py tests/test_fp32_gpu.py -m gpu -q -x -k "gemm or conv or adapter or vae or split" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest.log | cut -c1-300
echo "== microbench new / old / new / old"; date
for lib in new old new old; do
  if [ $lib = old ]; then export DWM_HIP_LIB=$OLD DWM_SKIP_SOURCE_HASH=1; else unset DWM_HIP_LIB DWM_SKIP_SOURCE_HASH; fi
  timeout 600 python scripts/microbench.py gemmt 2>&1 | grep gemm_tiles | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$lib', d['case'], d['tflops'])
" | tee -a $OUT/gemm_ab.log
done
echo "== bench new / old / new / old"; date
for lib in new old new old; do
  if [ $lib = old ]; then export DWM_HIP_LIB=$OLD DWM_SKIP_SOURCE_HASH=1; else unset DWM_HIP_LIB DWM_SKIP_SOURCE_HASH; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-text-only-leg 2>> $OUT/bench.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('bench $lib', d['ms_per_step'], d['roofline']['achieved'], d['config']['finite'])
" | tee -a $OUT/bench_ab.log
done
date
