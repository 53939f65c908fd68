#!/bin/bash
# gpurun call E of round 3: the whole GPU suite, smoke, bench (default flags)
TAG=${1:-r3e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
echo "== pytest -m gpu"; date
timeout 2400 python -m pytest tests -m gpu -q -rf --tb=short -p no:cacheprovider --durations=15 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log
tail -45 $OUT/pytest.log | cut -c1-250
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
echo "== smoke"; date
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
echo "== bench"; date
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-2500 $OUT/bench.json; tail -3 $OUT/bench.err
date
