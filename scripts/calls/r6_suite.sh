#!/bin/bash
# the whole GPU suite as the driver runs it (durations recorded), then smoke().  usage: r6_suite.sh <tag> [extra pytest args, e.g. "--maxfail=5"]
TAG=${1:-r6s}
shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
date
timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=25 -rs ${@:--x} > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest.log
tail -45 $OUT/pytest.log | cut -c1-220
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
date
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
date
