#!/bin/bash
# round 6, call N: layernorm_kernel on the fp32 stream - nontemporal stores / loads, workgroup size (one row per wave): isolated timing
export TAG=${1:-r6n}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/opendwm_amd/variants
for rep in 1 2; do
echo "-- default"; timeout 200 python scripts/microbench.py ln32 2>&1 | grep '"ln-' | tee -a $OUT/ln_default.log
for lib in ln_nt1 ln_nt2 ln_b128 ln_b512 ln_b64nt2; do
  echo "-- $lib"; DWM_HIP_LIB=$V/libdwm_hip_$lib.so timeout 200 python scripts/microbench.py ln32 2>&1 | grep '"ln-' | tee -a $OUT/$lib.log
done
done
