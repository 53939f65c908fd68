#!/bin/bash
# gpurun call G of round 4: the temporal VAE in fp32 (27-tap dwm_gemm_f32 in groups of 9, fp32 spatial norm / frame mix), the whole
# fp32 file (the A-row count of dwm_gemm_f32 changed) and the bf16 temporal-VAE file (its kernels became templates)
TAG=${1:-r4g}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity.log
timeout 1200 python -m pytest tests/test_fp32_gpu.py tests/test_cogvideox_gpu.py -q -p no:cacheprovider --durations=8 > $OUT/pytest.log 2>&1; echo "exit $?"; tail -30 $OUT/pytest.log | cut -c1-400
cp gpurun_out/gpu_parity.log $OUT/gpu_parity.log 2>/dev/null
grep -i "temporal_vae" $OUT/gpu_parity.log | cut -c1-400
