// 4-wave (2 x 2, 128x128 per wave) instantiation of the GEMM kernel.  Built WITHOUT
// -amdgpu-mfma-vgpr-form: the 256 accumulator registers of a wave live in AGPRs, the fragments and
// addresses in the 256 arch VGPRs (one wave per SIMD owns the whole 512-entry file).
#include "gemm_kernel.h"

int dwm_gemm_launch_w4(const dwm_gemm_args* a, const dwm_gemm::ConvParams& cp, int ntm, int ntn, hipStream_t s) {
    return dwm_gemm::launch_variant<2>(a, cp, ntm, ntn, s);
}
