// C entry point of the bf16 GEMM and the 8-wave (2 x 4, 128x64 per wave) instantiation; the kernel is in gemm_kernel.h
#include "gemm_kernel.h"

using namespace dwm_gemm;

extern "C" int dwm_gemm_bf16(const dwm_gemm_args* a, void* stream) {
    if (a == nullptr || a->A == nullptr || a->W == nullptr || a->C == nullptr) return DWM_EINVAL;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->M >= (1ll << 31) || a->N >= (1ll << 31)) return DWM_EINVAL;
    if (a->K % BK != 0 || a->N % 8 != 0) return DWM_EUNSUPPORTED;
    if (a->lda % 8 != 0 || a->ldc % 8 != 0) return DWM_EALIGN;
    if (!dwm_aligned16(a->A) || !dwm_aligned16(a->W) || !dwm_aligned16(a->C)) return DWM_EALIGN;
    if (a->bias && (((uintptr_t)a->bias) & 7u)) return DWM_EALIGN;
    const int64_t nout = a->epilogue == DWM_EPI_GEGLU ? a->N / 2 : a->N;
    if (a->ldc < nout) return DWM_EINVAL;
    switch (a->epilogue) {
        case DWM_EPI_PLAIN: break;
        case DWM_EPI_GEGLU:
            if (a->N % 64 != 0 || (a->N / 2) % 8 != 0) return DWM_EUNSUPPORTED;
            break;
        case DWM_EPI_RESID:
            if (a->gate && (a->rows_per_gate <= 0 || a->ld_gate % 8 != 0 || !dwm_aligned16(a->gate))) return DWM_EINVAL;
            if (a->res && (a->ld_res % 8 != 0 || !dwm_aligned16(a->res))) return DWM_EALIGN;
            if (a->blend && (a->alpha == nullptr || a->rows_per_alpha <= 0 || a->ld_blend % 8 != 0 || !dwm_aligned16(a->blend))) return DWM_EINVAL;
            break;
        case DWM_EPI_RMSHEAD:
            if (a->rms_w == nullptr || a->rms_ncols % 64 != 0 || a->N % 64 != 0) return DWM_EINVAL;
            break;
        default: return DWM_EINVAL;
    }
    ConvParams cp;
    auto mk = [](const dwm_rowmap2d& r, DevRowMap& d) -> bool {
        d.enabled = r.rw > 0;
        d.xstep = r.xstep > 0 ? (int)r.xstep : 1;
        if (!d.enabled) { d.rw = make_fastdiv(1); d.rh = make_fastdiv(1); d.rpitch = d.ipitch = d.origin = 0; return true; }
        if (r.rh <= 0 || r.rw >= (1ll << 30) || r.rh >= (1ll << 30)) return false;
        d.rw = make_fastdiv((uint32_t)r.rw); d.rh = make_fastdiv((uint32_t)r.rh);
        d.rpitch = r.rpitch; d.ipitch = r.ipitch; d.origin = r.origin;
        return true;
    };
    if (!mk(a->a_map, cp.a) || !mk(a->c_map, cp.c)) return DWM_EINVAL;
    const int ntaps = a->ntaps > 0 ? a->ntaps : 1;
    if (ntaps > 9) return DWM_EINVAL;
    const int64_t kpt = a->ntaps > 0 ? a->k_per_tap : a->K;
    if (kpt <= 0 || kpt % BK != 0 || kpt * ntaps != a->K) return DWM_EINVAL;
    cp.steps_per_tap = (int)(kpt / BK);
    cp.fd_steps = make_fastdiv((uint32_t)cp.steps_per_tap);
    for (int t = 0; t < 9; ++t) cp.tap_shift[t] = (a->ntaps > 0 && t < ntaps) ? a->tap_shift[t] : 0;
    if (a->lda < kpt) return DWM_EINVAL;
    const int ntm = (int)((a->M + BM - 1) / BM), ntn = (int)((a->N + BN - 1) / BN);
    hipStream_t s = (hipStream_t)stream;
    // default: the 8-wave kernel (2 waves per SIMD); reserved bit 10 selects the 4-wave kernel
    // (one wave per SIMD, AGPR accumulators) - measured slower, kept for experiments
    const bool w4 = (a->reserved & (1 << 10)) != 0;
    return w4 ? dwm_gemm_launch_w4(a, cp, ntm, ntn, s) : launch_variant<4>(a, cp, ntm, ntn, s);
}
