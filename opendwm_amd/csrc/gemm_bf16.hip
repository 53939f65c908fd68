// bf16 GEMM with fused epilogues for gfx950:  C[M,Nout] = epi(A[M,K] · W[N,K]^T)
//
// Block tile 256x256x64, 512 threads = 8 waves laid out 2 (M) x 4 (N); each wave
// owns a 128x64 output tile = 4x2 v_mfma_f32_32x32x16_bf16 accumulators (128 fp32
// VGPR/lane).  The MFMA is issued "swapped" (A-operand = W fragment, B-operand =
// activation fragment) so a lane owns ONE output row m and, per accumulator
// register group, FOUR CONSECUTIVE output columns n:
//     m = lane & 31,  n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)      (r = 0..15)
// That makes every epilogue lane-local: 8-byte bf16x4 stores, GEGLU pairs
// (value / gate sub-tiles of the same wave), and the per-head (64 col) RMSNorm of
// q/k needs one lane^32 exchange.
//
// Global -> LDS staging is the gfx950 LDS-DMA (global_load_lds_dwordx4): the LDS
// image of a tile is [256 rows][8 x 16-B chunks] with chunk XOR-swizzled by
// ((row >> 1) & 7); the swizzle is applied on the per-lane SOURCE address (the DMA
// destination is lane-linear) and again on the ds_read_b128 fragment reads, which
// makes those reads bank-conflict free.  Two LDS stages (128 KiB), one barrier per
// K step: tile k+1 streams in while tile k is on the matrix cores.
#include "common.h"
#include "dwm_hip.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int NTHREADS = 512;
constexpr int TILE_BYTES = BM * BK * 2;            // 32 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;        // A + W
constexpr int LDS_BYTES = 2 * STAGE_BYTES;         // double buffered = 128 KiB

template <int EPI>
__global__ void __launch_bounds__(NTHREADS, 2)
gemm_bf16_kernel(const dwm_gemm_args p, const int ntm, const int ntn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2;              // 0..1  (128-row slab)
    const int wn = wave & 3;               // 0..3  (64-col slab)
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int id = xcd_remap(blockIdx.x, ntm * ntn);
    const int tm = id / ntn, tn = id - tm * ntn;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int64_t M = p.M, N = p.N, K = p.K;

    // ---- staging addresses: wave w copies row groups (w*4 + j)*8 .. +8, j = 0..3
    const bf16_t* __restrict__ Ap = (const bf16_t*)p.A;
    const bf16_t* __restrict__ Wp = (const bf16_t*)p.W;
    const char* a_src[4];
    const char* w_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);          // logical 16-B chunk this lane fetches
        int64_t gm = m0 + row; gm = gm < M ? gm : M - 1;
        int64_t gn = n0 + row; gn = gn < N ? gn : N - 1;
        a_src[j] = (const char*)(Ap + gm * p.lda + c * 8);
        w_src[j] = (const char*)(Wp + gn * K + c * 8);
    }
    auto stage = [&](int buf, int kt) {
        char* la = smem + buf * STAGE_BYTES;
        char* lb = la + TILE_BYTES;
        const int64_t koff = (int64_t)kt * (BK * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r0 = (wave * 4 + j) * 8;
            glds16(a_src[j] + koff, la + r0 * 128);
            glds16(w_src[j] + koff, lb + r0 * 128);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets (bytes) inside a tile: row*128 + ((chunk ^ swz) << 4)
    const int swz = (lane >> 1) & 7;       // ((row >> 1) & 7) with row = 32*t + (lane & 31)
    const int a_row_off = (wm * 128 + l31) * 128;
    const int w_row_off = (wn * 64 + l31) * 128;

    const int nk = (int)(K / BK);
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // tile kt landed; buffer (kt+1)&1 is free
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const char* la = smem + (kt & 1) * STAGE_BYTES;
        const char* lb = la + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int coff = ((2 * ks + half) ^ swz) << 4;
            bf16x8 af[4], wf[2];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                af[mt] = *(const bf16x8*)(la + a_row_off + mt * (32 * 128) + coff);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                wf[nt] = *(const bf16x8*)(lb + w_row_off + nt * (32 * 128) + coff);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
        }
    }

    // ------------------------------------------------------------------ epilogue
    const bf16_t* __restrict__ bias = (const bf16_t*)p.bias;
    bf16_t* __restrict__ Cp = (bf16_t*)p.C;
    const int64_t nw = n0 + wn * 64;                      // first column of this wave's slab

    // bias for this lane's 2 x 16 columns
    float bv[2][16];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int64_t n = nw + nt * 32 + rg * 8 + half * 4;
            if (bias != nullptr && n < N) {
                const uint2 b = *(const uint2*)(bias + n);
                unpack4(b, &bv[nt][rg * 4]);
            } else {
                bv[nt][rg * 4 + 0] = bv[nt][rg * 4 + 1] = bv[nt][rg * 4 + 2] = bv[nt][rg * 4 + 3] = 0.f;
            }
        }

#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int64_t m = m0 + wm * 128 + mt * 32 + l31;
        const bool mok = m < M;

        if constexpr (EPI == DWM_EPI_PLAIN) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int64_t n = nw + nt * 32 + rg * 8 + half * 4;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x = acc[mt][nt][rg * 4 + j] + bv[nt][rg * 4 + j];
                        if (p.act == DWM_ACT_GELU_TANH) x = gelu_tanh_f(x);
                        else if (p.act == DWM_ACT_SILU) x = silu_f(x);
                        v[j] = x;
                    }
                    if (mok && n < N) *(uint2*)(Cp + m * p.ldc + n) = pack4(v);
                }
        } else if constexpr (EPI == DWM_EPI_GEGLU) {
            // value rows in sub-tile nt=0, gate rows in nt=1 (weight packed that way)
            const int64_t nout = (n0 >> 1) + wn * 32;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int64_t n = nout + rg * 8 + half * 4;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float hv = acc[mt][0][rg * 4 + j] + bv[0][rg * 4 + j];
                    const float gv = acc[mt][1][rg * 4 + j] + bv[1][rg * 4 + j];
                    v[j] = hv * gelu_erf_f(gv);
                }
                if (mok && n < (N >> 1)) *(uint2*)(Cp + m * p.ldc + n) = pack4(v);
            }
        } else if constexpr (EPI == DWM_EPI_RESID) {
            const bf16_t* __restrict__ gate = (const bf16_t*)p.gate;
            const bf16_t* __restrict__ res = (const bf16_t*)p.res;
            const bf16_t* __restrict__ blend = (const bf16_t*)p.blend;
            const int64_t mc = mok ? m : M - 1;
            const bf16_t* grow = gate ? gate + (mc / p.rows_per_gate) * p.ld_gate : nullptr;
            const bf16_t* rrow = res ? res + (p.res_mod > 0 ? mc % p.res_mod : mc) * p.ld_res : nullptr;
            const bf16_t* brow = blend ? blend + mc * p.ld_blend : nullptr;
            const float alpha = blend ? p.alpha[mc / p.rows_per_alpha] : 0.f;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int64_t n = nw + nt * 32 + rg * 8 + half * 4;
                    if (!(mok && n < N)) continue;
                    float v[4], t[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x = acc[mt][nt][rg * 4 + j] + bv[nt][rg * 4 + j];
                        if (p.act == DWM_ACT_GELU_TANH) x = gelu_tanh_f(x);
                        else if (p.act == DWM_ACT_SILU) x = silu_f(x);
                        v[j] = x;
                    }
                    if (grow) {
                        unpack4(*(const uint2*)(grow + n), t);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] *= t[j];
                    }
                    if (rrow) {
                        unpack4(*(const uint2*)(rrow + n), t);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += t[j];
                    }
                    if (brow) {
                        unpack4(*(const uint2*)(brow + n), t);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = alpha * t[j] + (1.f - alpha) * v[j];
                    }
                    *(uint2*)(Cp + m * p.ldc + n) = pack4(v);
                }
        } else {   // DWM_EPI_RMSHEAD: this wave's 64 columns are exactly one head
            const bool do_norm = nw < p.rms_ncols;         // wave-uniform
            float ss = 0.f;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float x = acc[mt][nt][r] + bv[nt][r];
                    acc[mt][nt][r] = x;
                    ss += x * x;
                }
            ss += __shfl_xor(ss, 32, 64);                   // other half of the row lives in lane^32
            const float rinv = do_norm ? rsqrtf(ss * (1.f / 64.f) + p.rms_eps) : 1.f;
            const bf16_t* __restrict__ rw = (const bf16_t*)p.rms_w;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int64_t n = nw + nt * 32 + rg * 8 + half * 4;
                    float v[4], t[4] = {1.f, 1.f, 1.f, 1.f};
                    if (do_norm && n < N) unpack4(*(const uint2*)(rw + n), t);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][rg * 4 + j] * rinv * t[j];
                    if (mok && n < N) *(uint2*)(Cp + m * p.ldc + n) = pack4(v);
                }
        }
    }
}

}  // namespace

extern "C" int dwm_gemm_bf16(const dwm_gemm_args* a, void* stream) {
    if (a == nullptr || a->A == nullptr || a->W == nullptr || a->C == nullptr) return DWM_EINVAL;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return DWM_EINVAL;
    if (a->K % BK != 0 || a->N % 8 != 0) return DWM_EUNSUPPORTED;
    if (a->lda % 8 != 0 || a->ldc % 4 != 0 || a->lda < a->K) return DWM_EALIGN;
    if (!dwm_aligned16(a->A) || !dwm_aligned16(a->W) || (((uintptr_t)a->C) & 7u)) return DWM_EALIGN;
    if (a->bias && (((uintptr_t)a->bias) & 7u)) return DWM_EALIGN;
    const int64_t nout = a->epilogue == DWM_EPI_GEGLU ? a->N / 2 : a->N;
    if (a->ldc < nout) return DWM_EINVAL;
    switch (a->epilogue) {
        case DWM_EPI_PLAIN: break;
        case DWM_EPI_GEGLU:
            if (a->N % 64 != 0) return DWM_EUNSUPPORTED;
            break;
        case DWM_EPI_RESID:
            if (a->gate && (a->rows_per_gate <= 0 || a->ld_gate % 4 != 0)) return DWM_EINVAL;
            if (a->res && a->ld_res % 4 != 0) return DWM_EALIGN;
            if (a->blend && (a->alpha == nullptr || a->rows_per_alpha <= 0 || a->ld_blend % 4 != 0)) return DWM_EINVAL;
            break;
        case DWM_EPI_RMSHEAD:
            if (a->rms_w == nullptr || a->rms_ncols % 64 != 0 || a->N % 64 != 0) return DWM_EINVAL;
            break;
        default: return DWM_EINVAL;
    }
    const int ntm = (int)((a->M + BM - 1) / BM), ntn = (int)((a->N + BN - 1) / BN);
    const dim3 grid((unsigned)(ntm * ntn)), block(NTHREADS);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
#define DWM_LAUNCH(EPI)                                                                              \
    do {                                                                                             \
        static bool attr_set = false;                                                                \
        if (!attr_set) {                                                                             \
            e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI>,                              \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);          \
            if (e != hipSuccess) return (int)e;                                                      \
            attr_set = true;                                                                         \
        }                                                                                            \
        hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, grid, block, LDS_BYTES, s, *a, ntm, ntn);          \
    } while (0)
    switch (a->epilogue) {
        case DWM_EPI_PLAIN: DWM_LAUNCH(DWM_EPI_PLAIN); break;
        case DWM_EPI_GEGLU: DWM_LAUNCH(DWM_EPI_GEGLU); break;
        case DWM_EPI_RESID: DWM_LAUNCH(DWM_EPI_RESID); break;
        default: DWM_LAUNCH(DWM_EPI_RMSHEAD); break;
    }
#undef DWM_LAUNCH
    e = hipGetLastError();
    return e == hipSuccess ? DWM_OK : (int)e;
}
