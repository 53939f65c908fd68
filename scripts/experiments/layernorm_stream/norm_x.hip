// LayerNorm family + per-head RMSNorm for gfx950.  HBM-bound: one wave per row, the
// row lives in registers (16-B bf16x8 loads/stores), fp32 two-pass statistics via
// wave64 butterfly shuffles; optional fused "x + per-image embedding" prologue and up
// to two modulated outputs (AdaLN-Zero-X) from one read of x.  XF32: the input row (and the optional `x + embedding`
// output) is the fp32 residual stream of the bf16 forward (dwm_layernorm_x32); outputs and parameters stay bf16.
#include "common.h"
#include "dwm_hip.h"

#ifndef LN_NT
#define LN_NT 0          // experiment: 1 = nontemporal stores of y / y2, 2 = and nontemporal loads of the fp32 row
#endif
#ifndef LN_BLOCK
#define LN_BLOCK 256     // experiment: threads per workgroup (one row per wave)
#endif
namespace {
typedef unsigned int __attribute__((ext_vector_type(4))) nt_u32x4;
typedef float __attribute__((ext_vector_type(4))) nt_f32x4;
__device__ __forceinline__ void ln_store16(bf16_t* p, const uint4 v) {
#if LN_NT >= 1
    const nt_u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (nt_u32x4*)p);
#else
    *(uint4*)p = v;
#endif
}
__device__ __forceinline__ float4 ln_load16f(const float* p) {
#if LN_NT >= 2
    const nt_f32x4 t = __builtin_nontemporal_load((const nt_f32x4*)p);
    return make_float4(t.x, t.y, t.z, t.w);
#else
    return *(const float4*)p;
#endif
}

template <int NI, bool XF32 = false>   // NI = ceil(D / 512): 8-element chunks per lane
__global__ void __launch_bounds__(LN_BLOCK)
layernorm_kernel(const dwm_layernorm_args p) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (LN_BLOCK / 64) + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int D = p.D;
    const bf16_t* __restrict__ x = (const bf16_t*)p.x + (XF32 ? 0 : row * p.ldx);
    const float* __restrict__ x32 = (const float*)p.x + (XF32 ? row * p.ldx : 0);

    float v[NI][8];
    bool ok[NI];
    const bf16_t* addv = p.addvec ? (const bf16_t*)p.addvec + (row / p.rows_per_add) * p.ld_add : nullptr;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = (i * 64 + lane) * 8;
        ok[i] = c < D;
        if (ok[i]) {
            if constexpr (XF32) {
                const float4 a0 = ln_load16f(x32 + c), a1 = ln_load16f(x32 + c + 4);
                v[i][0] = a0.x; v[i][1] = a0.y; v[i][2] = a0.z; v[i][3] = a0.w;
                v[i][4] = a1.x; v[i][5] = a1.y; v[i][6] = a1.z; v[i][7] = a1.w;
            } else {
                unpack8(*(const uint4*)(x + c), v[i]);
            }
            if (addv) {
                float a[8];
                unpack8(*(const uint4*)(addv + c), a);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] += a[j];
                if constexpr (XF32) {       // the sum is the VT block's residual stream: kept in fp32
                    if (p.xsum) {
                        float* xs = (float*)p.xsum + row * p.ldxsum + c;
                        *(float4*)xs = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
                        *(float4*)(xs + 4) = make_float4(v[i][4], v[i][5], v[i][6], v[i][7]);
                    }
                } else {                    // ... or rounded to bf16 once, here
                    const uint4 r = pack8(v[i]);
                    unpack8(r, v[i]);
                    if (p.xsum) *(uint4*)((bf16_t*)p.xsum + row * p.ldxsum + c) = r;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
        if (ok[i]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + p.eps);

    const bf16_t* __restrict__ w = (const bf16_t*)p.weight;
    const bf16_t* __restrict__ b = (const bf16_t*)p.bias;
    const int64_t mrow = p.rows_per_mod > 0 ? row / p.rows_per_mod : 0;
    const bf16_t* sc = p.scale ? (const bf16_t*)p.scale + mrow * p.ld_mod : nullptr;
    const bf16_t* sh = p.shift ? (const bf16_t*)p.shift + mrow * p.ld_mod : nullptr;
    const bf16_t* sc2 = p.scale2 ? (const bf16_t*)p.scale2 + mrow * p.ld_mod : nullptr;
    const bf16_t* sh2 = p.shift2 ? (const bf16_t*)p.shift2 + mrow * p.ld_mod : nullptr;
    bf16_t* __restrict__ y = (bf16_t*)p.y + row * p.ldy;
    bf16_t* __restrict__ y2 = p.y2 ? (bf16_t*)p.y2 + row * p.ldy2 : nullptr;

#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (!ok[i]) continue;
        const int c = (i * 64 + lane) * 8;
        float n[8], o[8], t[8], u[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) n[j] = (v[i][j] - mean) * rstd;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = n[j];
        if (w) {
            unpack8(*(const uint4*)(w + c), t);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] *= t[j];
        }
        if (b) {
            unpack8(*(const uint4*)(b + c), t);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += t[j];
        }
        if (sc) {
            unpack8(*(const uint4*)(sc + c), t);
            unpack8(*(const uint4*)(sh + c), u);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = o[j] * (1.f + t[j]) + u[j];
        }
        ln_store16(y + c, pack8(o));
        if (y2) {
            unpack8(*(const uint4*)(sc2 + c), t);
            unpack8(*(const uint4*)(sh2 + c), u);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = n[j] * (1.f + t[j]) + u[j];
            ln_store16(y2 + c, pack8(o));
        }
    }
}

// in-place per-head RMSNorm: one 8-lane group per (row, head) (8 lanes x 8 bf16 = 64)
__global__ void __launch_bounds__(256)
rmsnorm_heads_kernel(bf16_t* __restrict__ x, int64_t ldx, int64_t rows, int64_t nheads,
                     const bf16_t* __restrict__ w, float eps) {
    const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    const bool ok = g < rows * nheads;
    const int64_t gc = ok ? g : 0;
    const int64_t row = gc / nheads, head = gc - row * nheads;
    bf16_t* ptr = x + row * ldx + head * 64 + sub * 8;
    float v[8], t[8];
    unpack8(*(const uint4*)ptr, v);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    ss += __shfl_xor(ss, 4, 64);
    const float r = rsqrtf(ss * (1.f / 64.f) + eps);
    unpack8(*(const uint4*)(w + head * 64 + sub * 8), t);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] * r * t[j];
    if (ok) *(uint4*)ptr = pack8(v);
}

}  // namespace

static int layernorm_launch(const dwm_layernorm_args* a, void* stream, bool x32) {
    if (a == nullptr || a->x == nullptr || a->y == nullptr || a->rows <= 0 || a->D <= 0) return DWM_EINVAL;
    if (a->D % 8 != 0 || a->D > 2048) return DWM_EUNSUPPORTED;
    if (a->ldx % (x32 ? 4 : 8) != 0 || a->ldy % 8 != 0 || !dwm_aligned16(a->x) || !dwm_aligned16(a->y)) return DWM_EALIGN;
    if ((a->scale == nullptr) != (a->shift == nullptr)) return DWM_EINVAL;
    if (a->scale && (a->rows_per_mod <= 0 || a->ld_mod % 8 != 0 || !dwm_aligned16(a->scale) || !dwm_aligned16(a->shift)))
        return DWM_EALIGN;
    if (a->y2 && (a->scale2 == nullptr || a->shift2 == nullptr || a->rows_per_mod <= 0 || a->ldy2 % 8 != 0 ||
                  !dwm_aligned16(a->y2) || !dwm_aligned16(a->scale2) || !dwm_aligned16(a->shift2)))
        return DWM_EALIGN;
    if (a->addvec && (a->rows_per_add <= 0 || a->ld_add % 8 != 0 || !dwm_aligned16(a->addvec))) return DWM_EALIGN;
    if (a->xsum && (a->addvec == nullptr || a->ldxsum % (x32 ? 4 : 8) != 0 || !dwm_aligned16(a->xsum))) return DWM_EALIGN;
    if ((a->weight && !dwm_aligned16(a->weight)) || (a->bias && !dwm_aligned16(a->bias))) return DWM_EALIGN;
    const dim3 grid((unsigned)((a->rows + (LN_BLOCK / 64) - 1) / (LN_BLOCK / 64))), block(LN_BLOCK);
    hipStream_t s = (hipStream_t)stream;
    const int ni = (a->D + 511) / 512;
    if (x32) {
        switch (ni) {
            case 1: hipLaunchKernelGGL((layernorm_kernel<1, true>), grid, block, 0, s, *a); break;
            case 2: hipLaunchKernelGGL((layernorm_kernel<2, true>), grid, block, 0, s, *a); break;
            case 3: hipLaunchKernelGGL((layernorm_kernel<3, true>), grid, block, 0, s, *a); break;
            default: hipLaunchKernelGGL((layernorm_kernel<4, true>), grid, block, 0, s, *a); break;
        }
    } else {
        switch (ni) {
            case 1: hipLaunchKernelGGL((layernorm_kernel<1, false>), grid, block, 0, s, *a); break;
            case 2: hipLaunchKernelGGL((layernorm_kernel<2, false>), grid, block, 0, s, *a); break;
            case 3: hipLaunchKernelGGL((layernorm_kernel<3, false>), grid, block, 0, s, *a); break;
            default: hipLaunchKernelGGL((layernorm_kernel<4, false>), grid, block, 0, s, *a); break;
        }
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DWM_OK : (int)e;
}

extern "C" int dwm_layernorm(const dwm_layernorm_args* a, void* stream) { return layernorm_launch(a, stream, false); }
extern "C" int dwm_layernorm_x32(const dwm_layernorm_args* a, void* stream) { return layernorm_launch(a, stream, true); }

extern "C" int dwm_rmsnorm_heads(void* x, int64_t ldx, int64_t rows, int64_t ncols, const void* w,
                                 float eps, void* stream) {
    if (x == nullptr || w == nullptr || rows <= 0 || ncols <= 0) return DWM_EINVAL;
    if (ncols % 64 != 0 || ldx % 8 != 0 || !dwm_aligned16(x) || !dwm_aligned16(w)) return DWM_EALIGN;
    const int64_t nheads = ncols / 64;
    const int64_t nthreads = rows * nheads * 8;
    hipLaunchKernelGGL(rmsnorm_heads_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (bf16_t*)x, ldx, rows, nheads, (const bf16_t*)w, eps);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DWM_OK : (int)e;
}
