// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libdwm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef uint16_t bf16_t;                                         // raw bfloat16 storage
typedef __attribute__((ext_vector_type(8))) short bf16x8;        // MFMA A/B operand (8 bf16, 4 VGPR)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;       // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;

#define DWM_DEVINL __device__ __forceinline__

DWM_DEVINL float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// two fp32 -> packed bf16x2 (round-to-nearest-even), lowers to v_cvt_pk_bf16_f32
DWM_DEVINL uint32_t pack_bf16x2(float lo, float hi) {
    f32x2 v = {lo, hi};
    bf16x2_hw b = __builtin_convertvector(v, bf16x2_hw);
    return *reinterpret_cast<uint32_t*>(&b);
}
DWM_DEVINL bf16_t f32_to_bf16(float v) { return (bf16_t)(pack_bf16x2(v, 0.f) & 0xffffu); }

DWM_DEVINL float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
DWM_DEVINL float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// 8 bf16 (16 B) -> 8 fp32
DWM_DEVINL void unpack8(const uint4& v, float* f) {
    f[0] = bf16lo(v.x); f[1] = bf16hi(v.x); f[2] = bf16lo(v.y); f[3] = bf16hi(v.y);
    f[4] = bf16lo(v.z); f[5] = bf16hi(v.z); f[6] = bf16lo(v.w); f[7] = bf16hi(v.w);
}
DWM_DEVINL uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
    return v;
}
DWM_DEVINL void unpack4(const uint2& v, float* f) {
    f[0] = bf16lo(v.x); f[1] = bf16hi(v.x); f[2] = bf16lo(v.y); f[3] = bf16hi(v.y);
}
DWM_DEVINL uint2 pack4(const float* f) {
    uint2 v; v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]); return v;
}

// 8 consecutive elements of a bf16 or fp32 row <-> 8 fp32 registers (16 / 32 bytes, 16-byte aligned): the element-type
// generic form of the token-major kernels that exist for both the bf16 path and the fp32 accuracy path
template <typename T> DWM_DEVINL void load8(const T* p, float* f);
template <> DWM_DEVINL void load8<bf16_t>(const bf16_t* p, float* f) { unpack8(*(const uint4*)p, f); }
template <> DWM_DEVINL void load8<float>(const float* p, float* f) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <typename T> DWM_DEVINL void store8(T* p, const float* f);
template <> DWM_DEVINL void store8<bf16_t>(bf16_t* p, const float* f) { *(uint4*)p = pack8(f); }
template <> DWM_DEVINL void store8<float>(float* p, const float* f) {
    *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
    *(float4*)(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}
template <typename T> DWM_DEVINL void copy8(T* dst, const T* src);
template <> DWM_DEVINL void copy8<bf16_t>(bf16_t* dst, const bf16_t* src) { *(uint4*)dst = *(const uint4*)src; }
template <> DWM_DEVINL void copy8<float>(float* dst, const float* src) {
    *(float4*)dst = *(const float4*)src;
    *(float4*)(dst + 4) = *(const float4*)(src + 4);
}

DWM_DEVINL float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The same sum by DPP row operations instead of six LDS-routed shuffles (ds_bpermute: an LDS round trip each): quad swaps,
// half-row / row mirrors, then the two cross-row broadcasts of GFX9 (row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and
// 3); the total lands in lane 63 and is read back as a scalar.  For kernels whose row loop is latency bound on its reductions
// (LayerNorm backward: four per row).
DWM_DEVINL float wave_sum_dpp(float v) {
    auto add_dpp = [](float x, auto ctrl, auto row_mask) {
        const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false);
        return x + __int_as_float(y);
    };
    using std::integral_constant;
    v = add_dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{});       // quad_perm [1,0,3,2]
    v = add_dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{});       // quad_perm [2,3,0,1]
    v = add_dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{});      // row_half_mirror
    v = add_dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{});      // row_mirror: every lane holds its row's sum
    v = add_dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});      // row_bcast15 -> rows 1, 3
    v = add_dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});      // row_bcast31 -> rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// 0.5 x (1 + tanh(u)) = x * sigmoid(2u),  u = sqrt(2/pi) (x + 0.044715 x^3); one v_exp + one v_rcp
DWM_DEVINL float gelu_tanh_f(float x) {
    const float z = x * (2.302208198f + 0.1029432397f * x * x);          // 2 u log2(e)
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-z));
}
// erf by Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7, far below bf16 output resolution)
DWM_DEVINL float erf_fast_f(float x) {
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = 1.f - p * t * e;
    return __builtin_copysignf(r, x);
}
DWM_DEVINL float gelu_erf_f(float x) { return 0.5f * x * (1.f + erf_fast_f(x * 0.7071067811865476f)); }
DWM_DEVINL float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

// ---- packed-fp32 forms (v_pk_mul/add/fma_f32 process two lanes' worth of values per issue slot; the
//      transcendental ops stay scalar).  Used by the GEMM epilogues, which are VALU-issue bound.
DWM_DEVINL f32x2 splat2(float v) { return (f32x2){v, v}; }
DWM_DEVINL f32x2 exp2_2(f32x2 z) { return (f32x2){__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])}; }
DWM_DEVINL f32x2 rcp_2(f32x2 d) { return (f32x2){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])}; }
DWM_DEVINL f32x2 gelu_tanh2(f32x2 x) {
    const f32x2 u = (x * x) * splat2(-0.1029432397f) + splat2(-2.302208198f);    // -(2 log2e sqrt(2/pi)) (1 + 0.044715 x^2)
    return x * rcp_2(exp2_2(x * u) + splat2(1.f));
}
DWM_DEVINL f32x2 silu2(f32x2 x) { return x * rcp_2(exp2_2(x * splat2(-1.4426950408889634f)) + splat2(1.f)); }
DWM_DEVINL f32x2 relu2(f32x2 x) { return (f32x2){fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)}; }
DWM_DEVINL f32x2 gelu_erf2(f32x2 x) {
    const f32x2 ax = (f32x2){__builtin_fabsf(x[0]), __builtin_fabsf(x[1])} * splat2(0.7071067811865476f);   // |x| / sqrt 2
    const f32x2 t = rcp_2(ax * splat2(0.3275911f) + splat2(1.f));
    f32x2 pl = t * splat2(1.061405429f) + splat2(-1.453152027f);
    pl = pl * t + splat2(1.421413741f);
    pl = pl * t + splat2(-0.284496736f);
    pl = pl * t + splat2(0.254829592f);
    const f32x2 e = exp2_2((ax * ax) * splat2(-1.4426950408889634f));
    const f32x2 r = splat2(1.f) - (pl * t) * e;                                   // erf(|x| / sqrt 2), Abramowitz-Stegun 7.1.26
    const f32x2 hx = x * splat2(0.5f);
    const f32x2 er = (f32x2){__builtin_copysignf(r[0], x[0]), __builtin_copysignf(r[1], x[1])};
    return hx * er + hx;
}

// async global -> LDS, 16 B per lane; LDS destination = wave-uniform base + lane*16
// (the tiled and the resident forward attention kernels use this builtin form: with the requests as inline asm - the form of
//  the GEMMs, the backward and the cross-view kernels, whose waits the compiler otherwise degrades - the resident kernel ran
//  12 % SLOWER and the tiled one the same, profiles/r4i_microbench_attention_asm_dma_in_forward_kernels.log)
DWM_DEVINL void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gsrc,
        (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware bijective block remap (8 XCDs, block b runs on XCD b % 8): gives every
// XCD a contiguous range of logical ids so neighbouring tiles share that XCD's L2.
DWM_DEVINL int xcd_remap(int bid, int nblk) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Division by a launch-invariant divisor d >= 1 for n < 2^31 (Granlund-Montgomery round-up
// form): q = (mulhi(m, n) + n) >> s,  s = ceil(log2 d),  m = floor(2^32 (2^s - d) / d) + 1.
// Replaces ~40-instruction runtime integer divides in per-block address arithmetic.
struct FastDiv {
    uint32_t m, s, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    f.s = s;
    f.m = (uint32_t)((((1ull << s) - d) << 32) / d + 1);
    return f;
}
DWM_DEVINL uint32_t fdiv(uint32_t n, const FastDiv& f) { return (__umulhi(f.m, n) + n) >> f.s; }
DWM_DEVINL uint32_t fmod_u(uint32_t n, const FastDiv& f) { return n - fdiv(n, f) * f.d; }

static inline bool dwm_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
