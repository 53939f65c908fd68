// Small HBM-bound glue kernels of the denoise step (gfx950): 16-B vector accesses,
// grid sized to the data, no LDS.
#include "common.h"
#include "dwm_hip.h"

namespace {

__global__ void __launch_bounds__(256)
silu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float v[8];
    unpack8(*(const uint4*)(x + i * 8), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = silu_f(v[j]);
    *(uint4*)(y + i * 8) = pack8(v);
}

// out[i, j] = cos(t_i f_j) for j < C/2, sin(t_i f_{j - C/2}) otherwise
__global__ void __launch_bounds__(256)
sinusoid_kernel(const float* __restrict__ t, int64_t n, int C, bf16_t* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int half = C >> 1;
    if (idx >= n * half) return;
    const int64_t i = idx / half;
    const int j = (int)(idx - i * half);
    const float f = expf(-9.210340371976184f * (float)j / (float)half);   // ln(10000)
    const float a = t[i] * f;
    out[i * C + j] = f32_to_bf16(cosf(a));
    out[i * C + half + j] = f32_to_bf16(sinf(a));
}

DWM_DEVINL float ld_as_f32(const float* p, int64_t i) { return p[i]; }
DWM_DEVINL float ld_as_f32(const bf16_t* p, int64_t i) { return bf16_to_f32(p[i]); }

// one thread per output element of the im2col matrix
template <typename TIN>
__global__ void __launch_bounds__(256)
patchify_kernel(const TIN* __restrict__ x, int64_t I, int C, int H, int W, int p,
                bf16_t* __restrict__ out, int64_t ldo) {
    const int h = H / p, w = W / p;
    const int cols = C * p * p;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = I * h * w * ldo;
    if (idx >= total) return;
    const int64_t tok = idx / ldo;
    const int col = (int)(idx - tok * ldo);
    float v = 0.f;
    if (col < cols) {
        const int px = col % p, py = (col / p) % p, c = col / (p * p);
        const int ww = (int)(tok % w), hh = (int)((tok / w) % h);
        const int64_t img = tok / ((int64_t)w * h);
        v = ld_as_f32(x, ((img * C + c) * H + hh * p + py) * W + ww * p + px);
    }
    out[idx] = f32_to_bf16(v);
}
// out[n, c, hh*p+py, ww*p+px] = x[(n*h + hh)*w + ww, (py*p + px)*C + c]; one thread per output element
__global__ void __launch_bounds__(256)
unpatchify_kernel(const bf16_t* __restrict__ x, int64_t ldx, int64_t I, int C, int h, int w, int p,
                  bf16_t* __restrict__ out) {
    const int H = h * p, W = w * p;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= I * C * H * W) return;
    const int X = (int)(idx % W), Y = (int)((idx / W) % H);
    const int c = (int)((idx / ((int64_t)W * H)) % C);
    const int64_t n = idx / ((int64_t)W * H * C);
    const int ww = X / p, px = X - ww * p, hh = Y / p, py = Y - hh * p;
    out[idx] = x[((n * h + hh) * w + ww) * ldx + (py * p + px) * C + c];
}

__global__ void __launch_bounds__(256)
cfg_euler_kernel(const bf16_t* __restrict__ pred, float* __restrict__ lat, bf16_t* __restrict__ model_in,
                 int64_t n, float guidance, float dsigma, const float* __restrict__ dsigma_group, int64_t group_elems) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (dsigma_group != nullptr) dsigma = dsigma_group[i / group_elems];     // per-frame step (diffusion forcing)
    float u[4], c[4];
    unpack4(*(const uint2*)(pred + i), u);
    unpack4(*(const uint2*)(pred + n + i), c);
    float4 l = *(const float4*)(lat + i);
    float o[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] += dsigma * (u[j] + guidance * (c[j] - u[j]));
    *(float4*)(lat + i) = make_float4(o[0], o[1], o[2], o[3]);
    if (model_in) {
        const uint2 b = pack4(o);
        *(uint2*)(model_in + i) = b;
        *(uint2*)(model_in + n + i) = b;
    }
}

// classifier-free guidance + one linear multistep scheduler update (DPM-Solver++ 1st / 2nd order, DDIM, ...):
//   out = u + g (c - u);  x0 = kx * x + ko * out;  x' = A * x + B * x0 + C * x0_prev;  x0_prev' = x0
template <typename T>      // T: bf16_t, or float (the fp32 accuracy path: dwm_cfg_multistep_f32)
__global__ void __launch_bounds__(256)
cfg_multistep_kernel(const T* __restrict__ pred, float* __restrict__ lat, float* __restrict__ x0_prev,
                     T* __restrict__ model_in, int64_t n, float guidance, float kx, float ko, float A, float B, float Cc) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float u[4], c[4];
    if constexpr (sizeof(T) == 4) {
        const float4 a = *(const float4*)(pred + i), b = *(const float4*)(pred + n + i);
        u[0] = a.x; u[1] = a.y; u[2] = a.z; u[3] = a.w; c[0] = b.x; c[1] = b.y; c[2] = b.z; c[3] = b.w;
    } else {
        unpack4(*(const uint2*)(pred + i), u);
        unpack4(*(const uint2*)(pred + n + i), c);
    }
    const float4 l = *(const float4*)(lat + i);
    const float4 pv = *(const float4*)(x0_prev + i);
    const float x[4] = {l.x, l.y, l.z, l.w}, pr[4] = {pv.x, pv.y, pv.z, pv.w};
    float o[4], z[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        z[j] = kx * x[j] + ko * (u[j] + guidance * (c[j] - u[j]));
        o[j] = A * x[j] + B * z[j] + Cc * pr[j];
    }
    *(float4*)(lat + i) = make_float4(o[0], o[1], o[2], o[3]);
    *(float4*)(x0_prev + i) = make_float4(z[0], z[1], z[2], z[3]);
    if (model_in) {
        if constexpr (sizeof(T) == 4) {
            *(float4*)(model_in + i) = make_float4(o[0], o[1], o[2], o[3]);
            *(float4*)(model_in + n + i) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            const uint2 b = pack4(o);
            *(uint2*)(model_in + i) = b;
            *(uint2*)(model_in + n + i) = b;
        }
    }
}

// per-frame affine combination of two fp32 tensors (coefficients per group of `group_elems` consecutive elements):
//   out = c[g][0] * x + c[g][1] * y      - DDPMScheduler.add_noise / get_velocity with a timestep per (sample, frame, view)
//   (src/dwm/schedulers/temporal_independent.py:8-45)
__global__ void __launch_bounds__(256)
frame_affine_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ coef,
                    float* __restrict__ out, bf16_t* __restrict__ out_bf16, int64_t n, int64_t group_elems) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const int64_t g = i / group_elems;
    const float c0 = coef[2 * g], c1 = coef[2 * g + 1];
    const float4 a = *(const float4*)(x + i), b = *(const float4*)(y + i);
    const float o[4] = {c0 * a.x + c1 * b.x, c0 * a.y + c1 * b.y, c0 * a.z + c1 * b.z, c0 * a.w + c1 * b.w};
    if (out) *(float4*)(out + i) = make_float4(o[0], o[1], o[2], o[3]);
    if (out_bf16) *(uint2*)(out_bf16 + i) = pack4(o);
}

// classifier-free guidance + the tensor-timestep DDIM update (src/dwm/schedulers/temporal_independent.py:67-170), one
// coefficient row per group of `group_elems` elements (= per (sample, frame, view) timestep):
//   coef[g] = { sqrt(a_t), sqrt(1 - a_t), sqrt(a_prev), sqrt(1 - a_prev - std^2), std, 0 }
//   m = u + g (c - u) | u;   (x0, eps) from (sample, m) by prediction type;   x0 clipped;   eps re-derived if asked;
//   prev = sqrt(a_prev) x0 + dir eps + std noise
template <typename PT>
__global__ void __launch_bounds__(256)
cfg_ddim_kernel(const PT* __restrict__ pred, int64_t cond_offset, float* __restrict__ lat, bf16_t* __restrict__ model_in,
                float* __restrict__ x0_out, const float* __restrict__ noise, const float* __restrict__ coef, int64_t n,
                int64_t group_elems, float guidance, int ptype, float clip, int use_clipped) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float* cg = coef + 6 * (i / group_elems);
    const float sa = cg[0], sb = cg[1], sap = cg[2], dir = cg[3], sd = cg[4];
    float m[4];
    if constexpr (sizeof(PT) == 2) {
        unpack4(*(const uint2*)(pred + i), m);
        if (cond_offset) {
            float c[4];
            unpack4(*(const uint2*)(pred + cond_offset + i), c);
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] += guidance * (c[j] - m[j]);
        }
    } else {
        const float4 u = *(const float4*)(pred + i);
        m[0] = u.x; m[1] = u.y; m[2] = u.z; m[3] = u.w;
        if (cond_offset) {
            const float4 c = *(const float4*)(pred + cond_offset + i);
            m[0] += guidance * (c.x - m[0]); m[1] += guidance * (c.y - m[1]);
            m[2] += guidance * (c.z - m[2]); m[3] += guidance * (c.w - m[3]);
        }
    }
    const float4 l = *(const float4*)(lat + i);
    const float s[4] = {l.x, l.y, l.z, l.w};
    float nz[4] = {0.f, 0.f, 0.f, 0.f};
    if (noise) {
        const float4 q = *(const float4*)(noise + i);
        nz[0] = q.x; nz[1] = q.y; nz[2] = q.z; nz[3] = q.w;
    }
    float o[4], z[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x0, eps;
        if (ptype == 0) { x0 = (s[j] - sb * m[j]) / sa; eps = m[j]; }                       // epsilon
        else if (ptype == 1) { x0 = m[j]; eps = (s[j] - sa * x0) / sb; }                    // sample
        else { x0 = sa * s[j] - sb * m[j]; eps = sa * m[j] + sb * s[j]; }                   // v_prediction
        if (clip > 0.f) x0 = fminf(fmaxf(x0, -clip), clip);
        if (use_clipped) eps = (s[j] - sa * x0) / sb;
        z[j] = x0;
        o[j] = sap * x0 + dir * eps + sd * nz[j];
    }
    *(float4*)(lat + i) = make_float4(o[0], o[1], o[2], o[3]);
    if (x0_out) *(float4*)(x0_out + i) = make_float4(z[0], z[1], z[2], z[3]);
    if (model_in) {
        const uint2 b = pack4(o);
        *(uint2*)(model_in + i) = b;
        if (cond_offset) *(uint2*)(model_in + n + i) = b;
    }
}

// Explicit perspective modelling (crossview_temporal_dit.py:11-102): per latent token, the positional encoding of the camera
// origin (3 coordinates x 8 octaves) and of the unit view ray through the token centre (3 x 4 octaves) - the 72 inputs of
// RayEncoder.proj.  cam[i] = { Kinv (9, row-major inverse of the token-resolution intrinsics), R (9, camera->ego
// rotation), o (3, camera origin) } fp32.  out[(i*h + y)*w + x][0:72] = [ sin(o_d 2^k pi) (d-major, 24) | cos (24) |
// sin(r_d 2^k pi) (12) | cos (12) ], columns [72, ldo) zero.  fp32 math with full-range sinf / cosf (arguments reach
// 128 pi |o|).
template <typename T>
__global__ void __launch_bounds__(256)
ray_features_kernel(const float* __restrict__ cam, int64_t I, int h, int w, T* __restrict__ out, int64_t ldo) {
    auto cv = [](float v) { if constexpr (std::is_same<T, float>::value) return v; else return f32_to_bf16(v); };
    const int64_t tok = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (tok >= I * h * w) return;
    const int x = (int)(tok % w), y = (int)((tok / w) % h);
    const float* c = cam + (tok / ((int64_t)h * w)) * 21;
    const float pi = x + 0.5f, pj = y + 0.5f;
    float d[3], r[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) d[a] = c[3 * a] * pi + c[3 * a + 1] * pj + c[3 * a + 2];
#pragma unroll
    for (int a = 0; a < 3; ++a) r[a] = c[9 + 3 * a] * d[0] + c[9 + 3 * a + 1] * d[1] + c[9 + 3 * a + 2] * d[2];
    const float inv = 1.f / sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    T* o = out + tok * ldo;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float pos = c[18 + a], ray = r[a] * inv;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = pos * (3.14159265358979323846f * (float)(1 << k));
            o[a * 8 + k] = cv(sinf(v));
            o[24 + a * 8 + k] = cv(cosf(v));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v = ray * (3.14159265358979323846f * (float)(1 << k));
            o[48 + a * 4 + k] = cv(sinf(v));
            o[60 + a * 4 + k] = cv(cosf(v));
        }
    }
    for (int64_t k = 72; k < ldo; ++k) o[k] = T(0);
}

__global__ void __launch_bounds__(256)
cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 l = *(const float4*)(src + i);
    const float o[4] = {l.x, l.y, l.z, l.w};
    *(uint2*)(dst + i) = pack4(o);
}

// PixelUnshuffle(r) of an NCHW image batch written token-major:
//   out[(i*h + y)*w + x][c*r*r + dy*r + dx] = in[i][c][y*r + dy][x*r + dx],  h = H/r, w = W/r
template <typename TIN, typename TOUT = bf16_t>
__global__ void __launch_bounds__(256)
unshuffle_tokens_kernel(const TIN* __restrict__ x, int64_t I, int C, int H, int W, int r,
                        TOUT* __restrict__ out, int64_t ldo) {
    const int h = H / r, w = W / r, cols = C * r * r;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= I * h * w * ldo) return;
    const int64_t tok = idx / ldo;
    const int col = (int)(idx - tok * ldo);
    float v = 0.f;
    if (col < cols) {
        const int dx = col % r, dy = (col / r) % r, c = col / (r * r);
        const int xx = (int)(tok % w), yy = (int)((tok / w) % h);
        const int64_t img = tok / ((int64_t)w * h);
        v = ld_as_f32(x, ((img * C + c) * H + yy * r + dy) * W + xx * r + dx);
    }
    if constexpr (sizeof(TOUT) == 4) out[idx] = v;
    else out[idx] = f32_to_bf16(v);
}

// AvgPool2d(2, stride 2) on token-major [I, h, w, C] -> [I, h/2, w/2, C]; 8 channels per thread
template <typename T>
__global__ void __launch_bounds__(256)
avgpool2_tokens_kernel(const T* __restrict__ x, int64_t I, int h, int w, int C8,
                       T* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int ho = h / 2, wo = w / 2;
    if (idx >= I * ho * wo * C8) return;
    const int c8 = (int)(idx % C8);
    const int64_t tok = idx / C8;
    const int xo = (int)(tok % wo), yo = (int)((tok / wo) % ho);
    const int64_t img = tok / ((int64_t)wo * ho);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t[8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int64_t src = ((img * h + 2 * yo + a) * w + 2 * xo + b) * (int64_t)C8 + c8;
            load8<T>(x + src * 8, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += t[j];
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= 0.25f;
    store8<T>(out + idx * 8, acc);
}

__global__ void __launch_bounds__(256)
add_inplace_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ x, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float a[8], b[8];
    unpack8(*(const uint4*)(y + i * 8), a);
    unpack8(*(const uint4*)(x + i * 8), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += b[j];
    *(uint4*)(y + i * 8) = pack8(a);
}

// y (bf16) += x (fp32): one rounding of the fp32 sum (the cached fp32 layout residuals of the ImageAdapter)
__global__ void __launch_bounds__(256)
add_f32_inplace_kernel(bf16_t* __restrict__ y, const float* __restrict__ x, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float a[8];
    unpack8(*(const uint4*)(y + i * 8), a);
    const float4 b0 = *(const float4*)(x + i * 8), b1 = *(const float4*)(x + i * 8 + 4);
    a[0] += b0.x; a[1] += b0.y; a[2] += b0.z; a[3] += b0.w;
    a[4] += b1.x; a[5] += b1.y; a[6] += b1.z; a[7] += b1.w;
    *(uint4*)(y + i * 8) = pack8(a);
}

// y (fp32) += x (fp32): cached layout residuals onto the fp32 hidden stream
__global__ void __launch_bounds__(256)
add_f32_f32_inplace_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 a = *(const float4*)(y + i * 4);
    const float4 b = *(const float4*)(x + i * 4);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    *(float4*)(y + i * 4) = a;
}

inline int finish() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DWM_OK : (int)e;
}
inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int dwm_abi_version(void) { return DWM_ABI_VERSION; }
#ifndef DWM_SOURCE_HASH
#define DWM_SOURCE_HASH "unknown"
#endif
extern "C" const char* dwm_source_hash(void) { return DWM_SOURCE_HASH; }

extern "C" int dwm_silu(const void* x, void* y, int64_t n, void* stream) {
    if (x == nullptr || y == nullptr || n <= 0) return DWM_EINVAL;
    if (n % 8 != 0 || !dwm_aligned16(x) || !dwm_aligned16(y)) return DWM_EALIGN;
    hipLaunchKernelGGL(silu_kernel, dim3(blocks_for(n / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (bf16_t*)y, n / 8);
    return finish();
}

extern "C" int dwm_timestep_sinusoid(const float* t, int64_t n, int32_t C, void* out, void* stream) {
    if (t == nullptr || out == nullptr || n <= 0 || C <= 0) return DWM_EINVAL;
    if (C % 2 != 0) return DWM_EUNSUPPORTED;
    hipLaunchKernelGGL(sinusoid_kernel, dim3(blocks_for(n * (C / 2))), dim3(256), 0, (hipStream_t)stream,
                       t, n, C, (bf16_t*)out);
    return finish();
}

extern "C" int dwm_patchify(const void* x, int32_t x_is_f32, int64_t I, int32_t C, int32_t H, int32_t W,
                            int32_t p, void* out, int64_t ldo, void* stream) {
    if (x == nullptr || out == nullptr || I <= 0 || C <= 0 || H <= 0 || W <= 0 || p <= 0) return DWM_EINVAL;
    if (H % p != 0 || W % p != 0 || ldo < (int64_t)C * p * p) return DWM_EINVAL;
    const int64_t total = I * (H / p) * (W / p) * ldo;
    if (x_is_f32)
        hipLaunchKernelGGL(patchify_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, I, C, H, W, p, (bf16_t*)out, ldo);
    else
        hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, I, C, H, W, p, (bf16_t*)out, ldo);
    return finish();
}

extern "C" int dwm_unpatchify(const void* x, int64_t ldx, int64_t I, int32_t C, int32_t h, int32_t w,
                              int32_t p, void* out, void* stream) {
    if (x == nullptr || out == nullptr || I <= 0 || C <= 0 || h <= 0 || w <= 0 || p <= 0) return DWM_EINVAL;
    if (ldx < (int64_t)C * p * p) return DWM_EINVAL;
    const int64_t total = I * C * h * p * w * p;
    hipLaunchKernelGGL(unpatchify_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, ldx, I, C, h, w, p, (bf16_t*)out);
    return finish();
}

extern "C" int dwm_cfg_euler_step(const void* pred, float* latents, void* model_in, int64_t n,
                                  float guidance, float dsigma, void* stream) {
    if (pred == nullptr || latents == nullptr || n <= 0) return DWM_EINVAL;
    if (n % 4 != 0 || (((uintptr_t)pred) & 7u) || !dwm_aligned16(latents) || (model_in && (((uintptr_t)model_in) & 7u)))
        return DWM_EALIGN;
    hipLaunchKernelGGL(cfg_euler_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)pred, latents, (bf16_t*)model_in, n, guidance, dsigma, (const float*)nullptr, (int64_t)1);
    return finish();
}

extern "C" int dwm_cfg_euler_step_grouped(const void* pred, float* latents, void* model_in, int64_t n, float guidance,
                                          const float* dsigma, int64_t group_elems, void* stream) {
    if (pred == nullptr || latents == nullptr || dsigma == nullptr || n <= 0 || group_elems <= 0) return DWM_EINVAL;
    if (n % 4 != 0 || group_elems % 4 != 0 || n % group_elems != 0 || (((uintptr_t)pred) & 7u) || !dwm_aligned16(latents) ||
        (model_in && (((uintptr_t)model_in) & 7u)))
        return DWM_EALIGN;
    hipLaunchKernelGGL(cfg_euler_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)pred, latents, (bf16_t*)model_in, n, guidance, 0.f, dsigma, group_elems);
    return finish();
}

extern "C" int dwm_cfg_multistep(const void* pred, float* latents, float* x0_prev, void* model_in, int64_t n, float guidance,
                                 float kx, float ko, float A, float B, float C, void* stream) {
    if (pred == nullptr || latents == nullptr || x0_prev == nullptr || n <= 0) return DWM_EINVAL;
    if (n % 4 != 0 || (((uintptr_t)pred) & 7u) || !dwm_aligned16(latents) || !dwm_aligned16(x0_prev) ||
        (model_in && (((uintptr_t)model_in) & 7u)))
        return DWM_EALIGN;
    hipLaunchKernelGGL(cfg_multistep_kernel<bf16_t>, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pred,
                       latents, x0_prev, (bf16_t*)model_in, n, guidance, kx, ko, A, B, C);
    return finish();
}

extern "C" int dwm_cfg_multistep_f32(const float* pred, float* latents, float* x0_prev, float* model_in, int64_t n, float guidance,
                                     float kx, float ko, float A, float B, float C, void* stream) {
    if (pred == nullptr || latents == nullptr || x0_prev == nullptr || n <= 0) return DWM_EINVAL;
    if (n % 4 != 0 || !dwm_aligned16(pred) || !dwm_aligned16(latents) || !dwm_aligned16(x0_prev) || (model_in && !dwm_aligned16(model_in)))
        return DWM_EALIGN;
    hipLaunchKernelGGL(cfg_multistep_kernel<float>, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, pred,
                       latents, x0_prev, model_in, n, guidance, kx, ko, A, B, C);
    return finish();
}

extern "C" int dwm_frame_affine(const float* x, const float* y, const float* coef, float* out, void* out_bf16, int64_t n,
                                int64_t group_elems, void* stream) {
    if (x == nullptr || y == nullptr || coef == nullptr || (out == nullptr && out_bf16 == nullptr) || n <= 0 || group_elems <= 0) return DWM_EINVAL;
    if (n % 4 != 0 || group_elems % 4 != 0 || n % group_elems != 0 || !dwm_aligned16(x) || !dwm_aligned16(y) || (out && !dwm_aligned16(out)) ||
        (out_bf16 && (((uintptr_t)out_bf16) & 7u)))
        return DWM_EALIGN;
    hipLaunchKernelGGL(frame_affine_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, x, y, coef, out,
                       (bf16_t*)out_bf16, n, group_elems);
    return finish();
}

extern "C" int dwm_cfg_ddim_step(const void* pred, int32_t pred_is_f32, int32_t cfg, float* latents, void* model_in, float* x0_out,
                                 const float* noise, const float* coef, int64_t n, int64_t group_elems, float guidance,
                                 int32_t prediction_type, float clip_range, int32_t use_clipped_model_output, void* stream) {
    if (pred == nullptr || latents == nullptr || coef == nullptr || n <= 0 || group_elems <= 0) return DWM_EINVAL;
    if (prediction_type < 0 || prediction_type > 2) return DWM_EINVAL;
    if (n % 4 != 0 || group_elems % 4 != 0 || n % group_elems != 0 || !dwm_aligned16(latents) || (x0_out && !dwm_aligned16(x0_out)) ||
        (noise && !dwm_aligned16(noise)) || (model_in && (((uintptr_t)model_in) & 7u)) ||
        (((uintptr_t)pred) & (pred_is_f32 ? 15u : 7u)))
        return DWM_EALIGN;
    const int64_t off = cfg ? n : 0;
    if (pred_is_f32)
        hipLaunchKernelGGL(cfg_ddim_kernel<float>, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)pred, off,
                           latents, (bf16_t*)model_in, x0_out, noise, coef, n, group_elems, guidance, prediction_type, clip_range,
                           use_clipped_model_output);
    else
        hipLaunchKernelGGL(cfg_ddim_kernel<bf16_t>, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pred, off,
                           latents, (bf16_t*)model_in, x0_out, noise, coef, n, group_elems, guidance, prediction_type, clip_range,
                           use_clipped_model_output);
    return finish();
}

extern "C" int dwm_ray_features(const float* cam, int64_t I, int32_t h, int32_t w, void* out, int64_t ldo, void* stream) {
    if (cam == nullptr || out == nullptr || I <= 0 || h <= 0 || w <= 0 || ldo < 72) return DWM_EINVAL;
    hipLaunchKernelGGL(ray_features_kernel<bf16_t>, dim3(blocks_for(I * h * w)), dim3(256), 0, (hipStream_t)stream, cam, I, h, w,
                       (bf16_t*)out, ldo);
    return finish();
}

extern "C" int dwm_ray_features_f32(const float* cam, int64_t I, int32_t h, int32_t w, float* out, int64_t ldo, void* stream) {
    if (cam == nullptr || out == nullptr || I <= 0 || h <= 0 || w <= 0 || ldo < 72) return DWM_EINVAL;
    hipLaunchKernelGGL(ray_features_kernel<float>, dim3(blocks_for(I * h * w)), dim3(256), 0, (hipStream_t)stream, cam, I, h, w, out, ldo);
    return finish();
}

extern "C" int dwm_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    if (src == nullptr || dst == nullptr || n <= 0) return DWM_EINVAL;
    if (n % 4 != 0 || !dwm_aligned16(src) || (((uintptr_t)dst) & 7u)) return DWM_EALIGN;
    hipLaunchKernelGGL(cast_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, src,
                       (bf16_t*)dst, n);
    return finish();
}

extern "C" int dwm_unshuffle_tokens(const void* x, int32_t x_is_f32, int64_t I, int32_t C, int32_t H, int32_t W,
                                    int32_t r, void* out, int64_t ldo, void* stream) {
    if (x == nullptr || out == nullptr || I <= 0 || C <= 0 || H <= 0 || W <= 0 || r <= 0) return DWM_EINVAL;
    if (H % r != 0 || W % r != 0 || ldo < (int64_t)C * r * r) return DWM_EINVAL;
    const int64_t total = I * (H / r) * (W / r) * ldo;
    if (x_is_f32)
        hipLaunchKernelGGL(unshuffle_tokens_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, I, C, H, W, r, (bf16_t*)out, ldo);
    else
        hipLaunchKernelGGL(unshuffle_tokens_kernel<bf16_t>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, I, C, H, W, r, (bf16_t*)out, ldo);
    return finish();
}

static int avgpool2_impl(const void* x, int64_t I, int32_t h, int32_t w, int32_t C, void* out, void* stream, bool f32) {
    if (x == nullptr || out == nullptr || I <= 0 || h <= 0 || w <= 0 || C <= 0) return DWM_EINVAL;
    if (h % 2 != 0 || w % 2 != 0 || C % 8 != 0) return DWM_EUNSUPPORTED;
    if (!dwm_aligned16(x) || !dwm_aligned16(out)) return DWM_EALIGN;
    const int64_t total = I * (h / 2) * (w / 2) * (C / 8);
    if (f32) hipLaunchKernelGGL(avgpool2_tokens_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                                (const float*)x, I, h, w, C / 8, (float*)out);
    else hipLaunchKernelGGL(avgpool2_tokens_kernel<bf16_t>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                            (const bf16_t*)x, I, h, w, C / 8, (bf16_t*)out);
    return finish();
}
extern "C" int dwm_avgpool2_tokens(const void* x, int64_t I, int32_t h, int32_t w, int32_t C, void* out, void* stream) {
    return avgpool2_impl(x, I, h, w, C, out, stream, false);
}
extern "C" int dwm_avgpool2_tokens_f32(const float* x, int64_t I, int32_t h, int32_t w, int32_t C, float* out, void* stream) {
    return avgpool2_impl(x, I, h, w, C, out, stream, true);
}

// the fp32 accuracy path's pixel-unshuffle: fp32 NCHW images -> fp32 token-major rows (same index map)
extern "C" int dwm_unshuffle_tokens_f32(const float* x, int64_t I, int32_t C, int32_t H, int32_t W, int32_t r, float* out, int64_t ldo,
                                        void* stream) {
    if (x == nullptr || out == nullptr || I <= 0 || C <= 0 || H <= 0 || W <= 0 || r <= 0) return DWM_EINVAL;
    if (H % r != 0 || W % r != 0 || ldo < (int64_t)C * r * r) return DWM_EINVAL;
    const int64_t total = I * (H / r) * (W / r) * ldo;
    hipLaunchKernelGGL((unshuffle_tokens_kernel<float, float>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       x, I, C, H, W, r, out, ldo);
    return finish();
}

extern "C" int dwm_add_f32_inplace(void* y, const float* x, int64_t n, void* stream) {
    if (x == nullptr || y == nullptr || n <= 0) return DWM_EINVAL;
    if (n % 8 != 0 || !dwm_aligned16(x) || !dwm_aligned16(y)) return DWM_EALIGN;
    hipLaunchKernelGGL(add_f32_inplace_kernel, dim3(blocks_for(n / 8)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)y, x, n / 8);
    return finish();
}

extern "C" int dwm_add_f32_f32_inplace(float* y, const float* x, int64_t n, void* stream) {
    if (x == nullptr || y == nullptr || n <= 0) return DWM_EINVAL;
    if (n % 4 != 0 || !dwm_aligned16(x) || !dwm_aligned16(y)) return DWM_EALIGN;
    hipLaunchKernelGGL(add_f32_f32_inplace_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, y, x, n / 4);
    return finish();
}

extern "C" int dwm_add_inplace(void* y, const void* x, int64_t n, void* stream) {
    if (x == nullptr || y == nullptr || n <= 0) return DWM_EINVAL;
    if (n % 8 != 0 || !dwm_aligned16(x) || !dwm_aligned16(y)) return DWM_EALIGN;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks_for(n / 8)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)y, (const bf16_t*)x, n / 8);
    return finish();
}


// ---- block permutation (frame-shard pack / unpack, see dwm_hip.h): one workgroup row per destination block
namespace {
struct BlockPermP {
    const char* src; char* dst;
    int64_t block_bytes, sstride[4];
    FastDiv d1, d2, d3;                           // divisors n[1] * n[2] * n[3], n[2] * n[3], n[3]
    uint32_t chunks;                              // workgroups per block
};
__global__ void __launch_bounds__(256)
block_permute_kernel(const BlockPermP p) {
    const uint32_t blk = blockIdx.x / p.chunks, ch = blockIdx.x - blk * p.chunks;
    const uint32_t i0 = fdiv(blk, p.d1), r0 = blk - i0 * p.d1.d;
    const uint32_t i1 = fdiv(r0, p.d2), r1 = r0 - i1 * p.d2.d;
    const uint32_t i2 = fdiv(r1, p.d3), i3 = r1 - i2 * p.d3.d;
    const char* __restrict__ s = p.src + ((int64_t)i0 * p.sstride[0] + (int64_t)i1 * p.sstride[1] + (int64_t)i2 * p.sstride[2] + (int64_t)i3 * p.sstride[3]) * p.block_bytes;
    char* __restrict__ d = p.dst + (int64_t)blk * p.block_bytes;
    const int64_t per = ((p.block_bytes / 16 + p.chunks - 1) / p.chunks + 255) / 256 * 256 * 16;      // bytes of this block per workgroup
    const int64_t lo = (int64_t)ch * per, hi = lo + per < p.block_bytes ? lo + per : p.block_bytes;
    for (int64_t o = lo + (int64_t)threadIdx.x * 16; o < hi; o += 4 * 256 * 16) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (o + u * 4096 < hi) v[u] = *(const uint4*)(s + o + u * 4096);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (o + u * 4096 < hi) *(uint4*)(d + o + u * 4096) = v[u];
    }
}
}  // namespace

extern "C" int dwm_block_permute(const dwm_block_permute_args* a, void* stream) {
    if (a == nullptr || a->src == nullptr || a->dst == nullptr) return DWM_EINVAL;
    if (a->block_bytes <= 0 || a->block_bytes % 16 != 0 || !dwm_aligned16(a->src) || !dwm_aligned16(a->dst)) return DWM_EALIGN;
    int64_t nb = 1;
    for (int i = 0; i < 4; ++i) {
        if (a->n[i] <= 0 || a->sstride[i] < 0) return DWM_EINVAL;
        nb *= a->n[i];
    }
    if (nb >= (1ll << 24)) return DWM_EUNSUPPORTED;
    BlockPermP p;
    p.src = (const char*)a->src; p.dst = (char*)a->dst; p.block_bytes = a->block_bytes;
    for (int i = 0; i < 4; ++i) p.sstride[i] = a->sstride[i];
    p.d1 = make_fastdiv((uint32_t)(a->n[1] * a->n[2] * a->n[3]));
    p.d2 = make_fastdiv((uint32_t)(a->n[2] * a->n[3]));
    p.d3 = make_fastdiv((uint32_t)a->n[3]);
    // enough workgroups to fill the chip: >= 2048 in all, each with >= 16 KiB where the blocks are that large
    int64_t chunks = (2048 + nb - 1) / nb;
    const int64_t maxc = (a->block_bytes + 16383) / 16384;
    if (chunks > maxc) chunks = maxc;
    if (chunks < 1) chunks = 1;
    if (nb * chunks >= (1ll << 31)) return DWM_EUNSUPPORTED;
    p.chunks = (uint32_t)chunks;
    hipLaunchKernelGGL(block_permute_kernel, dim3((unsigned)(nb * chunks)), dim3(256), 0, (hipStream_t)stream, p);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DWM_OK : (int)e;
}
