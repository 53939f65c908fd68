// Kernels for the AutoencoderKL decode / encode blocks on token-major activations (gfx950):
// GroupNorm statistics + fused normalise/affine/SiLU (optionally writing into the zero-padded
// token grid that feeds the implicit-GEMM 3x3 convolutions), nearest 2x upsample into a padded
// grid, and a row softmax for the single-head mid-block attention.  All HBM-bound, 16-B accesses.
#include "common.h"
#include "dwm_hip.h"

namespace {

// ---- optional (image, pixel) -> token-row map: lets one GroupNorm call normalise over a strided set of
// rows, e.g. the (T, H, W) extent of every (batch, view) pair of a [(b t v), (h w), C] tensor
// (TemporalResnetBlock's GroupNorm over [B*V, C, T, H, W]).
struct ImgMap { int enabled; FastDiv iv, pn; int64_t s_ihi, s_ilo, s_phi; };
DWM_DEVINL int64_t img_row(const ImgMap& m, int64_t i, int64_t p, int64_t P) {
    if (!m.enabled) return i * P + p;
    const uint32_t ihi = fdiv((uint32_t)i, m.iv), ilo = (uint32_t)i - ihi * m.iv.d;
    const uint32_t phi = fdiv((uint32_t)p, m.pn), plo = (uint32_t)p - phi * m.pn.d;
    return (int64_t)ihi * m.s_ihi + (int64_t)ilo * m.s_ilo + (int64_t)phi * m.s_phi + plo;
}

// ---- GroupNorm statistics: x token-major [rows, C], G groups of CG = C/G channels (CG == 4 or CG >= 8:
// the 8 channels of a 16-B chunk then span at most two groups).  grid (chunks, I); each block reduces
// ppb pixels for all groups of image i into part[i][chunk][2G]; gn_finalize_kernel adds the chunks.  Every sum runs
// in a fixed order (per-thread partials -> LDS -> one thread per group -> one thread per statistic): the result does
// not depend on scheduling, so the whole UNet / VAE is bit-reproducible from run to run.
template <typename T>     // T: bf16_t, or float (the fp32 accuracy path of the UNet / VAE: dwm_groupnorm_silu_f32)
__global__ void __launch_bounds__(256)
gn_stats_kernel(const T* __restrict__ x, int64_t P, int C, int G, int64_t ppb, float* __restrict__ part, ImgMap im) {
    extern __shared__ float red[];            // [pstep][C8][4]: (sum, sumsq) of the two groups a chunk can touch
    const int i = blockIdx.y;
    const int C8 = C >> 3, CG = C / G;
    const int64_t p0 = (int64_t)blockIdx.x * ppb;
    const int64_t p1 = p0 + ppb < P ? p0 + ppb : P;
    const int TW = C8 < 256 ? C8 : 256;       // threads across channel chunks
    const int prow = threadIdx.x / TW, pstep = 256 / TW;
    if (prow < pstep) {
        for (int c8 = threadIdx.x % TW; c8 < C8; c8 += TW) {
            const int g0 = (c8 * 8) / CG;
            const int bnd = (g0 + 1) * CG - c8 * 8;     // first element of the chunk that belongs to group g0 + 1
            float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
            for (int64_t p = p0 + prow; p < p1; p += pstep) {
                float v[8];
                load8<T>(x + img_row(im, i, p, P) * C + c8 * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j < bnd) { s0 += v[j]; q0 += v[j] * v[j]; }
                    else { s1 += v[j]; q1 += v[j] * v[j]; }
                }
            }
            float* r = red + ((size_t)prow * C8 + c8) * 4;
            r[0] = s0; r[1] = q0; r[2] = s1; r[3] = q1;
        }
    }
    __syncthreads();
    // group g collects chunks [g*CG/8, ((g+1)*CG - 1)/8]: first slot of a chunk if the chunk starts in g, else second
    for (int g = threadIdx.x; g < G; g += 256) {
        float s = 0.f, q = 0.f;
        const int c_lo = (g * CG) >> 3, c_hi = ((g + 1) * CG - 1) >> 3;
        for (int c8 = c_lo; c8 <= c_hi; ++c8) {
            const int slot = ((c8 * 8) / CG == g) ? 0 : 2;
            for (int pr = 0; pr < pstep; ++pr) {
                const float* r = red + ((size_t)pr * C8 + c8) * 4 + slot;
                s += r[0]; q += r[1];
            }
        }
        float* o = part + (((int64_t)i * gridDim.x + blockIdx.x) * G + g) * 2;
        o[0] = s; o[1] = q;
    }
}

__global__ void __launch_bounds__(64)
gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats, int nchunks, int G2) {
    const int i = blockIdx.x;
    for (int t = threadIdx.x; t < G2; t += 64) {
        float s = 0.f;
        for (int c = 0; c < nchunks; ++c) s += part[((int64_t)i * nchunks + c) * G2 + t];
        stats[(int64_t)i * G2 + t] = s;
    }
}

// ---- y = silu?( (x - mean) * rstd * gamma + beta ), written compact or into a padded grid.
struct PadMap { int enabled; FastDiv rw, rh; int64_t rpitch, ipitch, origin; };
DWM_DEVINL int64_t pad_row(const PadMap& m, int64_t r) {
    if (!m.enabled) return r;
    const uint32_t q = fdiv((uint32_t)r, m.rw), xx = (uint32_t)r - q * m.rw.d;
    const uint32_t i = fdiv(q, m.rh), yy = q - i * m.rh.d;
    return (int64_t)i * m.ipitch + (int64_t)yy * m.rpitch + xx + m.origin;
}

template <typename T>
__global__ void __launch_bounds__(256)
gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t I, int64_t P, int C, int G,
                const float* __restrict__ stats, const T* __restrict__ gamma, const T* __restrict__ beta,
                float eps, int silu, PadMap pm, ImgMap im) {
    const int C8 = C >> 3, CG = C / G;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= I * P * C8) return;
    const int c8 = (int)(idx % C8);
    const int64_t ip = idx / C8;
    const int64_t i = ip / P, pp = ip - i * P;
    const int64_t r = img_row(im, i, pp, P);
    const float n = (float)P * (float)CG;
    const int g0 = (c8 * 8) / CG;
    const int bnd = (g0 + 1) * CG - c8 * 8;
    float mean[2], rstd[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int g = g0 + hf < G ? g0 + hf : G - 1;
        mean[hf] = stats[(i * G + g) * 2] / n;
        const float var = fmaxf(stats[(i * G + g) * 2 + 1] / n - mean[hf] * mean[hf], 0.f);
        rstd[hf] = rsqrtf(var + eps);
    }
    float v[8], ga[8], be[8];
    load8<T>(x + r * C + c8 * 8, v);
    load8<T>(gamma + c8 * 8, ga);
    load8<T>(beta + c8 * 8, be);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int hf = j < bnd ? 0 : 1;
        float t = (v[j] - mean[hf]) * rstd[hf] * ga[j] + be[j];
        if (sizeof(T) == 4) v[j] = silu ? t / (1.f + expf(-t)) : t;       // fp32 path: libm-accurate SiLU
        else v[j] = silu ? silu_f(t) : t;
    }
    store8<T>(y + pad_row(pm, r) * C + c8 * 8, v);
}

// ---- CogVideoXSpatialNorm3D apply: GroupNorm statistics as above, then * conv_y(zq) + conv_b(zq) gathered from the
//      latent-resolution modulation rows (nearest resize folded into the index), optional SiLU
struct ZMap {
    const void* mod; int64_t ld_mod;
    FastDiv n, w, b;          // pixels per frame, width, videos
    int hz, wz, shift;
    int zt[32];
};
template <typename T>     // (float: dwm_groupnorm_spatial_f32, the fp32 accuracy path of the temporal VAE)
__global__ void __launch_bounds__(256)
gn_spatial_apply_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t I, int64_t P, int C, int G,
                        const float* __restrict__ stats, const T* __restrict__ gamma, const T* __restrict__ beta,
                        float eps, int silu, PadMap pm, ImgMap im, ZMap zm) {
    const int C8 = C >> 3, CG = C / G;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= I * P * C8) return;
    const int c8 = (int)(idx % C8);
    const int64_t ip = idx / C8;
    const int64_t i = ip / P, pp = ip - i * P;
    const int64_t r = img_row(im, i, pp, P);
    const float n = (float)P * (float)CG;
    const int g0 = (c8 * 8) / CG;
    const int bnd = (g0 + 1) * CG - c8 * 8;
    float mean[2], rstd[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int g = g0 + hf < G ? g0 + hf : G - 1;
        mean[hf] = stats[(i * G + g) * 2] / n;
        const float var = fmaxf(stats[(i * G + g) * 2 + 1] / n - mean[hf] * mean[hf], 0.f);
        rstd[hf] = rsqrtf(var + eps);
    }
    // token row r = ((t*B + b)*h + yy)*w + xx  ->  modulation row at latent resolution
    const uint32_t f = fdiv((uint32_t)r, zm.n), pix = (uint32_t)r - f * zm.n.d;
    const uint32_t t = fdiv(f, zm.b), b = f - t * zm.b.d;
    const uint32_t yy = fdiv(pix, zm.w), xx = pix - yy * zm.w.d;
    const int64_t zr = (((int64_t)zm.zt[t] * zm.b.d + b) * zm.hz + (yy >> zm.shift)) * zm.wz + (xx >> zm.shift);
    float v[8], ga[8], be[8], my[8], mb[8];
    const T* __restrict__ mod = (const T*)zm.mod;
    load8<T>(x + r * C + c8 * 8, v);
    load8<T>(gamma + c8 * 8, ga);
    load8<T>(beta + c8 * 8, be);
    load8<T>(mod + zr * zm.ld_mod + c8 * 8, my);
    load8<T>(mod + zr * zm.ld_mod + C + c8 * 8, mb);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int hf = j < bnd ? 0 : 1;
        const float tv = ((v[j] - mean[hf]) * rstd[hf] * ga[j] + be[j]) * my[j] + mb[j];
        if (sizeof(T) == 4) v[j] = silu ? tv / (1.f + expf(-tv)) : tv;       // fp32 path: libm-accurate SiLU
        else v[j] = silu ? silu_f(tv) : tv;
    }
    store8<T>(y + pad_row(pm, r) * C + c8 * 8, v);
}

// ---- out frame j = w0 * x[f0] + w1 * x[f1]
struct FrameMix { int n_out; int f0[64], f1[64]; float w0[64], w1[64]; };
template <typename T>
__global__ void __launch_bounds__(256)
frame_mix_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t fe8, FrameMix fm) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= fe8) return;
    const int j = blockIdx.y;
    float a[8], b[8];
    load8<T>(x + ((int64_t)fm.f0[j] * fe8 + idx) * 8, a);
    if (fm.w1[j] != 0.f) {
        load8<T>(x + ((int64_t)fm.f1[j] * fe8 + idx) * 8, b);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = fm.w0[j] * a[k] + fm.w1[j] * b[k];
    } else if (fm.w0[j] != 1.f) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] *= fm.w0[j];
    }
    store8<T>(y + ((int64_t)j * fe8 + idx) * 8, a);
}

// ---- copy compact token rows into a (zero-bordered) padded grid
template <typename T>
__global__ void __launch_bounds__(256)
pad_tokens_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows, int C8, PadMap pm) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C8) return;
    const int c8 = (int)(idx % C8);
    const int64_t r = idx / C8;
    copy8<T>(y + (pad_row(pm, r) * C8 + c8) * 8, x + idx * 8);
}

// ---- nearest 2x upsample of token-major [I, h, w, C] into the padded grid of the [I, 2h, 2w] image
template <typename T>
__global__ void __launch_bounds__(256)
upsample2_pad_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t I, int h, int w, int C8) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int H = 2 * h, W = 2 * w;
    if (idx >= I * H * W * C8) return;
    const int c8 = (int)(idx % C8);
    const int64_t r = idx / C8;
    const int X = (int)(r % W), Y = (int)((r / W) % H);
    const int64_t i = r / ((int64_t)W * H);
    const int64_t orow = i * (int64_t)(H + 2) * (W + 2) + (int64_t)(Y + 1) * (W + 2) + X + 1;
    copy8<T>(y + (orow * C8 + c8) * 8, x + (((i * h + (Y >> 1)) * w + (X >> 1)) * (int64_t)C8 + c8) * 8);
}

// ---- row softmax (fp32 math) of bf16 x[rows, L] * scale, one wave per row, L <= 4096, L % 8 == 0
template <int NI, typename T>
__global__ void __launch_bounds__(256)
softmax_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows, int L, int64_t ld, float scale_log2) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NI][8];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < L) {
            load8<T>(x + row * ld + c, v[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] *= scale_log2; mx = fmaxf(mx, v[i][j]); }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = -INFINITY;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[i][j] = sizeof(T) == 4 ? exp2f(v[i][j] - mx) : __builtin_amdgcn_exp2f(v[i][j] - mx); s += v[i][j]; }
    s = wave_sum(s);
    const float inv = sizeof(T) == 4 ? 1.f / s : __builtin_amdgcn_rcpf(s);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < L) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] *= inv;
            store8<T>(y + row * ld + c, v[i]);
        }
    }
}

// ---- GroupNorm (+ SiLU) backward, two passes over (x, dz) with the forward's statistics recomputed first:
//   y = xh * gamma + beta, xh = (x - mean) * rstd;  z = silu(y);  dy = dz * silu'(y)
//   dgamma_c += sum dy * xh,  dbeta_c += sum dy                                   (fp32 atomics, one per block and channel)
//   S1_g = sum gamma dy,  S2_g = sum gamma dy xh  per (image, group)               (fixed-order reduction like the forward)
//   dx = rstd * (gamma dy - (S1_g + xh S2_g) / n)
// dz is read through the same padded-grid map the forward wrote z with (the gradient of a 3x3 convolution's input grid).
DWM_DEVINL float silu_grad_f(float y, float dz) {
    const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y));
    return dz * sg * (1.f + y * (1.f - sg));
}

__global__ void __launch_bounds__(256)
gn_bwd_stats_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dz, int64_t P, int C, int G, int64_t ppb,
                    const float* __restrict__ fstats, const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta, float eps,
                    int silu, float* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta, PadMap pm, ImgMap im) {
    extern __shared__ float red[];            // [pstep][C8][4] group partials, then [pstep][C][2] channel partials
    const int i = blockIdx.y;
    const int C8 = C >> 3, CG = C / G;
    const int TW = C8 < 256 ? C8 : 256;
    const int prow = threadIdx.x / TW, pstep = 256 / TW;
    float* red2 = red + (size_t)pstep * C8 * 4;
    const int64_t p0 = (int64_t)blockIdx.x * ppb;
    const int64_t p1 = p0 + ppb < P ? p0 + ppb : P;
    const float n = (float)P * (float)CG;
    if (prow < pstep) {
        for (int c8 = threadIdx.x % TW; c8 < C8; c8 += TW) {
            const int g0 = (c8 * 8) / CG;
            const int bnd = (g0 + 1) * CG - c8 * 8;
            float mean[2], rstd[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int g = g0 + hf < G ? g0 + hf : G - 1;
                mean[hf] = fstats[((int64_t)i * G + g) * 2] / n;
                const float var = fmaxf(fstats[((int64_t)i * G + g) * 2 + 1] / n - mean[hf] * mean[hf], 0.f);
                rstd[hf] = rsqrtf(var + eps);
            }
            float ga[8], be[8], dg[8], db[8];
            unpack8(*(const uint4*)(gamma + c8 * 8), ga);
            unpack8(*(const uint4*)(beta + c8 * 8), be);
#pragma unroll
            for (int j = 0; j < 8; ++j) dg[j] = db[j] = 0.f;
            float a0 = 0.f, b0 = 0.f, a1 = 0.f, b1 = 0.f;
            for (int64_t p = p0 + prow; p < p1; p += pstep) {
                const int64_t r = img_row(im, i, p, P);
                float v[8], d[8];
                unpack8(*(const uint4*)(x + r * C + c8 * 8), v);
                unpack8(*(const uint4*)(dz + pad_row(pm, r) * C + c8 * 8), d);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int hf = j < bnd ? 0 : 1;
                    const float xh = (v[j] - mean[hf]) * rstd[hf];
                    const float dy = silu ? silu_grad_f(xh * ga[j] + be[j], d[j]) : d[j];
                    dg[j] += dy * xh;
                    db[j] += dy;
                    const float gd = ga[j] * dy;
                    if (hf == 0) { a0 += gd; b0 += gd * xh; } else { a1 += gd; b1 += gd * xh; }
                }
            }
            float* r4 = red + ((size_t)prow * C8 + c8) * 4;
            r4[0] = a0; r4[1] = b0; r4[2] = a1; r4[3] = b1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red2[((size_t)prow * C + c8 * 8 + j) * 2] = dg[j];
                red2[((size_t)prow * C + c8 * 8 + j) * 2 + 1] = db[j];
            }
        }
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += 256) {
        float a = 0.f, b = 0.f;
        const int c_lo = (g * CG) >> 3, c_hi = ((g + 1) * CG - 1) >> 3;
        for (int c8 = c_lo; c8 <= c_hi; ++c8) {
            const int slot = ((c8 * 8) / CG == g) ? 0 : 2;
            for (int pr = 0; pr < pstep; ++pr) {
                const float* r4 = red + ((size_t)pr * C8 + c8) * 4 + slot;
                a += r4[0]; b += r4[1];
            }
        }
        float* o = part + (((int64_t)i * gridDim.x + blockIdx.x) * G + g) * 2;
        o[0] = a; o[1] = b;
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        float sg = 0.f, sb = 0.f;
        for (int pr = 0; pr < pstep; ++pr) { sg += red2[((size_t)pr * C + c) * 2]; sb += red2[((size_t)pr * C + c) * 2 + 1]; }
        atomicAdd(dgamma + c, sg);
        atomicAdd(dbeta + c, sb);
    }
}

__global__ void __launch_bounds__(256)
gn_bwd_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dz, bf16_t* __restrict__ dx, int64_t I, int64_t P, int C,
                    int G, const float* __restrict__ fstats, const float* __restrict__ bstats, const bf16_t* __restrict__ gamma,
                    const bf16_t* __restrict__ beta, float eps, int silu, int accumulate, PadMap pm, ImgMap im) {
    const int C8 = C >> 3, CG = C / G;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= I * P * C8) return;
    const int c8 = (int)(idx % C8);
    const int64_t ip = idx / C8;
    const int64_t i = ip / P, pp = ip - i * P;
    const int64_t r = img_row(im, i, pp, P);
    const float n = (float)P * (float)CG;
    const int g0 = (c8 * 8) / CG;
    const int bnd = (g0 + 1) * CG - c8 * 8;
    float mean[2], rstd[2], s1[2], s2[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int g = g0 + hf < G ? g0 + hf : G - 1;
        mean[hf] = fstats[(i * G + g) * 2] / n;
        const float var = fmaxf(fstats[(i * G + g) * 2 + 1] / n - mean[hf] * mean[hf], 0.f);
        rstd[hf] = rsqrtf(var + eps);
        s1[hf] = bstats[(i * G + g) * 2] / n;
        s2[hf] = bstats[(i * G + g) * 2 + 1] / n;
    }
    float v[8], d[8], ga[8], be[8], o[8];
    unpack8(*(const uint4*)(x + r * C + c8 * 8), v);
    unpack8(*(const uint4*)(dz + pad_row(pm, r) * C + c8 * 8), d);
    unpack8(*(const uint4*)(gamma + c8 * 8), ga);
    unpack8(*(const uint4*)(beta + c8 * 8), be);
    if (accumulate) unpack8(*(const uint4*)(dx + r * C + c8 * 8), o);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int hf = j < bnd ? 0 : 1;
        const float xh = (v[j] - mean[hf]) * rstd[hf];
        const float dy = silu ? silu_grad_f(xh * ga[j] + be[j], d[j]) : d[j];
        const float g = rstd[hf] * (ga[j] * dy - (s1[hf] + xh * s2[hf]));
        o[j] = accumulate ? o[j] + g : g;
    }
    *(uint4*)(dx + r * C + c8 * 8) = pack8(o);
}

inline int finish() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DWM_OK : (int)e;
}

}  // namespace

static int64_t gn_pixels_per_block(int64_t I, int64_t P) {
    int64_t ppb = (I * P + 2047) / 2048;
    return ppb < 64 ? 64 : ppb > 2048 ? 2048 : ppb;
}

extern "C" int64_t dwm_groupnorm_stats_floats(int64_t I, int64_t P, int32_t G) {
    if (I <= 0 || P <= 0 || G <= 0) return 0;
    const int64_t ppb = gn_pixels_per_block(I, P);
    return 2 * (int64_t)G * I * (1 + (P + ppb - 1) / ppb);
}

static int groupnorm_impl(const void* x, void* y, int64_t I, int64_t P, int32_t C, int32_t G, float eps,
                          const void* gamma, const void* beta, int32_t silu, float* stats,
                          const dwm_rowmap2d* out_map, const dwm_gn_imgmap* img_map, const dwm_gn_zmap* zmap, void* stream,
                          bool f32 = false) {
    if (x == nullptr || y == nullptr || gamma == nullptr || beta == nullptr || stats == nullptr) return DWM_EINVAL;
    if (I <= 0 || P <= 0 || C <= 0 || G <= 0 || C % G != 0) return DWM_EINVAL;
    const int CG = C / G;
    if (C % 8 != 0 || !(CG == 4 || CG >= 8) || I > 65535 || I * P >= (1ll << 31)) return DWM_EUNSUPPORTED;
    if (!dwm_aligned16(x) || !dwm_aligned16(y) || !dwm_aligned16(gamma) || !dwm_aligned16(beta)) return DWM_EALIGN;
    ImgMap im;
    im.enabled = img_map != nullptr && img_map->iv > 0;
    if (im.enabled) {
        if (img_map->pn <= 0 || img_map->iv >= (1ll << 30) || img_map->pn >= (1ll << 30)) return DWM_EINVAL;
        im.iv = make_fastdiv((uint32_t)img_map->iv); im.pn = make_fastdiv((uint32_t)img_map->pn);
        im.s_ihi = img_map->s_ihi; im.s_ilo = img_map->s_ilo; im.s_phi = img_map->s_phi;
    } else {
        im.iv = make_fastdiv(1); im.pn = make_fastdiv(1); im.s_ihi = im.s_ilo = im.s_phi = 0;
    }
    hipStream_t s = (hipStream_t)stream;
    // pixels per statistics block: enough blocks to fill the chip even for a dozen images (temporal GroupNorm: I = B*V)
    const int64_t ppb = gn_pixels_per_block(I, P);
    const int nchunks = (int)((P + ppb - 1) / ppb);
    const int C8 = C / 8, TW = C8 < 256 ? C8 : 256;
    const size_t lds = sizeof(float) * 4 * (size_t)(256 / TW) * C8;
    if (lds > 64 * 1024) return DWM_EUNSUPPORTED;
    float* part = stats + 2 * (int64_t)G * I;             // [I][nchunks][2G] behind the final statistics
    const dim3 grid((unsigned)nchunks, (unsigned)I);
    if (f32) hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), lds, s, (const float*)x, P, C, G, ppb, part, im);
    else hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, dim3(256), lds, s, (const bf16_t*)x, P, C, G, ppb, part, im);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)I), dim3(64), 0, s, (const float*)part, stats, nchunks, 2 * G);
    PadMap pm;
    pm.enabled = out_map != nullptr && out_map->rw > 0;
    if (pm.enabled) {
        if (out_map->rh <= 0 || (!im.enabled && out_map->rw * out_map->rh != P)) return DWM_EINVAL;
        pm.rw = make_fastdiv((uint32_t)out_map->rw); pm.rh = make_fastdiv((uint32_t)out_map->rh);
        pm.rpitch = out_map->rpitch; pm.ipitch = out_map->ipitch; pm.origin = out_map->origin;
    } else {
        pm.rw = make_fastdiv(1); pm.rh = make_fastdiv(1); pm.rpitch = pm.ipitch = pm.origin = 0;
    }
    const int64_t total = I * P * (C / 8);
    if (zmap != nullptr) {
        if (zmap->mod == nullptr || zmap->frames <= 0 || zmap->frames > 32 || zmap->videos <= 0 || zmap->h <= 0 || zmap->w <= 0 ||
            zmap->shift < 0 || zmap->shift > 8 || zmap->ld_mod < 2 * C || zmap->ld_mod % (f32 ? 4 : 8) != 0 || !dwm_aligned16(zmap->mod))
            return DWM_EINVAL;
        if ((int64_t)zmap->frames * zmap->videos * zmap->h * zmap->w != I * P) return DWM_EINVAL;
        if ((zmap->h >> zmap->shift) << zmap->shift != zmap->h || (zmap->w >> zmap->shift) << zmap->shift != zmap->w) return DWM_EINVAL;
        ZMap zm;
        zm.mod = zmap->mod; zm.ld_mod = zmap->ld_mod;
        zm.n = make_fastdiv((uint32_t)(zmap->h * zmap->w)); zm.w = make_fastdiv((uint32_t)zmap->w);
        zm.b = make_fastdiv((uint32_t)zmap->videos);
        zm.hz = zmap->h >> zmap->shift; zm.wz = zmap->w >> zmap->shift; zm.shift = zmap->shift;
        for (int t = 0; t < 32; ++t) zm.zt[t] = t < zmap->frames ? zmap->zt[t] : 0;
        if (f32)
            hipLaunchKernelGGL(gn_spatial_apply_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)x,
                               (float*)y, I, P, C, G, stats, (const float*)gamma, (const float*)beta, eps, silu, pm, im, zm);
        else
            hipLaunchKernelGGL(gn_spatial_apply_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const bf16_t*)x,
                               (bf16_t*)y, I, P, C, G, stats, (const bf16_t*)gamma, (const bf16_t*)beta, eps, silu, pm, im, zm);
        return finish();
    }
    if (f32)
        hipLaunchKernelGGL(gn_apply_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)x,
                           (float*)y, I, P, C, G, stats, (const float*)gamma, (const float*)beta, eps, silu, pm, im);
    else
        hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const bf16_t*)x,
                           (bf16_t*)y, I, P, C, G, stats, (const bf16_t*)gamma, (const bf16_t*)beta, eps, silu, pm, im);
    return finish();
}

extern "C" int dwm_groupnorm_bwd(const void* x, const void* dz, void* dx, int64_t I, int64_t P, int32_t C, int32_t G, float eps,
                                 const void* gamma, const void* beta, int32_t silu, int32_t accumulate, float* stats,
                                 float* dgamma, float* dbeta, const dwm_rowmap2d* dz_map, const dwm_gn_imgmap* img_map, void* stream) {
    if (x == nullptr || dz == nullptr || dx == nullptr || gamma == nullptr || beta == nullptr || stats == nullptr || dgamma == nullptr ||
        dbeta == nullptr)
        return DWM_EINVAL;
    if (I <= 0 || P <= 0 || C <= 0 || G <= 0 || C % G != 0) return DWM_EINVAL;
    const int CG = C / G;
    if (C % 8 != 0 || !(CG == 4 || CG >= 8) || I > 65535 || I * P >= (1ll << 31)) return DWM_EUNSUPPORTED;
    if (!dwm_aligned16(x) || !dwm_aligned16(dz) || !dwm_aligned16(dx) || !dwm_aligned16(gamma) || !dwm_aligned16(beta)) return DWM_EALIGN;
    ImgMap im;
    im.enabled = img_map != nullptr && img_map->iv > 0;
    if (im.enabled) {
        if (img_map->pn <= 0 || img_map->iv >= (1ll << 30) || img_map->pn >= (1ll << 30)) return DWM_EINVAL;
        im.iv = make_fastdiv((uint32_t)img_map->iv); im.pn = make_fastdiv((uint32_t)img_map->pn);
        im.s_ihi = img_map->s_ihi; im.s_ilo = img_map->s_ilo; im.s_phi = img_map->s_phi;
    } else {
        im.iv = make_fastdiv(1); im.pn = make_fastdiv(1); im.s_ihi = im.s_ilo = im.s_phi = 0;
    }
    PadMap pm;
    pm.enabled = dz_map != nullptr && dz_map->rw > 0;
    if (pm.enabled) {
        if (dz_map->rh <= 0 || (!im.enabled && dz_map->rw * dz_map->rh != P)) return DWM_EINVAL;
        pm.rw = make_fastdiv((uint32_t)dz_map->rw); pm.rh = make_fastdiv((uint32_t)dz_map->rh);
        pm.rpitch = dz_map->rpitch; pm.ipitch = dz_map->ipitch; pm.origin = dz_map->origin;
    } else {
        pm.rw = make_fastdiv(1); pm.rh = make_fastdiv(1); pm.rpitch = pm.ipitch = pm.origin = 0;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t ppb = gn_pixels_per_block(I, P);
    const int nchunks = (int)((P + ppb - 1) / ppb);
    const int C8 = C / 8, TW = C8 < 256 ? C8 : 256, pstep = 256 / TW;
    const size_t lds_f = sizeof(float) * 4 * (size_t)pstep * C8;
    const size_t lds_b = lds_f + sizeof(float) * 2 * (size_t)pstep * C;
    if (lds_b > 64 * 1024) return DWM_EUNSUPPORTED;
    // scratch: [forward statistics | forward partials | backward statistics | backward partials]
    const int64_t half = dwm_groupnorm_stats_floats(I, P, G);
    float* fstats = stats;
    float* fpart = stats + 2 * (int64_t)G * I;
    float* bstats = stats + half;
    float* bpart = bstats + 2 * (int64_t)G * I;
    const dim3 grid((unsigned)nchunks, (unsigned)I);
    hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, dim3(256), lds_f, s, (const bf16_t*)x, P, C, G, ppb, fpart, im);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)I), dim3(64), 0, s, (const float*)fpart, fstats, nchunks, 2 * G);
    hipLaunchKernelGGL(gn_bwd_stats_kernel, grid, dim3(256), lds_b, s, (const bf16_t*)x, (const bf16_t*)dz, P, C, G, ppb,
                       (const float*)fstats, (const bf16_t*)gamma, (const bf16_t*)beta, eps, silu, bpart, dgamma, dbeta, pm, im);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)I), dim3(64), 0, s, (const float*)bpart, bstats, nchunks, 2 * G);
    const int64_t total = I * P * (C / 8);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dz,
                       (bf16_t*)dx, I, P, C, G, (const float*)fstats, (const float*)bstats, (const bf16_t*)gamma, (const bf16_t*)beta,
                       eps, silu, accumulate, pm, im);
    return finish();
}

extern "C" int dwm_groupnorm_silu_mapped(const void* x, void* y, int64_t I, int64_t P, int32_t C, int32_t G, float eps,
                                         const void* gamma, const void* beta, int32_t silu, float* stats,
                                         const dwm_rowmap2d* out_map, const dwm_gn_imgmap* img_map, void* stream) {
    return groupnorm_impl(x, y, I, P, C, G, eps, gamma, beta, silu, stats, out_map, img_map, nullptr, stream);
}

extern "C" int dwm_groupnorm_silu_f32(const float* x, float* y, int64_t I, int64_t P, int32_t C, int32_t G, float eps,
                                      const float* gamma, const float* beta, int32_t silu, float* stats,
                                      const dwm_rowmap2d* out_map, const dwm_gn_imgmap* img_map, void* stream) {
    return groupnorm_impl(x, y, I, P, C, G, eps, gamma, beta, silu, stats, out_map, img_map, nullptr, stream, true);
}

extern "C" int dwm_groupnorm_spatial(const void* x, void* y, int64_t I, int64_t P, int32_t C, int32_t G, float eps,
                                     const void* gamma, const void* beta, int32_t silu, float* stats,
                                     const dwm_rowmap2d* out_map, const dwm_gn_imgmap* img_map, const dwm_gn_zmap* zmap,
                                     void* stream) {
    if (zmap == nullptr) return DWM_EINVAL;
    return groupnorm_impl(x, y, I, P, C, G, eps, gamma, beta, silu, stats, out_map, img_map, zmap, stream);
}

extern "C" int dwm_groupnorm_spatial_f32(const float* x, float* y, int64_t I, int64_t P, int32_t C, int32_t G, float eps,
                                         const float* gamma, const float* beta, int32_t silu, float* stats,
                                         const dwm_rowmap2d* out_map, const dwm_gn_imgmap* img_map, const dwm_gn_zmap* zmap,
                                         void* stream) {
    if (zmap == nullptr) return DWM_EINVAL;
    return groupnorm_impl(x, y, I, P, C, G, eps, gamma, beta, silu, stats, out_map, img_map, zmap, stream, true);
}

static int frame_mix_impl(const void* x, void* y, int64_t frame_elems, const dwm_frame_mix* mix, void* stream, bool f32) {
    if (x == nullptr || y == nullptr || mix == nullptr || frame_elems <= 0 || mix->n_out <= 0 || mix->n_out > 64) return DWM_EINVAL;
    if (frame_elems % 8 != 0) return DWM_EUNSUPPORTED;
    if (!dwm_aligned16(x) || !dwm_aligned16(y)) return DWM_EALIGN;
    FrameMix fm;
    fm.n_out = mix->n_out;
    for (int j = 0; j < 64; ++j) {
        const bool ok = j < mix->n_out;
        if (ok && (mix->f0[j] < 0 || mix->f1[j] < 0)) return DWM_EINVAL;
        fm.f0[j] = ok ? mix->f0[j] : 0; fm.f1[j] = ok ? mix->f1[j] : 0;
        fm.w0[j] = ok ? mix->w0[j] : 0.f; fm.w1[j] = ok ? mix->w1[j] : 0.f;
    }
    const int64_t fe8 = frame_elems / 8;
    const dim3 grid((unsigned)((fe8 + 255) / 256), (unsigned)mix->n_out);
    if (f32) hipLaunchKernelGGL(frame_mix_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, fe8, fm);
    else hipLaunchKernelGGL(frame_mix_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, fe8, fm);
    return finish();
}
extern "C" int dwm_frame_mix_bf16(const void* x, void* y, int64_t frame_elems, const dwm_frame_mix* mix, void* stream) {
    return frame_mix_impl(x, y, frame_elems, mix, stream, false);
}
extern "C" int dwm_frame_mix_f32(const float* x, float* y, int64_t frame_elems, const dwm_frame_mix* mix, void* stream) {
    return frame_mix_impl(x, y, frame_elems, mix, stream, true);
}

extern "C" int dwm_groupnorm_silu(const void* x, void* y, int64_t I, int64_t P, int32_t C, int32_t G, float eps,
                                  const void* gamma, const void* beta, int32_t silu, float* stats,
                                  const dwm_rowmap2d* out_map, void* stream) {
    return dwm_groupnorm_silu_mapped(x, y, I, P, C, G, eps, gamma, beta, silu, stats, out_map, nullptr, stream);
}

static int upsample2_impl(const void* x, void* y, int64_t I, int32_t h, int32_t w, int32_t C, void* stream, bool f32) {
    if (x == nullptr || y == nullptr || I <= 0 || h <= 0 || w <= 0 || C <= 0) return DWM_EINVAL;
    if (C % 8 != 0) return DWM_EUNSUPPORTED;
    if (!dwm_aligned16(x) || !dwm_aligned16(y)) return DWM_EALIGN;
    const int64_t total = I * 4 * h * w * (C / 8);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (f32) hipLaunchKernelGGL(upsample2_pad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, I, h, w, C / 8);
    else hipLaunchKernelGGL(upsample2_pad_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, I, h, w, C / 8);
    return finish();
}
extern "C" int dwm_upsample2_padded(const void* x, void* y, int64_t I, int32_t h, int32_t w, int32_t C, void* stream) {
    return upsample2_impl(x, y, I, h, w, C, stream, false);
}
extern "C" int dwm_upsample2_padded_f32(const float* x, float* y, int64_t I, int32_t h, int32_t w, int32_t C, void* stream) {
    return upsample2_impl(x, y, I, h, w, C, stream, true);
}

template <typename T>
static int softmax_rows_impl(const T* x, T* y, int64_t rows, int32_t L, int64_t ld, float scale, void* stream) {
    if (x == nullptr || y == nullptr || rows <= 0 || L <= 0) return DWM_EINVAL;
    if (L % 8 != 0 || L > 4096 || ld % 8 != 0 || ld < L) return DWM_EUNSUPPORTED;
    if (!dwm_aligned16(x) || !dwm_aligned16(y)) return DWM_EALIGN;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const float sl = scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    const int ni = (L + 511) / 512;
    if (ni <= 2) hipLaunchKernelGGL((softmax_rows_kernel<2, T>), grid, block, 0, s, x, y, rows, L, ld, sl);
    else if (ni <= 4) hipLaunchKernelGGL((softmax_rows_kernel<4, T>), grid, block, 0, s, x, y, rows, L, ld, sl);
    else hipLaunchKernelGGL((softmax_rows_kernel<8, T>), grid, block, 0, s, x, y, rows, L, ld, sl);
    return finish();
}
extern "C" int dwm_softmax_rows(const void* x, void* y, int64_t rows, int32_t L, int64_t ld, float scale, void* stream) {
    return softmax_rows_impl<bf16_t>((const bf16_t*)x, (bf16_t*)y, rows, L, ld, scale, stream);
}
extern "C" int dwm_softmax_rows_f32(const float* x, float* y, int64_t rows, int32_t L, int64_t ld, float scale, void* stream) {
    return softmax_rows_impl<float>(x, y, rows, L, ld, scale, stream);
}

static int pad_tokens_impl(const void* x, void* y, int64_t rows, int32_t C, const dwm_rowmap2d* map, void* stream, bool f32);
extern "C" int dwm_pad_tokens(const void* x, void* y, int64_t rows, int32_t C, const dwm_rowmap2d* map, void* stream) {
    return pad_tokens_impl(x, y, rows, C, map, stream, false);
}
extern "C" int dwm_pad_tokens_f32(const float* x, float* y, int64_t rows, int32_t C, const dwm_rowmap2d* map, void* stream) {
    return pad_tokens_impl(x, y, rows, C, map, stream, true);
}
static int pad_tokens_impl(const void* x, void* y, int64_t rows, int32_t C, const dwm_rowmap2d* map, void* stream, bool f32) {
    if (x == nullptr || y == nullptr || map == nullptr || rows <= 0 || C <= 0 || map->rw <= 0 || map->rh <= 0) return DWM_EINVAL;
    if (C % 8 != 0 || rows % (map->rw * map->rh) != 0) return DWM_EUNSUPPORTED;
    if (!dwm_aligned16(x) || !dwm_aligned16(y)) return DWM_EALIGN;
    PadMap pm;
    pm.enabled = 1;
    pm.rw = make_fastdiv((uint32_t)map->rw); pm.rh = make_fastdiv((uint32_t)map->rh);
    pm.rpitch = map->rpitch; pm.ipitch = map->ipitch; pm.origin = map->origin;
    const int64_t total = rows * (C / 8);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (f32) hipLaunchKernelGGL(pad_tokens_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, rows, C / 8, pm);
    else hipLaunchKernelGGL(pad_tokens_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, rows, C / 8, pm);
    return finish();
}
