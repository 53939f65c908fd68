// fp32 accuracy path of the CTSD MMDiT forward (BASELINE north_star: "within 1e-3 rel fp32"): fp32-I/O forms of the
// non-GEMM kernels - LayerNorm family, attention, the element-wise glue.  Same argument structs and semantics as the bf16
// entry points (include/dwm_hip.h), every tensor pointer fp32, libm-accurate transcendental functions, fp32 MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 products, 1/16 of the bf16 rate) for the attention contractions.  Written for
// exactness and clarity, not speed: this mode exists to show that the path reproduces the reference's fp32 numbers; the
// GEMMs of the mode are dwm_gemm_f32 (gemm_bf16.hip).
#include "attention_common.h"

using namespace dwm_attn;

namespace {

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }
inline int finish() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DWM_OK : (int)e;
}

// ---------------------------------------------------------------------------------------------------- LayerNorm family
// one wave per row, three passes over the row (mean, centred variance, output): the row is re-read from L2
__global__ void __launch_bounds__(256)
layernorm_f32_kernel(const dwm_layernorm_args p) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int D = p.D;
    const float* __restrict__ x = (const float*)p.x + row * p.ldx;
    const float* addv = p.addvec ? (const float*)p.addvec + (row / p.rows_per_add) * p.ld_add : nullptr;
    float* xs = p.xsum ? (float*)p.xsum + row * p.ldxsum : nullptr;
    float s = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        float4 v = *(const float4*)(x + c);
        if (addv) {
            const float4 a = *(const float4*)(addv + c);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            if (xs) *(float4*)(xs + c) = v;
        }
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = wave_sum(s) / (float)D;
    auto load = [&](int c) {
        float4 v = *(const float4*)(x + c);
        if (addv) {
            const float4 a = *(const float4*)(addv + c);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        return v;
    };
    float q = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 v = load(c);
        const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + p.eps);
    const float* __restrict__ w = (const float*)p.weight;
    const float* __restrict__ b = (const float*)p.bias;
    const int64_t mrow = p.rows_per_mod > 0 ? row / p.rows_per_mod : 0;
    const float* sc = p.scale ? (const float*)p.scale + mrow * p.ld_mod : nullptr;
    const float* sh = p.shift ? (const float*)p.shift + mrow * p.ld_mod : nullptr;
    const float* sc2 = p.scale2 ? (const float*)p.scale2 + mrow * p.ld_mod : nullptr;
    const float* sh2 = p.shift2 ? (const float*)p.shift2 + mrow * p.ld_mod : nullptr;
    float* y = (float*)p.y + row * p.ldy;
    float* y2 = p.y2 ? (float*)p.y2 + row * p.ldy2 : nullptr;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 v = load(c);
        float n[4] = {(v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd};
        float o[4], o2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = n[j];
            if (w) t = t * w[c + j] + (b ? b[c + j] : 0.f);
            if (sc) t = t * (1.f + sc[c + j]) + sh[c + j];
            o[j] = t;
            if (y2) o2[j] = n[j] * (1.f + sc2[c + j]) + sh2[c + j];
        }
        *(float4*)(y + c) = make_float4(o[0], o[1], o[2], o[3]);
        if (y2) *(float4*)(y2 + c) = make_float4(o2[0], o2[1], o2[2], o2[3]);
    }
}

// ---------------------------------------------------------------------------------------------------- attention
// One wave per 32 queries of one (problem, head), keys in tiles of 32, online softmax (no deferral), everything fp32.
//   S^T = K Q^T with v_mfma_f32_32x32x2_f32: lane (l31, half) supplies K[key l31][32 half + kk] / Q[query l31][32 half + kk]
//   for instruction kk = 0..31 (the two k-slots of an instruction are the two halves of the head dimension), and receives, for
//   its query l31, the scores of keys (r & 3) + 8 (r >> 2) + 4 half - the accumulator layout of the bf16 kernel.
//   O^T += V^T P^T: instruction t multiplies keys key(t, half) = (t & 3) + 8 (t >> 2) + 4 half, whose probabilities ARE the
//   accumulator registers st[t] of the two half-waves; the A operand V[key(t, half)][dt*32 + l31] comes from the wave's 8 KiB
//   LDS image of the V tile.
// Row maps / segments / masks as attn_fwd_kernel (attention.hip).
__global__ void __launch_bounds__(256)
attn_f32_kernel(const AttnParams P) {
    __shared__ __attribute__((aligned(16))) float vs[4][32 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    float* const vt = vs[wave];

    uint32_t id = blockIdx.x;
    const uint32_t id1 = fdiv(id, P.fd_nqb);
    const int qb = (int)(id - id1 * P.fd_nqb.d);
    const int prob = (int)fdiv(id1, P.fd_heads);
    const int head = (int)(id1 - (uint32_t)prob * P.fd_heads.d);
    const int L = P.L, L0 = P.L0;
    const int q0i = qb * 128 + wave * 32;
    if (q0i >= P.qend) return;                                     // wave-uniform; the kernel has no block-level barrier
    const int64_t base0 = seg0_base(P.rm, prob);
    const float* Q0 = (const float*)P.q0;
    const float* K0 = (const float*)P.k0;
    const float* V0 = (const float*)P.v0;
    auto tok_off = [&](int l, int64_t ld0, int64_t ld1, int64_t delta) -> int64_t {      // element offset of token l's row
        return l < L0 ? seg0_row(P.rm, base0, l) * ld0 : ((int64_t)prob * P.L1 + (l - L0)) * ld1 + delta;
    };
    const int lq = q0i + l31;
    const bool qok = lq < P.qend;
    const int lqc = qok ? lq : P.qend - 1;
    const float* qp = Q0 + tok_off(lqc, P.ld0, P.ld1, P.seg1_delta) + head * 64 + half * 32;
    float qf[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 t = *(const float4*)(qp + 4 * i);
        qf[4 * i] = t.x * P.scale_log2; qf[4 * i + 1] = t.y * P.scale_log2; qf[4 * i + 2] = t.z * P.scale_log2; qf[4 * i + 3] = t.w * P.scale_log2;
    }
    uint32_t gbits = 0xffffffffu;
    const uint8_t* dense_row = nullptr;
    if (P.mask_mode == 1) {
        const int gq = (int)fmod_u(fdiv((uint32_t)lqc, P.fd_gs), P.fd_G);
        const uint8_t* mrow = P.mask + ((int64_t)fdiv((uint32_t)prob, P.fd_ppm) * P.mask_G + gq) * P.mask_G;
        gbits = 0;
        for (int g = 0; g < P.mask_G; ++g) gbits |= (mrow[g] ? 1u : 0u) << g;
    } else if (P.mask_mode == 2) {
        dense_row = P.mask + ((int64_t)prob * L + lqc) * L;
    }

    f32x16 ot[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int kb = P.kbeg; kb < L; kb += 32) {
        const int lk = kb + l31 < L ? kb + l31 : L - 1;
        const int64_t ko = tok_off(lk, P.ld0, P.ld1, P.seg1_delta) + head * 64 + half * 32;
        float kf[32];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 t = *(const float4*)(K0 + ko + 4 * i);
            kf[4 * i] = t.x; kf[4 * i + 1] = t.y; kf[4 * i + 2] = t.z; kf[4 * i + 3] = t.w;
            *(float4*)(vt + l31 * 64 + half * 32 + 4 * i) = *(const float4*)(V0 + ko + 4 * i);
        }
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[kk], qf[kk], st, 0, 0, 0);
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * half;
            bool ok = key < L;
            if (P.mask_mode == 1) {
                const int g = (int)fmod_u(fdiv((uint32_t)(ok ? key : 0), P.fd_gs), P.fd_G);
                ok = ok && ((gbits >> g) & 1u);
            } else if (P.mask_mode == 2) {
                ok = ok && dense_row[ok ? key : 0] != 0;
            }
            if (!ok) st[r] = -INFINITY;
            mx = fmaxf(mx, st[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        // a fully masked tile before the first finite score: m_new = -inf; keep everything at zero
        const float alpha = m_new == -INFINITY ? 1.f : exp2f(m_run - m_new);
        m_run = m_new;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = m_new == -INFINITY ? 0.f : exp2f(st[r] - m_new);
            sum += st[r];
        }
        l_run = l_run * alpha + sum;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
        // same-wave LDS ops complete in order: the V rows written above are visible to the reads below
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int key = (t & 3) + 8 * (t >> 2) + 4 * half;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                ot[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vt[key * 64 + dt * 32 + l31], st[t], ot[dt], 0, 0, 0);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (P.lse != nullptr && qok && half == 0) P.lse[((int64_t)prob * P.heads + head) * L + lq] = -m_run - log2f(l_tot);
    if (!qok) return;
    float* op = lqc < L0 ? (float*)P.o0 + seg0_row(P.rm, base0, lqc) * P.ldo0 + head * 64
                         : (float*)P.o1 + ((int64_t)prob * P.L1 + (lqc - L0)) * P.ldo1 + head * 64;
    // lane (q, half) reg r of ot[dt] -> d = dt*32 + (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
            *(float4*)(op + dt * 32 + rg * 8 + half * 4) = make_float4(ot[dt][rg * 4] * inv, ot[dt][rg * 4 + 1] * inv,
                                                                       ot[dt][rg * 4 + 2] * inv, ot[dt][rg * 4 + 3] * inv);
}

// ---------------------------------------------------------------------------------------------------- element-wise glue
__global__ void __launch_bounds__(256)
silu_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = x[i] / (1.f + expf(-x[i]));
}

// diffusers Timesteps(C, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin] of t * exp(-ln(10000) k / (C/2))
__global__ void __launch_bounds__(256)
sinusoid_f32_kernel(const float* __restrict__ t, int64_t n, int C, float* __restrict__ out) {
    const int half = C / 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * half) return;
    const int64_t r = i / half;
    const int k = (int)(i - r * half);
    const float e = t[r] * expf(-9.210340371976184f * (float)k / (float)half);
    out[r * C + k] = cosf(e);
    out[r * C + half + k] = sinf(e);
}

// x [I, C, H, W] fp32 -> tokens [I*h*w, ldo] with column (c*p + py)*p + px (the PatchEmbed Conv2d weight flattened: im2col)
__global__ void __launch_bounds__(256)
patchify_f32_kernel(const float* __restrict__ x, int64_t I, int C, int H, int W, int p, float* __restrict__ out, int64_t ldo) {
    const int h = H / p, w = W / p;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= I * h * w * ldo) return;
    const int col = (int)(idx % ldo);
    const int64_t tok = idx / ldo;
    float v = 0.f;
    if (col < C * p * p) {
        const int px = col % p, py = (col / p) % p, c = col / (p * p);
        const int ww = (int)(tok % w), hh = (int)((tok / w) % h);
        const int64_t n = tok / ((int64_t)w * h);
        v = x[((n * C + c) * H + hh * p + py) * W + ww * p + px];
    }
    out[idx] = v;
}

__global__ void __launch_bounds__(256)
unpatchify_f32_kernel(const float* __restrict__ x, int64_t ldx, int64_t I, int C, int h, int w, int p, float* __restrict__ out) {
    const int H = h * p, W = w * p;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= I * C * H * W) return;
    const int X = (int)(idx % W), Y = (int)((idx / W) % H);
    const int c = (int)((idx / ((int64_t)W * H)) % C);
    const int64_t n = idx / ((int64_t)W * H * C);
    const int ww = X / p, px = X - ww * p, hh = Y / p, py = Y - hh * p;
    out[idx] = x[((n * h + hh) * w + ww) * ldx + (py * p + px) * C + c];
}

// latents += dsigma * (u + g (c - u)), pred = [uncond; cond] fp32; the next model input = the latents, twice, fp32
__global__ void __launch_bounds__(256)
cfg_euler_f32_kernel(const float* __restrict__ pred, float* __restrict__ lat, float* __restrict__ model_in, int64_t n, float guidance,
                     float dsigma, const float* __restrict__ dsigma_group, int64_t group_elems) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (dsigma_group != nullptr) dsigma = dsigma_group[i / group_elems];
    const float u = pred[i], c = pred[n + i];
    const float o = lat[i] + dsigma * (u + guidance * (c - u));
    lat[i] = o;
    if (model_in) { model_in[i] = o; model_in[n + i] = o; }
}

}  // namespace

extern "C" int dwm_layernorm_f32(const dwm_layernorm_args* a, void* stream) {
    if (a == nullptr || a->x == nullptr || a->y == nullptr || a->rows <= 0 || a->D <= 0) return DWM_EINVAL;
    if (a->D % 4 != 0 || a->ldx % 4 != 0 || a->ldy % 4 != 0 || !dwm_aligned16(a->x) || !dwm_aligned16(a->y)) return DWM_EALIGN;
    if ((a->scale == nullptr) != (a->shift == nullptr) || (a->scale && (a->ld_mod % 4 != 0 || a->rows_per_mod <= 0))) return DWM_EINVAL;
    if (a->y2 && (a->scale2 == nullptr || a->shift2 == nullptr || a->ldy2 % 4 != 0 || !dwm_aligned16(a->y2))) return DWM_EINVAL;
    if (a->addvec && (a->rows_per_add <= 0 || a->ld_add % 4 != 0 || !dwm_aligned16(a->addvec))) return DWM_EALIGN;
    if (a->xsum && (a->addvec == nullptr || a->ldxsum % 4 != 0 || !dwm_aligned16(a->xsum))) return DWM_EALIGN;
    hipLaunchKernelGGL(layernorm_f32_kernel, dim3((unsigned)((a->rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
    return finish();
}

extern "C" int dwm_attention_f32(const dwm_attn_args* a, void* stream) {
    AttnParams P;
    dwm_attn_args b = *a;
    // fill_params checks bf16 strides (multiples of 8 elements); fp32 rows need multiples of 4
    const int rc = fill_params(&b, P);
    if (rc != DWM_OK) return rc;
    P.nqb = (int)((P.qend + 127) / 128);
    P.fd_nqb = make_fastdiv((uint32_t)P.nqb);
    P.fd_heads = make_fastdiv((uint32_t)P.heads);
    const int64_t nblk = (int64_t)P.n_problems * P.heads * P.nqb;
    if (nblk >= (1ll << 31)) return DWM_EUNSUPPORTED;
    // seg1_delta is computed by fill_params from bf16_t pointers (2-byte units): redo it in fp32 elements
    if (a->L1 > 0) {
        const int64_t dq = (const float*)a->q1 - (const float*)a->q0, dk = (const float*)a->k1 - (const float*)a->k0,
                      dv = (const float*)a->v1 - (const float*)a->v0;
        if (dq != dk || dk != dv) return DWM_EUNSUPPORTED;
        P.seg1_delta = dq;
    }
    hipLaunchKernelGGL(attn_f32_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, P);
    return finish();
}

extern "C" int dwm_silu_f32(const float* x, float* y, int64_t n, void* stream) {
    if (x == nullptr || y == nullptr || n <= 0) return DWM_EINVAL;
    hipLaunchKernelGGL(silu_f32_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return finish();
}

extern "C" int dwm_timestep_sinusoid_f32(const float* t, int64_t n, int32_t C, float* out, void* stream) {
    if (t == nullptr || out == nullptr || n <= 0 || C <= 0) return DWM_EINVAL;
    if (C % 2 != 0) return DWM_EUNSUPPORTED;
    hipLaunchKernelGGL(sinusoid_f32_kernel, dim3(blocks_for(n * (C / 2))), dim3(256), 0, (hipStream_t)stream, t, n, C, out);
    return finish();
}

extern "C" int dwm_patchify_f32(const float* x, int64_t I, int32_t C, int32_t H, int32_t W, int32_t p, float* out, int64_t ldo,
                                void* stream) {
    if (x == nullptr || out == nullptr || I <= 0 || C <= 0 || H <= 0 || W <= 0 || p <= 0) return DWM_EINVAL;
    if (H % p != 0 || W % p != 0 || ldo < (int64_t)C * p * p) return DWM_EINVAL;
    hipLaunchKernelGGL(patchify_f32_kernel, dim3(blocks_for(I * (H / p) * (W / p) * ldo)), dim3(256), 0, (hipStream_t)stream, x, I, C,
                       H, W, p, out, ldo);
    return finish();
}

extern "C" int dwm_unpatchify_f32(const float* x, int64_t ldx, int64_t I, int32_t C, int32_t h, int32_t w, int32_t p, float* out,
                                  void* stream) {
    if (x == nullptr || out == nullptr || I <= 0 || C <= 0 || h <= 0 || w <= 0 || p <= 0) return DWM_EINVAL;
    if (ldx < (int64_t)C * p * p) return DWM_EINVAL;
    hipLaunchKernelGGL(unpatchify_f32_kernel, dim3(blocks_for(I * C * h * p * w * p)), dim3(256), 0, (hipStream_t)stream, x, ldx, I, C,
                       h, w, p, out);
    return finish();
}

extern "C" int dwm_cfg_euler_step_f32(const float* pred, float* latents, float* model_in, int64_t n, float guidance, float dsigma,
                                      const float* dsigma_group, int64_t group_elems, void* stream) {
    if (pred == nullptr || latents == nullptr || n <= 0) return DWM_EINVAL;
    if (dsigma_group != nullptr && (group_elems <= 0 || n % group_elems != 0)) return DWM_EINVAL;
    hipLaunchKernelGGL(cfg_euler_f32_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, pred, latents, model_in, n, guidance,
                       dsigma, dsigma_group, group_elems > 0 ? group_elems : 1);
    return finish();
}
