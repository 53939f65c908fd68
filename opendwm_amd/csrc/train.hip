// Backward-pass / optimizer kernels of the CTSD train step (reference: src/dwm/pipelines/ctsd.py
// :1195-1437, backward of the modules listed in include/dwm_hip.h).  All HBM-bound: 16-B bf16x8
// accesses, fp32 arithmetic, reductions accumulated with fp32 atomics into caller-zeroed buffers.
#include "common.h"
#include "dwm_hip.h"

namespace {

// ---------------------------------------------------------------------------------- transpose
// 64 x 64 tile through LDS (row pitch 66 elements: conflict-free column reads); out[c][r] = in[r][c],
// output columns [rows, rows_pad) are zero-filled.
__global__ void __launch_bounds__(256)
transpose_kernel(const bf16_t* __restrict__ in, int64_t ld_in, int64_t rows, int64_t cols,
                 bf16_t* __restrict__ out, int64_t ld_out, int64_t rows_pad) {
    __shared__ bf16_t tile[64][66];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t r = r0 + ty * 16 + i, c = c0 + tx;
        tile[ty * 16 + i][tx] = (r < rows && c < cols) ? in[r * ld_in + c] : (bf16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t c = c0 + ty * 16 + i, r = r0 + tx;
        if (c < cols && r < rows_pad) out[c * ld_out + r] = tile[tx][ty * 16 + i];
    }
}

// Vector form (cols % 8 == 0, rows_pad % 8 == 0, 16-byte aligned rows on both sides): a thread owns an 8 x 8 block -
// eight 16-byte row loads, a register transpose (v_perm_b32), eight 16-byte stores - so no LDS round trip and every
// store instruction of a wave writes 4 output rows x 256 contiguous bytes.  Workgroup = 128 rows x 128 cols.
__global__ void __launch_bounds__(256)
transpose8_kernel(const bf16_t* __restrict__ in, int64_t ld_in, int64_t rows, int64_t cols,
                  bf16_t* __restrict__ out, int64_t ld_out, int64_t rows_pad) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 128 + wave * 32 + (lane & 3) * 8;
    const int64_t r = (int64_t)blockIdx.y * 128 + (lane >> 2) * 8;
    if (c >= cols || r >= rows_pad) return;
    uint32_t v[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if (r + i < rows) q = *(const uint4*)(in + (r + i) * ld_in + c);
        v[i][0] = q.x; v[i][1] = q.y; v[i][2] = q.z; v[i][3] = q.w;
    }
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
        const uint32_t sel = (cc & 1) ? 0x07060302u : 0x05040100u;     // high / low halves of the two source dwords
        uint4 o;
        o.x = __builtin_amdgcn_perm(v[1][cc >> 1], v[0][cc >> 1], sel);
        o.y = __builtin_amdgcn_perm(v[3][cc >> 1], v[2][cc >> 1], sel);
        o.z = __builtin_amdgcn_perm(v[5][cc >> 1], v[4][cc >> 1], sel);
        o.w = __builtin_amdgcn_perm(v[7][cc >> 1], v[6][cc >> 1], sel);
        *(uint4*)(out + (c + cc) * ld_out + r) = o;
    }
}

// ---------------------------------------------------------------------------------- segmented sums
// out[g][n] += sum_{r in group g} a[r][n] * (b ? b[r][n] : 1).  Block = 256 threads = 32 column
// chunks (8 columns each) x 8 row lanes; a block covers 256 columns x up to RB rows of ONE group.
constexpr int SEG_RB = 128;
__global__ void __launch_bounds__(256)
segsum_kernel(const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ b, int64_t ldb,
              const bf16_t* __restrict__ b2, int64_t ldb2,
              int64_t rows, int64_t ncols, int64_t rpg, int chunks_per_group, float* __restrict__ out, int64_t ld_out) {
    __shared__ float red[8][256];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int64_t g = blockIdx.y / chunks_per_group, ch = blockIdx.y % chunks_per_group;
    const int64_t rbeg = g * rpg + ch * SEG_RB;
    int64_t rend = rbeg + SEG_RB;
    const int64_t gend = (g + 1) * rpg < rows ? (g + 1) * rpg : rows;
    if (rend > gend) rend = gend;
    const int64_t c = (int64_t)blockIdx.x * 256 + cx * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < ncols) {
        for (int64_t r = rbeg + ry; r < rend; r += 8) {
            float va[8], vb[8];
            unpack8(*(const uint4*)(a + r * lda + c), va);
            if (b2) {           // a * (b - b2): the difference is taken in fp32 before the product (AlphaBlender d(alpha))
                float vc[8];
                unpack8(*(const uint4*)(b + r * ldb + c), vb);
                unpack8(*(const uint4*)(b2 + r * ldb2 + c), vc);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += va[j] * (vb[j] - vc[j]);
            } else if (b) {
                unpack8(*(const uint4*)(b + r * ldb + c), vb);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += va[j] * vb[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += va[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[ry][cx * 8 + j] = acc[j];
    __syncthreads();
    const int col = threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += red[i][col];
    const int64_t cc = (int64_t)blockIdx.x * 256 + col;
    if (cc < ncols && rbeg < rend) atomicAdd(out + g * ld_out + cc, s);
}

// ---------------------------------------------------------------------------------- activations
DWM_DEVINL float gelu_tanh_grad(float x) {
    // d/dx [x * sigmoid(2u)],  u = k (x + 0.044715 x^3)
    const float z = x * (2.302208198f + 0.1029432397f * x * x);                 // 2 u log2(e)
    const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-z));   // sigmoid(2u)
    const float du2 = 1.5957691216f + 0.2140624845f * x * x;                    // d(2u)/dx = 2k (1 + 3*0.044715 x^2)
    return s + x * s * (1.f - s) * du2;
}
DWM_DEVINL float silu_grad(float x) {
    const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
    return s * (1.f + x * (1.f - s));
}
DWM_DEVINL float gelu_erf_grad(float x) {
    // 0.5 (1 + erf(x / sqrt 2)) + x * exp(-x^2 / 2) / sqrt(2 pi)
    return 0.5f * (1.f + erf_fast_f(x * 0.7071067811865476f)) +
           x * 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.7213475204444817f * x * x);
}

template <int ACT, bool BWD>
__global__ void __launch_bounds__(256)
act_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float v[8], d[8];
        unpack8(*(const uint4*)(x + i * 8), v);
        if (BWD) unpack8(*(const uint4*)(dy + i * 8), d);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (!BWD) v[j] = ACT == DWM_ACT_GELU_TANH ? gelu_tanh_f(v[j]) : ACT == DWM_ACT_RELU ? fmaxf(v[j], 0.f) : silu_f(v[j]);
            else v[j] = ACT == DWM_ACT_RELU ? (v[j] > 0.f ? d[j] : 0.f)          // x may be the pre- or the post-activation
                                            : d[j] * (ACT == DWM_ACT_GELU_TANH ? gelu_tanh_grad(v[j]) : silu_grad(v[j]));
        }
        *(uint4*)(out + i * 8) = pack8(v);
    }
}

// GEGLU: u [rows, 2*inner] = [value | gate];  g = value * gelu_erf(gate)
template <bool BWD>
__global__ void __launch_bounds__(256)
geglu_kernel(const bf16_t* __restrict__ u, int64_t ldu, const bf16_t* __restrict__ dg, int64_t lddg,
             bf16_t* __restrict__ out, int64_t ldo, int64_t rows, int64_t inner) {
    const int64_t c8 = inner / 8, total = rows * c8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / c8, c = (i - r * c8) * 8;
        float hv[8], gt[8];
        unpack8(*(const uint4*)(u + r * ldu + c), hv);
        unpack8(*(const uint4*)(u + r * ldu + inner + c), gt);
        if (!BWD) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = hv[j] * gelu_erf_f(gt[j]);
            *(uint4*)(out + r * ldo + c) = pack8(o);
        } else {
            float d[8], dv[8], dgt[8];
            unpack8(*(const uint4*)(dg + r * lddg + c), d);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dv[j] = d[j] * gelu_erf_f(gt[j]);
                dgt[j] = d[j] * hv[j] * gelu_erf_grad(gt[j]);
            }
            *(uint4*)(out + r * ldo + c) = pack8(dv);
            *(uint4*)(out + r * ldo + inner + c) = pack8(dgt);
        }
    }
}

// out[r][n] = ca(r) * a[r][n] + cb(r) * b[r][n]
//   ca(r) = (ga ? ga[r / rpa][n] : 1) * (fa ? fa[r / rfa] : 1);  cb likewise without a per-column vector
__global__ void __launch_bounds__(256)
rowcombine_kernel(const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ ga, int64_t ld_ga, int64_t rpa,
                  const float* __restrict__ fa, int64_t rfa,
                  const bf16_t* __restrict__ b, int64_t ldb, const float* __restrict__ fb, int64_t rfb,
                  bf16_t* __restrict__ out, int64_t ldo, int64_t rows, int64_t ncols) {
    const int64_t c8 = ncols / 8, total = rows * c8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / c8, c = (i - r * c8) * 8;
        float va[8], o[8];
        unpack8(*(const uint4*)(a + r * lda + c), va);
        const float sa = fa ? fa[r / rfa] : 1.f;
        if (ga) {
            float vg[8];
            unpack8(*(const uint4*)(ga + (r / rpa) * ld_ga + c), vg);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = va[j] * vg[j] * sa;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = va[j] * sa;
        }
        if (b) {
            float vb[8];
            unpack8(*(const uint4*)(b + r * ldb + c), vb);
            const float sb = fb ? fb[r / rfb] : 1.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += vb[j] * sb;
        }
        *(uint4*)(out + r * ldo + c) = pack8(o);
    }
}

// ---------------------------------------------------------------------------------- LayerNorm backward
// One wave per row (row in registers), RB/4 rows per wave sequentially; per-lane fp32 partial
// sums of the parameter / modulation gradients over the wave's rows, reduced over the 4 waves in
// LDS and flushed with one atomic per column and workgroup.
constexpr int LN_BWD_RB = 64;        // (round 4: 128 -> 64.  One training sample is 43 008 rows = 336 blocks of 128: 1.3 blocks per CU, 32
                                     //  rows per wave one after the other, each a chain of three memory round trips - 219 us for 400 MB)
template <int NI>
__global__ void __launch_bounds__(256)
layernorm_bwd_kernel(const dwm_layernorm_bwd_args p, int chunks_per_group) {
    constexpr int RB = LN_BWD_RB;           // rows per block (a quarter per wave)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = p.D;
    const int64_t rpg = p.rows_per_mod > 0 ? p.rows_per_mod : p.rows;
    const int64_t g = blockIdx.x / chunks_per_group, ch = blockIdx.x % chunks_per_group;
    const int64_t rbeg = g * rpg + ch * RB;
    int64_t rend = rbeg + RB;
    const int64_t gend = (g + 1) * rpg < p.rows ? (g + 1) * rpg : p.rows;
    if (rend > gend) rend = gend;

    const bf16_t* __restrict__ w = (const bf16_t*)p.weight;
    const bf16_t* sc = p.scale ? (const bf16_t*)p.scale + g * p.ld_mod : nullptr;
    const bf16_t* sc2 = p.scale2 ? (const bf16_t*)p.scale2 + g * p.ld_mod : nullptr;
    float gam[NI][8], gam2[NI][8];          // effective gamma of output 1 / 2
    bool ok[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = (i * 64 + lane) * 8;
        ok[i] = c < D;
#pragma unroll
        for (int j = 0; j < 8; ++j) { gam[i][j] = 1.f; gam2[i][j] = 1.f; }
        if (ok[i]) {
            float t[8];
            if (w) { unpack8(*(const uint4*)(w + c), t);
#pragma unroll
                for (int j = 0; j < 8; ++j) gam[i][j] = t[j]; }
            if (sc) { unpack8(*(const uint4*)(sc + c), t);
#pragma unroll
                for (int j = 0; j < 8; ++j) gam[i][j] *= 1.f + t[j]; }
            if (sc2) { unpack8(*(const uint4*)(sc2 + c), t);
#pragma unroll
                for (int j = 0; j < 8; ++j) gam2[i][j] = 1.f + t[j]; }
        }
    }
    float sg[NI][8], sb[NI][8], sg2[NI][8], sb2[NI][8];   // sum dy*xhat, sum dy (outputs 1 / 2)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { sg[i][j] = sb[i][j] = sg2[i][j] = sb2[i][j] = 0.f; }

    for (int64_t row = rbeg + wave; row < rend; row += 4) {
        const bf16_t* __restrict__ x = (const bf16_t*)p.x + row * p.ldx;
        const bf16_t* addv = p.addvec ? (const bf16_t*)p.addvec + (row / p.rows_per_add) * p.ld_add : nullptr;
        float v[NI][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = (i * 64 + lane) * 8;
            if (ok[i]) {
                unpack8(*(const uint4*)(x + c), v[i]);
                if (addv) {
                    float a[8];
                    unpack8(*(const uint4*)(addv + c), a);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[i][j] += a[j];
                    const uint4 r = pack8(v[i]);          // same rounding point as the forward kernel
                    unpack8(r, v[i]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[i][j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
            }
        }
        // the incoming gradients are requested NOW, before the two wave reductions of the statistics (they do not depend on
        // them): one memory round trip per row less on the critical path
        const bf16_t* __restrict__ dy = (const bf16_t*)p.dy + row * p.lddy;
        const bf16_t* __restrict__ dy2 = p.dy2 ? (const bf16_t*)p.dy2 + row * p.lddy2 : nullptr;
        uint4 dyr[NI], dy2r[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = ok[i] ? (i * 64 + lane) * 8 : 0;
            dyr[i] = *(const uint4*)(dy + c);
            dy2r[i] = dy2 ? *(const uint4*)(dy2 + c) : dyr[i];
        }
        const float mean = wave_sum_dpp(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (ok[i]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q += d * d; }
            }
        const float rstd = rsqrtf(wave_sum_dpp(q) / (float)D + p.eps);

        float dxh[NI][8];                    // d loss / d xhat
        float m1 = 0.f, m2 = 0.f;            // sum dxhat, sum dxhat * xhat
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = (i * 64 + lane) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) dxh[i][j] = 0.f;
            if (!ok[i]) continue;
            float d[8];
            unpack8(dyr[i], d);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (v[i][j] - mean) * rstd;
                v[i][j] = xh;
                sg[i][j] += d[j] * xh;
                sb[i][j] += d[j];
                dxh[i][j] = d[j] * gam[i][j];
            }
            if (dy2) {
                unpack8(dy2r[i], d);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sg2[i][j] += d[j] * v[i][j];
                    sb2[i][j] += d[j];
                    dxh[i][j] += d[j] * gam2[i][j];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { m1 += dxh[i][j]; m2 += dxh[i][j] * v[i][j]; }
        }
        m1 = wave_sum_dpp(m1) / (float)D;
        m2 = wave_sum_dpp(m2) / (float)D;
        bf16_t* __restrict__ dx = (bf16_t*)p.dx + row * p.lddx;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (!ok[i]) continue;
            const int c = (i * 64 + lane) * 8;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rstd * (dxh[i][j] - m1 - v[i][j] * m2);
            if (p.accumulate) {
                float t[8];
                unpack8(*(const uint4*)(dx + c), t);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += t[j];
            }
            *(uint4*)(dx + c) = pack8(o);
        }
    }
    // flush: dgamma[gg][c], dbeta[gg][c]; gg = group for modulation gradients, 0 for affine parameters
    __shared__ float red[4][NI * 512];
    const int64_t gg = p.grad_per_group ? g : 0;
    float* const outs[4] = {p.dgamma, p.dbeta, p.dgamma2, p.dbeta2};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (outs[q] == nullptr) continue;               // kernel-argument uniform
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                red[wave][(i * 64 + lane) * 8 + j] = q == 0 ? sg[i][j] : q == 1 ? sb[i][j] : q == 2 ? sg2[i][j] : sb2[i][j];
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) {
            const float v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
            if (rbeg < rend) atomicAdd(outs[q] + gg * p.ld_grad + c, v);
        }
    }
}

// ---------------------------------------------------------------------------------- per-head RMSNorm
// forward (in place) with the reciprocal RMS kept for the backward; one wave per row, a lane pair per
// ... simpler: 8 lanes per head (8 elements each), ncols/8 lanes busy per row pass of 512 columns.
__global__ void __launch_bounds__(256)
rmsnorm_heads_train_kernel(bf16_t* __restrict__ x, int64_t ldx, int64_t rows, int64_t ncols,
                           const bf16_t* __restrict__ w, float eps, float* __restrict__ rinv_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t nh = ncols / 64;
    for (int64_t c = lane * 8; c < ncols; c += 512) {
        float v[8], t[8];
        unpack8(*(const uint4*)(x + row * ldx + c), v);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
        ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
        const float rinv = rsqrtf(ss * (1.f / 64.f) + eps);
        unpack8(*(const uint4*)(w + c), t);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= rinv * t[j];
        *(uint4*)(x + row * ldx + c) = pack8(v);
        if (rinv_out && (lane & 7) == 0) rinv_out[row * nh + c / 64] = rinv;
    }
}

// backward in place on dy (-> dx); y = normalised output (xhat * w), rinv from the forward.
// dw[c] += sum_rows dy * xhat (fp32 atomics; block-local LDS reduction first).
__global__ void __launch_bounds__(256)
rmsnorm_heads_bwd_kernel(const bf16_t* __restrict__ y, int64_t ldy, const float* __restrict__ rinv,
                         const bf16_t* __restrict__ w, bf16_t* __restrict__ dy, int64_t lddy,
                         int64_t rows, int64_t ncols, float* __restrict__ dw) {
    extern __shared__ float dwl[];            // [ncols]
    for (int i = threadIdx.x; i < ncols; i += 256) dwl[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nh = ncols / 64;
    const int64_t rbeg = (int64_t)blockIdx.x * 32;
    for (int64_t c = lane * 8; c < ncols; c += 512) {
        float wv[8], acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        unpack8(*(const uint4*)(w + c), wv);
        for (int64_t row = rbeg + wave; row < rbeg + 32 && row < rows; row += 4) {
            float yv[8], d[8], xh[8], dxh[8];
            unpack8(*(const uint4*)(y + row * ldy + c), yv);
            unpack8(*(const uint4*)(dy + row * lddy + c), d);
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh[j] = wv[j] != 0.f ? yv[j] / wv[j] : 0.f;
                acc[j] += d[j] * xh[j];
                dxh[j] = d[j] * wv[j];
                m += dxh[j] * xh[j];
            }
            m += __shfl_xor(m, 1, 64); m += __shfl_xor(m, 2, 64); m += __shfl_xor(m, 4, 64);
            m *= 1.f / 64.f;
            const float ri = rinv[row * nh + c / 64];
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = ri * (dxh[j] - xh[j] * m);
            *(uint4*)(dy + row * lddy + c) = pack8(d);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&dwl[c + j], acc[j]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncols; i += 256)
        if (dwl[i] != 0.f) atomicAdd(dw + i, dwl[i]);
}

// ---------------------------------------------------------------------------------- optimizer / casts
// AdamW (decoupled weight decay, torch.optim.AdamW semantics) on fp32 master parameters; refreshes
// the bf16 compute copy in the same pass.
__global__ void __launch_bounds__(256)
adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
             bf16_t* __restrict__ pb, int64_t n, float lr, float b1, float b2, float eps, float wd,
             float bc1, float bc2, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        pi -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (pb) pb[i] = f32_to_bf16(pi);
    }
}

// The same update for MANY tensors in one launch: block b works on elements [block_start[b], block_start[b] + chunk) of tensor
// block_item[b] (a model has 1.4-1.7 k parameter tensors: per-tensor launches made the optimizer step launch-bound).
__global__ void __launch_bounds__(256)
adamw_multi_kernel(const dwm_adamw_item* __restrict__ items, const int32_t* __restrict__ block_item,
                   const int64_t* __restrict__ block_start, int64_t chunk, float lr, float b1, float b2, float eps, float wd,
                   float bc1, float bc2, float gscale) {
    const dwm_adamw_item it = items[block_item[blockIdx.x]];
    const int64_t i0 = block_start[blockIdx.x];
    const int64_t i1 = i0 + chunk < it.n ? i0 + chunk : it.n;
    float* __restrict__ p = it.p;
    const float* __restrict__ g = it.g;
    float* __restrict__ m = it.m;
    float* __restrict__ v = it.v;
    bf16_t* __restrict__ pb = (bf16_t*)it.p_bf16;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const float gi = g[i] * gscale;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        pi -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (pb) pb[i] = f32_to_bf16(pi);
    }
}

__global__ void __launch_bounds__(256)
cast_bf16_to_f32_kernel(const bf16_t* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
                        int64_t rows, int64_t cols, int accumulate) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols, c = i - r * cols;
        const float v = bf16_to_f32(x[r * ldx + c]);
        float* d = y + r * ldy + c;
        *d = accumulate ? *d + v : v;
    }
}

// the same, 8 elements per thread (cols % 8 == 0, 16-byte aligned rows): the entry of the fp32 residual stream (132 M elements)
__global__ void __launch_bounds__(256)
cast_bf16_to_f32_vec_kernel(const bf16_t* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
                            int64_t rows, int64_t cols8, int accumulate) {
    const int64_t total = rows * cols8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols8, c = (i - r * cols8) * 8;
        float v[8];
        unpack8(*(const uint4*)(x + r * ldx + c), v);
        float* d = y + r * ldy + c;
        if (accumulate) {
            const float4 a = *(const float4*)d, b = *(const float4*)(d + 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        *(float4*)d = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

inline unsigned grid_for(int64_t work_items) {
    int64_t b = (work_items + 255) / 256;
    if (b < 1) b = 1;
    if (b > 256 * 32) b = 256 * 32;
    return (unsigned)b;
}

}  // namespace

#define DWM_RET()                                          \
    do {                                                   \
        const hipError_t e_ = hipGetLastError();           \
        return e_ == hipSuccess ? DWM_OK : (int)e_;        \
    } while (0)

extern "C" int dwm_transpose_bf16(const void* in, int64_t ld_in, int64_t rows, int64_t cols, void* out,
                                  int64_t ld_out, int64_t rows_pad, void* stream) {
    if (!in || !out || rows <= 0 || cols <= 0 || rows_pad < rows || ld_in < cols || ld_out < rows_pad) return DWM_EINVAL;
    if (cols % 8 == 0 && rows_pad % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && dwm_aligned16(in) && dwm_aligned16(out)) {
        const dim3 grid8((unsigned)((cols + 127) / 128), (unsigned)((rows_pad + 127) / 128));
        hipLaunchKernelGGL(transpose8_kernel, grid8, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, rows, cols,
                           (bf16_t*)out, ld_out, rows_pad);
        DWM_RET();
    }
    const dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows_pad + 63) / 64));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, rows, cols,
                       (bf16_t*)out, ld_out, rows_pad);
    DWM_RET();
}

extern "C" int dwm_segsum(const void* a, int64_t lda, const void* b, int64_t ldb, int64_t rows, int64_t ncols,
                          int64_t rows_per_group, float* out, int64_t ld_out, void* stream) {
    if (!a || !out || rows <= 0 || ncols <= 0 || ncols % 8 != 0 || rows_per_group <= 0) return DWM_EINVAL;
    if (lda % 8 != 0 || (b && ldb % 8 != 0) || !dwm_aligned16(a) || (b && !dwm_aligned16(b))) return DWM_EALIGN;
    const int64_t groups = (rows + rows_per_group - 1) / rows_per_group;
    const int64_t rpg = rows_per_group < rows ? rows_per_group : rows;
    const int cpg = (int)((rpg + SEG_RB - 1) / SEG_RB);
    if (groups * cpg >= 65536 * 16) return DWM_EUNSUPPORTED;
    const dim3 grid((unsigned)((ncols + 255) / 256), (unsigned)(groups * cpg));
    hipLaunchKernelGGL(segsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb,
                       (const bf16_t*)nullptr, (int64_t)0, rows, ncols, rows_per_group, cpg, out, ld_out);
    DWM_RET();
}

extern "C" int dwm_segsum_diff(const void* a, int64_t lda, const void* b, int64_t ldb, const void* b2, int64_t ldb2, int64_t rows,
                               int64_t ncols, int64_t rows_per_group, float* out, int64_t ld_out, void* stream) {
    if (!a || !b || !b2 || !out || rows <= 0 || ncols <= 0 || ncols % 8 != 0 || rows_per_group <= 0) return DWM_EINVAL;
    if (lda % 8 != 0 || ldb % 8 != 0 || ldb2 % 8 != 0 || !dwm_aligned16(a) || !dwm_aligned16(b) || !dwm_aligned16(b2)) return DWM_EALIGN;
    const int64_t groups = (rows + rows_per_group - 1) / rows_per_group;
    const int64_t rpg = rows_per_group < rows ? rows_per_group : rows;
    const int cpg = (int)((rpg + SEG_RB - 1) / SEG_RB);
    if (groups * cpg >= 65536 * 16) return DWM_EUNSUPPORTED;
    const dim3 grid((unsigned)((ncols + 255) / 256), (unsigned)(groups * cpg));
    hipLaunchKernelGGL(segsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb,
                       (const bf16_t*)b2, ldb2, rows, ncols, rows_per_group, cpg, out, ld_out);
    DWM_RET();
}

extern "C" int dwm_act_fwd(const void* x, void* y, int64_t n, int32_t act, void* stream) {
    if (!x || !y || n <= 0 || n % 8 != 0) return DWM_EINVAL;
    if (!dwm_aligned16(x) || !dwm_aligned16(y)) return DWM_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (act == DWM_ACT_GELU_TANH) hipLaunchKernelGGL((act_kernel<DWM_ACT_GELU_TANH, false>), dim3(grid_for(n / 8)), dim3(256), 0, s, (const bf16_t*)x, nullptr, (bf16_t*)y, n / 8);
    else if (act == DWM_ACT_SILU) hipLaunchKernelGGL((act_kernel<DWM_ACT_SILU, false>), dim3(grid_for(n / 8)), dim3(256), 0, s, (const bf16_t*)x, nullptr, (bf16_t*)y, n / 8);
    else if (act == DWM_ACT_RELU) hipLaunchKernelGGL((act_kernel<DWM_ACT_RELU, false>), dim3(grid_for(n / 8)), dim3(256), 0, s, (const bf16_t*)x, nullptr, (bf16_t*)y, n / 8);
    else return DWM_EINVAL;
    DWM_RET();
}

extern "C" int dwm_act_bwd(const void* x, const void* dy, void* dx, int64_t n, int32_t act, void* stream) {
    if (!x || !dy || !dx || n <= 0 || n % 8 != 0) return DWM_EINVAL;
    if (!dwm_aligned16(x) || !dwm_aligned16(dy) || !dwm_aligned16(dx)) return DWM_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (act == DWM_ACT_GELU_TANH) hipLaunchKernelGGL((act_kernel<DWM_ACT_GELU_TANH, true>), dim3(grid_for(n / 8)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n / 8);
    else if (act == DWM_ACT_SILU) hipLaunchKernelGGL((act_kernel<DWM_ACT_SILU, true>), dim3(grid_for(n / 8)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n / 8);
    else if (act == DWM_ACT_RELU) hipLaunchKernelGGL((act_kernel<DWM_ACT_RELU, true>), dim3(grid_for(n / 8)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n / 8);
    else return DWM_EINVAL;
    DWM_RET();
}

extern "C" int dwm_geglu_fwd(const void* u, int64_t ldu, int64_t rows, int64_t inner, void* g, int64_t ldg, void* stream) {
    if (!u || !g || rows <= 0 || inner <= 0 || inner % 8 != 0 || ldu < 2 * inner || ldg < inner) return DWM_EINVAL;
    if (ldu % 8 != 0 || ldg % 8 != 0 || !dwm_aligned16(u) || !dwm_aligned16(g)) return DWM_EALIGN;
    hipLaunchKernelGGL((geglu_kernel<false>), dim3(grid_for(rows * inner / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)u, ldu, nullptr, 0, (bf16_t*)g, ldg, rows, inner);
    DWM_RET();
}

extern "C" int dwm_geglu_bwd(const void* u, int64_t ldu, const void* dg, int64_t lddg, int64_t rows, int64_t inner,
                             void* du, int64_t lddu, void* stream) {
    if (!u || !dg || !du || rows <= 0 || inner <= 0 || inner % 8 != 0 || ldu < 2 * inner || lddg < inner || lddu < 2 * inner) return DWM_EINVAL;
    if (ldu % 8 != 0 || lddg % 8 != 0 || lddu % 8 != 0 || !dwm_aligned16(u) || !dwm_aligned16(dg) || !dwm_aligned16(du)) return DWM_EALIGN;
    hipLaunchKernelGGL((geglu_kernel<true>), dim3(grid_for(rows * inner / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)u, ldu, (const bf16_t*)dg, lddg, (bf16_t*)du, lddu, rows, inner);
    DWM_RET();
}

extern "C" int dwm_rowcombine(const dwm_rowcombine_args* a, void* stream) {
    if (!a || !a->a || !a->out || a->rows <= 0 || a->ncols <= 0 || a->ncols % 8 != 0) return DWM_EINVAL;
    if (a->lda % 8 != 0 || a->ldo % 8 != 0 || (a->b && a->ldb % 8 != 0) || (a->gate_a && a->ld_gate_a % 8 != 0)) return DWM_EALIGN;
    if (!dwm_aligned16(a->a) || !dwm_aligned16(a->out) || (a->b && !dwm_aligned16(a->b)) || (a->gate_a && !dwm_aligned16(a->gate_a))) return DWM_EALIGN;
    if ((a->gate_a && a->rows_per_gate_a <= 0) || (a->coef_a && a->rows_per_coef_a <= 0) || (a->coef_b && a->rows_per_coef_b <= 0)) return DWM_EINVAL;
    hipLaunchKernelGGL(rowcombine_kernel, dim3(grid_for(a->rows * a->ncols / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)a->a, a->lda, (const bf16_t*)a->gate_a, a->ld_gate_a, a->rows_per_gate_a,
                       a->coef_a, a->rows_per_coef_a, (const bf16_t*)a->b, a->ldb, a->coef_b, a->rows_per_coef_b,
                       (bf16_t*)a->out, a->ldo, a->rows, a->ncols);
    DWM_RET();
}

extern "C" int dwm_layernorm_bwd(const dwm_layernorm_bwd_args* a, void* stream) {
    if (!a || !a->x || !a->dy || !a->dx || a->rows <= 0 || a->D <= 0 || a->D % 8 != 0 || a->D > 2048) return DWM_EINVAL;
    if (a->ldx % 8 != 0 || a->lddy % 8 != 0 || a->lddx % 8 != 0 || (a->dy2 && a->lddy2 % 8 != 0)) return DWM_EALIGN;
    if ((a->scale || a->scale2) && (a->rows_per_mod <= 0 || a->ld_mod % 8 != 0)) return DWM_EINVAL;
    if (a->addvec && (a->rows_per_add <= 0 || a->ld_add % 8 != 0)) return DWM_EINVAL;
    if ((a->dgamma || a->dbeta || a->dgamma2 || a->dbeta2) && a->ld_grad < a->D) return DWM_EINVAL;
    const int64_t rpg = a->rows_per_mod > 0 ? a->rows_per_mod : a->rows;
    const int64_t groups = (a->rows + rpg - 1) / rpg;
    const int cpg = (int)(((rpg < a->rows ? rpg : a->rows) + LN_BWD_RB - 1) / LN_BWD_RB);
    if (groups * cpg >= (1ll << 31)) return DWM_EUNSUPPORTED;
    const dim3 grid((unsigned)(groups * cpg));
    hipStream_t s = (hipStream_t)stream;
    const int ni = (a->D + 511) / 512;
    switch (ni) {
        case 1: hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, dim3(256), 0, s, *a, cpg); break;
        case 2: hipLaunchKernelGGL(layernorm_bwd_kernel<2>, grid, dim3(256), 0, s, *a, cpg); break;
        case 3: hipLaunchKernelGGL(layernorm_bwd_kernel<3>, grid, dim3(256), 0, s, *a, cpg); break;
        default: hipLaunchKernelGGL(layernorm_bwd_kernel<4>, grid, dim3(256), 0, s, *a, cpg); break;
    }
    DWM_RET();
}

extern "C" int dwm_rmsnorm_heads_train(void* x, int64_t ldx, int64_t rows, int64_t ncols, const void* w, float eps,
                                       float* rinv, void* stream) {
    if (!x || !w || rows <= 0 || ncols <= 0 || ncols % 64 != 0 || ldx % 8 != 0) return DWM_EINVAL;
    if (!dwm_aligned16(x) || !dwm_aligned16(w)) return DWM_EALIGN;
    hipLaunchKernelGGL(rmsnorm_heads_train_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)x, ldx, rows, ncols, (const bf16_t*)w, eps, rinv);
    DWM_RET();
}

extern "C" int dwm_rmsnorm_heads_bwd(const void* y, int64_t ldy, const float* rinv, const void* w, void* dy, int64_t lddy,
                                     int64_t rows, int64_t ncols, float* dw, void* stream) {
    if (!y || !rinv || !w || !dy || !dw || rows <= 0 || ncols <= 0 || ncols % 64 != 0 || ncols > 8192) return DWM_EINVAL;
    if (ldy % 8 != 0 || lddy % 8 != 0 || !dwm_aligned16(y) || !dwm_aligned16(dy) || !dwm_aligned16(w)) return DWM_EALIGN;
    hipLaunchKernelGGL(rmsnorm_heads_bwd_kernel, dim3((unsigned)((rows + 31) / 32)), dim3(256), (size_t)ncols * sizeof(float),
                       (hipStream_t)stream, (const bf16_t*)y, ldy, rinv, (const bf16_t*)w, (bf16_t*)dy, lddy, rows, ncols, dw);
    DWM_RET();
}

extern "C" int dwm_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
                         float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, float grad_scale,
                         void* stream) {
    if (!p || !g || !m || !v || n <= 0 || bias_corr1 <= 0.f || bias_corr2 <= 0.f) return DWM_EINVAL;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16, n, lr,
                       beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, grad_scale);
    DWM_RET();
}

extern "C" int dwm_adamw_multi(const dwm_adamw_item* items, const int32_t* block_item, const int64_t* block_start, int64_t n_blocks,
                               int64_t chunk, float lr, float beta1, float beta2, float eps, float weight_decay, float bias_corr1,
                               float bias_corr2, float grad_scale, void* stream) {
    if (!items || !block_item || !block_start || n_blocks <= 0 || n_blocks >= (1ll << 31) || chunk <= 0 || bias_corr1 <= 0.f ||
        bias_corr2 <= 0.f) return DWM_EINVAL;
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, items, block_item, block_start,
                       chunk, lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, grad_scale);
    DWM_RET();
}

extern "C" int dwm_cast_bf16_to_f32(const void* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t cols,
                                    int32_t accumulate, void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0 || ldx < cols || ldy < cols) return DWM_EINVAL;
    if (cols % 8 == 0 && ldx % 8 == 0 && ldy % 4 == 0 && dwm_aligned16(x) && dwm_aligned16(y))
        hipLaunchKernelGGL(cast_bf16_to_f32_vec_kernel, dim3(grid_for(rows * cols / 8)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, ldx, y, ldy, rows, cols / 8, accumulate);
    else
        hipLaunchKernelGGL(cast_bf16_to_f32_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, ldx, y, ldy, rows, cols, accumulate);
    DWM_RET();
}
