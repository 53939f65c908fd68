// Attention backward for gfx950 (head_dim 64, bf16 in / out, fp32 math), the derivative of
// attention.hip with the same addressing (row maps, two segments, group / dense masks).
//
//   P  = exp2(c q.k - lse)            (recomputed from the saved log2-domain LSE, no running max)
//   D  = rowsum(dO * O)               (delta_kernel)
//   dS = P * (dO.V^T - D)
//   dQ = scale * dS K,   dK = scale * dS^T Q,   dV = P^T dO
//
// Two MFMA kernels, no atomics: dq_kernel owns 128 queries of one (problem, head) and walks the key
// tiles exactly like the forward; dkv_kernel owns 128 keys and walks the query tiles.  Both use the
// forward's swapped-operand trick so the softmax-side quantities are lane-local:
//   dq_kernel :  lane = query, registers = keys     (neg_lse, neg_D broadcast as MFMA C-init)
//   dkv_kernel:  lane = key,   registers = queries  (neg_lse, neg_D read from LDS as MFMA C-init)
// and the probability / dS registers feed the second MFMA of each product as its B operand without
// moving across lanes; the transposed A operands come from ds_read_b64_tr_b16 on a second,
// V-swizzled LDS image of the same tile.
#include "attention_common.h"

using namespace dwm_attn;

namespace {

constexpr int IMG = 8192;                    // one 64-row x 64-col bf16 image

// The LDS-DMA requests of this file go through inline asm, as in gemm_bf16.hip: the compiler models the builtin
// (global_load_lds) as a FLAT access that may touch the LDS, and with one of those pending it degrades every later LDS wait
// to lgkmcnt(0) - each fragment read then drained the whole LDS queue (26 of the 33 LDS waits of attn_dq_kernel were full
// drains).  Opaque requests keep its LDS bookkeeping exact (counted waits); the ordering the requests need is explicit here
// anyway: every stage is followed by "s_waitcnt vmcnt(0)" + __syncthreads() before anything reads it.
// lds_addr: wave-uniform LDS byte address (it goes to M0); per-lane destination = lds_addr + lane * bytes.
DWM_DEVINL void dma16(const void* gsrc, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory", "m0");
}
DWM_DEVINL void dma4(const void* gsrc, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory", "m0");
}
DWM_DEVINL uint32_t lds_address(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }

// row index (within its segment's buffers) of token l of problem prob
DWM_DEVINL int64_t token_row(const AttnParams& P, int64_t base0, int prob, int l) {
    return l < P.L0 ? seg0_row(P.rm, base0, l) : (int64_t)prob * P.L1 + (l - P.L0);
}

// ------------------------------------------------------------------------------------ delta
// neg_delta[p, h, l] = -sum_d dO[row, h*64 + d] * O[row, h*64 + d]; one wave per token row
__global__ void __launch_bounds__(256)
delta_kernel(const AttnParams P) {
    const int lane = threadIdx.x & 63;
    const int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= (int64_t)P.n_problems * P.L) return;
    const int prob = (int)(idx / P.L), l = (int)(idx - (int64_t)prob * P.L);
    if (l >= P.qend) return;                       // cross-attention: only the queries (segment 0) have outputs
    const int64_t row = token_row(P, seg0_base(P.rm, prob), prob, l);
    const bf16_t* o = P.o0 + (l < P.L0 ? row * P.ldo0 : P.oseg1_delta + row * P.ldo1);
    const bf16_t* d = P.do0 + (l < P.L0 ? row * P.ldo0 : P.doseg1_delta + row * P.ldo1);
    for (int c = lane * 8; c < P.heads * 64; c += 512) {
        float a[8], b[8];
        unpack8(*(const uint4*)(o + c), a);
        unpack8(*(const uint4*)(d + c), b);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += a[j] * b[j];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if ((lane & 7) == 0) P.delta[((int64_t)prob * P.heads + (c >> 6)) * P.L + l] = -s;
    }
}

// shared lane geometry of the fragment reads (see attention.hip)
struct Geo {
    int kswz, kfrag;          // normal (K-swizzled image) reads: row l31, chunk (2 ks + half) ^ kswz
    int vra[2], vrb[2];       // transposing reads of a V-swizzled image
};
DWM_DEVINL Geo make_geo(int lane) {
    Geo g;
    const int half = lane >> 5, l31 = lane & 31;
    g.kswz = (lane >> 1) & 7;
    g.kfrag = l31 * 128;
    const int tr_u = lane & 15, tr_g = (lane >> 4) & 1;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const int dcol = dt * 32 + tr_g * 16 + (tr_u & 3) * 4;
        const int rA = half * 4 + (tr_u >> 2), rB = rA + 8;
        g.vra[dt] = rA * 128 + (((dcol >> 3) ^ (((rA >> 1) & 1) << 2)) << 4) + ((dcol & 7) << 1);
        g.vrb[dt] = rB * 128 + (((dcol >> 3) ^ (((rB >> 1) & 1) << 2)) << 4) + ((dcol & 7) << 1);
    }
    return g;
}
DWM_DEVINL bf16x8 read_frag(const char* img, const Geo& g, int j, int ks, int half) {
    return *(const bf16x8*)(img + g.kfrag + j * 32 * 128 + (((2 * ks + half) ^ g.kswz) << 4));
}
DWM_DEVINL bf16x8 read_frag_t(const char* img, const Geo& g, int s, int dt) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + g.vra[dt] + s * (16 * 128)));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + g.vrb[dt] + s * (16 * 128)));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
DWM_DEVINL bf16x8 pack_frag(const f32x16& v, int s2) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = v[s2 * 8 + e];
    const uint4 pk = pack8(t);
    return *reinterpret_cast<const bf16x8*>(&pk);
}

// mask bits of one side: forward-style row of the group mask (queries) or its column (keys)
DWM_DEVINL uint32_t group_bits(const AttnParams& P, int prob, int l, bool column) {
    const int g0 = (int)fmod_u(fdiv((uint32_t)l, P.fd_gs), P.fd_G);
    const uint8_t* m = P.mask + (int64_t)fdiv((uint32_t)prob, P.fd_ppm) * P.mask_G * P.mask_G;
    uint32_t bits = 0;
    for (int g = 0; g < P.mask_G; ++g) bits |= ((column ? m[g * P.mask_G + g0] : m[g0 * P.mask_G + g]) ? 1u : 0u) << g;
    return bits;
}
DWM_DEVINL int group_of(const AttnParams& P, int l) {
    int g = (int)(((float)l + 0.5f) * P.inv_group_size);
    g -= P.mask_G * (int)(((float)g + 0.5f) * P.inv_G);
    return g;
}

// ------------------------------------------------------------------------------------ dQ
// Stage = K (K-swizzle) | K (V-swizzle, transposing reads) | V (K-swizzle); 2 stages.
constexpr int DQ_STAGE = 3 * IMG;
template <int MASK>
__global__ void __launch_bounds__(256, 2)
attn_dq_kernel(const AttnParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* __restrict__ rowidx = (int32_t*)(smem + 2 * DQ_STAGE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;

    uint32_t id = (uint32_t)xcd_remap(blockIdx.x, P.n_problems * P.heads * P.nqb);
    const uint32_t id1 = fdiv(id, P.fd_nqb);
    const int qb = (int)(id - id1 * P.fd_nqb.d);
    const int prob = (int)fdiv(id1, P.fd_heads);
    const int head = (int)(id1 - (uint32_t)prob * P.fd_heads.d);
    const int L = P.L, L0 = P.L0;
    const int64_t hoff = (int64_t)head * 64;
    {
        const int64_t base0 = seg0_base(P.rm, prob);
        for (int l = tid; l < L; l += 256) rowidx[l] = (int32_t)token_row(P, base0, prob, l);
    }
    __syncthreads();

    // queries are tokens [0, qend), keys tokens [kbeg, L): (L, 0), or (L0, L0) for cross-attention
    const int lq = qb * 128 + wave * 32 + l31;
    const bool qok = lq < P.qend;
    const int lqc = qok ? lq : P.qend - 1;
    const bool wave_active = qb * 128 + wave * 32 < P.qend;
    const int64_t qrow = rowidx[lqc];
    const bool qseg0 = lqc < L0;
    bf16x8 qf[4], dof[4];
    {
        const bf16_t* qptr = P.q0 + (qseg0 ? qrow * P.ld0 : P.seg1_delta + qrow * P.ld1) + hoff;
        const bf16_t* dptr = P.do0 + (qseg0 ? qrow * P.ldo0 : P.doseg1_delta + qrow * P.ldo1) + hoff;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = scale_frag(*(const bf16x8*)(qptr + ks * 16 + half * 8), P.scale_log2);
            dof[ks] = *(const bf16x8*)(dptr + ks * 16 + half * 8);
        }
    }
    const int64_t sidx = ((int64_t)prob * P.heads + head) * L + lqc;
    const float nl = P.lse[sidx], nd = P.delta[sidx];
    f32x16 neglse, negdel, acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { neglse[r] = nl; negdel[r] = nd; acc[0][r] = 0.f; acc[1][r] = 0.f; }
    uint32_t gbits = 0xffffffffu;
    const uint8_t* dense_row = nullptr;
    if (MASK == 1) gbits = group_bits(P, prob, lqc, false);
    else if (MASK == 2) dense_row = P.mask + ((int64_t)prob * L + lqc) * L;

    const Geo geo = make_geo(lane);
    const int nkt = (L - P.kbeg + KT - 1) / KT;
    const int srow0 = wave * 16 + (lane >> 3), srow1 = srow0 + 8;
    const int kc0 = ((lane & 7) ^ ((srow0 >> 1) & 7)) << 3, kc1 = ((lane & 7) ^ ((srow1 >> 1) & 7)) << 3;
    const int vc0 = ((lane & 7) ^ (((srow0 >> 1) & 1) << 2)) << 3, vc1 = ((lane & 7) ^ (((srow1 >> 1) & 1) << 2)) << 3;
    const int sdst = wave * 2048;
    const uint32_t lds0 = lds_address(smem);
#define DQ_DMA(kt_, stage_)                                                                  \
    do {                                                                                     \
        const int kb_ = P.kbeg + (kt_) * KT;                                                 \
        const int ra_ = kb_ + srow0 < L ? kb_ + srow0 : L - 1;                               \
        const int rb_ = kb_ + srow1 < L ? kb_ + srow1 : L - 1;                               \
        const int64_t oa_ = (ra_ < L0 ? (int64_t)rowidx[ra_] * P.ld0 : P.seg1_delta + (int64_t)rowidx[ra_] * P.ld1) + hoff; \
        const int64_t ob_ = (rb_ < L0 ? (int64_t)rowidx[rb_] * P.ld0 : P.seg1_delta + (int64_t)rowidx[rb_] * P.ld1) + hoff; \
        const uint32_t l_ = lds0 + (stage_) * DQ_STAGE + sdst;                               \
        dma16(P.k0 + oa_ + kc0, l_);             dma16(P.k0 + ob_ + kc1, l_ + 1024);         \
        dma16(P.k0 + oa_ + vc0, l_ + IMG);       dma16(P.k0 + ob_ + vc1, l_ + IMG + 1024);   \
        dma16(P.v0 + oa_ + kc0, l_ + 2 * IMG);   dma16(P.v0 + ob_ + kc1, l_ + 2 * IMG + 1024); \
    } while (0)

    DQ_DMA(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) DQ_DMA(kt + 1, (kt + 1) & 1);
        const char* kimg = smem + (kt & 1) * DQ_STAGE;
        const char* ktimg = kimg + IMG;
        const char* vimg = kimg + 2 * IMG;
        if (wave_active) {
            // S and dP: the four fragments of MFMA step ks + 1 are read while the four MFMAs of step ks run (two register sets,
            // fenced so that the reads stay in front of the MFMAs they overlap with).  Written as read-then-use pairs the
            // compiler reused ONE fragment register for all sixteen reads: every MFMA then waited for a full LDS round trip.
            f32x16 st[2], dp[2];
            bf16x8 fk[2][2], fv[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { fk[0][j] = read_frag(kimg, geo, j, 0, half); fv[0][j] = read_frag(vimg, geo, j, 0, half); }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        fk[(ks + 1) & 1][j] = read_frag(kimg, geo, j, ks + 1, half);
                        fv[(ks + 1) & 1][j] = read_frag(vimg, geo, j, ks + 1, half);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    st[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[ks & 1][j], qf[ks], ks == 0 ? neglse : st[j], 0, 0, 0);
                    dp[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv[ks & 1][j], dof[ks], ks == 0 ? negdel : dp[j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const int kbase = P.kbeg + kt * KT;
            if (kbase + KT > L) {
                asm volatile("");
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kbase + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half >= L) st[j][r] = -INFINITY;
            }
            if (MASK == 1) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (!((gbits >> group_of(P, kbase + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half)) & 1u)) st[j][r] = -INFINITY;
            } else if (MASK == 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kbase + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (key < L && dense_row[key] == 0) st[j][r] = -INFINITY;
                    }
            }
            bf16x8 dsf[4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[j][r] = __builtin_amdgcn_exp2f(st[j][r]) * dp[j][r];      // dS = P (dP - D)
                dsf[j * 2] = pack_frag(st[j], 0);
                dsf[j * 2 + 1] = pack_frag(st[j], 1);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_frag_t(ktimg, geo, s, dt), dsf[s], acc[dt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef DQ_DMA
    if (qok) {
        bf16_t* dq = P.dq0 + (qseg0 ? qrow * P.ld_d0 : P.dseg1_delta + qrow * P.ld_d1) + hoff;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[dt][rg * 4 + j] * P.scale;
                *(uint2*)(dq + dt * 32 + rg * 8 + half * 4) = pack4(v);
            }
    }
}

// ------------------------------------------------------------------------------------ dK, dV
// Stage = Q (K-swizzle) | Q (V-swizzle) | dO (K-swizzle) | dO (V-swizzle) | neg_lse[64] | neg_D[64]
constexpr int DKV_STAGE = 4 * IMG + 512;
template <int MASK>
__global__ void __launch_bounds__(256, 2)
attn_dkv_kernel(const AttnParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* __restrict__ rowidx = (int32_t*)(smem + 2 * DKV_STAGE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;

    uint32_t id = (uint32_t)xcd_remap(blockIdx.x, P.n_problems * P.heads * P.nqb);
    const uint32_t id1 = fdiv(id, P.fd_nqb);
    const int kb = (int)(id - id1 * P.fd_nqb.d);
    const int prob = (int)fdiv(id1, P.fd_heads);
    const int head = (int)(id1 - (uint32_t)prob * P.fd_heads.d);
    const int L = P.L, L0 = P.L0;
    const int64_t hoff = (int64_t)head * 64;
    {
        const int64_t base0 = seg0_base(P.rm, prob);
        for (int l = tid; l < L; l += 256) rowidx[l] = (int32_t)token_row(P, base0, prob, l);
    }
    __syncthreads();

    const int lk = P.kbeg + kb * 128 + wave * 32 + l31;
    const bool kok = lk < L;
    const int lkc = kok ? lk : L - 1;
    const bool wave_active = P.kbeg + kb * 128 + wave * 32 < L;
    const int QE = P.qend;                       // queries are tokens [0, qend)
    const int64_t krow = rowidx[lkc];
    const bool kseg0 = lkc < L0;
    bf16x8 kf[4], vf[4];
    {
        const int64_t off = (kseg0 ? krow * P.ld0 : P.seg1_delta + krow * P.ld1) + hoff;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = scale_frag(*(const bf16x8*)(P.k0 + off + ks * 16 + half * 8), P.scale_log2);
            vf[ks] = *(const bf16x8*)(P.v0 + off + ks * 16 + half * 8);
        }
    }
    f32x16 acck[2], accv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acck[0][r] = acck[1][r] = accv[0][r] = accv[1][r] = 0.f; }
    uint32_t kbits = 0xffffffffu;
    if (MASK == 1) kbits = group_bits(P, prob, lkc, true);

    const Geo geo = make_geo(lane);
    const int nqt = (QE + KT - 1) / KT;
    const int srow0 = wave * 16 + (lane >> 3), srow1 = srow0 + 8;
    const int kc0 = ((lane & 7) ^ ((srow0 >> 1) & 7)) << 3, kc1 = ((lane & 7) ^ ((srow1 >> 1) & 7)) << 3;
    const int vc0 = ((lane & 7) ^ (((srow0 >> 1) & 1) << 2)) << 3, vc1 = ((lane & 7) ^ (((srow1 >> 1) & 1) << 2)) << 3;
    const int sdst = wave * 2048;
    const uint32_t lds0 = lds_address(smem);
    const int64_t stat0 = ((int64_t)prob * P.heads + head) * L;
#define DKV_DMA(qt_, stage_)                                                                 \
    do {                                                                                     \
        const int qb_ = (qt_) * KT;                                                          \
        const int ra_ = qb_ + srow0 < QE ? qb_ + srow0 : QE - 1;                             \
        const int rb_ = qb_ + srow1 < QE ? qb_ + srow1 : QE - 1;                             \
        const int64_t ia_ = rowidx[ra_], ib_ = rowidx[rb_];                                  \
        const int64_t qa_ = (ra_ < L0 ? ia_ * P.ld0 : P.seg1_delta + ia_ * P.ld1) + hoff;   \
        const int64_t qb2_ = (rb_ < L0 ? ib_ * P.ld0 : P.seg1_delta + ib_ * P.ld1) + hoff;  \
        const int64_t da_ = (ra_ < L0 ? ia_ * P.ldo0 : P.doseg1_delta + ia_ * P.ldo1) + hoff; \
        const int64_t db_ = (rb_ < L0 ? ib_ * P.ldo0 : P.doseg1_delta + ib_ * P.ldo1) + hoff; \
        const uint32_t l_ = lds0 + (stage_) * DKV_STAGE + sdst;                              \
        dma16(P.q0 + qa_ + kc0, l_);               dma16(P.q0 + qb2_ + kc1, l_ + 1024);      \
        dma16(P.q0 + qa_ + vc0, l_ + IMG);         dma16(P.q0 + qb2_ + vc1, l_ + IMG + 1024); \
        dma16(P.do0 + da_ + kc0, l_ + 2 * IMG);    dma16(P.do0 + db_ + kc1, l_ + 2 * IMG + 1024); \
        dma16(P.do0 + da_ + vc0, l_ + 3 * IMG);    dma16(P.do0 + db_ + vc1, l_ + 3 * IMG + 1024); \
        if (wave < 2) {                                                                      \
            const int qi_ = qb_ + lane < QE ? qb_ + lane : QE - 1;                           \
            dma4((wave == 0 ? P.lse : P.delta) + stat0 + qi_, lds0 + (stage_) * DKV_STAGE + 4 * IMG + wave * 256); \
        }                                                                                    \
    } while (0)

    DKV_DMA(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the key / value fragments count as used HERE: left to their first use inside the loop, the compiler's waits for these
    // plain loads (which it counts; the LDS-DMA requests above it does not) land behind the requests of the next stage and
    // drain them in the middle of every tile
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(kf[ks]), "+v"(vf[ks]));
    __syncthreads();
    for (int qt = 0; qt < nqt; ++qt) {
        if (qt + 1 < nqt) DKV_DMA(qt + 1, (qt + 1) & 1);
        const char* qimg = smem + (qt & 1) * DKV_STAGE;
        const char* qtimg = qimg + IMG;
        const char* doimg = qimg + 2 * IMG;
        const char* dotimg = qimg + 3 * IMG;
        const float* nlse = (const float*)(qimg + 4 * IMG);
        const float* ndel = nlse + 64;
        if (wave_active) {
            f32x16 st[2], dp[2];
            constexpr bool kPipe = MASK != 2;       // (the dense-mask form is at 256 registers without the second fragment set)
            bf16x8 fq[2][2], fo[2][2];
            if constexpr (kPipe) {
#pragma unroll
                for (int j = 0; j < 2; ++j) { fq[0][j] = read_frag(qimg, geo, j, 0, half); fo[0][j] = read_frag(doimg, geo, j, 0, half); }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // C-init: register r belongs to query j*32 + (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 a = *(const float4*)(nlse + j * 32 + 8 * g + 4 * half);
                    const float4 b = *(const float4*)(ndel + j * 32 + 8 * g + 4 * half);
                    st[j][4 * g] = a.x; st[j][4 * g + 1] = a.y; st[j][4 * g + 2] = a.z; st[j][4 * g + 3] = a.w;
                    dp[j][4 * g] = b.x; dp[j][4 * g + 1] = b.y; dp[j][4 * g + 2] = b.z; dp[j][4 * g + 3] = b.w;
                }
            }
            // S^T and dP^T: fragments of MFMA step ks + 1 read under the MFMAs of step ks (see attn_dq_kernel)
            if constexpr (kPipe) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks < 3) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            fq[(ks + 1) & 1][j] = read_frag(qimg, geo, j, ks + 1, half);
                            fo[(ks + 1) & 1][j] = read_frag(doimg, geo, j, ks + 1, half);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        st[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[ks & 1][j], kf[ks], st[j], 0, 0, 0);
                        dp[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo[ks & 1][j], vf[ks], dp[j], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        st[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_frag(qimg, geo, j, ks, half), kf[ks], st[j], 0, 0, 0);
                        dp[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_frag(doimg, geo, j, ks, half), vf[ks], dp[j], 0, 0, 0);
                    }
            }
            const int qbase = qt * KT;
            if (qbase + KT > QE) {
                asm volatile("");
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (qbase + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half >= QE) st[j][r] = -INFINITY;
            }
            if (MASK == 1) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (!((kbits >> group_of(P, qbase + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half)) & 1u)) st[j][r] = -INFINITY;
            } else if (MASK == 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int q = qbase + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (q < L && P.mask[((int64_t)prob * L + q) * L + lkc] == 0) st[j][r] = -INFINITY;
                    }
            }
            bf16x8 pf[4], dsf[4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[j][r] = __builtin_amdgcn_exp2f(st[j][r]);          // P
                    dp[j][r] *= st[j][r];                                 // dS = P (dP - D)
                }
                pf[j * 2] = pack_frag(st[j], 0);  pf[j * 2 + 1] = pack_frag(st[j], 1);
                dsf[j * 2] = pack_frag(dp[j], 0); dsf[j * 2 + 1] = pack_frag(dp[j], 1);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    accv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_frag_t(dotimg, geo, s, dt), pf[s], accv[dt], 0, 0, 0);
                    acck[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_frag_t(qtimg, geo, s, dt), dsf[s], acck[dt], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef DKV_DMA
    if (kok) {
        const int64_t off = (kseg0 ? krow * P.ld_d0 : P.dseg1_delta + krow * P.ld_d1) + hoff;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float a[4], b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { a[j] = acck[dt][rg * 4 + j] * P.scale; b[j] = accv[dt][rg * 4 + j]; }
                *(uint2*)(P.dk0 + off + dt * 32 + rg * 8 + half * 4) = pack4(a);
                *(uint2*)(P.dv0 + off + dt * 32 + rg * 8 + half * 4) = pack4(b);
            }
    }
}

template <int MASK>
int launch_bwd(const AttnParams& P, hipStream_t s) {
    // dq: one workgroup per 128 queries of [0, qend); dk / dv: one per 128 keys of [kbeg, L)
    AttnParams Pk = P;
    Pk.nqb = (int)((P.L - P.kbeg + 127) / 128);
    Pk.fd_nqb = make_fastdiv((uint32_t)Pk.nqb);
    const int64_t nblk = (int64_t)P.n_problems * P.heads * P.nqb, nblk_k = (int64_t)P.n_problems * P.heads * Pk.nqb;
    const size_t tab = (size_t)((P.L + 3) & ~3) * sizeof(int32_t);
    const size_t lds_q = 2 * DQ_STAGE + tab, lds_kv = 2 * DKV_STAGE + tab;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_dq_kernel<MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_dkv_kernel<MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        attr_set = true;
    }
    if (lds_kv > 128 * 1024) return DWM_EUNSUPPORTED;
    hipLaunchKernelGGL(delta_kernel, dim3((unsigned)(((int64_t)P.n_problems * P.L + 3) / 4)), dim3(256), 0, s, P);
    hipLaunchKernelGGL((attn_dq_kernel<MASK>), dim3((unsigned)nblk), dim3(256), lds_q, s, P);
    hipLaunchKernelGGL((attn_dkv_kernel<MASK>), dim3((unsigned)nblk_k), dim3(256), lds_kv, s, Pk);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DWM_OK : (int)e;
}

}  // namespace

extern "C" int dwm_attention_bwd(const dwm_attn_bwd_args* b, void* stream) {
    if (b == nullptr) return DWM_EINVAL;
    AttnParams P;
    const int rc = fill_params(&b->fwd, P);
    if (rc != DWM_OK) return rc;
    if (b->fwd.lse == nullptr || b->delta == nullptr || b->do0 == nullptr || b->dq0 == nullptr || b->dk0 == nullptr || b->dv0 == nullptr)
        return DWM_EINVAL;
    if (P.ldo0 % 8 != 0 || b->ld_d0 % 8 != 0 || !dwm_aligned16(b->do0) || !dwm_aligned16(b->dq0) || !dwm_aligned16(b->dk0) ||
        !dwm_aligned16(b->dv0) || !dwm_aligned16(P.o0))
        return DWM_EALIGN;
    P.do0 = (const bf16_t*)b->do0; P.do1 = (const bf16_t*)b->do1;
    P.dq0 = (bf16_t*)b->dq0; P.dk0 = (bf16_t*)b->dk0; P.dv0 = (bf16_t*)b->dv0;
    P.ld_d0 = b->ld_d0; P.ld_d1 = b->ld_d1;
    P.dseg1_delta = 0;
    P.doseg1_delta = 0;
    if (P.L1 > 0) {
        if (b->do1 == nullptr || b->dq1 == nullptr || b->dk1 == nullptr || b->dv1 == nullptr) return DWM_EINVAL;
        if (b->fwd.cross && (P.mask_mode != 0)) return DWM_EUNSUPPORTED;
        if ((!b->fwd.cross && P.ldo1 % 8 != 0) || b->ld_d1 % 8 != 0 || !dwm_aligned16(b->do1) || !dwm_aligned16(b->dq1) || !dwm_aligned16(P.o1)) return DWM_EALIGN;
        const int64_t dq = (const bf16_t*)b->dq1 - P.dq0, dk = (const bf16_t*)b->dk1 - P.dk0, dv = (const bf16_t*)b->dv1 - P.dv0;
        if (dq != dk || dk != dv) return DWM_EUNSUPPORTED;
        P.dseg1_delta = dq;
        P.doseg1_delta = P.do1 - P.do0;
    }
    P.delta = b->delta;
    P.nqb = (int)((P.qend + 127) / 128);
    P.fd_nqb = make_fastdiv((uint32_t)P.nqb);
    if ((int64_t)P.n_problems * P.heads * P.nqb >= (1ll << 31)) return DWM_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (P.mask_mode == 0) return launch_bwd<0>(P, s);
    if (P.mask_mode == 1) return launch_bwd<1>(P, s);
    return launch_bwd<2>(P, s);
}
