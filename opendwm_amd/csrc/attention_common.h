// Shared pieces of the attention forward / backward kernels (attention.hip, attention_bwd.hip):
// tile geometry, the row-map addressing of dwm_attn_args, device helpers and the host-side
// validation / parameter fill.
#pragma once
#include "common.h"
#include "dwm_hip.h"

namespace dwm_attn {

constexpr int KT = 64;                        // keys per tile
constexpr int K_TILE_BYTES = KT * 128;        // 8192
constexpr int V_TILE_BYTES = KT * 128;        // 8192
constexpr int STAGE_BYTES = K_TILE_BYTES + V_TILE_BYTES;
constexpr int MAX_LDS_BYTES = 96 * 1024;
constexpr int NSTAGE = 3;                     // LDS ring depth (tiles t, t+1, t+2 resident or in flight)

typedef __attribute__((ext_vector_type(4))) short s16x4;

struct RowMap {
    FastDiv pdiv[3], pmod[3];
    int64_t pstride[3];
    FastDiv ldiv0, ldiv1;
    int64_t lstride[3];
};

struct AttnParams {
    const bf16_t *q0, *k0, *v0, *q1, *k1, *v1;
    bf16_t *o0, *o1;
    int64_t ld0, ld1, ldo0, ldo1;
    int L0, L1, L;
    int qend, kbeg;          // queries are tokens [0, qend), keys tokens [kbeg, L): (L, 0), or (L0, L0) for cross-attention
    int64_t seg1_delta;      // (q1 - q0) == (k1 - k0) == (v1 - v0) in elements
    float* lse;              // optional [n_problems, heads, L]: NEGATIVE log2-domain log-sum-exp of scale*log2(e)*q.k
    // backward only
    const bf16_t *do0, *do1;             // gradient of the outputs, addressed like o0 / o1
    bf16_t *dq0, *dk0, *dv0;             // gradients, addressed like q0 / k0 / v0 with ld_d0 / ld_d1
    int64_t ld_d0, ld_d1, dseg1_delta;   // (dq1 - dq0) == (dk1 - dk0) == (dv1 - dv0)
    int64_t oseg1_delta, doseg1_delta;   // o1 - o0, do1 - do0 in elements
    int stream_far;          // attn_stream_kernel only (set by its launcher): the segment displacements are added per row, not folded into the tables
    float* delta;                        // [n_problems, heads, L] -rowsum(dO * O)
    float scale;
    int n_problems, heads, nqb;
    int nwc;                 // attn_res_kernel: compute waves
    int safe_softmax;        // attn_res_kernel: online softmax (running max) for every unit instead of the max-free fast path
    int hpb;                 // heads per workgroup (forward); fd_heads then divides by heads / hpb
    FastDiv fd_nqb, fd_heads, fd_gs, fd_G, fd_ppm;
    float scale_log2;
    int mask_mode;
    const uint8_t* mask;
    int mask_G, group_size, p_per_mask;
    float inv_group_size, inv_G;
    RowMap rm;
};

DWM_DEVINL int64_t seg0_base(const RowMap& rm, int p) {
    int64_t b = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) b += (int64_t)fmod_u(fdiv((uint32_t)p, rm.pdiv[i]), rm.pmod[i]) * rm.pstride[i];
    return b;
}
DWM_DEVINL int64_t seg0_row(const RowMap& rm, int64_t base, int l) {
    const uint32_t q0 = fdiv((uint32_t)l, rm.ldiv0), lo = (uint32_t)l - q0 * rm.ldiv0.d;
    const uint32_t hi = fdiv(q0, rm.ldiv1), mid = q0 - hi * rm.ldiv1.d;
    return base + (int64_t)lo * rm.lstride[0] + (int64_t)mid * rm.lstride[1] + (int64_t)hi * rm.lstride[2];
}
DWM_DEVINL float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));   // fmaxf() adds a canonicalising v_max per MFMA output
    return r;
}
// max over the 32 scores a lane holds for one query (two 32-key sub-tiles), combined with the
// partner lane (lane ^ 32) that holds the other keys of the same query
DWM_DEVINL float tile_max32(const f32x16& s0, const f32x16& s1) {
    float mxa = max3f(s0[0], s0[1], s0[2]), mxb = max3f(s0[3], s0[4], s0[5]);
    float mxc = max3f(s1[0], s1[1], s1[2]), mxd = max3f(s1[3], s1[4], s1[5]);
#pragma unroll
    for (int r = 6; r < 14; r += 4) {
        mxa = max3f(mxa, s0[r], s0[r + 1]);
        mxb = max3f(mxb, s0[r + 2], s0[r + 3]);
        mxc = max3f(mxc, s1[r], s1[r + 1]);
        mxd = max3f(mxd, s1[r + 2], s1[r + 3]);
    }
    mxa = max3f(mxa, s0[14], s0[15]);
    mxc = max3f(mxc, s1[14], s1[15]);
    const float mx = max3f(max3f(mxa, mxb, mxc), mxd, -INFINITY);
    return max3f(mx, __shfl_xor(mx, 32, 64), -INFINITY);
}
// 8 bf16 * c -> 8 bf16 (folds softmax scale * log2(e) into the Q fragments)
DWM_DEVINL bf16x8 scale_frag(const bf16x8& v, float c) {
    const uint4 u = *reinterpret_cast<const uint4*>(&v);
    float f[8];
    unpack8(u, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] *= c;
    const uint4 o = pack8(f);
    return *reinterpret_cast<const bf16x8*>(&o);
}


// ---- pieces shared by the resident forward kernels (attention.hip: attn_res_kernel; attention_res4.hip: attn_res4_kernel)
// One 32-key step by the textbook online softmax (running max m and sum l per lane, rescale every step): the fallback of a
// unit whose fast-path sums left the safe range.  It reads K and V from GLOBAL memory, not from the images: by the time a unit
// knows that it needs the fallback, the first sub-tiles of the images may already hold the NEXT head's rows (the refill point
// of attn_res_kernel).  K fragments are the lanes' own rows (16 B per lane); V fragments are gathered element by element in the
// MFMA A-operand layout with the key order of the P' registers.  Written for few registers and for correctness, not for speed.
struct ResGlobal {
    const bf16_t *k, *v;       // k0 / v0 + this head's column offset
    const int32_t* tab;        // row table (LDS)
    int64_t seg1_delta;
};
DWM_DEVINL void res_tile_safe(const ResGlobal& gm, int key0, int L, int L0, const bf16x8 (&qf)[4], f32x16 (&ot)[2], float& m_run, float& l_run,
                              int l31, int half) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto row_off = [&](int key) -> int64_t {
        key = key < L ? key : L - 1;
        return ((int64_t)gm.tab[key] << 3) + (key < L0 ? 0 : gm.seg1_delta);
    };
    f32x16 st;
    {
        const bf16_t* kp = gm.k + row_off(key0 + l31) + half * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kp + ks * 16), qf[ks], ks == 0 ? zero : st, 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (key0 + (r & 3) + 8 * (r >> 2) + 4 * half >= L) st[r] = -INFINITY;
        mx = fmaxf(mx, st[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);                  // finite: the first step of a sequence holds key 0
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // 0 at the first step (m_run = -inf)
    m_run = m_new;
    float pv[16], sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        pv[r] = __builtin_amdgcn_exp2f(st[r] - m_new);
        sum += pv[r];
    }
    l_run = l_run * alpha + sum;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const uint4 pk = pack8(pv + 8 * s2);
        const bf16x8 pf = *reinterpret_cast<const bf16x8*>(&pk);
        // B-operand k-slot (half, e) of pf holds P' of key (e & 3) + 8 ((8 s2 + e) >> 2) + 4 half: the A operand takes the same keys
        int64_t voff[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) voff[e] = row_off(key0 + (e & 3) + 8 * ((8 * s2 + e) >> 2) + 4 * half);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            bf16x8 vf;
#pragma unroll
            for (int e = 0; e < 8; ++e) vf[e] = (short)gm.v[voff[e] + dt * 32 + l31];
            ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, ot[dt], 0, 0, 0);
        }
    }
}

struct ResCtx {             // launch / workgroup invariants of attn_res_kernel's helpers
    const char *kimg, *vimg;
    const int32_t* rowtab;
    int L, L0, nsub, l31, half, kswz;     // nsub: 32-key sub-tiles of the sequence (the last one may be ragged)
    int vra[2], vrb[2];
};


// ---- host side: validate a dwm_attn_args and fill the launch-invariant parameter block
inline int fill_params(const dwm_attn_args* a, AttnParams& P) {
    if (a == nullptr || a->q0 == nullptr || a->k0 == nullptr || a->v0 == nullptr || a->o0 == nullptr) return DWM_EINVAL;
    if (a->head_dim != 64) return DWM_EUNSUPPORTED;
    if (a->L0 <= 0 || a->L1 < 0 || a->n_problems <= 0 || a->heads <= 0) return DWM_EINVAL;
    if (a->L1 > 0 && (a->q1 == nullptr || a->k1 == nullptr || a->v1 == nullptr || (a->o1 == nullptr && !a->cross))) return DWM_EINVAL;
    if (a->cross && a->L1 <= 0) return DWM_EINVAL;
    if (a->ld0 % 8 != 0 || a->ldo0 % 8 != 0 || (a->L1 > 0 && (a->ld1 % 8 != 0 || (!a->cross && a->ldo1 % 8 != 0)))) return DWM_EALIGN;
    if (!dwm_aligned16(a->q0) || !dwm_aligned16(a->k0) || !dwm_aligned16(a->v0) || !dwm_aligned16(a->o0)) return DWM_EALIGN;
    if (a->L1 > 0 && (!dwm_aligned16(a->q1) || !dwm_aligned16(a->k1) || !dwm_aligned16(a->v1) || (!a->cross && !dwm_aligned16(a->o1))))
        return DWM_EALIGN;
    if (a->ldiv[0] <= 0 || a->ldiv[1] <= 0) return DWM_EINVAL;
    if (a->mask_mode < 0 || a->mask_mode > 2) return DWM_EINVAL;
    if (a->mask_mode != 0 && a->mask == nullptr) return DWM_EINVAL;
    if (a->mask_mode == 1 && (a->mask_G <= 0 || a->mask_G > 32 || a->group_size <= 0 || a->p_per_mask <= 0)) return DWM_EINVAL;
    const int64_t L = a->L0 + a->L1;
    if (L >= (1 << 22) || a->n_problems >= (1ll << 30)) return DWM_EUNSUPPORTED;

    P = AttnParams();
    P.q0 = (const bf16_t*)a->q0; P.k0 = (const bf16_t*)a->k0; P.v0 = (const bf16_t*)a->v0;
    P.q1 = (const bf16_t*)a->q1; P.k1 = (const bf16_t*)a->k1; P.v1 = (const bf16_t*)a->v1;
    P.o0 = (bf16_t*)a->o0; P.o1 = (bf16_t*)a->o1;
    P.ld0 = a->ld0; P.ld1 = a->ld1; P.ldo0 = a->ldo0; P.ldo1 = a->ldo1;
    P.L0 = (int)a->L0; P.L1 = (int)a->L1; P.L = (int)L;
    P.qend = a->cross ? P.L0 : P.L;
    P.kbeg = a->cross ? P.L0 : 0;
    P.seg1_delta = 0;
    P.oseg1_delta = 0;
    if (a->L1 > 0) {
        // the kernels address both segments through one offset table relative to q0/k0/v0
        const int64_t dq = P.q1 - P.q0, dk = P.k1 - P.k0, dv = P.v1 - P.v0;
        if (dq != dk || dk != dv) return DWM_EUNSUPPORTED;
        P.seg1_delta = dq;
        P.oseg1_delta = a->cross ? 0 : P.o1 - P.o0;
    }
    P.lse = a->lse;
    P.n_problems = (int)a->n_problems; P.heads = a->heads;
    P.scale = a->scale;
    P.scale_log2 = a->scale * 1.4426950408889634f;
    if ((a->variant >> 15) & 1) {         // Q arrives with scale * log2(e) folded in by its producer: the scores are log2-domain already
        P.scale_log2 = 1.f;
        P.scale = 0.6931471805599453f;
    }
    P.mask_mode = a->mask_mode; P.mask = a->mask;
    P.mask_G = (int)a->mask_G; P.group_size = (int)a->group_size; P.p_per_mask = (int)a->p_per_mask;
    P.inv_group_size = a->group_size > 0 ? 1.f / (float)a->group_size : 0.f;
    P.inv_G = a->mask_G > 0 ? 1.f / (float)a->mask_G : 0.f;
    for (int i = 0; i < 3; ++i) {
        if (a->pdiv[i] <= 0 || a->pmod[i] <= 0 || a->pdiv[i] > (1ll << 30) || a->pmod[i] > (1ll << 30)) return DWM_EINVAL;
        P.rm.pdiv[i] = make_fastdiv((uint32_t)a->pdiv[i]); P.rm.pmod[i] = make_fastdiv((uint32_t)a->pmod[i]);
        P.rm.pstride[i] = a->pstride[i];
    }
    if (a->ldiv[0] > (1ll << 30) || a->ldiv[1] > (1ll << 30)) return DWM_EINVAL;
    P.rm.ldiv0 = make_fastdiv((uint32_t)a->ldiv[0]); P.rm.ldiv1 = make_fastdiv((uint32_t)a->ldiv[1]);
    for (int i = 0; i < 3; ++i) P.rm.lstride[i] = a->lstride[i];
    P.fd_heads = make_fastdiv((uint32_t)P.heads);
    P.fd_gs = make_fastdiv((uint32_t)(P.group_size > 0 ? P.group_size : 1));
    P.fd_G = make_fastdiv((uint32_t)(P.mask_G > 0 ? P.mask_G : 1));
    P.fd_ppm = make_fastdiv((uint32_t)(P.p_per_mask > 0 ? P.p_per_mask : 1));
    return DWM_OK;
}

}  // namespace dwm_attn
