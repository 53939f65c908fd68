// gemm4p_kernel: the RESID epilogue on the fp32 residual stream, moved UNDER a main loop.
//
//     C32[M, N] (fp32, in place over `res` allowed) = res + gate * (A[M, K] . W[N, K]^T + bias)        (gate optional)
//
// Why it exists.  On gemm4w_kernel (gemm_bf16_4w.hip) this launch at N = K = 1536 - the attention out-projection of every joint / VT
// block, 96 launches per denoise step - runs at 0.69 of the library's plain GEMM (profiles/r6_library_gemm_comparison.log: 885-921
// against 1327 TFLOP/s): its epilogue moves 512 KiB of fp32 stream per 256 x 256 tile through one CU at ~30 GB/s (a per-CU cap on bytes
// in flight over the HBM round trip: 256 CUs x 30 GB/s IS the HBM rate), 22 us behind a 38 us main loop, and all 256 workgroups do so at
// the same time (scripts/experiments/gemm4w_resid_prefetch/README.md: seven experiments, nothing inside the epilogue helps).  The stream
// traffic has to travel while the matrix pipes work.
//
// How.  Persistent workgroups (one per CU) walk 256 x 128 HALF tiles: 4 waves x 128 x 64, v_mfma_f32_16x16x32_bf16, 128 accumulator
// registers per wave - so TWO accumulator sets fit the 256 AGPRs.  While half tile h accumulates into set h & 1, the epilogue of half
// tile h - 1 drains set (h - 1) & 1: its 8 row passes are spread over the K steps of h's main loop as side instructions of the MFMA
// stream - pass p requests its residual / gate rows in K step 3 p, transposes its accumulators through 4 KiB of wave-private LDS in
// step 3 p + 1 and adds / stores in step 3 p + 2 (K >= 1536: 24 K steps).  Loads, stores and the LDS round trip all sit between MFMAs
// of the next half tile; only a workgroup's last half tile drains with nothing to hide under.
// LDS: A ring 3 x 32 KiB + W ring 2 x 16 KiB (the images, swizzles and counted waits of gemm_bf16_4w.hip) + 4 x 4 KiB transposes = 144 KiB.
// The two half tiles of a 256 x 256 tile run back to back on one workgroup (the A rows come from the L2 the second time); tiles are
// taken in gemm4w_kernel's XCD-aware raster, virtual block b + 256 i.
// Covered: EPI_RESID on the fp32 stream without a bf16 copy, residual or gate + residual, no row maps / taps / split-K,
// M % 256 == N % 256 == 0, K % 64 == 0, K >= 1536.  Everything else stays where it was.
#include <atomic>

#include "common.h"
#include "dwm_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 hbf16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int BM = 256, BNH = 128, BK = 64;
constexpr int TILE_A = BM * BK * 2;               // 32 KiB
constexpr int TILE_W = BNH * BK * 2;              // 16 KiB
constexpr int AST = 3, WST = 2;
constexpr int W_BASE = AST * TILE_A;              // 96 KiB
constexpr int SCR_BASE = W_BASE + WST * TILE_W;   // 128 KiB
constexpr int LDS_BYTES = SCR_BASE + 4 * 4096;    // 144 KiB
constexpr int NJA = 8, NJW = 4;                   // 1-KiB requests per wave and stage: A, W

struct G4PParams {
    int ntm, ntn, gm;                             // 256 x 256 tiles (two half tiles each), raster group
    int nt;                                       // tiles in all
    FastDiv fd_rpg;
};

template <int RS>                                 // 2: residual, 3: gate + residual
__global__ void __launch_bounds__(256, 1)
gemm4p_kernel(const dwm_gemm_args p, const G4PParams gp) {
    static_assert(RS == 2 || RS == 3, "residual / gate + residual");
    constexpr bool f_gate = (RS & 1) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    const int64_t K = p.K;
    const int nk = (int)(K / BK);
    const bf16_t* __restrict__ Ap = (const bf16_t*)p.A;
    const bf16_t* __restrict__ Wp = (const bf16_t*)p.W;
    auto make_rsrc = [](const void* ptr) {
        const uint64_t a = (uint64_t)ptr;
        i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
        r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu));
        r[2] = -1;
        r[3] = 0x00020000;
        return r;
    };
    // LDS-DMA requests (gemm_bf16_4w.hip): request j of this wave fills rows (wave * NJ + j) * 8 .. + 8 of a tile
    const int row0a = wave * (NJA * 8) + (lane >> 3), row0w = wave * (NJW * 8) + (lane >> 3);
    const uint32_t voff_a = (uint32_t)row0a * (uint32_t)(p.lda * 2) + (uint32_t)((lane & 7) ^ ((row0a >> 1) & 7)) * 16u;
    const uint32_t voff_w = (uint32_t)row0w * (uint32_t)(K * 2) + (uint32_t)((lane & 7) ^ ((row0w >> 1) & 7)) * 16u;
    const uint32_t step_a = 8u * (uint32_t)(p.lda * 2), step_w = 8u * (uint32_t)(K * 2);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    auto bufld = [&](const i32x4& rs, uint32_t vo, uint32_t so) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, 0x400" ::"v"(vo), "s"(rs), "s"(so) : "memory", "m0");
    };
    auto set_m0 = [&](uint32_t lds_off) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds0 + lds_off) : "memory", "m0"); };
    i32x4 rs_a, rs_w;
    auto stage_a = [&](int buf, int kt, int j) {
        if (j == 0) set_m0((uint32_t)(buf * TILE_A + wave * NJA * 1024));
        bufld(rs_a, (j & 1) ? (voff_a ^ 64u) : voff_a, (uint32_t)kt * (BK * 2) + (uint32_t)j * step_a);
    };
    auto stage_w = [&](int buf, int kt, int j) {
        if (j == 0) set_m0((uint32_t)(W_BASE + buf * TILE_W + wave * NJW * 1024));
        bufld(rs_w, (j & 1) ? (voff_w ^ 64u) : voff_w, (uint32_t)kt * (BK * 2) + (uint32_t)j * step_w);
    };
    // fragment reads: 16 rows x 32 k; lane = row l15, 16-byte chunk (4 kh + lg) ^ ((row >> 1) & 7)
    const int swz = (l15 >> 1) & 7;
    const int a_row = (wm * 128 + l15) * 128, w_row = (wn * 64 + l15) * 128;
    int coff[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) coff[kh] = ((4 * kh + lg) ^ swz) << 4;

    f32x4 acc[2][8][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[s][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                asm volatile("" : "+a"(acc[s][i][j]));
            }
    hbf16x8 af[2][8], wf[2][4];

    // ---- epilogue of the PREVIOUS half tile (accumulator set ES), spread over the current one's K steps
    char* const scr = smem + SCR_BASE + wave * 4096;           // this wave's transpose image: 16 rows x 256 B, chunks XOR-swizzled by the row
    const int rrow = lane >> 3, rc8 = lane & 7;               // row-major side: 8 lanes per row (8 columns each), 8 rows per step, 2 steps per pass
    int64_t em0 = 0, en0 = 0;                                 // the previous half tile
    bool have_prev = false;
    float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint4 gA[2], rA[2], rB[2];
    float xv[2][8];
    // role 0: bias (pass 0) and the operand rows of pass ep
    auto epi_load = [&](int ep) {
        const uint32_t ocol = (uint32_t)(en0 + wn * 64 + rc8 * 8);
        if (ep == 0 && p.bias != nullptr) unpack8(*(const uint4*)((const bf16_t*)p.bias + ocol), b8);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const uint32_t m = (uint32_t)(em0 + wm * 128 + ep * 16 + st * 8 + rrow);
            if constexpr (f_gate) gA[st] = *(const uint4*)((const bf16_t*)p.gate + ((uint64_t)fdiv(m, gp.fd_rpg) * (uint32_t)p.ld_gate + ocol));
            const float* rp = (const float*)p.res + ((uint64_t)m * (uint32_t)p.ld_res + ocol);
            rA[st] = *(const uint4*)rp;
            rB[st] = *(const uint4*)(rp + 4);
        }
    };
    // role 1: this pass's 16 rows of the drained set into the transpose image (raw fp32 sums: 16 bytes at chunk 4 j + lg)
    auto epi_write = [&](auto es_tag, int ep, int j) {
        constexpr int ES = decltype(es_tag)::value;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (ep == i) v = acc[ES][i][j];
        *(float4*)(scr + l15 * 256 + (((4 * j + lg) ^ l15) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
    };
    // role 2: row-major reads, bias / gate / residual, stores - in pieces of one step each
    auto epi_read = [&](int st) {
        const int r = st * 8 + rrow;
        const float4 x0 = *(const float4*)(scr + r * 256 + (((2 * rc8) ^ r) << 4));
        const float4 x1 = *(const float4*)(scr + r * 256 + (((2 * rc8 + 1) ^ r) << 4));
        xv[st][0] = x0.x; xv[st][1] = x0.y; xv[st][2] = x0.z; xv[st][3] = x0.w;
        xv[st][4] = x1.x; xv[st][5] = x1.y; xv[st][6] = x1.z; xv[st][7] = x1.w;
    };
    auto epi_store = [&](int ep, int st) {
        float t[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) xv[st][c] += b8[c];
        if constexpr (f_gate) {
            unpack8(gA[st], t);
#pragma unroll
            for (int c = 0; c < 8; ++c) xv[st][c] *= t[c];
        }
        const float4 ta = *reinterpret_cast<const float4*>(&rA[st]), tb = *reinterpret_cast<const float4*>(&rB[st]);
        const uint32_t m = (uint32_t)(em0 + wm * 128 + ep * 16 + st * 8 + rrow);
        float* o32 = (float*)p.C32 + ((uint64_t)m * (uint32_t)p.ldc32 + (uint32_t)(en0 + wn * 64 + rc8 * 8));
        *(float4*)o32 = make_float4(xv[st][0] + ta.x, xv[st][1] + ta.y, xv[st][2] + ta.z, xv[st][3] + ta.w);
        *(float4*)(o32 + 4) = make_float4(xv[st][4] + tb.x, xv[st][5] + tb.y, xv[st][6] + tb.z, xv[st][7] + tb.w);
    };

    // ---- one K step of the current half tile (accumulator set S): gemm4w_kernel's, with 4 column blocks per wave and - ROLE > 0 - a
    //      slice of the previous half tile's epilogue as side instructions.  MODE 0: steady state; 1: second-to-last step (nothing
    //      requested); 2: last step.  FIRST: kt == 0 (the accumulators start from C = 0).
    int sa = 0, kt = 0;
    auto k_step = [&](auto s_tag, auto mode_tag, auto role_tag, auto first_tag, int ep) {
        constexpr int S = decltype(s_tag)::value, MODE = decltype(mode_tag)::value, ROLE = decltype(role_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        const int sa1 = sa == AST - 1 ? 0 : sa + 1, sa2 = sa1 == AST - 1 ? 0 : sa1 + 1;
        const char* la = smem + sa * TILE_A;
        const char* lw = smem + W_BASE + (kt & 1) * TILE_W;
        const char* lan = smem + sa1 * TILE_A;
        const char* lwn = smem + W_BASE + ((kt + 1) & 1) * TILE_W;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            if (kh == 1 && MODE != 2) {
                if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NJA) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                    acc[S][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kh][j], af[kh][i], FIRST && kh == 0 ? zero4 : acc[S][i][j], 0, 0, 0);
                    const int idx = i * 4 + j;
                    // side instructions, at most one group per MFMA slot: the 12 fragment reads of the next half (slots 1..12), this half's
                    // requests (even slots from 14), the epilogue slice of ROLE (odd slots from 13 / the slots behind the requests)
                    if (!(MODE == 2 && kh == 1)) {
                        const char* fa = kh == 0 ? la : lan;
                        const char* fw = kh == 0 ? lw : lwn;
                        const int khn = kh ^ 1;
                        if (idx >= 1 && idx <= 4) wf[khn][idx - 1] = *(const hbf16x8*)(fw + w_row + (idx - 1) * 2048 + coff[khn]);
                        if (idx >= 5 && idx <= 12) af[khn][idx - 5] = *(const hbf16x8*)(fa + a_row + (idx - 5) * 2048 + coff[khn]);
                    }
                    if (MODE == 0 && idx >= 14 && (idx & 1) == 0) {
                        const int r = (idx - 14) >> 1;
                        if (kh == 0 && r < NJA) stage_a(sa2, kt + 2, r);
                        if (kh == 1 && r < NJW) stage_w(kt & 1, kt + 2, r);
                    }
                    if constexpr (ROLE == 1) {                       // operand rows of pass ep: second half, behind its four W requests
                        if (kh == 1 && idx == 23) epi_load(ep);
                    } else if constexpr (ROLE == 2) {                // accumulators of pass ep into the transpose image
                        if (kh == 1 && idx >= 23 && idx <= 29 && (idx & 1) == 1) epi_write(s_tag_other(s_tag), ep, (idx - 23) >> 1);
                    } else if constexpr (ROLE == 3) {                // row-major side
                        if (kh == 0 && idx == 13) epi_read(0);
                        if (kh == 0 && idx == 15) epi_read(1);
                        if (kh == 1 && idx == 23) epi_store(ep, 0);
                        if (kh == 1 && idx == 27) epi_store(ep, 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        sa = sa1;
        ++kt;
    };
    (void)k_step;
}

}  // namespace
