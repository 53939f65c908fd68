// EXPERIMENT (not part of the product library): the 256 x 256 x 64 bf16 GEMM main loop in the geometry of hipBLASLt's gfx950
// kernel for these shapes (Custom_Cijk_Alik_Bljk_..._MT256x256x64_MI16x16x1, disassembled from the ROCm install: 4 waves, one per
// SIMD, 128 x 128 per wave, v_mfma_f32_16x16x32_bf16, 256 accumulator registers in AGPRs, 16 fragment reads + 8 LDS-DMA requests
// per 64 MFMAs) - it runs the bench's GEMM shapes 10-20 % faster than gemm_bf16.hip's 8-wave loop (profiles/r4n_*).  This file
// keeps gemm_bf16.hip's LDS image ([256 rows][8 x 16-B chunks], chunk ^ ((row >> 1) & 7)), its 3 + 2 stage LDS-DMA ring and its
// counted waits, and changes the wave geometry and the MFMA shape.  PLAIN epilogue only (C = A W^T in bf16, stored straight from
// the accumulators), M % 256 == N % 256 == 0, K % 64 == 0.      build + run: scripts/experiments/gemm4w/run.py
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define DEVINL __device__ __forceinline__

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE = BM * BK * 2;                 // 32 KiB
constexpr int AST = 3, WST = 2;
constexpr int W_BASE = AST * TILE;
constexpr int LDS_BYTES = (AST + WST) * TILE;     // 160 KiB
constexpr int NJ = TILE / (4 * 1024);             // 1-KiB requests per wave, operand and stage: 8

DEVINL int xcd_remap(int bid, int nblk) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
DEVINL uint32_t pack2(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// BUF: the LDS-DMA requests as `buffer_load_dwordx4 v, s[rsrc], s_off offen lds` - one VGPR offset per lane for all requests of an
// operand, the request's row block in an SGPR offset, the K walk on the descriptor's base, M0 bumped by 1024 behind every request:
// 2 scalar-side instructions per request and no vector ones (the global_load_lds form: 64-bit vector add + s_add + s_mov m0 + s_nop)
typedef __attribute__((ext_vector_type(4))) int i32x4;
// TEPI: the output tile leaves through wave-private LDS (4 KiB per 16-row pass, 16-byte chunks XOR-swizzled by the row) so that every
// store instruction writes whole 256-byte row pieces (16 bytes per lane); otherwise 8 bytes per lane straight from the accumulators
template <bool BUF, bool TEPI>
__global__ void __launch_bounds__(256, 1)
gemm4w_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* __restrict__ C, int M, int N, int K, int ntm, int ntn, int gm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;                 // fragment row, 16-byte k chunk of a 32-wide K half

    // tile rasterisation of gemm_bf16.hip: XCD-contiguous ids, groups of gm row tiles x all column tiles
    int id = xcd_remap(blockIdx.x, ntm * ntn);
    const int per_group = gm * ntn;
    const int grp = id / per_group, in_grp = id - grp * per_group;
    const int first_m = grp * gm;
    const int gsize = ntm - first_m < gm ? ntm - first_m : gm;
    const int tn = in_grp / gsize, tm = first_m + in_grp % gsize;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    // staging: request j of this wave fills rows (wave*NJ + j)*8 .. +8 of a tile; 8 lanes per row; the chunk swizzle of the
    // image is applied on the source column (the LDS destination of a request is lane-linear)
    const char* a_src[NJ];
    const char* w_src[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int row = (wave * NJ + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        a_src[j] = (const char*)(A + (m0 + row) * (int64_t)K + c * 8);
        w_src[j] = (const char*)(W + (n0 + row) * (int64_t)K + c * 8);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    auto glds = [&](const char* src, uint32_t lds_off) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds0 + lds_off) : "memory", "m0");
    };
    // BUF form: descriptors of this tile's A / W row blocks (base = first row of the tile; 2^32 - 1 bytes: no range check wanted),
    // per-lane offset of request 0 (row = wave * 64 + lane / 8), request j adds j * 8 rows
    auto make_rsrc = [](const void* p) {
        const uint64_t a = (uint64_t)p;
        i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
        r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu));
        r[2] = -1;
        r[3] = 0x00020000;
        return r;
    };
    const i32x4 rs_a = make_rsrc(A + m0 * (int64_t)K), rs_w = make_rsrc(W + n0 * (int64_t)K);
    uint32_t voff;
    {
        const int row = wave * NJ * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);         // (the rows of request j are 8 j further: (row + 8 j) >> 1 & 7 differs by 4 j & 7 ...
        voff = (uint32_t)row * (uint32_t)(K * 2) + (uint32_t)c * 16u;
    }
    // ... so the swizzle term changes with j: chunk = (lane & 7) ^ (((row0 + 8 j) >> 1) & 7) = c0 ^ ((4 j) & 7) for row0 % 16 < 8 lanes -
    // 8 j rows add 4 j to (row >> 1): XOR with (4 j & 7) when no carry crosses bit 3, i.e. always (4 j & 7 is 0 or 4 and
    // ((row0 >> 1) & 7) + 4 only flips bit 2 modulo 8).  Two lane offsets are enough: even j and odd j.
    const uint32_t voff_odd = voff ^ 64u;                     // chunk ^ 4 -> byte offset ^ 64 (chunk is bits 4..6 of the offset)
    const uint32_t row_step = 8u * (uint32_t)(K * 2);         // bytes between the row blocks of consecutive requests
    auto bufld = [&](const i32x4& rs, uint32_t vo, uint32_t so) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, 0x400" ::"v"(vo), "s"(rs), "s"(so) : "memory", "m0");
    };
    auto set_m0 = [&](uint32_t lds_off) { asm volatile("s_mov_b32 m0, %0" ::"s"(lds0 + lds_off) : "memory", "m0"); };
    auto stage_a = [&](int buf, int kt, int j) {
        if constexpr (BUF) {
            if (j == 0) { set_m0((uint32_t)(buf * TILE + wave * NJ * 1024)); asm volatile("s_nop 0"); }
            bufld(rs_a, (j & 1) ? voff_odd : voff, (uint32_t)kt * (BK * 2) + (uint32_t)j * row_step);
        } else {
            glds(a_src[j] + (int64_t)kt * (BK * 2), (uint32_t)(buf * TILE + (wave * NJ + j) * 1024));
        }
    };
    auto stage_w = [&](int buf, int kt, int j) {
        if constexpr (BUF) {
            if (j == 0) { set_m0((uint32_t)(W_BASE + buf * TILE + wave * NJ * 1024)); asm volatile("s_nop 0"); }
            bufld(rs_w, (j & 1) ? voff_odd : voff, (uint32_t)kt * (BK * 2) + (uint32_t)j * row_step);
        } else {
            glds(w_src[j] + (int64_t)kt * (BK * 2), (uint32_t)(W_BASE + buf * TILE + (wave * NJ + j) * 1024));
        }
    };

    // fragment reads: 16 rows x 32 k; lane = row l15, 16-byte chunk (4 kh + lg) ^ ((row >> 1) & 7); rows of fragment i are
    // 16 i + l15, and (16 i + l15) >> 1 & 7 == (l15 >> 1) & 7: one swizzle term per lane
    const int swz = (l15 >> 1) & 7;
    const int a_row = (wm * 128 + l15) * 128, w_row = (wn * 128 + l15) * 128;
    int coff[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) coff[kh] = ((4 * kh + lg) ^ swz) << 4;

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    bf16x8 af[2][8], wf[2][8];
    // prologue: tiles 0 and 1 requested back to back, accumulators zeroed under their round trip
    // (whole groups of NJ requests per operand: the BUF form walks M0 through a group)
#pragma unroll
    for (int j = 0; j < NJ; ++j) stage_a(0, 0, j);
#pragma unroll
    for (int j = 0; j < NJ; ++j) stage_w(0, 0, j);
    if (nk > 1) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) stage_a(1, 1, j);
#pragma unroll
        for (int j = 0; j < NJ; ++j) stage_w(1, 1, j);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJ) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        af[0][i] = *(const bf16x8*)(smem + a_row + i * 2048 + coff[0]);
        wf[0][i] = *(const bf16x8*)(smem + W_BASE + w_row + i * 2048 + coff[0]);
    }

    int sa = 0;
    // one K step = two K halves of 64 MFMAs; half 0 reads the fragments of half 1 (same tile) and requests A(kt+2); half 1 starts
    // with the barrier (own reads of tile kt done, own shares of tile kt+1 landed), reads the first-half fragments of tile kt+1
    // and requests W(kt+2) into the slot of this tile.  MODE 0: steady state; 1: second-to-last step (nothing requested, the
    // barrier waits for everything); 2: last step (no requests, no barrier, no reads of a next tile)
    int kt = 0;
    auto k_step = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const int sa1 = sa == AST - 1 ? 0 : sa + 1, sa2 = sa1 == AST - 1 ? 0 : sa1 + 1;
        const char* la = smem + sa * TILE;
        const char* lw = smem + W_BASE + (kt & 1) * TILE;
        const char* lan = smem + sa1 * TILE;
        const char* lwn = smem + W_BASE + ((kt + 1) & 1) * TILE;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            if (kh == 1 && MODE != 2) {
                if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NJ) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // operands swapped (A-operand = W fragment): a lane owns output row m = 16 i + l15 and columns 16 j + 4 lg .. + 3
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kh][j], af[kh][i], acc[i][j], 0, 0, 0);
                    const int idx = i * 8 + j;
                    // side instructions, one per MFMA slot: 16 fragment reads of the next half (order of first use: W0..W7 are all
                    // needed by row i = 0, then A0, A1, ...), then this half's share of the requests
                    if (!(MODE == 2 && kh == 1)) {
                        const char* fa = kh == 0 ? la : lan;
                        const char* fw = kh == 0 ? lw : lwn;
                        const int khn = kh ^ 1;
                        if (idx >= 8 && idx < 24 && ((idx & 1) == 0)) {                  // W fragments: slots 8, 10, ... 22
                            const int f = (idx - 8) >> 1;
                            wf[khn][f] = *(const bf16x8*)(fw + w_row + f * 2048 + coff[khn]);
                        }
                        if (idx >= 9 && idx < 25 && ((idx & 1) == 1)) {                  // A fragments: slots 9, 11, ... 23
                            const int f = (idx - 9) >> 1;
                            af[khn][f] = *(const bf16x8*)(fa + a_row + f * 2048 + coff[khn]);
                        }
                    }
                    if (MODE == 0 && idx >= 28 && idx < 28 + 4 * NJ && ((idx - 28) & 3) == 0) {
                        const int r = (idx - 28) >> 2;
                        if (kh == 0) stage_a(sa2, kt + 2, r);                            // A(kt+2): the slot tile kt-1 left
                        else stage_w(kt & 1, kt + 2, r);                                 // W(kt+2): the slot of this tile (after the barrier)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        sa = sa1;
        ++kt;
    };
    // (fragment register sets: half kh uses set kh, so the reads issued during half kh fill set kh ^ 1: of this tile for kh = 0,
    //  of the next tile's first half for kh = 1)
    while (kt + 2 < nk) k_step(std::integral_constant<int, 0>{});
    if (nk > 1) k_step(std::integral_constant<int, 1>{});
    k_step(std::integral_constant<int, 2>{});

    if constexpr (TEPI) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                  // every wave is done with the operand tiles
        char* const scr = smem + wave * 4096;
        const int rrow = lane >> 4, rch = lane & 15;      // read side: 4 rows x 16 chunks of 16 bytes per pass
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint2 o;
                o.x = pack2(acc[i][j][0], acc[i][j][1]);
                o.y = pack2(acc[i][j][2], acc[i][j][3]);
                // row l15, logical 16-byte chunk 2 j + (lg >> 1), half (lg & 1); physical chunk = logical ^ row
                *(uint2*)(scr + l15 * 256 + (((2 * j + (lg >> 1)) ^ l15) << 4) + (lg & 1) * 8) = o;
            }
            // same-wave LDS operations complete in order: the reads below see the writes above
#pragma unroll
            for (int pss = 0; pss < 4; ++pss) {
                const int r = pss * 4 + rrow;
                const uint4 v = *(const uint4*)(scr + r * 256 + ((rch ^ r) << 4));
                const int64_t m = m0 + wm * 128 + i * 16 + r;
                *(uint4*)(C + m * N + n0 + wn * 128 + rch * 8) = v;
            }
        }
    } else {
    // PLAIN epilogue, straight from the accumulators: 8 bytes (4 consecutive columns) per lane and fragment
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t m = m0 + wm * 128 + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t n = n0 + wn * 128 + j * 16 + lg * 4;
            uint2 o;
            o.x = pack2(acc[i][j][0], acc[i][j][1]);
            o.y = pack2(acc[i][j][2], acc[i][j][3]);
            *(uint2*)(C + m * N + n) = o;
        }
    }
    }
}

}  // namespace

extern "C" int gemm4w_plain(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int buffer_form, void* stream) {
    if (M % BM || N % BN || K % BK || M <= 0 || N <= 0 || K <= 0 || 256 * K * 2 >= (1ll << 31)) return 1;
    // buffer_form: bit 0 = buffer_load...lds requests, bit 1 = output through LDS (whole-row stores)
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)gemm4w_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return 2;
        if (hipFuncSetAttribute((const void*)gemm4w_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return 2;
        if (hipFuncSetAttribute((const void*)gemm4w_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return 2;
        if (hipFuncSetAttribute((const void*)gemm4w_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return 2;
        attr = true;
    }
    const int ntm = (int)(M / BM), ntn = (int)(N / BN);
    const int gm = K >= 4096 ? 4 : 8;
    const dim3 grid((unsigned)(ntm * ntn)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bf16_t* a = (const bf16_t*)A; const bf16_t* w = (const bf16_t*)W; bf16_t* c = (bf16_t*)C;
    switch (buffer_form & 3) {
        case 0: hipLaunchKernelGGL((gemm4w_kernel<false, false>), grid, block, LDS_BYTES, st, a, w, c, (int)M, (int)N, (int)K, ntm, ntn, gm); break;
        case 1: hipLaunchKernelGGL((gemm4w_kernel<true, false>), grid, block, LDS_BYTES, st, a, w, c, (int)M, (int)N, (int)K, ntm, ntn, gm); break;
        case 2: hipLaunchKernelGGL((gemm4w_kernel<false, true>), grid, block, LDS_BYTES, st, a, w, c, (int)M, (int)N, (int)K, ntm, ntn, gm); break;
        default: hipLaunchKernelGGL((gemm4w_kernel<true, true>), grid, block, LDS_BYTES, st, a, w, c, (int)M, (int)N, (int)K, ntm, ntn, gm); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
