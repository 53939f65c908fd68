// Weight-gradient GEMM for gfx950, both operands in their natural row-major layout ("TN"):
//
//     out[n, t*C + c] = sum_{m < M}  A[m, n] * B[clamp(m + tap_shift[t], 0, b_rows - 1), c]
//
// A = dY [M, N] (gradient of a layer's output), B = X (the layer's input): dW = dY^T X of torch.nn.Linear (one tap, shift 0)
// and - with A and B on the same zero-bordered padded token grid - the weight gradient of a 3x3 / 3-tap / 27-tap convolution in
// ONE launch (tap t reads the grid rows shifted by tap_shift[t]; border rows of A are zero, so clamped rows contribute nothing).
// Replaces: two transposes (dY^T, X^T; a gather + a transpose per tap for convolutions) feeding the NT kernel of gemm_bf16.hip.
//
// The contraction index m is the ROW of both operands, i.e. the operand tiles arrive in LDS as [64 rows m][256 columns] and an MFMA
// fragment (32 columns x 16 m, a lane: one column, 8 consecutive m) is a TRANSPOSE of what sits there: two ds_read_b64_tr_b16 per
// fragment (16-lane group: 4 rows x 16 columns in, a lane receives its column's 4 rows).  Everything else follows gemm_bf16.hip:
// 256 x 256 output tile, 8 waves (2 x 4), 128 x 64 per wave, operands swapped so that a lane owns one output row (a column of A)
// and 4 consecutive output columns; LDS-DMA staging into the whole 160 KiB as a ring (3 stages of A, 2 of B; one counted vmcnt +
// bare barrier per K step; requests from inline asm so that the compiler's LDS waits stay counted); the contraction is always
// split over K ranges (one workgroup per (tile, range)): fp32 partial tiles to the workspace, tn_finish_kernel reduces them in a
// fixed order.  Same K order and ranges as the transposed path => bit-identical results.
//
// LDS image of a tile: row r at r*512, its 64-byte slot index XORed with (r & 3).  The LDS serves 256 bytes (64 banks) per cycle,
// i.e. 32 lanes of a ds_read_b64_tr_b16: two 16-lane groups = the same 4 rows x 64 contiguous bytes; the rows are 512 bytes
// apart (the same banks), the XOR puts them into the four 64-byte slots of the 256-byte bank space.  (A first version XORed the
// 32-byte granule instead: conflict-free per 16-lane group, but the two groups of a cycle then met in the same 128 bytes - the
// counters showed one conflict cycle per active cycle, profiles/README.md.)  Applied on the SOURCE chunk of the lane-linear
// LDS-DMA and again on the transposing reads.
#include "common.h"
#include "dwm_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int TBK = 64;                        // contraction rows per K step
constexpr int TILE = TBK * 512;                // 32 KiB: [64 rows][256 bf16]
constexpr int AST = 3, BST = 2;
constexpr int B_BASE = AST * TILE;
constexpr int TN_LDS = (AST + BST) * TILE;     // 160 KiB

struct TnParams {
    const bf16_t* A; int64_t lda;
    const bf16_t* B; int64_t ldb;
    int64_t M, N, C, b_rows;
    int ntaps, ctiles;                         // column tiles per tap
    int64_t tap_shift[27];
    float* ws; int64_t ws_slice;               // floats per K range: N * ntaps * C
    int ksplit, nk_all, ntm, ntn;
};

DWM_DEVINL int tn_swz(int r) { return r & 3; }

__global__ void __launch_bounds__(512, 2)
gemm_tn_kernel(const TnParams p) {
    constexpr int NJ = 4;                      // 1-KiB requests per wave, operand and stage (2 rows each)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int half = lane >> 5, l31 = lane & 31;

    const int ntiles = p.ntm * p.ntn;
    int id = xcd_remap(blockIdx.x, ntiles * p.ksplit);
    const int slice = id / ntiles;             // range-major: the tiles of one K range run together (they share A / B rows in L2)
    id -= slice * ntiles;
    const int tn = id / p.ntm, tm = id - tn * p.ntm;
    const int tap = tn / p.ctiles;
    const int64_t n0 = (int64_t)tm * 256, c0 = (int64_t)(tn - tap * p.ctiles) * 256;
    const int kt0 = (int)((int64_t)slice * p.nk_all / p.ksplit);
    const int nk = (int)((int64_t)(slice + 1) * p.nk_all / p.ksplit) - kt0;
    const int64_t shift = p.tap_shift[tap];

    // ---- staging: request j of this wave fills rows (wave*4 + j)*2 + {0, 1} of a tile; lane -> row half, physical 16-B chunk
    const char* a_src[NJ];
    const char* b_base[NJ];                    // B + this lane's column
    int b_row[NJ];                             // row of B for K step 0 of this range, before the clamp (rows < 2^31)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = (wave * NJ + j) * 2 + (lane >> 5);
        const int lc = (lane & 31) ^ (tn_swz(r) << 2);                       // logical 16-B chunk this lane fetches (64-B slot XOR)
        const int64_t acol = n0 + lc * 8 < p.N ? n0 + lc * 8 : 0;           // columns past the matrix: any valid address (never stored)
        const int64_t bcol = c0 + lc * 8 < p.C ? c0 + lc * 8 : 0;
        a_src[j] = (const char*)(p.A + ((int64_t)kt0 * TBK + r) * p.lda + acol);
        b_base[j] = (const char*)(p.B + bcol);
        b_row[j] = (int)((int64_t)kt0 * TBK + r + shift);
    }
    const int64_t a_step = (int64_t)TBK * p.lda * 2;                       // bytes per K step
    const uint32_t ldb_bytes = (uint32_t)(p.ldb * 2);
    const int b_last = (int)(p.b_rows - 1);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    auto glds = [&](const char* src, uint32_t lds_off) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds0 + lds_off) : "memory", "m0");
    };
    auto stage_a = [&](int buf, int kt, int j) {
        glds(a_src[j] + (int64_t)kt * a_step, (uint32_t)(buf * TILE + (wave * NJ + j) * 1024));
    };
    auto stage_b = [&](int buf, int kt, int j) {
        // one v_med3 (clamp: tap shifts reach before / behind the grid only in the first / last K step) + one 32 x 32 -> 64-bit multiply-add
        int row = b_row[j] + kt * TBK;
        row = row < 0 ? 0 : row;
        row = row > b_last ? b_last : row;                     // (max + min: the compiler folds them into one v_med3_i32)
        glds(b_base[j] + (uint64_t)(uint32_t)row * ldb_bytes, (uint32_t)(B_BASE + buf * TILE + (wave * NJ + j) * 1024));
    };

    // ---- transposing fragment reads: 16-lane group (g1 = column half of the 32-column fragment, `half` = which 8 of the 16 rows),
    // lane u: address of row (u >> 2), 8-byte piece (u & 3) of the group's 32-byte column segment; receives column u, 4 rows
    const int u = lane & 15, g1 = (lane >> 4) & 1;
    const int sw = (u >> 2) << 6;                                           // swz(row) << 6: row & 3 = u >> 2 for every row this lane addresses
    const int row_off = half * 4096 + (u >> 2) * 512;                       // + ks * 8192 + rd * 2048
    int a_off[4], b_off[2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) a_off[mt] = row_off + wm * 256 + ((mt * 64 + g1 * 32 + (u & 3) * 8) ^ sw);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) b_off[nt] = row_off + (wn >> 1) * 256 + (((wn & 1) * 128 + nt * 64 + g1 * 32 + (u & 3) * 8) ^ sw);
    auto frag = [&](const char* tile, int off, int ks) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + off + ks * 8192));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + off + ks * 8192 + 2048));
        return (bf16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- main loop (the structure of gemm_bf16_kernel, configuration 0): per K step 4 sub-steps of 8 MFMAs; sub-step s reads the
    // fragments of sub-step s+1 (order of first use: B0 A0 B1 A1 A2 A3); A(kt+2) is requested in sub-step 0 into the slot tile
    // kt-1 left, B(kt+2) in the last sub-step - after the barrier - into the slot of this tile
    bf16x8 af[2][4], bfr[2][2];
    int sa = 0;
    {
#pragma unroll
        for (int j = 0; j < NJ; ++j) { stage_a(0, 0, j); stage_b(0, 0, j); }
        const int k1 = nk > 1 ? 1 : 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) stage_a(1, k1, j);
#pragma unroll
        for (int j = 0; j < NJ; ++j) stage_b(1, k1, j);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[i][j]));
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJ) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        bfr[0][0] = frag(smem + B_BASE, b_off[0], 0);
        af[0][0] = frag(smem, a_off[0], 0);
        bfr[0][1] = frag(smem + B_BASE, b_off[1], 0);
#pragma unroll
        for (int mt = 1; mt < 4; ++mt) af[0][mt] = frag(smem, a_off[mt], 0);
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int sa1 = sa == AST - 1 ? 0 : sa + 1, sa2 = sa1 == AST - 1 ? 0 : sa1 + 1;
        const char* la = smem + sa * TILE;
        const char* lb = smem + B_BASE + (kt & 1) * TILE;
        const char* lan = smem + sa1 * TILE;
        const char* lbn = smem + B_BASE + ((kt + 1) & 1) * TILE;
        const int kt2 = kt + 2 < nk ? kt + 2 : nk - 1;                       // past the end: a redundant reload nobody reads
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (ks == 3 && c == 0) {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NJ) : "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                acc[c >> 1][c & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks & 1][c & 1], af[ks & 1][c >> 1], acc[c >> 1][c & 1], 0, 0, 0);
                if (c < 6) {
                    const char* fa = ks < 3 ? la : lan;
                    const char* fb = ks < 3 ? lb : lbn;
                    const int kn = ks < 3 ? ks + 1 : 0;
                    const bool is_b = c == 0 || c == 2;
                    const int fi = c == 0 ? 0 : c == 1 ? 0 : c == 2 ? 1 : c - 2;
                    if (is_b) bfr[(ks + 1) & 1][fi] = frag(fb, b_off[fi], kn);
                    else af[(ks + 1) & 1][fi] = frag(fa, a_off[fi], kn);
                }
                if (ks == 0 && c < NJ) stage_a(sa2, kt2, c);
                if (ks == 3 && c >= 1 && c <= NJ) stage_b(kt & 1, kt2, c - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        sa = sa1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- fp32 partial tile of this K range: lane = output row n0 + wm*128 + mt*32 + l31, 4 consecutive columns per register group
    const int64_t ncols = (int64_t)p.ntaps * p.C;
    float* __restrict__ ws = p.ws + (int64_t)slice * p.ws_slice;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int64_t n = n0 + wm * 128 + mt * 32 + l31;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int64_t c = c0 + wn * 64 + nt * 32 + rg * 8 + half * 4;
                if (n < p.N && c < p.C)
                    *(float4*)(ws + n * ncols + (int64_t)tap * p.C + c) =
                        make_float4(acc[mt][nt][rg * 4], acc[mt][nt][rg * 4 + 1], acc[mt][nt][rg * 4 + 2], acc[mt][nt][rg * 4 + 3]);
            }
    }
}

// out[n][c..c+8) = sum over the K ranges, in range order (the result does not depend on scheduling)
__global__ void __launch_bounds__(256)
tn_finish_kernel(const float* __restrict__ ws, int64_t ws_slice, int ksplit, int64_t N, int64_t ncols, bf16_t* __restrict__ out, int64_t ldo) {
    const int64_t c8 = ncols >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * c8) return;
    const int64_t n = i / c8, c = (i - n * c8) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < ksplit; ++s) {
        const float* src = ws + (int64_t)s * ws_slice + n * ncols + c;
        const float4 x0 = *(const float4*)src, x1 = *(const float4*)(src + 4);
        v[0] += x0.x; v[1] += x0.y; v[2] += x0.z; v[3] += x0.w; v[4] += x1.x; v[5] += x1.y; v[6] += x1.z; v[7] += x1.w;
    }
    *(uint4*)(out + n * ldo + c) = pack8(v);
}

}  // namespace

extern "C" int dwm_gemm_tn(const dwm_gemm_tn_args* a, void* stream) {
    if (a == nullptr || a->A == nullptr || a->B == nullptr || a->out == nullptr || a->workspace == nullptr) return DWM_EINVAL;
    if (a->M <= 0 || a->N <= 0 || a->C <= 0 || a->b_rows <= 0 || a->M >= (1ll << 31) || a->b_rows >= (1ll << 31)) return DWM_EINVAL;
    if (a->M % TBK != 0 || a->N % 8 != 0 || a->C % 8 != 0) return DWM_EUNSUPPORTED;
    const int ntaps = a->ntaps > 0 ? a->ntaps : 1;
    if (ntaps > 27) return DWM_EINVAL;
    if (a->ntaps <= 0 && a->b_rows < a->M) return DWM_EINVAL;      // no taps: row m of B pairs with row m of A, nothing may be clamped
    if (a->lda < a->N || a->ldb < a->C || a->lda % 8 != 0 || a->ldb % 8 != 0 || a->ldb >= (1ll << 30) || a->ldo % 8 != 0 || a->ldo < (int64_t)ntaps * a->C) return DWM_EALIGN;
    if (!dwm_aligned16(a->A) || !dwm_aligned16(a->B) || !dwm_aligned16(a->out) || !dwm_aligned16(a->workspace)) return DWM_EALIGN;
    TnParams p;
    p.A = (const bf16_t*)a->A; p.lda = a->lda;
    p.B = (const bf16_t*)a->B; p.ldb = a->ldb;
    p.M = a->M; p.N = a->N; p.C = a->C; p.b_rows = a->b_rows;
    p.ntaps = ntaps;
    p.ctiles = (int)((a->C + 255) / 256);
    for (int t = 0; t < 27; ++t) p.tap_shift[t] = (a->ntaps > 0 && t < ntaps) ? a->tap_shift[t] : 0;
    p.ntm = (int)((a->N + 255) / 256);
    p.ntn = ntaps * p.ctiles;
    p.nk_all = (int)(a->M / TBK);
    const int64_t tiles = (int64_t)p.ntm * p.ntn;
    const int64_t slice_floats = a->N * (int64_t)ntaps * a->C;
    // K ranges: every (tile, range) is one workgroup and a CU holds one, so the launch runs in ceil(tiles * ranges / CUs) rounds.
    // Automatic choice: the range count (ranges of >= 8 K steps, <= 32, within the workspace) with the smallest modelled time
    //     rounds * (K steps per range * 1.5 us + 25 us per workgroup) + ranges * partial bytes / 4 TB/s   (the ordered reduction)
    // - e.g. the 144 tiles of a 6144 x 1536 weight gradient: 1 range = 144 of 256 CUs busy for 1344 steps, 7 ranges = 1008
    // workgroups = 3.94 rounds of 192 steps; but 108 tiles over 462 steps stay at 2 ranges (26 ranges would fill the last round
    // better and pay 11 rounds of fixed costs for it; measured).
    int kmax = p.nk_all / 8;
    if (kmax > 32) kmax = 32;
    if ((int64_t)kmax * slice_floats * 4 > a->workspace_bytes) kmax = (int)(a->workspace_bytes / (slice_floats * 4));
    if (kmax < 1) kmax = slice_floats * 4 <= a->workspace_bytes ? 1 : 0;
    if (kmax < 1) return DWM_EINVAL;
    int ksplit = a->split_k;
    if (ksplit <= 0) {
        static int ncu = 0;
        if (ncu == 0) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        }
        double best = 0.0;
        for (int k = 1; k <= kmax; ++k) {
            const int64_t rounds = (tiles * k + ncu - 1) / ncu;
            const double steps = (double)((p.nk_all + k - 1) / k);
            const double us = (double)rounds * (steps * 1.5 + 25.0) + (double)k * (double)slice_floats * 4.0 / 4.0e6;
            if (k == 1 || us < best) { best = us; ksplit = k; }
        }
    } else if (ksplit > kmax) {
        return DWM_EINVAL;
    }
    p.ksplit = ksplit;
    p.ws = (float*)a->workspace;
    p.ws_slice = slice_floats;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    {
        static bool attr_set = false;
        if (!attr_set) {
            e = hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
    }
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)(tiles * ksplit)), dim3(512), TN_LDS, s, p);
    const int64_t nthr = a->N * (((int64_t)ntaps * a->C) >> 3);
    hipLaunchKernelGGL(tn_finish_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, p.ws, p.ws_slice, ksplit, a->N,
                       (int64_t)ntaps * a->C, (bf16_t*)a->out, a->ldo);
    e = hipGetLastError();
    return e == hipSuccess ? DWM_OK : (int)e;
}
