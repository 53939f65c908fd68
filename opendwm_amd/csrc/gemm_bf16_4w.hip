// bf16 GEMM with the fused epilogues of gemm_bf16.hip on a 4-WAVE main loop:  C[M,Nout] = epi(A[M,K] · W[N,K]^T)
//
// Used where the caller asks for it (dwm_gemm_args.tile == 3 / 4: the MMDiT inference forward and the MMDiT train step; the Python
// host side maps environment DWM_GEMM4W=1 to tile 3 on every call - the library itself reads no environment).  The other callers (UNet, VAEs) keep the 8-wave kernels of gemm_bf16.hip: measured slower
// on this tile for the UNet's N = 320 / 640 shapes (profiles/r5a_*: 73.9 -> 75.6 ms per step).
//
// Geometry = what hipBLASLt's gfx950 kernel for these shapes does (Custom_Cijk_Alik_Bljk_..._MT256x256x64_MI16x16x1, disassembled
// from the ROCm install; it runs the bench's GEMM shapes 10-20 % faster than the 8-wave loop, profiles/README.md): the same
// 256 x 256 x 64 tile, but 4 waves - one per SIMD - of 128 x 128 each, v_mfma_f32_16x16x32_bf16, the 256 accumulator registers of a
// wave in AGPRs (this file is compiled WITHOUT -amdgpu-mfma-vgpr-form, build.py AGPR_SOURCES).  Per K step and wave: 128 MFMAs, 32
// ds_read_b128 (a 16 x 32 fragment is reused by 8 MFMAs: a third fewer LDS bytes per flop than 128 x 64 wave tiles of 32 x 32 MFMAs)
// and 16 LDS-DMA requests issued as `buffer_load_dwordx4 v, s[rsrc], s_off offen lds`: one VGPR offset per lane for all requests of an
// operand (two: the chunk swizzle alternates with the request's row block), the row block in an SGPR offset, the K walk in the
// same SGPR, M0 bumped by 1 KiB behind every request - two scalar-side instructions per request, no vector ones.
// LDS image, 3 + 2 stage ring and counted waits are gemm_bf16.hip's: [256 rows][8 x 16-B chunks], chunk ^ ((row >> 1) & 7) (conflict
// free for 16-row fragments too: a ds_read_b128 lane group covers row pairs with 8 distinct chunk slots x 2 rows).
//
// MFMA operands are swapped (A-operand = W fragment), so a lane owns ONE output row m and, per 16-column block j, FOUR CONSECUTIVE
// columns:    m = 16 i + (lane & 15),   n = 16 j + 4 (lane >> 4) + r      (i, j = 0..7 blocks of the wave's 128 x 128, r = 0..3)
// Epilogues (per 16-row pass i): stage A in that layout - bias, activation, GEGLU product (value blocks j, gate blocks j + 2 of a
// 64-column slab: the same lane), q / k RMSNorm per 64-column head (16 values per lane, two lane exchanges: xor 16, xor 32); stage B
// after a transpose through 8 KiB of wave-private LDS (16-byte chunks XOR-swizzled by the row): row-major 16-byte accesses for the
// gate / residual / blend rows and the stores.  RESID comes in the compile-time operand forms (RS) only: residual, gate + residual,
// residual + blend, on the fp32 stream (RF32, no bf16 copy) or in bf16.
// Covered launches (everything else stays on gemm_bf16.hip): no row maps, no taps, no split-K, M % 256 == N % 256 == K % 64 == 0.
// General form (template parameter GEN): ragged M / N, the A row map, taps, the per-image residual row - see dwm_gemm4w_try.
#include <atomic>

#include "common.h"
#include "dwm_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 hbf16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE = BM * BK * 2;                 // 32 KiB per operand tile
constexpr int AST = 3, WST = 2;
constexpr int W_BASE = AST * TILE;
constexpr int LDS_BYTES = (AST + WST) * TILE;     // 160 KiB
constexpr int NJ = TILE / (4 * 1024);             // 1-KiB requests per wave, operand and stage: 8

struct G4Map {              // dwm_rowmap2d with fast divisors (gemm_bf16.hip's DevRowMap)
    FastDiv rw, rh;
    int64_t rpitch, ipitch, origin;
    int enabled, xstep;
};
struct G4Params {
    int ntm, ntn, gm;
    FastDiv fd_rpg, fd_rpa;
    // general form (GEN): the A row map, the taps of an implicit convolution as byte offsets from the row of the smallest shift
    // (the A resource starts `a_base_rows` rows from p.A, so every offset is >= 0), the divisor of the per-image residual row
    FastDiv fd_rmod;
    G4Map amap;
    int steps_per_tap;
    int64_t a_base_rows;
    uint32_t tap_off[27];
};
DWM_DEVINL int64_t map_row4(const G4Map& rm, int64_t m) {
    if (!rm.enabled) return m;
    const uint32_t q = fdiv((uint32_t)m, rm.rw), x = (uint32_t)m - q * rm.rw.d;
    const uint32_t i = fdiv(q, rm.rh), y = q - i * rm.rh.d;
    return (int64_t)i * rm.ipitch + (int64_t)y * rm.rpitch + (int64_t)x * rm.xstep + rm.origin;
}

// GEN (general form, see dwm_gemm4w_try): ragged M / N (operand rows clamped to the last one, stores guarded), A rows through
// the row map `a_map`, K walked tap by tap (implicit convolution), RS bit 16 (the residual row is m / |res_mod|).  The per-lane
// request offsets then differ from request to request (one VGPR each instead of two per operand) and the K walk of A is a scalar
// that jumps at tap boundaries (the table sits in a VGPR, one lane per tap, read by v_readlane: gemm_bf16.hip's reason).
template <int EPI, bool RF32, int RS, bool GEN = false>
__global__ void __launch_bounds__(256, 1)
gemm4w_kernel(const dwm_gemm_args p, const G4Params gp) {
    static_assert(EPI == DWM_EPI_RESID || (!RF32 && RS == 0), "RF32 / RS belong to RESID");
    static_assert(EPI != DWM_EPI_RESID || RS == 2 || RS == 3 || RS == 6 || (GEN && !RF32 && RS == 18),
                  "RESID: residual / gate + residual / residual + blend / (general form) residual row per image");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;                 // fragment row; 16-byte k chunk of a 32-wide K half = column group of 4
    const int64_t M = p.M, N = p.N, K = p.K;
    uint32_t tapv = 0;                                         // GEN: lane t keeps the byte offset of tap t
    if constexpr (GEN) tapv = gp.tap_off[lane < 27 ? lane : 0];

    // tile rasterisation of gemm_bf16.hip: XCD-contiguous ids, groups of gm row tiles x all column tiles
    const int ntm = gp.ntm, ntn = gp.ntn, gm = gp.gm;
    const int id = xcd_remap(blockIdx.x, ntm * ntn);
    const int per_group = gm * ntn;
    const int grp = id / per_group, in_grp = id - grp * per_group;
    const int first_m = grp * gm;
    const int gsize = ntm - first_m < gm ? ntm - first_m : gm;
    const int tn = in_grp / gsize, tm = first_m + in_grp % gsize;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    // ---- LDS-DMA requests (see the header): request j of this wave fills rows (wave * 8 + j) * 8 .. + 8 of a tile
    const bf16_t* __restrict__ Ap = (const bf16_t*)p.A;
    const bf16_t* __restrict__ Wp = (const bf16_t*)p.W;
    auto make_rsrc = [](const void* ptr) {
        const uint64_t a = (uint64_t)ptr;
        i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
        r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu));
        r[2] = -1;                                        // no range check wanted (the tile is inside the matrix)
        r[3] = 0x00020000;
        return r;
    };
    const i32x4 rs_a = make_rsrc(GEN ? Ap + gp.a_base_rows * p.lda : Ap + m0 * p.lda), rs_w = make_rsrc(GEN ? Wp : Wp + n0 * K);
    // per-lane byte offset of request 0 (row = wave * 64 + lane / 8, chunk swizzled by the row); request j is 8 j rows further, which
    // XORs the chunk with (4 j) & 7 = 4 (j & 1): two offsets per operand (A and W differ in their row pitch)
    const int row0 = wave * (NJ * 8) + (lane >> 3);
    const uint32_t ch0 = (uint32_t)((lane & 7) ^ ((row0 >> 1) & 7)) * 16u;
    const uint32_t voff_a = (uint32_t)row0 * (uint32_t)(p.lda * 2) + ch0, voff_w = (uint32_t)row0 * (uint32_t)(K * 2) + ch0;
    const uint32_t step_a = 8u * (uint32_t)(p.lda * 2), step_w = 8u * (uint32_t)(K * 2);        // bytes between consecutive requests' rows
    // GEN: one offset per request - the row (clamped to the last one, mapped for A) times the pitch, plus the swizzled chunk
    uint32_t va[NJ], vw[NJ];
    if constexpr (GEN) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = row0 + j * 8;
            const uint32_t ch = (uint32_t)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
            int64_t gmr = m0 + row; gmr = gmr < M ? gmr : M - 1;
            int64_t gnr = n0 + row; gnr = gnr < N ? gnr : N - 1;
            va[j] = (uint32_t)map_row4(gp.amap, gmr) * (uint32_t)(p.lda * 2) + ch;
            vw[j] = (uint32_t)gnr * (uint32_t)(K * 2) + ch;
        }
        // (the tap table's load is consumed HERE, before the first request: its wait behind them would drain them)
        asm volatile("" : "+v"(tapv));
    }
    // GEN: the walk of the A source over K (gemm_bf16.hip): inside a tap one tile further per step, at a tap boundary the next
    // tap's offset; walk_a belongs to the tile requested next
    int walk_left = 0, walk_tap = 0;
    uint32_t walk_a = 0;
    auto walk_next = [&]() {
        if (--walk_left == 0) {
            ++walk_tap;
            walk_left = gp.steps_per_tap;
            walk_a = (uint32_t)__builtin_amdgcn_readlane((int)tapv, walk_tap);
        } else {
            walk_a += BK * 2;
        }
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    auto bufld = [&](const i32x4& rs, uint32_t vo, uint32_t so) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, 0x400" ::"v"(vo), "s"(rs), "s"(so) : "memory", "m0");
    };
    auto set_m0 = [&](uint32_t lds_off) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds0 + lds_off) : "memory", "m0"); };
    // (a GROUP of NJ requests per operand and stage is issued in order j = 0 .. NJ - 1: M0 walks through it)
    // (GEN: the A requests of a stage take the walk's offset `wa` of their tile)
    auto stage_a = [&](int buf, int kt, int j, uint32_t wa) {
        if (j == 0) set_m0((uint32_t)(buf * TILE + wave * NJ * 1024));
        if constexpr (GEN) bufld(rs_a, va[j], wa);
        else bufld(rs_a, (j & 1) ? (voff_a ^ 64u) : voff_a, (uint32_t)kt * (BK * 2) + (uint32_t)j * step_a);
    };
    auto stage_w = [&](int buf, int kt, int j) {
        if (j == 0) set_m0((uint32_t)(W_BASE + buf * TILE + wave * NJ * 1024));
        if constexpr (GEN) bufld(rs_w, vw[j], (uint32_t)kt * (BK * 2));
        else bufld(rs_w, (j & 1) ? (voff_w ^ 64u) : voff_w, (uint32_t)kt * (BK * 2) + (uint32_t)j * step_w);
    };

    // fragment reads: 16 rows x 32 k; lane = row l15, 16-byte chunk (4 kh + lg) ^ ((row >> 1) & 7); rows of fragment f are 16 f + l15
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

    const int nk = (int)(K / BK);
    hbf16x8 af[2][8], wf[2][8];
    if constexpr (GEN) {
        walk_left = gp.steps_per_tap;
        walk_a = gp.tap_off[0];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) stage_a(0, 0, j, walk_a);
#pragma unroll
    for (int j = 0; j < NJ; ++j) stage_w(0, 0, j);
    if (nk > 1) {
        if constexpr (GEN) walk_next();
#pragma unroll
        for (int j = 0; j < NJ; ++j) stage_a(1, 1, j, walk_a);
#pragma unroll
        for (int j = 0; j < NJ; ++j) stage_w(1, 1, j);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));      // zeroed here, under the round trip of the first requests
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJ) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        af[0][f] = *(const hbf16x8*)(smem + a_row + f * 2048 + coff[0]);
        wf[0][f] = *(const hbf16x8*)(smem + W_BASE + w_row + f * 2048 + coff[0]);
    }

    // One K step = two K halves of 64 MFMAs.  Half 0 reads the fragments of half 1 (same tile) and requests A(kt+2) into the slot tile
    // kt-1 left; half 1 starts with the barrier (own reads of tile kt done, own shares of tile kt+1 landed: vmcnt(NJ) leaves the
    // requests of A(kt+2) in flight), reads the first-half fragments of tile kt+1 and requests W(kt+2) into the slot of this tile.
    // MODE 0: steady state; 1: second-to-last step (nothing requested, the barrier waits for everything); 2: last step.
    int sa = 0, kt = 0;
    auto k_step = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const int sa1 = sa == AST - 1 ? 0 : sa + 1, sa2 = sa1 == AST - 1 ? 0 : sa1 + 1;
        const char* la = smem + sa * TILE;
        const char* lw = smem + W_BASE + (kt & 1) * TILE;
        const char* lan = smem + sa1 * TILE;
        const char* lwn = smem + W_BASE + ((kt + 1) & 1) * TILE;
        if constexpr (GEN && MODE == 0) walk_next();            // tile kt + 2
        const uint32_t wa2 = walk_a;
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kh][j], af[kh][i], acc[i][j], 0, 0, 0);
                    const int idx = i * 8 + j;
                    // side instructions, one per MFMA slot: the 16 fragment reads of the next half, then this half's 8 requests
                    if (!(MODE == 2 && kh == 1)) {
                        const char* fa = kh == 0 ? la : lan;
                        const char* fw = kh == 0 ? lw : lwn;
                        const int khn = kh ^ 1;
                        if (idx >= 8 && idx < 24 && ((idx & 1) == 0)) {
                            const int f = (idx - 8) >> 1;
                            wf[khn][f] = *(const hbf16x8*)(fw + w_row + f * 2048 + coff[khn]);
                        }
                        if (idx >= 9 && idx < 25 && ((idx & 1) == 1)) {
                            const int f = (idx - 9) >> 1;
                            af[khn][f] = *(const hbf16x8*)(fa + a_row + f * 2048 + coff[khn]);
                        }
                    }
                    if (MODE == 0 && idx >= 28 && idx < 28 + 4 * NJ && ((idx - 28) & 3) == 0) {
                        const int r = (idx - 28) >> 2;
                        if (kh == 0) stage_a(sa2, kt + 2, r, wa2);
                        else stage_w(kt & 1, kt + 2, r);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        sa = sa1;
        ++kt;
    };
    while (kt + 2 < nk) k_step(std::integral_constant<int, 0>{});
    if (nk > 1) k_step(std::integral_constant<int, 1>{});
    k_step(std::integral_constant<int, 2>{});

    // ------------------------------------------------------------------------------------------------ epilogue
    constexpr bool kGeglu = EPI == DWM_EPI_GEGLU, kResid = EPI == DWM_EPI_RESID;
    constexpr int OW = kGeglu ? 64 : 128;                     // output columns of this wave
    constexpr int ESZ = kResid ? 4 : 2;                       // bytes per value crossing the LDS
    constexpr int RB = OW * ESZ;                              // bytes per row of the transpose image: 512 / 256 / 128
    constexpr int NCH = RB / 16;                              // 16-byte chunks per row: 32 / 16 / 8
    constexpr int LPR = OW / 8;                               // row-major side: lanes per row (8 columns each): 16 / 8
    constexpr int RPS = 64 / LPR;                             // rows per step: 4 / 8
    constexpr int NST = 16 / RPS;                             // steps per 16-row pass: 4 / 2
    const int64_t Nout = kGeglu ? (N >> 1) : N;
    const int64_t ocol0 = (kGeglu ? (n0 >> 1) + wn * 64 : n0 + wn * 128);       // first output column of this wave
    // bias (and RMS weights) of this lane's 8 x 4 columns in the MFMA layout, requested before the operand tiles are released
    float bv[8][4];
    float rw[8][4];
    bool do_norm[2] = {false, false};
    if constexpr (!kResid) {
        const bf16_t* __restrict__ bias = (const bf16_t*)p.bias;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int64_t n = n0 + wn * 128 + j * 16 + lg * 4;
            if constexpr (GEN) n = n < N ? n : 0;               // columns past N: any valid address (never stored)
            if (bias != nullptr) unpack4(*(const uint2*)(bias + n), bv[j]);
            else { bv[j][0] = bv[j][1] = bv[j][2] = bv[j][3] = 0.f; }
        }
        if constexpr (EPI == DWM_EPI_RMSHEAD) {
#pragma unroll
            for (int s = 0; s < 2; ++s) do_norm[s] = n0 + wn * 128 + s * 64 < p.rms_ncols;      // wave-uniform: slab s is a q / k head
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int64_t n = n0 + wn * 128 + j * 16 + lg * 4;
                if constexpr (GEN) n = n < N ? n : 0;
                if (do_norm[j >> 2]) unpack4(*(const uint2*)((const bf16_t*)p.rms_w + n), rw[j]);
                else { rw[j][0] = rw[j][1] = rw[j][2] = rw[j][3] = 1.f; }
            }
        }
    }
    // row-major side: this lane's 8 output columns, and (RESID) their bias
    const int rrow = lane / LPR, rc8 = lane % LPR;
    const int64_t ocol_raw = ocol0 + rc8 * 8;
    const bool nok = !GEN || ocol_raw < Nout;                 // GEN: this lane's 8 columns exist (N % 8 == 0)
    const int64_t ocol = nok ? ocol_raw : 0;                  // (loads of absent columns: a valid address, the values are dropped)
    float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (kResid) {
        if (p.bias != nullptr) unpack8(*(const uint4*)((const bf16_t*)p.bias + ocol), b8);
    }
    constexpr bool f_gate = (RS & 1) != 0, f_blend = (RS & 4) != 0;          // (the residual is always there: RS & 2)
    constexpr bool f_rowdiv = (RS & 16) != 0;                 // the residual row is m / |res_mod| (one row per image)

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                          // every wave is done with the operand tiles
    char* const scr = smem + wave * 8192;                     // this wave's transpose image: 16 rows x RB bytes

    // RESID operand rows of one 16-row pass (NST steps x one row piece per lane), requested one pass ahead
    uint4 gA[NST], rA[NST], rB[NST], blA[NST], blB[NST];
    float alA[NST];
    auto issue_resid = [&](int i) {
        if constexpr (kResid) {
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                uint32_t m = (uint32_t)(m0 + wm * 128 + i * 16 + st * RPS + rrow);
                if constexpr (GEN) m = m < (uint32_t)M ? m : (uint32_t)(M - 1);
                if constexpr (f_gate) gA[st] = *(const uint4*)((const bf16_t*)p.gate + ((uint64_t)fdiv(m, gp.fd_rpg) * (uint32_t)p.ld_gate + (uint32_t)ocol));
                if constexpr (RF32) {
                    const float* rp = (const float*)p.res + ((uint64_t)m * (uint32_t)p.ld_res + (uint32_t)ocol);
                    rA[st] = *(const uint4*)rp;
                    rB[st] = *(const uint4*)(rp + 4);
                    if constexpr (f_blend) {
                        const float* bp = (const float*)p.blend + ((uint64_t)m * (uint32_t)p.ld_blend + (uint32_t)ocol);
                        blA[st] = *(const uint4*)bp;
                        blB[st] = *(const uint4*)(bp + 4);
                    }
                } else {
                    rA[st] = *(const uint4*)((const bf16_t*)p.res + ((uint64_t)(f_rowdiv ? fdiv(m, gp.fd_rmod) : m) * (uint32_t)p.ld_res + (uint32_t)ocol));
                    if constexpr (f_blend) blA[st] = *(const uint4*)((const bf16_t*)p.blend + ((uint64_t)m * (uint32_t)p.ld_blend + (uint32_t)ocol));
                }
                if constexpr (f_blend) alA[st] = p.alpha[fdiv(m, gp.fd_rpa)];
            }
        }
    };
    issue_resid(0);

#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // ---- stage A (MFMA layout) + transpose writes: row l15, logical 16-byte chunk, XOR-swizzled by the row
        if constexpr (kGeglu) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int jv = 4 * s + jj, jg = jv + 2;
                    float y[4];
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2 hv = (f32x2){acc[i][jv][r], acc[i][jv][r + 1]} + (f32x2){bv[jv][r], bv[jv][r + 1]};
                        const f32x2 x = hv * gelu_erf2((f32x2){acc[i][jg][r], acc[i][jg][r + 1]} + (f32x2){bv[jg][r], bv[jg][r + 1]});
                        y[r] = x[0]; y[r + 1] = x[1];
                    }
                    const int b = 2 * s + jj;                                  // output block of 16 columns: 8 bytes at chunk 2 b + (lg >> 1)
                    *(uint2*)(scr + l15 * RB + (((2 * b + (lg >> 1)) ^ (l15 & (NCH - 1))) << 4) + (lg & 1) * 8) = pack4(y);
                }
        } else if constexpr (EPI == DWM_EPI_RMSHEAD) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float x[4][4];
                f32x2 ss2 = {0.f, 0.f};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2 v = (f32x2){acc[i][4 * s + jj][r], acc[i][4 * s + jj][r + 1]} + (f32x2){bv[4 * s + jj][r], bv[4 * s + jj][r + 1]};
                        x[jj][r] = v[0]; x[jj][r + 1] = v[1];
                        ss2 += v * v;
                    }
                float ss = ss2[0] + ss2[1];
                ss += __shfl_xor(ss, 16, 64);                 // the other twelve values of the row's head live in the lanes of the
                ss += __shfl_xor(ss, 32, 64);                 // other three column groups
                const float rinv = do_norm[s] ? rsqrtf(ss * (1.f / 64.f) + p.rms_eps) : 1.f;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    float y[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = x[jj][r] * rinv * rw[4 * s + jj][r];
                    const int b = 4 * s + jj;
                    *(uint2*)(scr + l15 * RB + (((2 * b + (lg >> 1)) ^ (l15 & (NCH - 1))) << 4) + (lg & 1) * 8) = pack4(y);
                }
            }
        } else if constexpr (kResid) {
#pragma unroll
            for (int j = 0; j < 8; ++j)                          // raw fp32 sums: 16 bytes at chunk 4 j + lg
                *(float4*)(scr + l15 * RB + (((4 * j + lg) ^ l15) << 4)) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        } else {
#define DWM4_ACT_PASS(FN_)                                                                                         \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                        \
                float y[4];                                                                                        \
                _Pragma("unroll") for (int r = 0; r < 4; r += 2) {                                                 \
                    const f32x2 x = FN_((f32x2){acc[i][j][r], acc[i][j][r + 1]} + (f32x2){bv[j][r], bv[j][r + 1]}); \
                    y[r] = x[0]; y[r + 1] = x[1];                                                                  \
                }                                                                                                  \
                *(uint2*)(scr + l15 * RB + (((2 * j + (lg >> 1)) ^ (l15 & (NCH - 1))) << 4) + (lg & 1) * 8) = pack4(y); \
            }
            if (p.act == DWM_ACT_GELU_TANH) { DWM4_ACT_PASS(gelu_tanh2) }
            else if (p.act == DWM_ACT_SILU) { DWM4_ACT_PASS(silu2) }
            else if (p.act == DWM_ACT_RELU) { DWM4_ACT_PASS(relu2) }
            else { DWM4_ACT_PASS() }
#undef DWM4_ACT_PASS
        }
        // same-wave LDS operations complete in order: the reads below see the writes above
        // ---- stage B (row-major: 8 columns per lane)
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int r = st * RPS + rrow;                       // row inside this 16-row pass
            const int64_t m = m0 + wm * 128 + i * 16 + r;
            if constexpr (!kResid) {
                const uint4 o = *(const uint4*)(scr + r * RB + ((rc8 ^ (r & (NCH - 1))) << 4));
                if (!GEN || (m < M && nok)) *(uint4*)((bf16_t*)p.C + ((uint64_t)(uint32_t)m * (uint32_t)p.ldc + (uint32_t)ocol)) = o;
            } else {
                const float4 x0 = *(const float4*)(scr + r * RB + (((2 * rc8) ^ r) << 4));
                const float4 x1 = *(const float4*)(scr + r * RB + (((2 * rc8 + 1) ^ r) << 4));
                float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                float t[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] += b8[c];
                if constexpr (f_gate) {
                    unpack8(gA[st], t);
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] *= t[c];
                }
                if constexpr (RF32) {
                    const float4 ta = *reinterpret_cast<const float4*>(&rA[st]), tb = *reinterpret_cast<const float4*>(&rB[st]);
                    t[0] = ta.x; t[1] = ta.y; t[2] = ta.z; t[3] = ta.w; t[4] = tb.x; t[5] = tb.y; t[6] = tb.z; t[7] = tb.w;
                } else {
                    unpack8(rA[st], t);
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] += t[c];
                if constexpr (f_blend) {
                    if constexpr (RF32) {
                        const float4 ta = *reinterpret_cast<const float4*>(&blA[st]), tb = *reinterpret_cast<const float4*>(&blB[st]);
                        t[0] = ta.x; t[1] = ta.y; t[2] = ta.z; t[3] = ta.w; t[4] = tb.x; t[5] = tb.y; t[6] = tb.z; t[7] = tb.w;
                    } else {
                        unpack8(blA[st], t);
                    }
                    const float al = alA[st];
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = al * t[c] + (1.f - al) * v[c];
                }
                if (!GEN || (m < M && nok)) {
                    if constexpr (RF32) {
                        float* o32 = (float*)p.C32 + ((uint64_t)(uint32_t)m * (uint32_t)p.ldc32 + (uint32_t)ocol);
                        *(float4*)o32 = make_float4(v[0], v[1], v[2], v[3]);
                        *(float4*)(o32 + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    } else {
                        *(uint4*)((bf16_t*)p.C + ((uint64_t)(uint32_t)m * (uint32_t)p.ldc + (uint32_t)ocol)) = pack8(v);
                    }
                }
            }
        }
        if (i + 1 < 8) issue_resid(i + 1);
    }
}

// launches served by this file in this process (dwm_gemm4w_launches) and, of them, in the general form: diagnostics counters, the only
// process-wide state of this file - atomic (relaxed), any thread may launch (autograd's backward thread does)
std::atomic<int64_t> g_launches{0};
std::atomic<int64_t> g_launches_gen{0};

template <int EPI, bool RF32, int RS, bool GEN = false>
int launch4w(const dwm_gemm_args* a, const G4Params& gp, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        const hipError_t e = hipFuncSetAttribute((const void*)gemm4w_kernel<EPI, RF32, RS, GEN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL((gemm4w_kernel<EPI, RF32, RS, GEN>), dim3((unsigned)(gp.ntm * gp.ntn)), dim3(256), LDS_BYTES, s, *a, gp);
    g_launches.fetch_add(1, std::memory_order_relaxed);
    if (GEN) g_launches_gen.fetch_add(1, std::memory_order_relaxed);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DWM_OK : (int)e;
}

}  // namespace

// Called by dwm_gemm_bf16 (gemm_bf16.hip) after its argument validation when the caller (tile == 3) or DWM_GEMM4W asks for it.
// Returns -1 if this launch is not one the 4-wave kernels cover (the caller then continues with the 8-wave kernels), otherwise the
// launch status.
//   fast form: no row maps, no taps, no split-K, M % 256 == N % 256 == K % 64 == 0.
//   general form (GEN; validated on the GPU in round 5, profiles/r5a_*: 22-call battery against fp64 / conv2d and the 8-wave kernels,
//   393.8 -> 389.4 ms per denoise step on one box): ragged M / N (N % 8 == 0; GEGLU / RMSHEAD: N % 64 == 0), an A row map, taps
//   (implicit convolution), the per-image residual row (res_mod < 0) - no OUTPUT row map, no split-K.
int dwm_gemm4w_try(const dwm_gemm_args* a, void* stream, bool fast_only) {
    // fast_only (dwm_gemm_args.tile == 4): the fast form only (A/B measurements of the general form)
    const int64_t lim = 1ll << 31;
    if (a->c_map.rw > 0 || a->split_k > 1 || a->tile == 1 || a->tile == 2) return -1;        // (tile 1 / 2: an 8-wave configuration was asked for)
    if (a->K % BK != 0 || a->K < 2 * BK) return -1;        // (one K step: a known bad corner of the request form)
    const bool row_div = a->epilogue == DWM_EPI_RESID && a->res_mod < 0;
    const bool gen = a->a_map.rw > 0 || a->ntaps > 0 || a->M % BM != 0 || a->N % BN != 0 || row_div;
    if (gen && fast_only) return -1;
    if (a->ldc >= lim || a->ldc32 >= lim || a->ld_res >= lim || a->ld_blend >= lim || a->ld_gate >= lim) return -1;
    // the automatic split-K rule of dwm_gemm_bf16 (small tile grids with a long K) keeps its kernels
    const int ntm = (int)((a->M + BM - 1) / BM), ntn = (int)((a->N + BN - 1) / BN);
    if ((a->epilogue == DWM_EPI_PLAIN || a->epilogue == DWM_EPI_RESID) && a->workspace != nullptr && a->split_k == 0 &&
        (int64_t)ntm * ntn <= 128 && a->K / BK >= 16 && a->C32 == nullptr)
        return -1;
    G4Params gp;
    gp.ntm = ntm; gp.ntn = ntn;
    gp.gm = a->K >= 4096 ? 4 : 8;
    gp.fd_rpg = make_fastdiv((uint32_t)(a->rows_per_gate > 0 ? a->rows_per_gate : 1));
    gp.fd_rpa = make_fastdiv((uint32_t)(a->rows_per_alpha > 0 ? a->rows_per_alpha : 1));
    gp.fd_rmod = make_fastdiv((uint32_t)(a->res_mod < 0 ? -a->res_mod : 1));
    gp.amap.enabled = 0; gp.amap.xstep = 1; gp.amap.rw = gp.amap.rh = make_fastdiv(1); gp.amap.rpitch = gp.amap.ipitch = gp.amap.origin = 0;
    gp.steps_per_tap = (int)(a->K / BK);
    gp.a_base_rows = 0;
    for (int t = 0; t < 27; ++t) gp.tap_off[t] = 0;
    if (!gen) {
        if (a->lda * 2 * BM >= lim || a->K * 2 * BN >= lim || a->lda * 2 >= (1ll << 28)) return -1;            // 32-bit request offsets
    } else {
        // anything gemm_bf16.hip would reject is left to it (its error codes)
        if (a->M <= 0 || a->N <= 0 || a->N % 8 != 0) return -1;
        if ((a->epilogue == DWM_EPI_GEGLU || a->epilogue == DWM_EPI_RMSHEAD) && a->N % 64 != 0) return -1;
        if (a->M >= (1ll << 30) || a->res_mod < -(1ll << 30) || a->rows_per_gate > (1ll << 30) || a->rows_per_alpha > (1ll << 30)) return -1;
        const int ntaps = a->ntaps > 0 ? a->ntaps : 1;
        const int64_t kpt = a->ntaps > 0 ? a->k_per_tap : a->K;
        if (ntaps > 27 || kpt <= 0 || kpt % BK != 0 || kpt * ntaps != a->K || a->lda < kpt) return -1;
        gp.steps_per_tap = (int)(kpt / BK);
        if (a->a_map.rw > 0) {
            const dwm_rowmap2d& r = a->a_map;
            if (r.rh <= 0 || r.rw >= (1ll << 30) || r.rh >= (1ll << 30)) return -1;
            gp.amap.enabled = 1;
            gp.amap.xstep = r.xstep > 0 ? (int)r.xstep : 1;
            gp.amap.rw = make_fastdiv((uint32_t)r.rw); gp.amap.rh = make_fastdiv((uint32_t)r.rh);
            gp.amap.rpitch = r.rpitch; gp.amap.ipitch = r.ipitch; gp.amap.origin = r.origin;
        }
        // rows of A the launch reaches, counted from the row of the smallest tap shift: every 32-bit offset must stay below 2^31
        int64_t smin = 0, smax = 0;
        for (int t = 0; t < ntaps && a->ntaps > 0; ++t) {
            smin = a->tap_shift[t] < smin ? a->tap_shift[t] : smin;
            smax = a->tap_shift[t] > smax ? a->tap_shift[t] : smax;
        }
        int64_t last = a->M - 1;                             // the map is monotone: its largest row is the last pixel's
        if (a->a_map.rw > 0) {
            const dwm_rowmap2d& r = a->a_map;
            const int64_t q = last / r.rw, x = last % r.rw, img = q / r.rh, y = q % r.rh;
            last = img * r.ipitch + y * r.rpitch + x * (r.xstep > 0 ? r.xstep : 1) + r.origin;
        }
        if (last < 0 || (last + smax - smin + 1) * a->lda * 2 + a->K * 2 >= lim || a->N * a->K * 2 >= lim) return -1;
        gp.a_base_rows = smin;
        for (int t = 0; t < ntaps && a->ntaps > 0; ++t) gp.tap_off[t] = (uint32_t)((a->tap_shift[t] - smin) * a->lda * 2);
    }
    hipStream_t s = (hipStream_t)stream;
#define DWM4_GO(...) return gen ? launch4w<__VA_ARGS__, true>(a, gp, s) : launch4w<__VA_ARGS__, false>(a, gp, s)
    switch (a->epilogue) {
        case DWM_EPI_PLAIN:
            if (a->C32 != nullptr) return -1;
            DWM4_GO(DWM_EPI_PLAIN, false, 0);
        case DWM_EPI_GEGLU:
            DWM4_GO(DWM_EPI_GEGLU, false, 0);
        case DWM_EPI_RMSHEAD:
            DWM4_GO(DWM_EPI_RMSHEAD, false, 0);
        case DWM_EPI_RESID: {
            if (a->res_mod > 0 || a->act != DWM_ACT_NONE || a->res == nullptr) return -1;
            const int rs = (a->gate ? 1 : 0) | 2 | (a->blend ? 4 : 0) | (row_div ? 16 : 0);
            if (a->C32 != nullptr) {
                if (a->C != nullptr || row_div) return -1;        // (the form with the bf16 copy stays on the 8-wave kernel)
                if (rs == 2) DWM4_GO(DWM_EPI_RESID, true, 2);
                if (rs == 3) DWM4_GO(DWM_EPI_RESID, true, 3);
                if (rs == 6) DWM4_GO(DWM_EPI_RESID, true, 6);
                return -1;
            }
            if (rs == 2) DWM4_GO(DWM_EPI_RESID, false, 2);
            if (rs == 3) DWM4_GO(DWM_EPI_RESID, false, 3);
            if (rs == 6) DWM4_GO(DWM_EPI_RESID, false, 6);
            if (rs == 18) return launch4w<DWM_EPI_RESID, false, 18, true>(a, gp, s);
            return -1;
        }
        default: return -1;
    }
#undef DWM4_GO
}

extern "C" int64_t dwm_gemm4w_launches(void) { return g_launches.load(std::memory_order_relaxed); }
extern "C" int64_t dwm_gemm4w_launches_general(void) { return g_launches_gen.load(std::memory_order_relaxed); }
