// bf16 GEMM with fused epilogues for gfx950:  C[M,Nout] = epi(A[M,K] · W[N,K]^T)
//
// Block tile 256x256x64, 512 threads = 8 waves laid out 2 (M) x 4 (N); each wave
// owns a 128x64 output tile = 4x2 v_mfma_f32_32x32x16_bf16 accumulators (128 fp32
// VGPR/lane).  The MFMA is issued "swapped" (A-operand = W fragment, B-operand =
// activation fragment) so a lane owns ONE output row m and, per accumulator
// register group, FOUR CONSECUTIVE output columns n:
//     m = lane & 31,  n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)      (r = 0..15)
// That makes every epilogue lane-local: 8-byte bf16x4 stores, GEGLU pairs
// (value / gate sub-tiles of the same wave), and the per-head (64 col) RMSNorm of
// q/k needs one lane^32 exchange.
//
// Global -> LDS staging is the gfx950 LDS-DMA (global_load_lds_dwordx4): the LDS
// image of a tile is [256 rows][8 x 16-B chunks] with chunk XOR-swizzled by
// ((row >> 1) & 7); the swizzle is applied on the per-lane SOURCE address (the DMA
// destination is lane-linear) and again on the ds_read_b128 fragment reads, which
// makes those reads bank-conflict free.  The whole 160 KiB LDS of the CU is the operand ring: three
// stages for the activation tile (requested two K steps ahead: it is the operand that streams from HBM
// when K is long) and two for the weight tile (one step ahead, L2-resident); one barrier per K step with a
// counted vmcnt that leaves the newest activation requests in flight.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "dwm_hip.h"

namespace {

// dwm_gemm_args.reserved carries ablation / tuning knobs of development builds (-DDWM_DEV_HOOKS: bit 0 main loop without
// epilogue, bit 1 no stores, bit 2 general instead of FAST kernels, bits 4-8 group height, bit 9 the 256 x 128 tile, bit 10 pad its LDS); the shipped object ignores it
#ifdef DWM_DEV_HOOKS
#define DWM_RESERVED(x_) (x_)
#define DWM_DEV_LDS_PAD 16384      /* bit 10: this much unused LDS on top, so that only ONE 256 x 128 workgroup fits a CU */
#else
#define DWM_RESERVED(x_) 0
#define DWM_DEV_LDS_PAD 0
#endif

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;            // 32 KiB per operand tile
constexpr int A_STAGES = 3, W_STAGES = 2;          // the activation operand is requested two K steps ahead, the weights one
constexpr int W_BASE = A_STAGES * TILE_BYTES;
constexpr int LDS_BYTES = (A_STAGES + W_STAGES) * TILE_BYTES;   // 160 KiB: the whole LDS of a CU

// Tile configurations of gemm_bf16_kernel (template parameter TC).
//   0: the 256 x 256 x 64 tile described above (the constants above), one 8-wave workgroup per CU.
//   1: 256 x 128 x 32, four waves laid out 2 x 2 with the SAME 128 x 64 wave tile (so the epilogues are shared line for line),
//      three stages of each operand = 72 KiB and 256 VGPRs per wave: TWO workgroups per CU.  They run out of phase, so one
//      workgroup's epilogue (VALU / LDS transpose / stores, no MFMA) sits under the other one's main loop, and a workgroup
//      waiting at its K-step barrier leaves the matrix pipe to the other.  Costs: 1.5 x the LDS-DMA bytes per flop and
//      one barrier per 16 instead of 32 MFMAs.  LDS image: [rows][4 x 16-B chunks], chunk XOR-swizzled by ((row >> 2) & 3)
//      (16 consecutive rows x one logical chunk cover the 64 banks exactly once).  Also the better fit for N = 320 / 640
//      (SD 2.1 UNet levels): 17 % / 0 % padded columns instead of 37 % / 17 %.
template <int TC> struct TileCfg;
template <> struct TileCfg<0> { static constexpr int bm = 256, bn = 256, bk = 64, nwn = 4, nwaves = 8, ast = A_STAGES, wst = W_STAGES; };
template <> struct TileCfg<1> { static constexpr int bm = 256, bn = 128, bk = 32, nwn = 2, nwaves = 4, ast = 3, wst = 3; };
template <int TC> constexpr int tile_lds_bytes() { return (TileCfg<TC>::ast * TileCfg<TC>::bm + TileCfg<TC>::wst * TileCfg<TC>::bn) * TileCfg<TC>::bk * 2; }
static_assert(tile_lds_bytes<0>() == LDS_BYTES && 2 * tile_lds_bytes<1>() <= 160 * 1024, "LDS budget");

struct DevRowMap {
    FastDiv rw, rh;
    int64_t rpitch, ipitch, origin;
    int enabled, xstep;
};
struct ConvParams {
    DevRowMap a, c;
    int steps_per_tap;          // k_per_tap / BK
    FastDiv fd_steps;
    FastDiv fd_rpg, fd_rmod, fd_rpa;   // RESID: rows_per_gate, res_mod, rows_per_alpha
    int64_t tap_shift[27];
    // split-K: the K steps are cut into `ksplit` contiguous ranges, one workgroup per (tile, range); range s
    // writes its fp32 partial tile to ws + s * ws_slice ([M, N] row-major) and splitk_finish_kernel reduces
    int ksplit;
    float* ws;
    int64_t ws_slice;
    // tile rasterisation: groups of `gm` row tiles x all column tiles (fast divisors prepared on the host)
    int gm;
    FastDiv fd_pergroup, fd_gm;
};
// group height: 8 row tiles share a W panel in the XCD's L2 for K ~ 1.5 k; a long K (FF2: 6144) makes the A panel of
// 8 rows (25 MB) stream through it, 4 rows measured 3 % faster there
static inline void set_raster(ConvParams& cp, const dwm_gemm_args& a, int ntn) {
    int gm = (DWM_RESERVED(a.reserved) >> 4) & 31;
    if (gm == 0) gm = a.K >= 4096 ? 4 : 8;
    cp.gm = gm;
    cp.fd_pergroup = make_fastdiv((uint32_t)(gm * ntn));
    cp.fd_gm = make_fastdiv((uint32_t)gm);
}
constexpr int EPI_SPLITK = 100;     // internal epilogue id: fp32 partials to the workspace
DWM_DEVINL int64_t map_row(const DevRowMap& rm, int64_t m) {
    if (!rm.enabled) return m;
    const uint32_t q = fdiv((uint32_t)m, rm.rw), x = (uint32_t)m - q * rm.rw.d;
    const uint32_t i = fdiv(q, rm.rh), y = q - i * rm.rh.d;
    return (int64_t)i * rm.ipitch + (int64_t)y * rm.rpitch + (int64_t)x * rm.xstep + rm.origin;
}

// FAST: the linear layers of the transformer blocks - no output row map, every leading dimension < 2^31, and for RESID:
// residual row = output row, no activation - with those facts known at compile time: the per-step address arithmetic is
// one 32 x 32 -> 64 bit multiply-add per pointer instead of the row-map / modulo chains in 64-bit arithmetic, and the RESID
// activation switch is gone (the general RESID form spends ~350 instructions per 8-row step, 44 per output value).
// RF32 (RESID, both forms): the residual stream is kept in fp32 - `res` (and `blend`) are fp32 matrices (leading dimensions in
// fp32 elements), the result is written to `C32` in fp32 (in place over `res` / `blend` is fine: a lane reads exactly the
// elements it writes) and, when C is given, rounded to the bf16 matrix C as well.  For chains of residual blocks whose bf16
// storage rounding would otherwise accumulate block after block: the layout ImageAdapter (12 resnets, step-invariant input,
// hence a step-invariant error) and - round 4 - the hidden state of the MMDiT itself (~130 residual adds per forward, each a
// bf16 rounding of the whole stream: 1.1e-3 rms x sqrt(130) = the 1.3e-2 the bf16 forward showed against the fp32 oracle).
// RS (RESID): 0 = which of gate / residual / blend / bf16 mirror are present is read from the arguments at run time; otherwise a
// bit mask (1 gate, 2 residual, 4 blend, 8 bf16 mirror, 16 the residual row is m / |res_mod|: one row per image, the time-embedding
// term of the SD 2.1 UNet's resnets) known at compile time - the forms of the hidden-state stream of the
// MMDiT (gate + residual: the joint blocks; residual: inside a VT block; residual + blend: a VT block's last GEMM): the
// run-time form spends a third of its ~2000 epilogue instructions per wave on the selects and branches between the cases.
template <int EPI, bool FAST = false, bool RF32 = false, int TC = 0, int RS = 0>
__global__ void __launch_bounds__(TileCfg<TC>::nwaves * 64, 2)
gemm_bf16_kernel(const dwm_gemm_args p, const ConvParams cp, const int ntm, const int ntn) {
    using T = TileCfg<TC>;
    static_assert(RS == 0 || (EPI == DWM_EPI_RESID && FAST), "compile-time RESID operands: the FAST RESID kernels only");
    constexpr int BM = T::bm, BN = T::bn, BK = T::bk;        // (shadow the file-level constants of configuration 0)
    constexpr int NWN = T::nwn;                  // wave columns of the 2 x NWN wave grid
    constexpr int WCOLS = BN / NWN;              // output columns per wave: 64
    constexpr int NTW = WCOLS / 32;              // 32-column accumulator tiles per wave: 2
    constexpr int ROWB = BK * 2;                 // bytes per row of an LDS operand tile: 128 / 64
    constexpr int A_TILE = BM * ROWB, W_TILE = BN * ROWB;
    constexpr int A_STAGES = T::ast, W_BASE = A_STAGES * A_TILE;
    constexpr int NJA = A_TILE / (T::nwaves * 1024), NJW = W_TILE / (T::nwaves * 1024);   // 1-KiB LDS-DMA requests per wave and stage
    constexpr int RPG = 1024 / ROWB, CPR = ROWB / 16;        // rows per request, 16-B chunks per row
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN;             // 0..1  (128-row slab)
    const int wn = wave % NWN;             // column slab of WCOLS
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int64_t tap_raw = cp.tap_shift[lane < 27 ? lane : 0];        // lane t: row shift of tap t (see the K walk below)

    // XCD-contiguous ids, then grouped rasterisation: consecutive ids walk GM row-tiles down a
    // column before moving to the next column, so the ~32 tiles resident on one XCD at a time form
    // a compact GM x (32/GM) block that shares A and W panels in that XCD's L2.
    int id = xcd_remap(blockIdx.x, ntm * ntn * (EPI == EPI_SPLITK ? cp.ksplit : 1));
    int slice = 0;
    if constexpr (EPI == EPI_SPLITK) {          // slice-major: consecutive ids stay tile neighbours of one K range
        slice = id / (ntm * ntn);
        id -= slice * (ntm * ntn);
    }
    const int gm_ = cp.gm;
    const int per_group = gm_ * ntn;
    const int grp_id = (int)fdiv((uint32_t)id, cp.fd_pergroup), in_grp = id - grp_id * per_group;
    const int first_m = grp_id * gm_;
    const int gsize = ntm - first_m < gm_ ? ntm - first_m : gm_;
    int tm, tn;
    if (gsize == gm_) {                                  // (all groups but a ragged last one)
        tn = (int)fdiv((uint32_t)in_grp, cp.fd_gm);
        tm = first_m + in_grp - tn * gm_;
    } else {
        tn = in_grp / gsize;
        tm = first_m + in_grp % gsize;
    }
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int64_t M = p.M, N = p.N, K = p.K;

    // ---- staging addresses: wave w copies row groups (w*NJ + j)*8 .. +8, j = 0..NJ-1
    const bf16_t* __restrict__ Ap = (const bf16_t*)p.A;
    const bf16_t* __restrict__ Wp = (const bf16_t*)p.W;
    const char* a_src[NJA];
    const char* w_src[NJW];
    auto chunk_swz = [](int row) { return TC == 0 ? (row >> 1) & 7 : (row >> 2) & 3; };
#pragma unroll
    for (int j = 0; j < NJA; ++j) {
        const int row = (wave * NJA + j) * RPG + lane / CPR;
        const int c = (lane % CPR) ^ chunk_swz(row);          // logical 16-B chunk this lane fetches
        int64_t gm = m0 + row; gm = gm < M ? gm : M - 1;
        a_src[j] = (const char*)(Ap + map_row(cp.a, gm) * p.lda + c * 8);
    }
#pragma unroll
    for (int j = 0; j < NJW; ++j) {
        const int row = (wave * NJW + j) * RPG + lane / CPR;
        const int c = (lane % CPR) ^ chunk_swz(row);
        int64_t gn = n0 + row; gn = gn < N ? gn : N - 1;
        w_src[j] = (const char*)(Wp + gn * K + c * 8);
    }
    // K step kt covers tap t = kt / steps_per_tap and channels (kt % steps_per_tap)*64.. of it; the A
    // source moves by tap_shift[t] rows (0 for a plain GEMM), the W source is simply contiguous in K
    const int64_t lda_bytes = p.lda * 2;
    const int nk_all = (int)(K / BK);
    const int kt0 = EPI == EPI_SPLITK ? (int)((int64_t)slice * nk_all / cp.ksplit) : 0;
    const int nk = EPI == EPI_SPLITK ? (int)((int64_t)(slice + 1) * nk_all / cp.ksplit) - kt0 : nk_all;
    // The walk over K: wave-uniform byte offsets of the A / W sources of one K step, advanced step by step.  Inside a tap both
    // simply move on by one tile; the tap table is read at tap boundaries only (a plain GEMM has none), so no K step waits
    // for a scalar load and the per-lane source address is ONE 64-bit add per request.
    int64_t walk_a, walk_w;
    int walk_left, walk_tap;                           // K steps left in this tap (this one included), tap index
    // lane t keeps the byte shift of tap t; a tap boundary fetches it with v_readlane.  (A scalar load anywhere in the loop -
    // even on this rare path - makes the compiler wait for ALL outstanding LDS reads at every use of a fragment: scalar
    // loads return out of order on the counter they share with the LDS, so no counted lgkmcnt wait is possible any more.)
    int64_t tap_bytes = tap_raw * lda_bytes;
    // (the table load was issued at the top of the kernel; it is consumed HERE, before the first LDS-DMA request, so that
    // the compiler's wait for it cannot end up behind those requests - loads return in order, it would drain them)
    asm volatile("" : "+v"(tap_bytes));
    auto tap_offset = [&](int tap) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tap_bytes, tap);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)tap_bytes >> 32), tap);
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    auto walk_init = [&]() {
        const uint32_t tap = fdiv((uint32_t)kt0, cp.fd_steps);
        const int r = kt0 - (int)tap * cp.steps_per_tap;
        walk_tap = (int)tap;
        walk_left = cp.steps_per_tap - r;
        walk_w = (int64_t)kt0 * (BK * 2);
        walk_a = cp.tap_shift[tap] * lda_bytes + (int64_t)r * (BK * 2);     // (scalar load: the per-lane table may still be in flight)
    };
    auto walk_next = [&]() {
        walk_w += BK * 2;
        if (--walk_left == 0) {
            ++walk_tap;
            walk_left = cp.steps_per_tap;
            walk_a = tap_offset(walk_tap);
        } else {
            walk_a += BK * 2;
        }
    };

    f32x16 acc[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets (bytes) inside a tile: row*128 + ((chunk ^ swz) << 4)
    const int swz = chunk_swz(l31);        // rows of a fragment are 32*t + (lane & 31)
    const int a_row_off = (wm * 128 + l31) * ROWB;
    const int w_row_off = (wn * WCOLS + l31) * ROWB;
    constexpr int NKS = BK / 16;           // MFMA K steps per tile: 4 / 2
    int coff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) coff[ks] = ((2 * ks + half) ^ swz) << 4;


    // ---- main loop.  Per K tile: 4 sub-steps x 8 chunks of { NTW/2 MFMAs, one fragment read for the
    // next sub-step, a share of the DMA issue }, fenced with sched_barrier so LDS reads and DMA
    // issues sit in the shadow of the MFMAs.  The single barrier of a tile sits at the START of its
    // last sub-step: by then every wave holds its last fragments of tile kt in registers (the weight slot
    // kt&1 is free for tile kt+2) and its shares of tile kt+1 have landed, so the fragment reads of
    // tile kt+1's first sub-step and the barrier skew hide under the MFMAs of sub-step 3.
    // The LDS-DMA requests go through inline asm.  The compiler models global_load_lds as a FLAT access that may touch the LDS,
    // and with one of those "pending" it degrades every later LDS wait to lgkmcnt(0): each sub-step then began by draining all
    // six fragment reads, the newest issued two MFMAs earlier.  Opaque requests keep its LDS bookkeeping exact (counted
    // lgkmcnt waits); their vmcnt bookkeeping is done by hand here anyway (the counted waits at the barriers), and the
    // compiler's own vmcnt waits stay safe because loads return in order and no tracked load is older than an untracked one
    // that it must not wait for.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    auto glds = [&](const char* src, uint32_t lds_off) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds0 + lds_off) : "memory", "m0");
    };
    auto stage_a = [&](int buf, int64_t aoff, int j) {
        glds(a_src[j] + aoff, (uint32_t)(buf * A_TILE + (wave * NJA + j) * 1024));
    };
    auto stage_w = [&](int buf, int64_t koff, int j) {
        glds(w_src[j] + koff, (uint32_t)(W_BASE + buf * W_TILE + (wave * NJW + j) * 1024));
    };
    constexpr int MPC = NTW / 2;                       // MFMAs per chunk
    constexpr int NDS = 4 + NTW;                       // fragment reads per sub-step
    bf16x8 af[2][4], wf[2][NTW];
    if constexpr (TC == 0) {
        // Request order per wave (loads return in order, so the barrier's counted wait follows it):
        //   ... A(kt+1) [sub-step 0 of step kt-1], W(kt+1) [sub-step 3 of step kt-1], A(kt+2) [sub-step 0 of step kt] ...
        // The barrier of step kt needs A(kt+1) and W(kt+1): vmcnt(NJA) leaves exactly the NJA requests of A(kt+2) in flight,
        // which therefore have almost two K steps to arrive (the activation panel is the operand that streams from HBM);
        // the weights have one step (they are shared by every row tile and sit in L2).
        int sa = 0;                                        // A stage of tile kt (kt % 3)
        {
            walk_init();
            int64_t aoff = walk_a, koff = walk_w;
#pragma unroll
            for (int j = 0; j < NJA; ++j) { stage_a(0, aoff, j); stage_w(0, koff, j); }
            // tile 1 is requested right behind tile 0, before the first wait: it then has the whole round trip of tile 0 plus
            // most of K step 0 to arrive, instead of starting its own round trip only after tile 0 has landed
            if (nk > 1) {
                walk_next();
                aoff = walk_a; koff = walk_w;
#pragma unroll
                for (int j = 0; j < NJA; ++j) stage_a(1, aoff, j);
#pragma unroll
                for (int j = 0; j < NJA; ++j) stage_w(1, koff, j);
            }
            // (the accumulators are zeroed HERE, under the round trip of the first requests: left to itself the compiler
            // sinks the 128 moves behind the barrier, ~1 us of a 45 us tile with nothing else to run)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) asm volatile("" : "+v"(acc[i][j]));
            if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJA) : "memory");   // tile 0 landed (loads return in order)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[0][mt] = *(const bf16x8*)(smem + a_row_off + mt * (32 * 128) + coff[0]);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) wf[0][nt] = *(const bf16x8*)(smem + W_BASE + w_row_off + nt * (32 * 128) + coff[0]);
        }
        // One K step.  MODE 0: tiles kt+1 and kt+2 exist (the steady state described above).  MODE 1: the second-to-last step -
        // nothing is requested any more, so the barrier's wait is vmcnt(0) (tile kt+1, the last, must have landed).  MODE 2: the
        // last step - no requests, no barrier, no fragment reads of a next tile.  (Until round 3 the tail re-requested the last
        // tile twice to keep the counted waits uniform: 8 % of a K = 1536 tile's LDS-DMA traffic for nothing, and the wait for the
        // final redundant request - a full L2 round trip - sat between the last MFMA and the epilogue.)
        int kt = 0;
        auto k_step = [&](auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            const int sa1 = sa == A_STAGES - 1 ? 0 : sa + 1, sa2 = sa1 == A_STAGES - 1 ? 0 : sa1 + 1;
            const char* la = smem + sa * A_TILE;
            const char* lb = smem + W_BASE + (kt & 1) * A_TILE;
            const char* lan = smem + sa1 * A_TILE;                  // tile kt+1
            const char* lbn = smem + W_BASE + ((kt + 1) & 1) * A_TILE;
            if constexpr (MODE == 0) walk_next();                      // tile kt+2
            const int64_t aoff2 = walk_a, koff2 = walk_w;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (ks == 3 && c == 0 && MODE != 2) {
                        // own last fragments read; own shares of A(kt+1) and W(kt+1) landed (A(kt+2) may stay in flight)
                        // bare s_barrier, not __syncthreads(): the workgroup-scope release fence of __syncthreads() makes the
                        // compiler append "s_waitcnt vmcnt(0)" (LDS-DMA writes LDS and is tracked by vmcnt), which drains
                        // A(kt+2) at every K step; the counted wait above is the ordering this barrier needs
                        if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NJA) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
#pragma unroll
                    for (int u = 0; u < MPC; ++u) {
                        const int idx = c * MPC + u, mt = idx / NTW, nt = idx % NTW;
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][nt], af[ks & 1][mt], acc[mt][nt], 0, 0, 0);
                    }
                    if (c < NDS && !(MODE == 2 && ks == 3)) {
                        // order of first use in the next sub-step (MFMA c uses A[c / 2], W[c % 2]): W0 A0 W1 A1 A2 A3 - every
                        // read then has eight MFMAs to land and the compiler's counted lgkmcnt waits never drain the queue
                        const char* fa = ks < 3 ? la : lan;
                        const char* fb = ks < 3 ? lb : lbn;
                        const int kn = ks < 3 ? ks + 1 : 0;
                        const bool is_w = c == 0 || c == 2;
                        const int fi = c == 0 ? 0 : c == 1 ? 0 : c == 2 ? 1 : c - 2;
                        if (is_w) wf[(ks + 1) & 1][fi] = *(const bf16x8*)(fb + w_row_off + fi * (32 * 128) + coff[kn]);
                        else af[(ks + 1) & 1][fi] = *(const bf16x8*)(fa + a_row_off + fi * (32 * 128) + coff[kn]);
                    }
                    if (MODE == 0 && ks == 0 && c < NJA) stage_a(sa2, aoff2, c);                       // A(kt+2): the slot tile kt-1 left
                    if (MODE == 0 && ks == 3 && c >= 1 && c <= NJA) stage_w(kt & 1, koff2, c - 1);     // W(kt+2): the slot of this tile
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            sa = sa1;
            ++kt;
        };
        while (kt + 2 < nk) k_step(std::integral_constant<int, 0>{});
        if (nk > 1) k_step(std::integral_constant<int, 1>{});
        k_step(std::integral_constant<int, 2>{});
    } else {
        // Configuration 1 (BK = 32: two MFMA K steps per tile).  Per K step and wave: sub-step 0 = 8 MFMAs on the first
        // half of tile kt while the second-half fragments are read and the 6 requests of tile kt+2 are issued (into the slot
        // tile kt-1 left: every wave passed the barrier of step kt-1 after its last reads of it); sub-step 1 = counted wait
        // (own shares of tile kt+1 landed, tile kt+2 stays in flight) + barrier, then 8 MFMAs on the second half while the
        // first-half fragments of tile kt+1 are read.  Both operands are requested two steps ahead (3 + 3 stages).
        // Fragment read order = first use: W0 A0 W1 A1 A2 A3 (MFMA c uses A[c / 2], W[c % 2]).
        static_assert(TC == 0 || (NJA + NJW <= 8 && NKS == 2 && T::wst == T::ast), "configuration 1 main loop");
        int sa = 0;
        {
            walk_init();
            int64_t aoff = walk_a, koff = walk_w;
#pragma unroll
            for (int j = 0; j < NJA; ++j) stage_a(0, aoff, j);
#pragma unroll
            for (int j = 0; j < NJW; ++j) stage_w(0, koff, j);
            if (nk > 1) walk_next();                                   // (tile 1 right behind tile 0, as in configuration 0)
            aoff = walk_a; koff = walk_w;
#pragma unroll
            for (int j = 0; j < NJA; ++j) stage_a(1, aoff, j);
#pragma unroll
            for (int j = 0; j < NJW; ++j) stage_w(1, koff, j);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) asm volatile("" : "+v"(acc[i][j]));
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJA + NJW) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[0][mt] = *(const bf16x8*)(smem + a_row_off + mt * (32 * ROWB) + coff[0]);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) wf[0][nt] = *(const bf16x8*)(smem + W_BASE + w_row_off + nt * (32 * ROWB) + coff[0]);
        }
        for (int kt = 0; kt < nk; ++kt) {
            const int sa1 = sa == A_STAGES - 1 ? 0 : sa + 1, sa2 = sa1 == A_STAGES - 1 ? 0 : sa1 + 1;
            const char* la = smem + sa * A_TILE;
            const char* lb = smem + W_BASE + sa * W_TILE;
            const char* lan = smem + sa1 * A_TILE;                      // tile kt+1
            const char* lbn = smem + W_BASE + sa1 * W_TILE;
            if (kt + 2 < nk) walk_next();                              // tile kt+2 (past the end: a redundant reload nobody reads)
            const int64_t aoff2 = walk_a, koff2 = walk_w;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (ks == 1 && c == 0) {
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NJA + NJW) : "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                    acc[c >> 1][c & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][c & 1], af[ks][c >> 1], acc[c >> 1][c & 1], 0, 0, 0);
                    if (c < 6) {
                        const char* fa = ks == 0 ? la : lan;
                        const char* fb = ks == 0 ? lb : lbn;
                        const int kn = ks ^ 1;                         // sub-step 0 reads the second half of this tile, 1 the first half of the next
                        const bool is_w = c == 0 || c == 2;
                        const int fi = c == 0 ? 0 : c == 1 ? 0 : c == 2 ? 1 : c - 2;
                        if (is_w) wf[kn][fi] = *(const bf16x8*)(fb + w_row_off + fi * (32 * ROWB) + coff[kn]);
                        else af[kn][fi] = *(const bf16x8*)(fa + a_row_off + fi * (32 * ROWB) + coff[kn]);
                    }
                    if (ks == 0 && c < NJA) stage_a(sa2, aoff2, c);
                    if (ks == 0 && c >= NJA && c < NJA + NJW) stage_w(sa2, koff2, c - NJA);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            sa = sa1;
        }
    }
    // bias (and, RMSHEAD, per-column norm weights) of this lane's 2 x 16 columns in the MFMA layout, requested NOW so that
    // the round trip overlaps the drain of the last MFMAs and the barrier below.  All 8 (16) loads are issued back to back,
    // branch-free, from clamped addresses and unpacked behind counted waits: a per-load `if (n < N)` made every load its own
    // exec-masked block with `s_waitcnt vmcnt(0)` behind it - 8 to 16 serialised L2 round trips per tile (8-14 us of a
    // 39 us tile at K = 1536)
    static_assert(NTW == 2, "one 64-column slab per wave");
    const bf16_t* __restrict__ bias = (EPI == EPI_SPLITK || EPI == DWM_EPI_RESID) ? nullptr : (const bf16_t*)p.bias;
    const bool hb = bias != nullptr;                       // wave-uniform
    bool do_norm = false;
    uint2 braw[2][4], rraw[2][4];
    bool nv[2][4];
    {
        const int64_t nw = n0 + wn * WCOLS;
        const bf16_t* __restrict__ bsrc = hb ? bias : (const bf16_t*)p.W;            // any valid, 8-byte aligned address
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int64_t n = nw + nt * 32 + rg * 8 + half * 4;
                nv[nt][rg] = n < N;
                braw[nt][rg] = *(const uint2*)(bsrc + ((hb && nv[nt][rg]) ? n : 0));
            }
        if constexpr (EPI == DWM_EPI_RMSHEAD) {
            do_norm = nw < p.rms_ncols;                        // wave-uniform: this slab is a q/k head
            const bf16_t* __restrict__ rsrc = do_norm ? (const bf16_t*)p.rms_w : (const bf16_t*)p.W;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int64_t n = nw + nt * 32 + rg * 8 + half * 4;
                    rraw[nt][rg] = *(const uint2*)(rsrc + ((do_norm && nv[nt][rg]) ? n : 0));
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the redundant last DMA must not land in the epilogue scratch

    // ------------------------------------------------------------------ epilogue
    // Stage A (MFMA layout, lane = one row x 4-column groups): bias, activation, GEGLU product,
    // per-head RMSNorm.  Stage B: each wave transposes its tile through a private, XOR-swizzled
    // 8 KiB LDS region (32 rows per pass: 64 fp32 for RESID / split-K, whose arithmetic follows the
    // transpose; 64 bf16 for the other epilogues) so that gate / residual / blend loads and
    // the bf16 stores are row-major 16-B accesses (8 rows x 128 B per wave instruction).
    if (DWM_RESERVED(p.reserved) & 1) {        // ablation knob (development builds only): main loop without the epilogue
        float sink = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) sink += acc[i][j][0] + acc[i][j][7] + acc[i][j][15];
        if (sink == 123.456f) ((float*)(p.C ? p.C : p.C32))[0] = sink;
        return;
    }
    bf16_t* __restrict__ Cp = (bf16_t*)p.C;

    __syncthreads();                                       // every wave is done with the operand tiles
    char* scr = smem + wave * 8192;                        // this wave's transpose region
    // a wave's WCOLS columns are handled as NTW/2 slabs of 64 (one q/k head, one GEGLU value/gate
    // pair, one 8 KiB transpose pass each)
#pragma unroll
    for (int ch = 0; ch < NTW / 2; ++ch) {
        const int64_t nw = n0 + wn * WCOLS + ch * 64;         // first column of this slab
        // bias (and, RMSHEAD, the per-column norm weights) for this lane's 2 x 16 columns (MFMA layout).  All 8 (16) loads are
        // issued back to back, branch-free, from clamped addresses, and unpacked afterwards behind ONE wait: a per-load
        // `if (n < N)` makes every load its own exec-masked block with `s_waitcnt vmcnt(0)` behind it - 8 to 16 serialised L2
        // round trips per tile (measured: 8-14 us of a 39 us tile at K = 1536)
        float bv[2][16];
        float rw[2][16];                                       // RMSHEAD: per-column norm weights
        {
            // absent values as bit masks on the raw words (0.0 for the bias, bf16 1.0 for the norm weight): no control flow
    #pragma unroll
            for (int nt = 0; nt < 2; ++nt)
    #pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const uint32_t mb = (hb && nv[nt][rg]) ? 0xffffffffu : 0u;
                    uint2 b = braw[nt][rg];
                    b.x &= mb; b.y &= mb;
                    unpack4(b, &bv[nt][rg * 4]);
                    if constexpr (EPI == DWM_EPI_RMSHEAD) {
                        const uint32_t mr = (do_norm && nv[nt][rg]) ? 0xffffffffu : 0u;
                        uint2 r = rraw[nt][rg];
                        r.x = (r.x & mr) | (~mr & 0x3f803f80u); r.y = (r.y & mr) | (~mr & 0x3f803f80u);
                        unpack4(r, &rw[nt][rg * 4]);
                    }
                }
        }

        constexpr bool kGeglu = EPI == DWM_EPI_GEGLU;
        constexpr int CW = kGeglu ? 32 : 64;                   // output columns of this wave's slab
        const int64_t ncol0 = kGeglu ? (n0 >> 1) + wn * (WCOLS / 2) + ch * 32 : nw;
        const int64_t Nout = kGeglu ? (N >> 1) : N;
        // stage-B lane geometry: 8 fp32 (two 16-B chunks) per lane; LPR lanes per row
        constexpr int LPR = CW / 8;                            // 8 (or 4 for GEGLU)
        constexpr int RPS = 64 / LPR;                          // rows per step: 8 (16)
        const int brow = lane / LPR, bc8 = lane % LPR;
        const int64_t ncol = ncol0 + bc8 * 8;
        const bool nok = ncol < Nout;

        // RESID operand rows of one 32-row pass: gate (or, without a gate, blend), residual, alpha.  One register set,
        // refilled in place: as soon as step st of pass mt has consumed its rows, the rows of step st of pass mt + 1 are
        // requested into the same registers, so four steps' worth of loads are always in flight behind the math and
        // the stores.  (Plain arrays + a macro: a second set, or structs behind lambda references, end up in scratch.)
        constexpr int NSTEP = 32 / RPS;
        uint4 gbA[NSTEP], gbB[RF32 ? NSTEP : 1], rA[NSTEP], rB[RF32 ? NSTEP : 1];      // (gbB / rB: second half of an fp32 row piece)
        float alA[NSTEP];
        // branch-free: always two 16-byte loads and one alpha load per step (absent operands read a valid dummy row of C),
        // so the compiler's counted waits stay exact and never degrade to "everything outstanding, stores included"
        // (RF32: C may be absent - the dummy is then C32 with its own pitch, read as if it were bf16: in bounds a fortiori;
        //  the blend rows are fp32 and travel as two 16-byte pieces, gbA / gbB; the gate stays bf16)
        constexpr bool kSpec = RS != 0;                 // operands known at compile time (see RS above)
        const bool f_gate = kSpec ? (RS & 1) != 0 : p.gate != nullptr;
        const bool f_res = kSpec ? (RS & 2) != 0 : p.res != nullptr;
        const bool f_blend = kSpec ? (RS & 4) != 0 : p.blend != nullptr;
        const bool f_mirror = kSpec ? (RS & 8) != 0 : p.C != nullptr;
        // run-time form: every load below is always issued (see "branch-free" above); compile-time form: only what is used
        const bool ld_gb = kSpec ? (f_gate || f_blend) : true, ld_gb2 = kSpec ? f_blend : true, ld_r = kSpec ? f_res : true,
                   ld_al = kSpec ? f_blend : true;
        const bf16_t* const dummy = (RF32 && !f_mirror) ? (const bf16_t*)p.C32 : (const bf16_t*)p.C;
        const int64_t dummy_ld = (RF32 && !f_mirror) ? p.ldc32 : p.ldc;
        const bool blend32 = RF32 && f_blend;
        const bf16_t* const gb_ptr = f_gate ? (const bf16_t*)p.gate : (f_blend && !RF32) ? (const bf16_t*)p.blend : dummy;
        const int64_t gb_ld = f_gate ? p.ld_gate : (f_blend && !RF32) ? p.ld_blend : dummy_ld;
        const float* const bl32_ptr = blend32 ? (const float*)p.blend : (const float*)p.C32;
        const int64_t bl32_ld = blend32 ? p.ld_blend : p.ldc32;
        const bf16_t* const r_ptr = f_res ? (const bf16_t*)p.res : dummy;
        const int64_t r_ld = f_res ? p.ld_res : dummy_ld;
        const float* const al_ptr = f_blend ? p.alpha : (const float*)dummy;
        float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // RESID: bias of this lane's 8 output columns
        if constexpr (EPI == DWM_EPI_RESID) {
            if (p.bias != nullptr && nok) unpack8(*(const uint4*)((const bf16_t*)p.bias + ncol), b8);
        }
#define DWM_ISSUE_RESID(MT_, ST_)                                                                                 \
        if constexpr (FAST) {                                                                                     \
            uint32_t m_ = (uint32_t)(m0 + wm * 128 + (MT_) * 32 + (ST_) * RPS + brow);                            \
            m_ = m_ < (uint32_t)M ? m_ : (uint32_t)(M - 1);                                                       \
            const uint32_t nc_ = nok ? (uint32_t)ncol : 0u;                                                       \
            const uint32_t grow_ = f_gate ? fdiv(m_, cp.fd_rpg) : m_;                                             \
            if constexpr (RF32) {       /* (selects, no branches: the load count per step stays fixed) */           \
                const char* g0_ = blend32 ? (const char*)(bl32_ptr + ((uint64_t)m_ * (uint32_t)bl32_ld + nc_))    \
                                          : (const char*)(gb_ptr + ((uint64_t)grow_ * (uint32_t)gb_ld + nc_));    \
                if (ld_gb) gbA[ST_] = *(const uint4*)g0_;                                                         \
                if (ld_gb2) gbB[ST_] = *(const uint4*)(g0_ + (blend32 ? 16 : 0));                                 \
                const float* rp_ = (const float*)(f_res ? p.res : p.C32) + ((uint64_t)m_ * (uint32_t)(f_res ? r_ld : p.ldc32) + nc_); \
                if (ld_r) {                                                                                       \
                    rA[ST_] = *(const uint4*)rp_;                                                                 \
                    rB[ST_] = *(const uint4*)(rp_ + 4);                                                           \
                }                                                                                                 \
            } else {                                                                                              \
                if (ld_gb) gbA[ST_] = *(const uint4*)(gb_ptr + ((uint64_t)grow_ * (uint32_t)gb_ld + nc_));        \
                if (ld_r) rA[ST_] = *(const uint4*)(r_ptr + ((uint64_t)((RS & 16) ? fdiv(m_, cp.fd_rmod) : m_) * (uint32_t)r_ld + nc_)); \
            }                                                                                                     \
            if (ld_al) alA[ST_] = al_ptr[f_blend ? fdiv(m_, cp.fd_rpa) : 0u];                                     \
        } else {                                                                                                  \
            int64_t m_ = m0 + wm * 128 + (MT_) * 32 + (ST_) * RPS + brow;                                         \
            m_ = m_ < M ? m_ : M - 1;                                                                             \
            const int64_t nc_ = nok ? ncol : 0;                                                                   \
            const int64_t mr_ = map_row(cp.c, m_);                                                                \
            const int64_t grow_ = p.gate ? (int64_t)fdiv((uint32_t)m_, cp.fd_rpg) : mr_;                          \
            const int64_t rr_ = !p.res ? mr_ : p.res_mod > 0 ? (int64_t)fmod_u((uint32_t)m_, cp.fd_rmod)          \
                                             : p.res_mod < 0 ? (int64_t)fdiv((uint32_t)m_, cp.fd_rmod) : mr_;     \
            if constexpr (RF32) {                                                                                 \
                const char* g0_ = blend32 ? (const char*)(bl32_ptr + mr_ * bl32_ld + nc_)                         \
                                          : (const char*)(gb_ptr + grow_ * gb_ld + nc_);                          \
                gbA[ST_] = *(const uint4*)g0_;                                                                    \
                gbB[ST_] = *(const uint4*)(g0_ + (blend32 ? 16 : 0));                                             \
                const float* rp_ = (const float*)(p.res ? p.res : p.C32) + rr_ * (p.res ? r_ld : p.ldc32) + nc_;  \
                rA[ST_] = *(const uint4*)rp_;                                                                     \
                rB[ST_] = *(const uint4*)(rp_ + 4);                                                               \
            } else {                                                                                              \
                gbA[ST_] = *(const uint4*)(gb_ptr + grow_ * gb_ld + nc_);                                         \
                rA[ST_] = *(const uint4*)(r_ptr + rr_ * r_ld + nc_);                                              \
            }                                                                                                     \
            alA[ST_] = al_ptr[p.blend ? (int64_t)fdiv((uint32_t)m_, cp.fd_rpa) : 0];                              \
        }
        if constexpr (EPI == DWM_EPI_RESID) {
    #pragma unroll
            for (int st = 0; st < NSTEP; ++st) DWM_ISSUE_RESID(0, st)
        }

    #pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            // ---- stage A (packed fp32: the epilogue is VALU-issue bound; the activation switch is
            // taken once per 32-row pass, not per element)
#define DWM_PAIR(v_, r_) ((f32x2){(v_)[r_], (v_)[(r_) + 1]})
            if constexpr (EPI == DWM_EPI_RMSHEAD) {
                f32x2 ss2 = {0.f, 0.f};
    #pragma unroll
                for (int nt = 0; nt < 2; ++nt)
    #pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 x = DWM_PAIR(acc[mt][ch * 2 + nt], r) + DWM_PAIR(bv[nt], r);
                        acc[mt][ch * 2 + nt][r] = x[0];
                        acc[mt][ch * 2 + nt][r + 1] = x[1];
                        ss2 += x * x;
                    }
                float ss = ss2[0] + ss2[1];
                ss += __shfl_xor(ss, 32, 64);                   // other half of the row lives in lane^32
                const float rinv = do_norm ? rsqrtf(ss * (1.f / 64.f) + p.rms_eps) : 1.f;
    #pragma unroll
                for (int nt = 0; nt < 2; ++nt)
    #pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 x = (DWM_PAIR(acc[mt][ch * 2 + nt], r) * splat2(rinv)) * DWM_PAIR(rw[nt], r);
                        acc[mt][ch * 2 + nt][r] = x[0];
                        acc[mt][ch * 2 + nt][r + 1] = x[1];
                    }
            } else if constexpr (kGeglu) {
    #pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 hv = DWM_PAIR(acc[mt][ch * 2], r) + DWM_PAIR(bv[0], r);
                    const f32x2 x = hv * gelu_erf2(DWM_PAIR(acc[mt][ch * 2 + 1], r) + DWM_PAIR(bv[1], r));
                    acc[mt][ch * 2][r] = x[0];
                    acc[mt][ch * 2][r + 1] = x[1];
                }
            } else if constexpr (EPI == EPI_SPLITK) {
                // raw fp32 partial sums: bias / activation / residual are applied by the finishing kernel
            } else if constexpr (EPI == DWM_EPI_RESID) {
                // bias + activation are applied after the transpose: a row-major lane owns 8 fixed columns, i.e. 8 bias
                // registers instead of the 32 of the MFMA layout - the registers the residual pipeline needs
            } else {
#define DWM_ACT_PASS(FN_)                                                                           \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                     \
                    _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                              \
                        const f32x2 x = FN_(DWM_PAIR(acc[mt][ch * 2 + nt], r) + DWM_PAIR(bv[nt], r)); \
                        acc[mt][ch * 2 + nt][r] = x[0];                                              \
                        acc[mt][ch * 2 + nt][r + 1] = x[1];                                          \
                    }
                if (p.act == DWM_ACT_GELU_TANH) { DWM_ACT_PASS(gelu_tanh2) }
                else if (p.act == DWM_ACT_SILU) { DWM_ACT_PASS(silu2) }
                else if (p.act == DWM_ACT_RELU) { DWM_ACT_PASS(relu2) }
                else { DWM_ACT_PASS() }
#undef DWM_ACT_PASS
            }
#undef DWM_PAIR
            // ---- transpose.  PLAIN / GEGLU / RMSHEAD: the values are final after stage A, so they cross the LDS as bf16
            // (half the LDS bytes, one 16-byte read per lane and step, no conversion after it): row l31 of [32][CW] bf16, a
            // lane's 4 consecutive columns = 8 bytes at 16-B chunk (nt*4 + rg), half `half` of it, chunk swizzled by the row.
            // RESID / split-K: fp32 (the gate / residual / blend math follows the transpose): row l31, 16-B chunk
            // c = (nt*32 + rg*8 + half*4) / 4, swizzled by the row
            constexpr bool kT16 = EPI != DWM_EPI_RESID && EPI != EPI_SPLITK;
            constexpr int RB16 = CW * 2;                                           // bytes per bf16 row: 128 (64 for GEGLU)
            auto swz16 = [](int r) { return RB16 == 128 ? (r >> 1) & 7 : (r >> 2) & 3; };
    #pragma unroll
            for (int nt = 0; nt < (kGeglu ? 1 : 2); ++nt)
    #pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    if constexpr (kT16) {
                        uint2 w;
                        w.x = pack_bf16x2(acc[mt][ch * 2 + nt][rg * 4], acc[mt][ch * 2 + nt][rg * 4 + 1]);
                        w.y = pack_bf16x2(acc[mt][ch * 2 + nt][rg * 4 + 2], acc[mt][ch * 2 + nt][rg * 4 + 3]);
                        *(uint2*)(scr + l31 * RB16 + (((nt * 4 + rg) ^ swz16(l31)) << 4) + half * 8) = w;
                    } else {
                        const int c = nt * 8 + rg * 2 + half;
                        const float4 v = make_float4(acc[mt][ch * 2 + nt][rg * 4], acc[mt][ch * 2 + nt][rg * 4 + 1],
                                                     acc[mt][ch * 2 + nt][rg * 4 + 2], acc[mt][ch * 2 + nt][rg * 4 + 3]);
                        *(float4*)(scr + l31 * (CW * 4) + ((c ^ (l31 & (CW / 4 - 1))) << 4)) = v;
                    }
                }
            // same-wave LDS ops complete in order; the reads below see the writes above
            // ---- stage B.  RESID: the gate / residual / blend rows were requested one pass ahead (see above)
            constexpr int NST = 32 / RPS;
    #pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int r = st * RPS + brow;                 // row inside this 32-row pass
                const int64_t m = m0 + wm * 128 + mt * 32 + r;
                const int64_t mrow = FAST ? (m < M ? m : M - 1) : map_row(cp.c, m < M ? m : M - 1);
                if constexpr (kT16) {
                    const uint4 o = *(const uint4*)(scr + r * RB16 + ((bc8 ^ swz16(r)) << 4));
                    if (m < M && nok && !((DWM_RESERVED(p.reserved) & 2) && m >= 0)) {
                        if constexpr (FAST) *(uint4*)(Cp + ((uint64_t)(uint32_t)mrow * (uint32_t)p.ldc + (uint32_t)ncol)) = o;
                        else *(uint4*)(Cp + mrow * p.ldc + ncol) = o;
                    }
                    continue;
                }
                const float4 x0 = *(const float4*)(scr + r * (CW * 4) + (((2 * bc8) ^ (r & (CW / 4 - 1))) << 4));
                const float4 x1 = *(const float4*)(scr + r * (CW * 4) + (((2 * bc8 + 1) ^ (r & (CW / 4 - 1))) << 4));
                float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                if constexpr (EPI == DWM_EPI_RESID) {
                    float t[8];
    #pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        f32x2 y = (f32x2){v[j], v[j + 1]} + (f32x2){b8[j], b8[j + 1]};
                        if constexpr (!FAST) {
                            if (p.act != DWM_ACT_NONE) y = p.act == DWM_ACT_GELU_TANH ? gelu_tanh2(y) : p.act == DWM_ACT_SILU ? silu2(y) : relu2(y);
                        }
                        v[j] = y[0]; v[j + 1] = y[1];
                    }
                    if (f_gate) {
                        unpack8(gbA[st], t);
    #pragma unroll
                        for (int j = 0; j < 8; j += 2) {
                            const f32x2 y = (f32x2){v[j], v[j + 1]} * (f32x2){t[j], t[j + 1]};
                            v[j] = y[0]; v[j + 1] = y[1];
                        }
                    }
                    if (f_res) {
                        if constexpr (RF32) {
                            const float4 ta = *reinterpret_cast<const float4*>(&rA[st]), tb = *reinterpret_cast<const float4*>(&rB[st]);
                            t[0] = ta.x; t[1] = ta.y; t[2] = ta.z; t[3] = ta.w; t[4] = tb.x; t[5] = tb.y; t[6] = tb.z; t[7] = tb.w;
                        } else {
                            unpack8(rA[st], t);
                        }
    #pragma unroll
                        for (int j = 0; j < 8; j += 2) {
                            const f32x2 y = (f32x2){v[j], v[j + 1]} + (f32x2){t[j], t[j + 1]};
                            v[j] = y[0]; v[j + 1] = y[1];
                        }
                    }
                    if (f_blend) {
                        if constexpr (RF32) {            // (gate and blend together are rejected by the entry point)
                            const float4 ta = *reinterpret_cast<const float4*>(&gbA[st]), tb = *reinterpret_cast<const float4*>(&gbB[st]);
                            t[0] = ta.x; t[1] = ta.y; t[2] = ta.z; t[3] = ta.w; t[4] = tb.x; t[5] = tb.y; t[6] = tb.z; t[7] = tb.w;
                        } else {
                            unpack8(gbA[st], t);
                        }
                        const float al = alA[st];
                        const f32x2 a2 = splat2(al), b2 = splat2(1.f - al);
    #pragma unroll
                        for (int j = 0; j < 8; j += 2) {
                            const f32x2 y = a2 * (f32x2){t[j], t[j + 1]} + b2 * (f32x2){v[j], v[j + 1]};
                            v[j] = y[0]; v[j + 1] = y[1];
                        }
                    }
                }
                if constexpr (EPI == EPI_SPLITK) {
                    if (m < M && nok) {
                        float* wp = cp.ws + slice * cp.ws_slice + m * N + ncol;
                        *(float4*)wp = x0;
                        *(float4*)(wp + 4) = x1;
                    }
                } else {
                    if (m < M && nok && !((DWM_RESERVED(p.reserved) & 2) && m >= 0)) {
                        if (!RF32 || f_mirror) {               // (RF32: the bf16 mirror is optional)
                            if constexpr (FAST) *(uint4*)(Cp + ((uint64_t)(uint32_t)mrow * (uint32_t)p.ldc + (uint32_t)ncol)) = pack8(v);
                            else *(uint4*)(Cp + mrow * p.ldc + ncol) = pack8(v);
                        }
                        if constexpr (RF32) {
                            float* o32 = FAST ? (float*)p.C32 + ((uint64_t)(uint32_t)mrow * (uint32_t)p.ldc32 + (uint32_t)ncol)
                                              : (float*)p.C32 + mrow * p.ldc32 + ncol;
                            *(float4*)o32 = make_float4(v[0], v[1], v[2], v[3]);
                            *(float4*)(o32 + 4) = make_float4(v[4], v[5], v[6], v[7]);
                        }
                    }
                }
                if constexpr (EPI == DWM_EPI_RESID) {
                    if (mt + 1 < 4) { DWM_ISSUE_RESID(mt + 1, st) }
                }
            }
        }
#undef DWM_ISSUE_RESID
    }
}

// Split-K finish: out[m][n..n+8) = epilogue(sum_s ws[s][m][n..n+8)) for the PLAIN and RESID epilogues (one thread per
// 8 columns; fixed summation order, so the result does not depend on scheduling).
__global__ void __launch_bounds__(256)
splitk_finish_kernel(const dwm_gemm_args p, const ConvParams cp) {
    const int64_t n8 = p.N >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.M * n8) return;
    const int64_t m = i / n8, n = (i - m * n8) << 3;
    const float* w = cp.ws + m * p.N + n;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < cp.ksplit; ++s) {
        const float4 a = *(const float4*)(w + s * cp.ws_slice), b = *(const float4*)(w + s * cp.ws_slice + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
        v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    float t[8];
    if (p.bias) {
        unpack8(*(const uint4*)((const bf16_t*)p.bias + n), t);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += t[j];
    }
    if (p.act != DWM_ACT_NONE) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x2 x = {v[j], v[j + 1]};
            const f32x2 y = p.act == DWM_ACT_GELU_TANH ? gelu_tanh2(x) : p.act == DWM_ACT_SILU ? silu2(x) : relu2(x);
            v[j] = y[0]; v[j + 1] = y[1];
        }
    }
    const int64_t mr = map_row(cp.c, m);
    if (p.epilogue == DWM_EPI_RESID) {
        if (p.gate) {
            unpack8(*(const uint4*)((const bf16_t*)p.gate + (int64_t)fdiv((uint32_t)m, cp.fd_rpg) * p.ld_gate + n), t);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= t[j];
        }
        if (p.res) {
            const int64_t rr = p.res_mod > 0 ? (int64_t)fmod_u((uint32_t)m, cp.fd_rmod) : p.res_mod < 0 ? (int64_t)fdiv((uint32_t)m, cp.fd_rmod) : mr;
            unpack8(*(const uint4*)((const bf16_t*)p.res + rr * p.ld_res + n), t);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += t[j];
        }
        if (p.blend) {
            unpack8(*(const uint4*)((const bf16_t*)p.blend + mr * p.ld_blend + n), t);
            const float al = p.alpha[fdiv((uint32_t)m, cp.fd_rpa)];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = al * t[j] + (1.f - al) * v[j];
        }
    }
    *(uint4*)((bf16_t*)p.C + mr * p.ldc + n) = pack8(v);
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 accuracy path (BASELINE north_star: "within 1e-3 rel fp32"): C = epi(A W^T) for fp32 A / W on the SAME bf16 MFMA main
// loop.  Each fp32 operand is split into two bf16 planes x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 significant bits)
// and the three leading products hi*hi + hi*lo + lo*hi are ONE GEMM over K' = 3K that accumulates in the fp32 MFMA
// accumulators: the A side walks the planes [A_hi ; A_lo] as three "taps" (row shifts 0, 0, M - the implicit-GEMM tap
// mechanism), the W side is the pre-split [W_hi | W_lo | W_hi] (N x 3K, packed once per weight).  The dropped lo*lo term and
// the plane rounding are ~2^-16 relative per product.  The main loop leaves raw fp32 sums in the workspace (the split-K form);
// f32_finish_kernel applies bias / activation / GEGLU / q-k RMSNorm / gate, residual, blend in fp32 with libm-accurate
// functions and writes fp32.  Three times the MFMA work plus two extra passes over the output: an accuracy mode, not the
// throughput path.
__global__ void __launch_bounds__(256)
split_planes_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, int64_t K, bf16_t* __restrict__ planes) {
    const int64_t k8 = K >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * k8) return;
    const int64_t m = i / k8, k = (i - m * k8) << 3;
    const float4 a = *(const float4*)(x + m * ldx + k), b = *(const float4*)(x + m * ldx + k + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float hi[8], lo[8];
    const uint4 ph = pack8(v);
    unpack8(ph, hi);
#pragma unroll
    for (int j = 0; j < 8; ++j) lo[j] = v[j] - hi[j];
    *(uint4*)(planes + m * K + k) = ph;
    *(uint4*)(planes + (M + m) * K + k) = pack8(lo);
}

DWM_DEVINL float act_f32(float x, int act) {
    if (act == DWM_ACT_GELU_TANH) return 0.5f * x * (1.f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
    if (act == DWM_ACT_SILU) return x / (1.f + expf(-x));
    if (act == DWM_ACT_RELU) return fmaxf(x, 0.f);
    return x;
}

// one thread per 8 OUTPUT columns; slices of a split-K run are summed in a fixed order
template <int EPI>
__global__ void __launch_bounds__(256)
f32_finish_kernel(const dwm_gemm_args p, const ConvParams cp) {
    constexpr bool kGeglu = EPI == DWM_EPI_GEGLU;
    const int64_t Nout = kGeglu ? (p.N >> 1) : p.N;
    const int64_t n8 = Nout >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < p.M * n8;
    const int64_t m = live ? i / n8 : 0, n = live ? (i - m * n8) << 3 : 0;
    // accumulator columns of this thread: GEGLU groups of 64 = [32 value | 32 gate]
    const int64_t nv = kGeglu ? ((n >> 5) << 6) + (n & 31) : n;
    auto load8 = [&](int64_t col, float* v) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        const float* w = cp.ws + m * p.N + col;
        for (int s = 0; s < cp.ksplit; ++s) {
            const float4 a = *(const float4*)(w + s * cp.ws_slice), b = *(const float4*)(w + s * cp.ws_slice + 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
            v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        if (p.bias) {
            const float* bp = (const float*)p.bias + col;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += bp[j];
        }
    };
    float v[8];
    load8(nv, v);
    if constexpr (kGeglu) {
        float g[8];
        load8(nv + 32, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= 0.5f * g[j] * (1.f + erff(g[j] * 0.7071067811865476f));      // exact-erf GELU
    } else if constexpr (EPI == DWM_EPI_RMSHEAD) {
        // per-head (64 columns = 8 neighbouring threads) RMSNorm of the q / k columns; N % 64 == 0 keeps a head inside a wave
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        ss += __shfl_xor(ss, 4, 64);
        if (n < p.rms_ncols) {
            const float rinv = 1.f / sqrtf(ss * (1.f / 64.f) + p.rms_eps);
            const float* rw = (const float*)p.rms_w + n;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * rinv * rw[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = act_f32(v[j], p.act);
    }
    if (!live) return;
    const int64_t mr = map_row(cp.c, m);
    if constexpr (EPI == DWM_EPI_RESID) {
        if (p.gate) {
            const float* t = (const float*)p.gate + (int64_t)fdiv((uint32_t)m, cp.fd_rpg) * p.ld_gate + n;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= t[j];
        }
        if (p.res) {
            const int64_t rr = p.res_mod > 0 ? (int64_t)fmod_u((uint32_t)m, cp.fd_rmod) : p.res_mod < 0 ? (int64_t)fdiv((uint32_t)m, cp.fd_rmod) : mr;
            const float* t = (const float*)p.res + rr * p.ld_res + n;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += t[j];
        }
        if (p.blend) {
            const float* t = (const float*)p.blend + mr * p.ld_blend + n;
            const float al = p.alpha[fdiv((uint32_t)m, cp.fd_rpa)];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = al * t[j] + (1.f - al) * v[j];
        }
    }
    float* o = (float*)p.C + mr * p.ldc + n;
    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

}  // namespace

extern "C" int dwm_gemm_f32(const dwm_gemm_args* a, void* stream) {
    if (a == nullptr || a->A == nullptr || a->W == nullptr || a->C == nullptr || a->workspace == nullptr) return DWM_EINVAL;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->M >= (1ll << 30) || a->N >= (1ll << 31)) return DWM_EINVAL;
    if (a->K % BK != 0 || a->N % 8 != 0) return DWM_EUNSUPPORTED;
    // implicit convolution (a_map / c_map / taps as in dwm_gemm_bf16): every tap is walked as three plane taps (A_hi x W_hi, A_hi x
    // W_lo, A_lo x W_hi), so 9 taps (a 3x3 kernel) fill the 27 tap slots of one main-loop launch; W is [N, ntaps * 3 * k_per_tap]
    // (ops.split_weight).  More taps (the 27 of a causal 3x3x3 convolution: the temporal VAE) run as GROUPS of 9 - one main-loop
    // launch per group, each into its own fp32 partial slices, all summed in order by the one finishing kernel; W then holds the
    // groups one after the other, group g = [N, taps_g * 3 * k_per_tap] contiguous.
    const int ctaps = a->ntaps > 0 ? a->ntaps : 1;
    if (ctaps > 27) return DWM_EUNSUPPORTED;
    const int ngroups = (ctaps + 8) / 9;
    const int64_t kpt = a->ntaps > 0 ? a->k_per_tap : a->K;
    if (kpt <= 0 || kpt % BK != 0 || kpt * ctaps != a->K) return DWM_EINVAL;
    // rows of the A buffer (both planes cover all of them: taps reach into the padded border)
    int64_t a_rows = a->M;
    if (a->a_map.rw > 0) {
        const int64_t ppi = a->a_map.rw * a->a_map.rh;
        if (a->a_map.rh <= 0 || a->M % ppi != 0) return DWM_EINVAL;
        // one past the last row any tap of any output pixel reads (every grid of opendwm_amd.ops ends exactly there: PaddedGrid
        // and its stride-2 maps, TimeGrid, and Grid3D, whose two context frames sit in FRONT of the mapped images)
        int64_t reach = 0;
        for (int t = 0; t < ctaps && a->ntaps > 0; ++t) reach = a->tap_shift[t] > reach ? a->tap_shift[t] : reach;
        a_rows = a->a_map.origin + (a->M / ppi - 1) * a->a_map.ipitch + (a->a_map.rh - 1) * a->a_map.rpitch +
                 (a->a_map.rw - 1) * (a->a_map.xstep > 0 ? a->a_map.xstep : 1) + reach + 1;
        if (a_rows <= 0) return DWM_EINVAL;
    } else if (a->ntaps > 0) return DWM_EUNSUPPORTED;            // taps need the padded-grid map
    if (a->lda % 4 != 0 || a->ldc % 4 != 0 || !dwm_aligned16(a->A) || !dwm_aligned16(a->W) || !dwm_aligned16(a->C) ||
        !dwm_aligned16(a->workspace)) return DWM_EALIGN;
    const int64_t nout = a->epilogue == DWM_EPI_GEGLU ? a->N / 2 : a->N;
    if (a->ldc < nout || a->lda < kpt) return DWM_EINVAL;
    switch (a->epilogue) {
        case DWM_EPI_PLAIN: break;
        case DWM_EPI_GEGLU: if (a->N % 64 != 0) return DWM_EUNSUPPORTED; break;
        case DWM_EPI_RESID:
            if (a->gate && (a->rows_per_gate <= 0 || a->ld_gate % 4 != 0 || !dwm_aligned16(a->gate))) return DWM_EINVAL;
            if (a->res && (a->ld_res % 4 != 0 || !dwm_aligned16(a->res))) return DWM_EALIGN;
            if (a->blend && (a->alpha == nullptr || a->rows_per_alpha <= 0 || a->ld_blend % 4 != 0 || !dwm_aligned16(a->blend))) return DWM_EINVAL;
            if (a->gate && a->blend) return DWM_EUNSUPPORTED;
            break;
        case DWM_EPI_RMSHEAD: if (a->rms_w == nullptr || a->rms_ncols % 64 != 0 || a->N % 64 != 0) return DWM_EINVAL; break;
        default: return DWM_EINVAL;
    }
    if (a->rows_per_gate > (1ll << 30) || a->res_mod > (1ll << 30) || a->res_mod < -(1ll << 30) || a->rows_per_alpha > (1ll << 30)) return DWM_EINVAL;
    // workspace: [A_hi ; A_lo] bf16 planes, then the fp32 partial sums
    const int64_t plane_bytes = ((2 * a_rows * kpt * 2 + 255) / 256) * 256;
    const int ntm = (int)((a->M + BM - 1) / BM), ntn = (int)((a->N + BN - 1) / BN);
    // (K ranges per launch: chosen for the shortest group, so that every range of every group has K steps)
    const int last_taps = ctaps - 9 * (ngroups - 1);
    const int64_t tiles = (int64_t)ntm * ntn, nk = 3 * kpt * last_taps / BK, slice_bytes = a->M * a->N * 4;
    int ksplit = 1;
    if (tiles * ngroups <= 128 && nk >= 16) {
        ksplit = (int)(256 / (tiles * ngroups));
        if (ksplit > nk / 8) ksplit = (int)(nk / 8);
        if (ksplit > 32 / ngroups) ksplit = 32 / ngroups;
        if (ksplit < 1) ksplit = 1;
    }
    while (ksplit > 1 && plane_bytes + (int64_t)ngroups * ksplit * slice_bytes > a->workspace_bytes) --ksplit;
    if (plane_bytes + (int64_t)ngroups * slice_bytes > a->workspace_bytes) return DWM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    bf16_t* planes = (bf16_t*)a->workspace;
    {
        const int64_t nthr = a_rows * (kpt >> 3);
        hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, (const float*)a->A, a->lda, a_rows,
                           kpt, planes);
    }
    dwm_gemm_args g = *a;
    g.A = planes; g.lda = kpt;
    ConvParams cp;
    auto mk = [](const dwm_rowmap2d& r, DevRowMap& d) -> bool {
        d.enabled = r.rw > 0;
        d.xstep = r.xstep > 0 ? (int)r.xstep : 1;
        if (!d.enabled) { d.rw = make_fastdiv(1); d.rh = make_fastdiv(1); d.rpitch = d.ipitch = d.origin = 0; return true; }
        if (r.rh <= 0 || r.rw >= (1ll << 30) || r.rh >= (1ll << 30)) return false;
        d.rw = make_fastdiv((uint32_t)r.rw); d.rh = make_fastdiv((uint32_t)r.rh);
        d.rpitch = r.rpitch; d.ipitch = r.ipitch; d.origin = r.origin;
        return true;
    };
    if (!mk(a->a_map, cp.a) || !mk(a->c_map, cp.c)) return DWM_EINVAL;
    cp.steps_per_tap = (int)(kpt / BK);
    cp.fd_steps = make_fastdiv((uint32_t)cp.steps_per_tap);
    cp.fd_rpg = make_fastdiv((uint32_t)(a->rows_per_gate > 0 ? a->rows_per_gate : 1));
    cp.fd_rmod = make_fastdiv((uint32_t)(a->res_mod > 0 ? a->res_mod : a->res_mod < 0 ? -a->res_mod : 1));
    cp.fd_rpa = make_fastdiv((uint32_t)(a->rows_per_alpha > 0 ? a->rows_per_alpha : 1));
    cp.ksplit = ksplit;
    float* const ws_base = (float*)((char*)a->workspace + plane_bytes);
    cp.ws_slice = a->M * a->N;
    hipError_t e;
    {
        static bool attr_set = false;
        if (!attr_set) {
            e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI_SPLITK>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
    }
    g.reserved = 0;
    for (int gi = 0; gi < ngroups; ++gi) {
        const int tg = gi + 1 < ngroups ? 9 : last_taps;
        g.K = 3 * kpt * tg;
        g.W = (const bf16_t*)a->W + (int64_t)a->N * (3 * kpt * 9) * gi;
        for (int t = 0; t < 27; ++t) cp.tap_shift[t] = 0;
        for (int t = 0; t < tg; ++t) {                // per tap: A_hi (x W_hi), A_hi (x W_lo), A_lo (x W_hi)
            const int64_t sh = a->ntaps > 0 ? a->tap_shift[9 * gi + t] : 0;
            cp.tap_shift[3 * t] = sh; cp.tap_shift[3 * t + 1] = sh; cp.tap_shift[3 * t + 2] = sh + a_rows;
        }
        cp.ws = ws_base + (int64_t)gi * ksplit * cp.ws_slice;
        set_raster(cp, g, ntn);
        hipLaunchKernelGGL(gemm_bf16_kernel<EPI_SPLITK>, dim3((unsigned)(ntm * ntn * ksplit)), dim3(512), LDS_BYTES, s, g, cp, ntm, ntn);
    }
    cp.ws = ws_base;
    cp.ksplit = ngroups * ksplit;                     // the finishing kernel sums every range of every group, in order
    const int64_t nthr = a->M * (nout >> 3);
    const dim3 fg((unsigned)((nthr + 255) / 256));
    switch (a->epilogue) {
        case DWM_EPI_PLAIN: hipLaunchKernelGGL(f32_finish_kernel<DWM_EPI_PLAIN>, fg, dim3(256), 0, s, *a, cp); break;
        case DWM_EPI_GEGLU: hipLaunchKernelGGL(f32_finish_kernel<DWM_EPI_GEGLU>, fg, dim3(256), 0, s, *a, cp); break;
        case DWM_EPI_RESID: hipLaunchKernelGGL(f32_finish_kernel<DWM_EPI_RESID>, fg, dim3(256), 0, s, *a, cp); break;
        default: hipLaunchKernelGGL(f32_finish_kernel<DWM_EPI_RMSHEAD>, fg, dim3(256), 0, s, *a, cp); break;
    }
    e = hipGetLastError();
    return e == hipSuccess ? DWM_OK : (int)e;
}

// gemm_bf16_4w.hip: the 4-wave main loop under the same epilogues (dwm_gemm_args.tile == 3 / 4); -1 = not a launch it covers
int dwm_gemm4w_try(const dwm_gemm_args* a, void* stream, bool fast_only);

extern "C" int dwm_gemm_bf16(const dwm_gemm_args* a_in, void* stream) {
    if (a_in == nullptr) return DWM_EINVAL;
    // tile == 3: "automatic, and the 4-wave kernels may serve the launch" (what the MMDiT inference forward asks for); 4: the same,
    // their fast form only
    dwm_gemm_args a_copy;
    const bool allow4w = a_in->tile == 3 || a_in->tile == 4;
    const bool fast_only4w = a_in->tile == 4;
    if (allow4w) { a_copy = *a_in; a_copy.tile = 0; }
    const dwm_gemm_args* a = allow4w ? &a_copy : a_in;
    if (a == nullptr || a->A == nullptr || a->W == nullptr || (a->C == nullptr && a->C32 == nullptr)) return DWM_EINVAL;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->M >= (1ll << 31) || a->N >= (1ll << 31)) return DWM_EINVAL;
    if (a->K % BK != 0 || a->N % 8 != 0) return DWM_EUNSUPPORTED;
    if (a->lda % 8 != 0 || (a->C != nullptr && a->ldc % 8 != 0)) return DWM_EALIGN;
    if (!dwm_aligned16(a->A) || !dwm_aligned16(a->W) || !dwm_aligned16(a->C)) return DWM_EALIGN;
    if (a->bias && (((uintptr_t)a->bias) & 7u)) return DWM_EALIGN;
    const int64_t nout = a->epilogue == DWM_EPI_GEGLU ? a->N / 2 : a->N;
    if (a->C != nullptr && a->ldc < nout) return DWM_EINVAL;
    switch (a->epilogue) {
        case DWM_EPI_PLAIN: break;
        case DWM_EPI_GEGLU:
            if (a->N % 64 != 0 || (a->N / 2) % 8 != 0) return DWM_EUNSUPPORTED;
            break;
        case DWM_EPI_RESID:
            if (a->gate && (a->rows_per_gate <= 0 || a->ld_gate % 8 != 0 || !dwm_aligned16(a->gate))) return DWM_EINVAL;
            if (a->res && (a->ld_res % (a->C32 ? 4 : 8) != 0 || !dwm_aligned16(a->res))) return DWM_EALIGN;
            if (a->blend && (a->alpha == nullptr || a->rows_per_alpha <= 0 || a->ld_blend % (a->C32 ? 4 : 8) != 0 || !dwm_aligned16(a->blend))) return DWM_EINVAL;
            if (a->gate && a->blend) return DWM_EUNSUPPORTED;      // one register set carries the gate OR the blend rows
            if (a->C32 != nullptr && ((a->res != nullptr && a->ld_res % 4 != 0) || a->ldc32 % 4 != 0 || a->ldc32 < a->N ||
                                      !dwm_aligned16(a->C32) || a->res_mod != 0 || a->split_k > 1)) return DWM_EINVAL;
            break;
        case DWM_EPI_RMSHEAD:
            if (a->rms_w == nullptr || a->rms_ncols % 64 != 0 || a->N % 64 != 0) return DWM_EINVAL;
            break;
        default: return DWM_EINVAL;
    }
    if (a->C32 != nullptr && a->epilogue != DWM_EPI_RESID) return DWM_EUNSUPPORTED;
    {
        // (kernel selection comes from the arguments only: the library reads no environment)
        const bool use4w = allow4w;
        if (use4w && !DWM_RESERVED(a->reserved) && a->lda % 64 == 0) {
            const int rc4 = dwm_gemm4w_try(a, stream, fast_only4w);
            if (rc4 >= 0) return rc4;
        }
    }
    ConvParams cp;
    auto mk = [](const dwm_rowmap2d& r, DevRowMap& d) -> bool {
        d.enabled = r.rw > 0;
        d.xstep = r.xstep > 0 ? (int)r.xstep : 1;
        if (!d.enabled) { d.rw = make_fastdiv(1); d.rh = make_fastdiv(1); d.rpitch = d.ipitch = d.origin = 0; return true; }
        if (r.rh <= 0 || r.rw >= (1ll << 30) || r.rh >= (1ll << 30)) return false;
        d.rw = make_fastdiv((uint32_t)r.rw); d.rh = make_fastdiv((uint32_t)r.rh);
        d.rpitch = r.rpitch; d.ipitch = r.ipitch; d.origin = r.origin;
        return true;
    };
    if (!mk(a->a_map, cp.a) || !mk(a->c_map, cp.c)) return DWM_EINVAL;
    const int ntaps = a->ntaps > 0 ? a->ntaps : 1;
    if (ntaps > 27) return DWM_EINVAL;
    const int64_t kpt = a->ntaps > 0 ? a->k_per_tap : a->K;
    if (kpt <= 0 || kpt % BK != 0 || kpt * ntaps != a->K) return DWM_EINVAL;
    cp.steps_per_tap = (int)(kpt / BK);
    cp.fd_steps = make_fastdiv((uint32_t)cp.steps_per_tap);
    if (a->rows_per_gate > (1ll << 30) || a->res_mod > (1ll << 30) || a->res_mod < -(1ll << 30) || a->rows_per_alpha > (1ll << 30)) return DWM_EINVAL;
    cp.fd_rpg = make_fastdiv((uint32_t)(a->rows_per_gate > 0 ? a->rows_per_gate : 1));
    cp.fd_rmod = make_fastdiv((uint32_t)(a->res_mod > 0 ? a->res_mod : a->res_mod < 0 ? -a->res_mod : 1));
    cp.fd_rpa = make_fastdiv((uint32_t)(a->rows_per_alpha > 0 ? a->rows_per_alpha : 1));
    for (int t = 0; t < 27; ++t) cp.tap_shift[t] = (a->ntaps > 0 && t < ntaps) ? a->tap_shift[t] : 0;
    if (a->lda < kpt) return DWM_EINVAL;
    if (a->tile < 0 || a->tile > 2) return DWM_EINVAL;
    int ntm = (int)((a->M + BM - 1) / BM), ntn = (int)((a->N + BN - 1) / BN);
    set_raster(cp, *a, ntn);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    // ---- split-K: a grid that fills less than half of the 256 CUs and a long K.  One K range per workgroup,
    // fp32 partials in the caller's workspace, deterministic reduction + epilogue in splitk_finish_kernel.
    int ksplit = 1;
    {
        const int64_t tiles = (int64_t)ntm * ntn, nk = a->K / BK;
        const bool can = (a->epilogue == DWM_EPI_PLAIN || a->epilogue == DWM_EPI_RESID) && a->workspace != nullptr &&
                         dwm_aligned16(a->workspace) && !(DWM_RESERVED(a->reserved) & 3) && a->C32 == nullptr;
        if (a->split_k > 1) {
            if (!can) return DWM_EUNSUPPORTED;
            ksplit = a->split_k;
        } else if (a->split_k == 0 && can && tiles <= 128 && nk >= 16) {
            ksplit = (int)(256 / tiles);
        }
        if (ksplit > 1) {
            const int64_t slice_bytes = a->M * a->N * 4;
            if (ksplit > nk / 8) ksplit = (int)(nk / 8);
            if (ksplit > 32) ksplit = 32;
            if ((int64_t)ksplit * slice_bytes > a->workspace_bytes) ksplit = (int)(a->workspace_bytes / slice_bytes);
            if (ksplit < 2) {
                if (a->split_k > 1) return DWM_EUNSUPPORTED;
                ksplit = 1;
            }
        }
    }
    cp.ksplit = ksplit;
    cp.ws = (float*)a->workspace;
    cp.ws_slice = a->M * a->N;
    if (ksplit > 1) {
        static bool attr_set = false;
        if (!attr_set) {
            e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI_SPLITK>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        hipLaunchKernelGGL(gemm_bf16_kernel<EPI_SPLITK>, dim3((unsigned)(ntm * ntn * ksplit)), dim3(512), LDS_BYTES, s, *a, cp, ntm, ntn);
        const int64_t nthr = a->M * (a->N >> 3);
        hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, *a, cp);
        e = hipGetLastError();
        return e == hipSuccess ? DWM_OK : (int)e;
    }
    // ---- tile configuration (TileCfg): 256 x 128 tiles (two workgroups per CU) on request, or automatically where they
    // cut the padded columns (N = 320 -> 384 instead of 512: the SD 2.1 UNet's first level) AND K is short, i.e. the
    // epilogue's share is large (measured, profiles/README.md: N = 320, K = 320: 1.17 x; with K >= 2880 the 256 x 256
    // main loop wins although it pads more)
    int tc = a->tile == 2 ? 1 : 0;
    if (a->tile == 0 && a->C32 == nullptr) {
        const int64_t c256 = (a->N + 255) / 256 * 256, c128 = (a->N + 127) / 128 * 128;
        if (c128 < c256 && a->K <= 640) tc = 1;
    }
    if (DWM_RESERVED(a->reserved) & 0x200) tc = 1;
    if (tc == 1 && a->C32 != nullptr) return DWM_EUNSUPPORTED;
    if (tc == 1) {
        ntn = (int)((a->N + TileCfg<1>::bn - 1) / TileCfg<1>::bn);
        cp.steps_per_tap = (int)(kpt / TileCfg<1>::bk);
        cp.fd_steps = make_fastdiv((uint32_t)cp.steps_per_tap);
        set_raster(cp, *a, ntn);
    }
    const dim3 grid((unsigned)(ntm * ntn)), block(tc == 1 ? TileCfg<1>::nwaves * 64 : 512);
#define DWM_LAUNCH_TC(EPI, FAST, TC_)                                                                \
    do {                                                                                             \
        static bool attr_set = false;                                                                \
        if (!attr_set) {                                                                             \
            e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI, FAST, false, TC_>,            \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, tile_lds_bytes<TC_>() + (TC_ ? DWM_DEV_LDS_PAD : 0)); \
            if (e != hipSuccess) return (int)e;                                                      \
            attr_set = true;                                                                         \
        }                                                                                            \
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI, FAST, false, TC_>), grid, block,                   \
                           tile_lds_bytes<TC_>() + ((TC_ && (DWM_RESERVED(a->reserved) & 0x400)) ? DWM_DEV_LDS_PAD : 0), s, *a, cp, ntm, ntn); \
    } while (0)
#define DWM_LAUNCH(EPI, FAST)                                                                        \
    do {                                                                                             \
        if (tc == 1) DWM_LAUNCH_TC(EPI, FAST, 1); else DWM_LAUNCH_TC(EPI, FAST, 0);                  \
    } while (0)
    // the transformer blocks' linear layers (see FAST above); reserved bit 2 keeps the general kernels (A/B measurements)
    const int64_t lim = 1ll << 31;
    bool fast = !cp.c.enabled && a->ldc < lim && !(DWM_RESERVED(a->reserved) & 4);
    const int rs16 = (a->gate ? 1 : 0) | (a->res ? 2 : 0) | (a->blend ? 4 : 0) | ((a->res && a->res_mod < 0) ? 16 : 0);
    if (a->epilogue == DWM_EPI_RESID)
        fast = fast && (a->res_mod == 0 || (rs16 == 18 && a->C32 == nullptr)) && a->act == DWM_ACT_NONE &&
               (a->gate == nullptr || a->ld_gate < lim) && (a->res == nullptr || a->ld_res < lim) && (a->blend == nullptr || a->ld_blend < lim);
#define DWM_LAUNCH2(EPI)                                                                             \
    do {                                                                                             \
        if (fast) DWM_LAUNCH(EPI, true); else DWM_LAUNCH(EPI, false);                                \
    } while (0)
    if (a->C32 != nullptr) {                 // fp32 residual stream: RESID with fp32 residual / blend rows and fp32 output
        static bool attr_set = false;
        if (!attr_set) {
            e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<DWM_EPI_RESID, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            if (e != hipSuccess) return (int)e;
            e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<DWM_EPI_RESID, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        // the transformer blocks' hidden-state stream takes the FAST form (32-bit row arithmetic, no row map, no activation), and
        // its three operand sets - gate + residual, residual, residual + blend, no bf16 mirror - their compile-time forms (RS)
        const int rs = (a->gate ? 1 : 0) | (a->res ? 2 : 0) | (a->blend ? 4 : 0) | (a->C ? 8 : 0);
#define DWM_LAUNCH_RS(RS_)                                                                                        \
        do {                                                                                                      \
            static bool set_ = false;                                                                             \
            if (!set_) {                                                                                          \
                e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<DWM_EPI_RESID, true, true, 0, RS_>,         \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);                  \
                if (e != hipSuccess) return (int)e;                                                               \
                set_ = true;                                                                                      \
            }                                                                                                     \
            hipLaunchKernelGGL((gemm_bf16_kernel<DWM_EPI_RESID, true, true, 0, RS_>), grid, block, LDS_BYTES, s, *a, cp, ntm, ntn); \
        } while (0)
        if (fast && a->ldc32 < lim) {
            if (rs == 3) DWM_LAUNCH_RS(3);
            else if (rs == 2) DWM_LAUNCH_RS(2);
            else if (rs == 6) DWM_LAUNCH_RS(6);
            else hipLaunchKernelGGL((gemm_bf16_kernel<DWM_EPI_RESID, true, true>), grid, block, LDS_BYTES, s, *a, cp, ntm, ntn);
        } else {
            hipLaunchKernelGGL((gemm_bf16_kernel<DWM_EPI_RESID, false, true>), grid, block, LDS_BYTES, s, *a, cp, ntm, ntn);
        }
#undef DWM_LAUNCH_RS
        e = hipGetLastError();
        return e == hipSuccess ? DWM_OK : (int)e;
    }
    // bf16 RESID, FAST form: the operand sets of the transformer blocks at compile time as well (RS: gate + residual, residual,
    // residual + blend) - the SD 2.1 UNet's K = 320 ... 1280 GEMMs are mostly epilogue
#define DWM_LAUNCH_RS16(TC_, RS_)                                                                                 \
    do {                                                                                                          \
        static bool set_ = false;                                                                                 \
        if (!set_) {                                                                                              \
            e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<DWM_EPI_RESID, true, false, TC_, RS_>,          \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, tile_lds_bytes<TC_>() + (TC_ ? DWM_DEV_LDS_PAD : 0)); \
            if (e != hipSuccess) return (int)e;                                                                   \
            set_ = true;                                                                                          \
        }                                                                                                         \
        hipLaunchKernelGGL((gemm_bf16_kernel<DWM_EPI_RESID, true, false, TC_, RS_>), grid, block,                 \
                           tile_lds_bytes<TC_>() + ((TC_ && (DWM_RESERVED(a->reserved) & 0x400)) ? DWM_DEV_LDS_PAD : 0), s, *a, cp, ntm, ntn); \
    } while (0)
    const bool spec16 = a->epilogue == DWM_EPI_RESID && fast && (rs16 == 2 || rs16 == 3 || rs16 == 6 || rs16 == 18);
    if (spec16) {
        if (tc == 1) {
            if (rs16 == 2) DWM_LAUNCH_RS16(1, 2); else if (rs16 == 3) DWM_LAUNCH_RS16(1, 3); else if (rs16 == 6) DWM_LAUNCH_RS16(1, 6);
            else DWM_LAUNCH_RS16(1, 18);
        } else {
            if (rs16 == 2) DWM_LAUNCH_RS16(0, 2); else if (rs16 == 3) DWM_LAUNCH_RS16(0, 3); else if (rs16 == 6) DWM_LAUNCH_RS16(0, 6);
            else DWM_LAUNCH_RS16(0, 18);
        }
    } else
    switch (a->epilogue) {
        case DWM_EPI_PLAIN: DWM_LAUNCH2(DWM_EPI_PLAIN); break;
        case DWM_EPI_GEGLU: DWM_LAUNCH2(DWM_EPI_GEGLU); break;
        case DWM_EPI_RESID: DWM_LAUNCH2(DWM_EPI_RESID); break;
        default: DWM_LAUNCH2(DWM_EPI_RMSHEAD); break;
    }
#undef DWM_LAUNCH_RS16
#undef DWM_LAUNCH2
#undef DWM_LAUNCH
#undef DWM_LAUNCH_TC
    e = hipGetLastError();
    return e == hipSuccess ? DWM_OK : (int)e;
}
