// attn_res4_kernel: the resident attention forward (attention.hip: attn_res_kernel) with ONE wave per SIMD.  Its own translation
// unit because of its flags: -fno-slp-vectorize (the row-sum adds must stay scalar, see res4_block) and no -amdgpu-mfma-vgpr-form
// (build.py AGPR_SOURCES: the builtin MFMAs accumulate in AGPRs).
#include "attention_common.h"

using namespace dwm_attn;

namespace {

// ---------------------------------------------------------------------------------------------------------------
// One-wave-per-SIMD resident form (round 5).  Images, swizzles, row tables, persistent workgroups, the two barriers per head, the
// maximum-free fast path with its acceptance test and fallback, and the register-exchange stores are attn_res_kernel's.  What changes
// is who walks the images: FOUR waves - one per SIMD, 512 registers each - and every wave takes ALL its query tiles of a head
// (NT = 2..5 adjacent tiles: 19 tiles -> 5 / 5 / 5 / 4, 14 -> 4 / 4 / 3 / 3) through ONE pass over the keys:
//   * every K / V fragment read from the images feeds NT MFMAs (attn_res_kernel: one; the 12-wave kernel spends 1.5 LDS instructions
//     with their address adds and waits per MFMA, 22.5 ns per MFMA slot against 17.8 without them, profiles/README.md);
//   * no co-resident waves: on this chip a wave's own VALU work hides under its own MFMAs, another wave's does not
//     (profiles/r3_mfma_valu_probe.txt; attn_res_kernel's three waves per SIMD run the same tile loop in 15.6 / 19.7 / 34.3 k cycles);
//   * 19 tiles on 4 waves leave 5 % of the SIMD time idle (12 waves: 21 %).
// Registers: the O accumulators of all tiles (NT x 32) live in AGPRs (builtin MFMAs; this file is compiled WITHOUT
// -amdgpu-mfma-vgpr-form), the S accumulators in arch VGPRs (inline-asm MFMAs, below), next to the Q fragments of all tiles (NT x 16),
// two S buffers, two P' buffers and ONE set of K / V fragments: 256 + 160 registers at NT = 5 - nothing else may be live across the
// tile loop (the next head's Q rows are requested in the head seam, behind the image copy, not across the loop; the fallback re-reads
// its Q rows), or the spills' scratch round trips - which queue behind the LDS-DMA and Q loads of the seam - cost more than the
// loop (measured, profiles/r5d_trace4_L602.txt: 20 k cycles of "stores", 27 k of "copy issue" per head).
// Schedule: the units u = (key step k, tile t), k-major, form ONE software pipeline; slot u holds
//     S(u+1) = K Q^T (4 MFMAs)  ||  E(u): P' = 2^S, row sums, bf16 pack  ||  PV(u-1): O^T += V^T P'^T (4 MFMAs)
// as one instruction stream of 8 chunks (one MFMA + one slice of E each, order pinned by sched_barrier); a fragment register is
// re-requested right behind its last reader (res4_block).  ILV: S and PV MFMAs alternate, so that consecutive MFMAs never share an
// accumulator (a filler instruction between two MFMAs on the SAME accumulator costs ~40 cycles, MI355X_MICROARCH.md; measured here:
// 45-54 cycles per MFMA with the 4 + 4 order).
// Keys past the end of a ragged sequence need no masking: the pad rows of both images are ZERO (written once at kernel start; the
// copies skip them by EXEC), so a pad key scores exactly 0, contributes P' = 2^0 = 1 to the row sum and nothing to O, and the row
// sum is corrected by the constant number of pad keys.  (A row sum below 2^-6 would lose precision in that subtraction: such a
// unit takes the fallback like one that leaves the fast path's range.)
template <int NT>
struct Res4Regs {
    bf16x8 qf[NT][4];
    f32x16 ot[NT][2];
    float ls[NT][2];
    f32x16 s[2];
    bf16x8 p[2][2];
    bf16x8 kf[4];
    bf16x8 vf[2][2];             // [16-key half s2][d tile dt]
};

// The S MFMAs are inline asm with arch-VGPR destinations: every builtin MFMA of this file (the PV accumulation) keeps its accumulator
// in AGPRs, managed - with all hazards - by the compiler, while the scores, which the VALU reads, must not pay a v_accvgpr_read each.
// What the compiler cannot see about the asm MFMAs holds by construction and is checked on the generated code by
// scripts/dev/check_res4_asm.py: a chain of four accumulates on one register tuple (no wait states needed), its destination is
// early-clobber (never overlaps the operands), and the first VALU read of a chain's result comes >= 11 wait states behind the
// chain's last MFMA (an 8-pass MFMA -> VALU read; another MFMA in between is worth 8: it cannot issue before the pipe is free).
DWM_DEVINL void res4_mfma_s_first(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
DWM_DEVINL void res4_mfma_s(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
DWM_DEVINL bf16x8 res4_kread(const ResCtx& c, const char* kl, int m) {
    return *(const bf16x8*)(kl + c.l31 * 128 + (((2 * m + c.half) ^ c.kswz) << 4));
}
DWM_DEVINL bf16x8 res4_vread(const ResCtx& c, const char* vl, int s2, int dt) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vl + c.vra[dt] + s2 * (16 * 128)));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vl + c.vrb[dt] + s2 * (16 * 128)));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// key step k of a unit: slots (k, 0) .. (k, NT - 1).  KP = k & 1 (the parity of unit (k, 0) follows from it), FIRST: k = 0 (no PV in
// slot 0), LAST: k = n - 1 (no S in the last slot, no K request, and the trailing PV).
// Fragment registers are single-buffered: a K fragment of step k + 1 is requested right behind the last MFMA that reads the same
// fragment of step k (S(k, NT-1) in slot NT - 2; its next reader is S(k+1, 0) a slot later), a V fragment of step k right behind the last
// PV MFMA of step k - 1 (PV(k-1, NT-1) in slot 0; next reader PV(k, 0) in slot 1): seven MFMAs of distance each.
template <int NT, int KP, bool FIRST, bool LAST, bool ILV>
DWM_DEVINL void res4_block(Res4Regs<NT>& r, const ResCtx& c, int k) {
    constexpr int PB = (NT & 1) ? KP : 0;                 // parity of unit (k, 0): k * NT mod 2
    const char* const kln = c.kimg + (k + 1) * 4096;
    const char* const vlc = c.vimg + k * 4096;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int par = (PB + t) & 1;                     // parity of this slot's unit: S buffer read, P' buffer written
        const bool do_s = !(LAST && t == NT - 1);
        const int ts = t + 1 == NT ? 0 : t + 1;           // S(u+1): tile
        const bool do_pv = !(FIRST && t == 0);
        const int tp = t == 0 ? NT - 1 : t - 1;           // PV(u-1): tile
        uint32_t pk[8];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            // the MFMA of this chunk, and the request of its fragment's successor
            const bool is_s = ILV ? (ch & 1) == 0 : ch < 4;
            const int mi = ILV ? ch >> 1 : ch & 3;
            if (is_s) {
                if (do_s) {
                    if (mi == 0) res4_mfma_s_first(r.s[par ^ 1], r.kf[mi], r.qf[ts][mi]);
                    else res4_mfma_s(r.s[par ^ 1], r.kf[mi], r.qf[ts][mi]);
                }
                if (!LAST && t == NT - 2) r.kf[mi] = res4_kread(c, kln, mi);
            } else {
                if (do_pv) r.ot[tp][mi & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.vf[mi >> 1][mi & 1], r.p[par ^ 1][mi >> 1], r.ot[tp][mi & 1], 0, 0, 0);
                if (!FIRST && t == 0) r.vf[mi >> 1][mi & 1] = res4_vread(c, vlc, mi >> 1, mi & 1);
            }
            // slice ch of E(u): scores 2 ch, 2 ch + 1
            {
                const float a = r.s[par][2 * ch], b = r.s[par][2 * ch + 1];
                const float pa = __builtin_amdgcn_exp2f(a), pb = __builtin_amdgcn_exp2f(b);
                // (scalar adds - this file is built with -fno-slp-vectorize: left alone the compiler packs the adds of two slices into
                //  v_pk_add_f32, bunched behind the later slice; packed fp32 VALU beside MFMAs costs more than the plain adds it
                //  replaces, MI355X_MICROARCH.md "price of one filler")
                float acc = r.ls[t][ch & 1];
                acc += pa;
                acc += pb;
                r.ls[t][ch & 1] = acc;
                uint32_t w = pack_bf16x2(pa, pb);
                asm volatile("" : "+v"(w));                 // pins the convert to its slice (res_step)
                pk[ch] = w;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // (alternating order, first slot of a unit: no PV MFMAs between the S chain's last MFMA and the next slot's first read of its
        //  result - the wait states the compiler cannot know about)
        if (ILV && FIRST && t == 0) asm volatile("s_nop 7" : "+v"(r.s[par ^ 1]));
        const uint4 lo = {pk[0], pk[1], pk[2], pk[3]}, hi = {pk[4], pk[5], pk[6], pk[7]};
        r.p[par][0] = *reinterpret_cast<const bf16x8*>(&lo);
        r.p[par][1] = *reinterpret_cast<const bf16x8*>(&hi);
    }
    if (LAST) {                                             // PV of the last unit (k, NT - 1)
        constexpr int par = (PB + NT - 1) & 1;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
            r.ot[NT - 1][mi & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.vf[mi >> 1][mi & 1], r.p[par][mi >> 1], r.ot[NT - 1][mi & 1], 0, 0, 0);
    }
}

// normalise and store one output tile (res_unit's store: lane (q, half) holds d = 32 dt + 8 g + 4 half + (0..3) in registers
// 4 g .. 4 g + 3 of o[dt]; after the exchange of one 8-byte piece with lane ^ 32 per pair of g, the lower lane owns the whole
// 16-byte chunk of the even g, the upper lane that of the odd g)
DWM_DEVINL void res_store_tile(const f32x16 (&o)[2], float l_tot, bf16_t* op, int half) {
    const float inv = __builtin_amdgcn_rcpf(l_tot);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            float a[4], b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = o[dt][gp * 8 + j] * inv;
                b[j] = o[dt][gp * 8 + 4 + j] * inv;
            }
            const uint2 pa = pack4(a), pb = pack4(b);
            const auto s0 = __builtin_amdgcn_permlane32_swap(pa.x, pb.x, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pa.y, pb.y, false, false);
            const uint4 val = {s0[0], s1[0], s0[1], s1[1]};
            *(uint4*)(op + dt * 32 + (2 * gp + half) * 8) = val;
        }
}

// one unit = the NT query tiles of this wave against the resident K / V images of one head (n = c.nsub >= 3 key steps).
// qraw: raw Q fragments (consumed: scaled into the unit's registers); out_ptr(t): this lane's output row of tile t; reload_q(t, dst):
// the raw Q fragments of tile t again (fallback only); n_pad: zero pad keys of the images (Lp - L); after_loop(): called once, when
// the tile loop is over (its fragment / score registers are free from here on)
template <int NT, bool ILV, class OutPtr, class ReloadQ, class AfterLoop>
DWM_DEVINL void res4_unit(const ResCtx& c, const bf16x8 (&qraw)[NT][4], float scale_log2, bool force_safe, const ResGlobal& gm, float n_pad,
                          OutPtr&& out_ptr, ReloadQ&& reload_q, AfterLoop&& after_loop, long long* tr = nullptr) {
    Res4Regs<NT> r;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) r.qf[t][ks] = scale_log2 == 1.f ? qraw[t][ks] : scale_frag(qraw[t][ks], scale_log2);
        r.ls[t][0] = r.ls[t][1] = 0.f;
        r.ot[t][0] = zero;
        r.ot[t][1] = zero;
    }
    // prologue: fragments of key step 0, S(0, 0)
#pragma unroll
    for (int m = 0; m < 4; ++m) r.kf[m] = res4_kread(c, c.kimg, m);
#pragma unroll
    for (int i = 0; i < 4; ++i) r.vf[i >> 1][i & 1] = res4_vread(c, c.vimg, i >> 1, i & 1);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (m == 0) res4_mfma_s_first(r.s[0], r.kf[m], r.qf[0][m]);
        else res4_mfma_s(r.s[0], r.kf[m], r.qf[0][m]);
    }
    asm volatile("s_nop 15" : "+v"(r.s[0]));                 // E(0, 0) follows at once: the wait states the compiler cannot know about
    const int n = c.nsub;
#ifdef DWM_ATTN_TRACE
    if (tr != nullptr) tr[4] = (long long)__builtin_readcyclecounter();
#endif
    res4_block<NT, 0, true, false, ILV>(r, c, 0);
    int k = 1;
    for (; k + 2 < n; k += 2) {
        res4_block<NT, 1, false, false, ILV>(r, c, k);
        res4_block<NT, 0, false, false, ILV>(r, c, k + 1);
    }
    if (k + 1 < n) {                                        // two steps left: k (odd), k + 1 = n - 1
        res4_block<NT, 1, false, false, ILV>(r, c, k);
        res4_block<NT, 0, false, true, ILV>(r, c, k + 1);
    } else {
        res4_block<NT, 1, false, true, ILV>(r, c, k);
    }
#ifdef DWM_ATTN_TRACE
    if (tr != nullptr) tr[5] = (long long)__builtin_readcyclecounter();
#endif
    after_loop();                                           // (the next head's K / V rows are requested here: res4_heads)
    // row sums: the two lanes of a query, minus the pad keys' contribution (exactly 1 each); acceptance test of the fast path
    bool ok = !force_safe;
    const float lmin = n_pad > 0.f ? 0.015625f : 5.421010862e-20f;
    float l_tot[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float l_half = r.ls[t][0] + r.ls[t][1];
        const auto lsw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_half), __float_as_uint(l_half), false, false);
        l_tot[t] = (__uint_as_float(lsw[0]) + __uint_as_float(lsw[1])) - n_pad;
        ok = ok && (l_tot[t] >= lmin) && (l_tot[t] <= 1.8446744e19f);
    }
    if (__all(ok)) {
        // tile by tile (the order is pinned: all accumulators at once would need 160 arch registers)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x16 o[2] = {r.ot[t][0], r.ot[t][1]};
            res_store_tile(o, l_tot[t], out_ptr(t), c.half);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {                                                // wave-uniform: redo the unit by the online softmax (res_tile_safe)
#pragma unroll 1
        for (int t = 0; t < NT; ++t) {
            bf16x8 q[4];
            reload_q(t, q);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) q[ks] = scale_log2 == 1.f ? q[ks] : scale_frag(q[ks], scale_log2);
            float m_run = -INFINITY, l_run = 0.f;
            f32x16 o[2] = {zero, zero};
            for (int kk = 0; kk < c.nsub; ++kk) res_tile_safe(gm, kk << 5, c.L, c.L0, q, o, m_run, l_run, c.l31, c.half);
            res_store_tile(o, l_run + __shfl_xor(l_run, 32, 64), out_ptr(t), c.half);
        }
    }
}

// The head seam as a function of its own (NOT inlined: inside the head loop - one function with the unrolled tile loops of all tile
// counts - the compiler kept the 152 staging registers in scratch memory and waited for every load before storing it there):
// this wave's pieces of one head's K and V rows (piece i - 8 rows of both images, 16 bytes per lane - belongs to wave i mod 4) are
// REQUESTED as plain global loads, all in flight at once, then - behind the barrier that says everybody is done with the current
// head's images (`sync`) - WRITTEN to the images.
// What the seam costs, measured (profiles/r5d .. r5j_trace4_*, cycles per head at L = 602, tile loop 34-35 k):
//   LDS-DMA from 4 waves, row table read between requests           22-28 k   (an LDS read behind a DMA waits for the DMA)
//   LDS-DMA, computed row offsets, nothing between the requests     20-23 k   (a wave keeps ~2 KiB of LDS-DMA in flight; 12 waves: 7 k)
//   register-staged (this function) + the Q rows                    12-13 k + 9 k  = 12-13 B / cycle / CU for 230 KiB
//   the same with a start stagger inside every XCD                  unchanged (not a contention effect)
//   the same with L2 touches of the next head in the last key steps unchanged seam, tile loop + 13 k (the touches stall the loop)
// i.e. a CU streams line-granular data at ~12 B / cycle whether it comes from HBM or the L2 (its vector L1's miss queue at these
// latencies), and a synchronous seam of 230 KiB per head cannot get under ~18 k cycles: the copy has to run UNDER the tile loop - into
// image rows that all four waves have passed - to beat attn_res_kernel, whose 12 waves hide it in their skew.  Not built; this kernel
// stays opt-in.
// Rows past the end of the sequence are NOT written (their lanes are switched off): they keep the zeros of the kernel's start.
// (19 named pieces, not an array: the compiler kept `uint4 kb[19]` in scratch memory even in this small function)
#define DWM_RES4_PIECES(X_) X_(0) X_(1) X_(2) X_(3) X_(4) X_(5) X_(6) X_(7) X_(8) X_(9) X_(10) X_(11) X_(12) X_(13) X_(14) X_(15) X_(16) X_(17) X_(18)
__device__ __attribute__((noinline)) void res4_copy_head(const bf16_t* __restrict__ k0g, const bf16_t* __restrict__ v0g, const int32_t* tab, int64_t seg1_delta,
                                                         int64_t ho, int L, int L0, int ni, char* kimg, char* vimg, int sync, int store) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int cl = lane & 7, rl = lane >> 3;
    const __attribute__((address_space(1))) bf16_t* const k0 = (const __attribute__((address_space(1))) bf16_t*)k0g;     // global, not flat, loads
    const __attribute__((address_space(1))) bf16_t* const v0 = (const __attribute__((address_space(1))) bf16_t*)v0g;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    typedef const __attribute__((address_space(1))) u32x4* gptr;
    const __attribute__((address_space(3))) int32_t* const ltab = (const __attribute__((address_space(3))) int32_t*)tab;     // LDS, not flat, reads
    // pieces past the last one re-load the last one: unconditional code
#define DWM_RES4_LOAD(J_)                                                                                   \
    u32x4 kb##J_, vb##J_;                                                                                   \
    {                                                                                                       \
        int i = wave + J_ * 4;                                                                              \
        i = i < ni ? i : ni - 1;                                                                            \
        const int r = i * 8 + rl;                                                                           \
        const int rc = r < L ? r : L - 1;                                                                   \
        const int64_t off = ((int64_t)ltab[rc] << 3) + (rc < L0 ? 0 : seg1_delta) + ho;                     \
        kb##J_ = *(gptr)(k0 + off + ((cl ^ ((r >> 1) & 7)) << 3));                                          \
        vb##J_ = *(gptr)(v0 + off + ((cl ^ (((r >> 1) & 1) << 2)) << 3));                                   \
    }
    DWM_RES4_PIECES(DWM_RES4_LOAD)
#undef DWM_RES4_LOAD
    if (sync) __syncthreads();                               // everybody is done with the current head's images
    if (store) {
#define DWM_RES4_STORE(J_)                                                                                  \
        {                                                                                                   \
            const int i = wave + J_ * 4;                                                                    \
            if (i < ni) {                                                                                   \
                const int r = i * 8 + rl;                                                                   \
                if (r < L) {                                                                                \
                    *(u32x4*)(kimg + i * 1024 + lane * 16) = kb##J_;                                        \
                    *(u32x4*)(vimg + i * 1024 + lane * 16) = vb##J_;                                        \
                }                                                                                           \
            }                                                                                               \
        }
        DWM_RES4_PIECES(DWM_RES4_STORE)
#undef DWM_RES4_STORE
    }
}
#undef DWM_RES4_PIECES

// the persistent head loop of one wave with NT query tiles per head (tiles t0 .. t0 + NT - 1)
template <int NT, bool ILV>
DWM_DEVINL void res4_heads(const AttnParams& P, char* smem, int t0) {
    constexpr int NW = 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int L = P.L, L0 = P.L0;
    const int Lp = (L + 31) & ~31;
    const int Lt = (L + 3) & ~3;
    char* const kimg = smem;
    char* const vimg = smem + Lp * 128;
    int32_t* const tabs = (int32_t*)(smem + 2 * Lp * 128);
    int32_t* const otab = tabs + 2 * Lt;

    ResCtx c;
    c.kimg = kimg; c.vimg = vimg; c.rowtab = tabs;
    c.L = L; c.L0 = L0; c.nsub = Lp >> 5;
    c.l31 = l31; c.half = half; c.kswz = (lane >> 1) & 7;
    {
        const int tr_u = lane & 15, tr_g = (lane >> 4) & 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int dcol = dt * 32 + tr_g * 16 + (tr_u & 3) * 4;
            const int keyA = half * 4 + (tr_u >> 2), keyB = keyA + 8;
            c.vra[dt] = keyA * 128 + (((dcol >> 3) ^ (((keyA >> 1) & 1) << 2)) << 4) + ((dcol & 7) << 1);
            c.vrb[dt] = keyB * 128 + (((dcol >> 3) ^ (((keyB >> 1) & 1) << 2)) << 4) + ((dcol & 7) << 1);
        }
    }
    const int hpb = P.hpb;
    const int n_items = P.n_problems * (int)P.fd_heads.d;
    const int n_my = ((int)blockIdx.x < n_items) ? (n_items - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int G = n_my * hpb;
    auto item_of = [&](int g, uint32_t& prob, int64_t& hoff) {
        const int it = g / hpb, hh = g - it * hpb;
        const uint32_t item = blockIdx.x + (uint32_t)it * gridDim.x;
        prob = fdiv(item, P.fd_heads);
        hoff = ((int64_t)(item - prob * P.fd_heads.d) * hpb + hh) * 64;
    };
    auto build_tab = [&](int32_t* tab, int32_t* ot, uint32_t prob) {
        const int64_t base0 = seg0_base(P.rm, (int)prob);
        for (int l = tid; l < L; l += NW * 64) {
            const int64_t r0 = l < L0 ? seg0_row(P.rm, base0, l) : 0;
            if (tab != nullptr) tab[l] = (int32_t)((l < L0 ? r0 * P.ld0 : ((int64_t)prob * P.L1 + (l - L0)) * P.ld1) >> 3);
            if (ot != nullptr) ot[l] = (int32_t)((l < L0 ? r0 * P.ldo0 : ((int64_t)prob * P.L1 + (l - L0)) * P.ldo1) >> 3);
        }
    };
    const int ni = (Lp >> 5) * 4;
    auto load_q = [&](bf16x8 (&qdst)[NT][4], const int32_t* tab, int64_t ho) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int lq = (t0 + t) * 32 + l31;
            lq = lq < P.qend ? lq : P.qend - 1;
            const bf16_t* qp = P.q0 + ((int64_t)tab[lq] << 3) + (lq < L0 ? 0 : P.seg1_delta) + ho + half * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qdst[t][ks] = *(const bf16x8*)(qp + ks * 16);
        }
    };
    auto q_ptr = [&](const int32_t* tab, int64_t ho, int t) -> const bf16_t* {
        int lq = (t0 + t) * 32 + l31;
        lq = lq < P.qend ? lq : P.qend - 1;
        return P.q0 + ((int64_t)tab[lq] << 3) + (lq < L0 ? 0 : P.seg1_delta) + ho + half * 8;
    };
    const bool force_safe = P.safe_softmax != 0;
    const float n_pad = (float)(Lp - L);
    if (G == 0) return;
    // development aid (-DDWM_ATTN_TRACE): shader-clock stamps of the 4 waves of workgroups 0-7 at 8 points of every head, written to the
    // (otherwise unused) lse buffer as int64 [8 workgroups][4 waves][64 heads][8] (scripts/experiments/attn_trace4.py)
#ifdef DWM_ATTN_TRACE
#define DWM_TR4(slot_) do { if (P.lse != nullptr && blockIdx.x < 8 && lane == 0 && g < 64) \
        ((long long*)P.lse)[(((int)blockIdx.x * NW + wave) * 64 + g) * 8 + (slot_)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define DWM_TR4(slot_) do {} while (0)
#endif
    // Start stagger (opt-in, P.stagger > 0: units of 8128 cycles per (workgroup / 8 mod 8), i.e. between the workgroups of one XCD): measured
    // without effect on the head seam (profiles/r5i_*) - the seam is not a contention effect, see below.
    if (P.stagger > 0) {
        const int steps = (int)((blockIdx.x >> 3) & 7u) * P.stagger;
        for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    uint32_t prob; int64_t hoff;
    item_of(0, prob, hoff);
    build_tab(tabs, otab, prob);
    // the pad rows of both images: zero for the life of the kernel (16 bytes per lane: 8 lanes per row)
    for (int i = tid; i < (Lp - L) * 8; i += NW * 64) {
        const int off = (L + (i >> 3)) * 128 + (i & 7) * 16;
        *(uint4*)(kimg + off) = make_uint4(0, 0, 0, 0);
        *(uint4*)(vimg + off) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    bf16x8 qn[NT][4];                                        // raw Q fragments of this wave's tiles of the coming head
    res4_copy_head(P.k0, P.v0, tabs, P.seg1_delta, hoff, L, L0, ni, kimg, vimg, 0, 1);
    load_q(qn, tabs, hoff);

    for (int g = 0; g < G; ++g) {
        const int it = g / hpb;
        const int32_t* const tab = tabs + (it & 1) * Lt;
        item_of(g, prob, hoff);
        const bool has_next = g + 1 < G;
        uint32_t nprob = prob; int64_t nhoff = hoff;
        const int32_t* ntab = tab;
        const bool new_item_next = has_next && (g + 1) / hpb != it;
        if (has_next) {
            item_of(g + 1, nprob, nhoff);
            if (new_item_next) {
                ntab = tabs + ((it + 1) & 1) * Lt;
                build_tab((int32_t*)ntab, nullptr, nprob);
            }
        }
        DWM_TR4(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): this wave's share of the head's rows is in the images
        DWM_TR4(1);
        __syncthreads();                                     // ... and everybody else's
        DWM_TR4(2);
        c.rowtab = tab;
        {
            auto out_ptr = [&](int t) -> bf16_t* {
                int lq = (t0 + t) * 32 + l31;
                lq = lq < P.qend ? lq : P.qend - 1;
                return P.o0 + ((int64_t)otab[lq] << 3) + (lq < L0 ? 0 : P.oseg1_delta) + hoff;
            };
            auto reload_q = [&](int t, bf16x8 (&dst)[4]) {
                const bf16_t* qp = q_ptr(tab, hoff, t);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) dst[ks] = *(const bf16x8*)(qp + ks * 16);
            };
            ResGlobal gm;
            gm.k = P.k0 + hoff; gm.v = P.v0 + hoff; gm.tab = tab; gm.seg1_delta = P.seg1_delta;
            auto nothing = [&]() {};
#ifdef DWM_ATTN_TRACE
            res4_unit<NT, ILV>(c, qn, P.scale_log2, force_safe, gm, n_pad, out_ptr, reload_q, nothing,
                               (P.lse != nullptr && blockIdx.x < 8 && lane == 0 && g < 64) ? (long long*)P.lse + (((int)blockIdx.x * NW + wave) * 64 + g) * 8 : nullptr);
#else
            res4_unit<NT, ILV>(c, qn, P.scale_log2, force_safe, gm, n_pad, out_ptr, reload_q, nothing);
#endif
        }
        // the next head's rows (res4_copy_head: requested when this wave's outputs are on their way, written behind the barrier that
        // says everybody is done with the current head's images; after the last head: the same rows once more, not written)
        DWM_TR4(3);
        res4_copy_head(P.k0, P.v0, ntab, P.seg1_delta, nhoff, L, L0, ni, kimg, vimg, 1, has_next ? 1 : 0);
        DWM_TR4(6);
        // (unconditional: a `qn` that is only conditionally redefined stays live across the whole tile loop)
        load_q(qn, ntab, nhoff);                             // waited for behind the barrier of the head top, under the unit's set-up
        if (new_item_next) build_tab(nullptr, otab, nprob);
        DWM_TR4(7);
    }
}

template <bool ILV>
__global__ void __launch_bounds__(256, 1)
attn_res4_kernel(const AttnParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // this wave's query tiles of every head: nqt / 4 (+ 1 for the first nqt % 4 waves) adjacent tiles; the host side launches
    // this kernel for 8 <= nqt <= 20 only (2..5 tiles per wave)
    const int nqt = (P.qend + 31) >> 5;
    const int q4 = nqt >> 2, x4 = nqt & 3;
    const int cnt = q4 + (wave < x4 ? 1 : 0);
    const int t0 = wave * q4 + (wave < x4 ? wave : x4);
    switch (cnt) {
        case 2: res4_heads<2, ILV>(P, smem, t0); break;
        case 3: res4_heads<3, ILV>(P, smem, t0); break;
        case 4: res4_heads<4, ILV>(P, smem, t0); break;
        default: res4_heads<5, ILV>(P, smem, t0); break;
    }
}

}  // namespace

// Called by dwm_attention_fwd (attention.hip) for the launches this kernel covers: unmasked self-attention whose K / V rows of a head
// fit the LDS, 8 <= query tiles <= 20 (225 <= L <= 608: two to five tiles per wave).  P, nblk, lds: as for attn_res_kernel.
int dwm_attn_res4_launch(const dwm_attn::AttnParams& P, unsigned nblk, size_t lds, bool ilv, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_res4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_res4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (ilv) hipLaunchKernelGGL(attn_res4_kernel<true>, dim3(nblk), dim3(256), lds, s, P);
    else hipLaunchKernelGGL(attn_res4_kernel<false>, dim3(nblk), dim3(256), lds, s, P);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DWM_OK : (int)e;
}
