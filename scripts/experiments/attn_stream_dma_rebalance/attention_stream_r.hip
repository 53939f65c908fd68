// attn_stream_kernel: the resident attention forward (attention.hip: attn_res_kernel) with ONE wave per SIMD and NO synchronous head
// seam.  Its own translation unit because of its flags: -fno-slp-vectorize (the row-sum adds of the tile loop must stay scalar) and no
// -amdgpu-mfma-vgpr-form (build.py AGPR_SOURCES: the builtin MFMAs accumulate in AGPRs).
//
// Round 5 built the one-wave-per-SIMD tile loop (attention_res4.hip, now under scripts/experiments/: every K / V fragment read feeds up to
// five MFMAs, 34-35 k cycles per head at L = 602 against ~45 k of the 12-wave kernel) and measured why it lost: K, V (154 KiB) and Q
// (77 KiB) of the next head were fetched BETWEEN two heads, 21 k cycles with nothing to hide under, plus 13-17 k of set-up and stores
// (DESIGN.md s8).  This kernel keeps that loop and removes the seam by taking the three operands off the one LDS path:
//   * V is the only operand that needs the LDS (the PV MFMA wants it transposed: ds_read_b64_tr_b16).  One head's V rows are 76 KiB, so
//     TWO images fit: the next head's V is copied by LDS-DMA into the other image under the tile loop - one 1-KiB request per wave and
//     key step - and one barrier per head swaps them.  No ring, no barrier inside the loop.
//   * K fragments are the lanes' own rows (16 B per lane and MFMA: row l31, 8 d of the 64): they come straight from global memory (L2),
//     requested two key steps ahead into a second fragment set.  Each wave reads a head's K once (76 KiB x 4 waves per head through the
//     vector L1: a tenth of its rate) - only possible because a wave's fragment feeds all its 2..5 query tiles.  The request pipeline
//     runs across the head boundary: the last two key steps request the first two of the next head.
//   * Q fragments live in AGPRs as the B operand of the S MFMAs (inline asm, "a" constraint).  That frees 80 arch VGPRs during the tile
//     loop (the second K set, the address arithmetic); the next head's Q rows are requested into arch VGPRs as soon as the loop is over,
//     travel under the normalisation of the outputs and move to the accumulator file at the head seam.  The softmax scale is expected
//     folded into Q by the producer (dwm_attn_args.variant bit 15, or scale * log2(e) == 1); otherwise they are scaled on that way.
//   * the output accumulators are not zeroed: the first PV MFMA of a tile takes the inline constant 0 as its C operand.
// Per head and wave what is left outside the tile loop: row sums, normalisation, 4 x NT 16-byte stores, one s_waitcnt vmcnt(0) (this
// wave's V requests, Q, the first K fragments), one barrier.
// Images, swizzles, row tables, persistent workgroups, the maximum-free fast path with its acceptance test and fallback, zero pad rows
// and the register-exchange stores are attn_res_kernel's / round 5's.
#include <atomic>

#include "attention_common.h"

using namespace dwm_attn;

namespace {

typedef const __attribute__((address_space(1))) bf16_t* gbf16p;           // global (not flat) loads
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int NT>
struct StRegs {
    bf16x8 qf[NT][4];            // AGPRs ("a" operands of the S MFMAs)
    f32x16 ot[NT][2];            // AGPRs (builtin MFMA accumulators)
    float ls[NT];                // row sums (one accumulator per tile: its two adds per chunk sit ~50 cycles apart - no chain to break)
    f32x16 s[2];
    bf16x8 p[2][2];
    bf16x8 kf[2][4];             // [key step parity][MFMA m]
    bf16x8 vf[2][2];             // [16-key half s2][d tile dt]
};

// what a wave needs to address the rows of the current and of the next head
struct StSrc {
    gbf16p k0, v0, q0;
    const int32_t *tab, *ntab;   // row tables (LDS) of the current / next head's item
    int64_t hoff, nhoff;         // column offset of the current / next head
    int L, L0, n;                // n: key steps per head
    int l31, half, lane, wave;
    int vsw;                     // this lane's swizzled 16-byte chunk of a V row in an LDS-DMA request (a lane constant: see st_vdma)
    bool has_next;
    uint32_t nv_lds;             // LDS byte address of the next head's V image
    int vp0;                     // first V piece (8 rows of a key step) this wave requests; it requests ND consecutive ones
    int64_t d1;                  // FAR form only: (q1 - q0) == (k1 - k0) == (v1 - v0) in elements, added to the rows of segment 1
};

// S MFMAs: inline asm, score accumulator in arch VGPRs (the VALU reads it), K fragment from arch VGPRs, Q fragment from AGPRs.
// What the compiler cannot see holds by construction and is checked on the generated code (scripts/dev/check_stream_asm.py): a chain
// of four accumulates on one tuple, early-clobber destination, and the first VALU read of a chain's result >= 11 wait states behind
// the chain's last MFMA (4 PV MFMAs or an explicit s_nop in between).
DWM_DEVINL void st_mfma_s_first(f32x16& acc, const bf16x8& k, const bf16x8& q) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(k), "a"(q));
}
DWM_DEVINL void st_mfma_s(f32x16& acc, const bf16x8& k, const bf16x8& q) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(k), "a"(q));
}
DWM_DEVINL bf16x8 st_vread(const ResCtx& c, const char* vl, int s2, int dt) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vl + c.vra[dt] + s2 * (16 * 128)));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vl + c.vrb[dt] + s2 * (16 * 128)));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
// Row tables of this kernel: one int32 per token, its row's offset from q0 / k0 / v0 in 16-byte units with the segment-1 displacement
// folded in (the host side refuses launches where that does not fit), padded to whole key steps with the last row's entry - no
// segment select and no clamp on the way from an entry to an address.
// row table entry of key step j (j >= n: step j - n of the next head), this lane's key row
DWM_DEVINL int32_t st_ktab(const StSrc& x, int j) {
    const bool nxt = j >= x.n;
    const int jj = nxt ? j - x.n : j;
    return (nxt ? x.ntab : x.tab)[jj * 32 + x.l31];
}
// one 128-byte row of zeros in device memory: the source of every pad row (keys past the end of a ragged sequence).  A pad key then
// scores exactly 0 (P' = 1, corrected in the row sum) and its V row is zero - without a branch or an EXEC mask anywhere in the tile loop
// (control flow inside the key step lets the compiler sink the row-sum adds to the end of the step: 80 live registers at 5 tiles)
__device__ const uint4 st_zero_row[8] = {};

// this lane's K row of key step j (its 16 bytes of MFMA m are at + m * 16 elements)
// FAR (here and below): the two segments of the launch lie further apart than the 32-bit table entries can fold in (+-16 GiB): the tables
// then hold each row's offset inside its own segment and the displacement is added per row (three more vector instructions per address)
template <bool FAR>
DWM_DEVINL gbf16p st_kptr(const StSrc& x, int j, int32_t tabv) {
    const bool nxt = j >= x.n;
    const int jj = nxt ? j - x.n : j;
    const int key = jj * 32 + x.l31;
    gbf16p row = (x.k0 + (nxt ? x.nhoff : x.hoff)) + ((int64_t)(tabv + x.half) << 3);             // (half: the lane's 8 of 16 k values = one unit)
    if constexpr (FAR) row += key >= x.L0 ? x.d1 : (int64_t)0;
    return key < x.L ? row : (gbf16p)st_zero_row + x.half * 8;
}
DWM_DEVINL void st_kload(bf16x8 (&kf)[4], const gbf16p kp, int m) {
    kf[m] = *(const __attribute__((address_space(1))) bf16x8*)(kp + m * 16);
}
// one LDS-DMA request of the next head's V rows: piece i = wave + 4 k (8 rows x 128 B, 16 B per lane, lane-linear destination; the chunk
// swizzle of the image is applied on the source column)
DWM_DEVINL int32_t st_vtab_p(const StSrc& x, int k, int p) {
    return x.ntab[k * 32 + p * 8 + (x.lane >> 3)];
}
DWM_DEVINL int32_t st_vtab(const StSrc& x, int k) { return st_vtab_p(x, k, x.wave); }
template <bool FAR>
DWM_DEVINL void st_vdma_p(const StSrc& x, int k, int32_t tabv, int p);
template <bool FAR>
DWM_DEVINL void st_vdma(const StSrc& x, int k, int32_t tabv) { st_vdma_p<FAR>(x, k, tabv, x.wave); }
template <bool FAR>
DWM_DEVINL void st_vdma_p(const StSrc& x, int k, int32_t tabv, int p) {
    const int r = k * 32 + p * 8 + (x.lane >> 3);
    // (the chunk swizzle (lane & 7) ^ 4 ((r >> 1) & 1) depends on bit 1 of the row only, which is bit 1 of lane >> 3: x.vsw)
    gbf16p row = (x.v0 + x.nhoff) + ((int64_t)(tabv + x.vsw) << 3);
    if constexpr (FAR) row += r >= x.L0 ? x.d1 : (int64_t)0;
    const gbf16p src = r < x.L ? row : (gbf16p)st_zero_row + x.vsw * 8;
    const uint32_t dst = x.nv_lds + (uint32_t)(p + 4 * k) * 1024u;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
}

// key step k of a unit: slots (k, 0) .. (k, NT - 1); slot (k, t) holds
//     S(u+1) = K Q^T (4 MFMAs)  ||  E(u): P' = 2^S, row sums, bf16 pack  ||  PV(u-1): O^T += V^T P'^T (4 MFMAs)        u = (k, t), k-major
// as one instruction stream of 8 chunks (one MFMA + one slice of E each, order pinned by sched_barrier).  KP = k & 1 (the parity of
// unit (k, 0) follows from it), FIRST: k = 0 (no PV in slot 0; the first PV MFMAs of a tile start from C = 0), LAST: k = n - 1 (no S in
// the last slot, the trailing PV).
// K fragments: S(k, t) reads set KP, S(k+1, 0) (slot NT - 1) set KP ^ 1; the fragments of step k + 2 are requested into set KP right
// behind the last MFMA that reads it (S(k, NT-1), slot NT - 2): a whole key step of distance.  V fragments are single-buffered: step
// k's are requested right behind the last PV MFMA of step k - 1 (slot 0).  The head of the step reads the row-table entries of the requests
// one step ahead; this wave's V request of the NEXT head goes out in the last slot, behind the K waits (tk: K row of step k + 3, tv: V row of step k + 1).
template <int NT, int KP, bool FIRST, bool LAST, bool FAR, int ND>
DWM_DEVINL void st_block(StRegs<NT>& r, const ResCtx& c, const StSrc& x, int k, int32_t& tk, int32_t (&tv)[ND > 0 ? ND : 1]) {
    constexpr int PB = (NT & 1) ? KP : 0;                 // parity of unit (k, 0): k * NT mod 2
    const char* const vlc = c.vimg + k * 4096;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // (the Q fragments as accumulator-class values at the head of every key step: should the register allocator hold a tile in arch
    //  VGPRs across a step, its copies into the accumulator file land HERE - far in front of the asm MFMAs that read them - and not as
    //  v_accvgpr_write right before an MFMA, whose wait states the hazard recogniser cannot place for an asm; check_stream_asm.py audits it)
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" : "+a"(r.qf[t][0]), "+a"(r.qf[t][1]), "+a"(r.qf[t][2]), "+a"(r.qf[t][3]));
    // (the requests of this step use the entries read a step ago; the reads for the next step are issued now)
#ifndef ST_X_NO_KLOAD
    const gbf16p kp2 = st_kptr<FAR>(x, k + 2, tk);
#endif
    int32_t tv_now[ND > 0 ? ND : 1];
#pragma unroll
    for (int i = 0; i < ND; ++i) tv_now[i] = tv[i];
    if (!LAST) {
#pragma unroll
        for (int i = 0; i < ND; ++i) tv[i] = st_vtab_p(x, k + 1, x.vp0 + i);
    }
#ifndef ST_X_NO_KLOAD
    tk = st_ktab(x, k + 3);
#endif
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int par = (PB + t) & 1;                     // parity of this slot's unit: S buffer read, P' buffer written
        const bool do_s = !(LAST && t == NT - 1);
        const int ts = t + 1 == NT ? 0 : t + 1;           // S(u+1): tile
        const int ks = t + 1 == NT ? (KP ^ 1) : KP;       // ... and its K fragment set
        const bool do_pv = !(FIRST && t == 0);
        const int tp = t == 0 ? NT - 1 : t - 1;           // PV(u-1): tile
        // the first PV MFMAs of tile tp (its two d tiles at s2 = 0) start from C = 0: tiles 0 .. NT-2 in step 0
        uint32_t pk[8];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            const bool is_s = ch < 4;
            const int mi = ch & 3;
            if (is_s) {
                if (do_s) {
                    if (mi == 0) st_mfma_s_first(r.s[par ^ 1], r.kf[ks][mi], r.qf[ts][mi]);
                    else st_mfma_s(r.s[par ^ 1], r.kf[ks][mi], r.qf[ts][mi]);
                }
#ifndef ST_X_NO_KLOAD
                if (t == NT - 2) st_kload(r.kf[KP], kp2, mi);
#endif
            } else {
                if (do_pv) {
                    const bool fresh = FIRST && (mi >> 1) == 0;        // (tile NT - 1 starts in slot 0 of step 1: zeroed by the caller)
                    r.ot[tp][mi & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.vf[mi >> 1][mi & 1], r.p[par ^ 1][mi >> 1], fresh ? zero : r.ot[tp][mi & 1], 0, 0, 0);
                }
                if (!FIRST && t == 0) r.vf[mi >> 1][mi & 1] = st_vread(c, vlc, mi >> 1, mi & 1);
#if !defined(ST_X_NO_DMA) && !defined(ST_DMA_EARLY)
                // this wave's V request of the next head: in the LAST slot, behind the S MFMAs that wait for the K fragments of step k + 1 -
                // the counter is in order, so those counted waits also cover every request in front of them in the queue: a V request
                // issued at the head of the step is waited for there, 2..4 slots after its issue (shorter than its latency with 2..4
                // tiles per wave); issued here it is a whole step old at the next wait.  Measured: L = 448 (4 / 3 tiles per wave) 719-741
                // against 715-720 TFLOP/s, head period 34.8 k against 36.9 k cycles in the trace builds; L = 602 834-855 against 831-836
                // (profiles/r6p_*).  Unconditional - the last head of a workgroup requests its own rows once more: a branch here splits the
                // key step into basic blocks, and the compiler then drains the K requests at the head of the second one.
                if (t == NT - 1 && mi < ND) st_vdma_p<FAR>(x, k, tv_now[mi < ND ? mi : 0], x.vp0 + mi);      // (one request per PV chunk of the last slot)
#endif
            }
            // slice ch of E(u): scores 2 ch, 2 ch + 1
            {
                const float a = r.s[par][2 * ch], b = r.s[par][2 * ch + 1];
                const float pa = __builtin_amdgcn_exp2f(a), pb = __builtin_amdgcn_exp2f(b);
                // (scalar adds - this file is built with -fno-slp-vectorize: packed fp32 VALU beside MFMAs costs more than the plain
                //  adds it replaces, MI355X_MICROARCH.md "price of one filler")
                float acc = r.ls[t];
                acc += pa;
                acc += pb;
                asm volatile("" : "+v"(acc));               // pins the adds to their slice too
                r.ls[t] = acc;
                uint32_t w = pack_bf16x2(pa, pb);
                asm volatile("" : "+v"(w));                 // pins the convert to its slice
                pk[ch] = w;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
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

// normalise and store one output tile (attn_res_kernel's store: lane (q, half) holds d = 32 dt + 8 g + 4 half + (0..3) in registers
// 4 g .. 4 g + 3 of o[dt]; after the exchange of one 8-byte piece with lane ^ 32 per pair of g, the lower lane owns the whole
// 16-byte chunk of the even g, the upper lane that of the odd g)
DWM_DEVINL void st_pack_tile(const f32x16 (&o)[2], float l_tot, uint4 (&out)[4]) {
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
            out[dt * 2 + gp] = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
}
DWM_DEVINL void st_store_tile(const uint4 (&v)[4], bf16_t* op, int half) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) *(uint4*)(op + dt * 32 + (2 * gp + half) * 8) = v[dt * 2 + gp];
}
// The same tile as FOUR stores of 16 rows x 64 contiguous bytes instead of four of 32 rows x 32 bytes.  A store instruction costs the
// CU's vector memory path one request per 128-byte line it touches (measured: ~140 cycles for a store of 32 rows, 77 KiB per head at
// ~9 B / cycle, three waves' worth at once at every head seam, profiles/r6g3_attn_stream_timeline_L602.txt); halving the lines per
// instruction halves that.  After st_pack_tile lane (q, half) holds the 16-byte chunks half, 2 + half, 4 + half, 6 + half of ITS row q
// (v[0..3]).  One v_permlane16_swap per dword between the lanes of rows q and q + 16 (q < 16) turns that into
//     lanes q:      row q  chunks half, 4 + half        and   row q + 16  chunks half, 4 + half
//     lanes q + 16: row q  chunks 2 + half, 6 + half    and   row q + 16  chunks 2 + half, 6 + half
// so that the four lanes (q, 0), (q, 1), (q + 16, 0), (q + 16, 1) store chunks 0..3 (then 4..7) of one row in one instruction.
// op_lo / op_hi: this lane's pointers to rows (l31 & 15) and (l31 & 15) + 16 of the tile.
DWM_DEVINL void st_store_tile64(const uint4 (&v)[4], bf16_t* op_lo, bf16_t* op_hi, int l31, int half) {
    uint32_t y[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, x[4] = {v[1].x, v[1].y, v[1].z, v[1].w};      // chunks half / 2 + half
    uint32_t z[4] = {v[2].x, v[2].y, v[2].z, v[2].w}, w[4] = {v[3].x, v[3].y, v[3].z, v[3].w};      // chunks 4 + half / 6 + half
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // lanes 16-31 (48-63) of the first operand <-> lanes 0-15 (32-47) of the second
        const auto a = __builtin_amdgcn_permlane16_swap(y[i], x[i], false, false);
        y[i] = a[0]; x[i] = a[1];             // y: rows q (lanes q: chunk half, lanes q + 16: chunk 2 + half); x: rows q + 16, the same chunks
        const auto b = __builtin_amdgcn_permlane16_swap(z[i], w[i], false, false);
        z[i] = b[0]; w[i] = b[1];             // z: rows q, chunks 4 + half / 6 + half; w: rows q + 16
    }
    const int ch = ((l31 >> 4) << 1) + half;                      // this lane's chunk inside a 64-byte run
    *(uint4*)(op_lo + ch * 8) = make_uint4(y[0], y[1], y[2], y[3]);
    *(uint4*)(op_lo + 32 + ch * 8) = make_uint4(z[0], z[1], z[2], z[3]);
    *(uint4*)(op_hi + ch * 8) = make_uint4(x[0], x[1], x[2], x[3]);
    *(uint4*)(op_hi + 32 + ch * 8) = make_uint4(w[0], w[1], w[2], w[3]);
}

// The fallback of one query tile (a unit whose fast-path sums left the safe range: the textbook online softmax from global memory,
// res_tile_safe) as a function of its own - NOT inlined: its ~200 registers would otherwise be part of the head loop's allocation
// problem (Q fragments spilled to scratch memory in the last key steps of EVERY head); behind the call only the cold path pays for
// saving what is live.
__device__ __attribute__((noinline)) void st_fallback_tile(const bf16_t* qp, bf16_t* op, const bf16_t* k, const bf16_t* v, const int32_t* tab,
                                                           int64_t seg1_delta, int L, int L0, int n, float scale_log2) {
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, half = lane >> 5;
    ResGlobal gm;
    gm.k = k; gm.v = v; gm.tab = tab; gm.seg1_delta = seg1_delta;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bf16x8 q[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) q[ks] = *(const bf16x8*)(qp + ks * 16);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) q[ks] = scale_log2 == 1.f ? q[ks] : scale_frag(q[ks], scale_log2);
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2] = {zero, zero};
    for (int kk = 0; kk < n; ++kk) res_tile_safe(gm, kk << 5, L, L0, q, o, m_run, l_run, l31, half);
    uint4 pkd[4];
    st_pack_tile(o, l_run + __shfl_xor(l_run, 32, 64), pkd);
    st_store_tile(pkd, op, half);
}

// the persistent head loop of one wave with NT query tiles per head (tiles t0 .. t0 + NT - 1); NODD: the number of key steps is odd
// (a template parameter, not a branch behind the main loop: the two tails - one or two peeled steps - would join with 160 accumulator
// registers live, and the register allocator reconciles the two paths through scratch memory)
#ifndef ST_QLOAD64
#define ST_QLOAD64 0                           // (experiment builds: 1 = Q rows requested as 64-byte runs)
#endif
#ifndef ST_STORE64_MAX_NT
#define ST_STORE64_MAX_NT 4                    // (experiment builds override: 5 = the 64-byte-run stores for every tile count, 0 = never)
#endif
template <int NT, bool NODD, bool FAR, int ND>
DWM_DEVINL void st_heads(const AttnParams& P, char* smem, int t0, int vp0) {
    constexpr int NW = 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int L = P.L, L0 = P.L0;
    const int Lp = (L + 31) & ~31;
    const int Lt = Lp;                                       // table pitch: whole key steps (pad entries = the last row's)
    char* const vimg0 = smem;                                // two V images, then the row tables
    int32_t* const tabs = (int32_t*)(smem + 2 * Lp * 128);
    int32_t* const otab = tabs + 2 * Lt;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;

    ResCtx c;
    c.kimg = nullptr; c.vimg = vimg0; c.rowtab = tabs;
    c.L = L; c.L0 = L0; c.nsub = Lp >> 5;
    c.kswz = 0;
    StSrc x;
    // this lane's constants of the tile loop (row / chunk offsets of its fragment reads and requests), derived from an OPAQUE copy of the
    // lane id at the top of every head: as kernel-lifetime values they are live across the head seam - where the next head's 80 Q
    // registers are in flight - and the 5-tile form spilled three of them there; their reloads in front of the tile loop are pending
    // loads in the compiler's bookkeeping at the loop's head, which made it drain the K requests (vmcnt(0)) in every trip
    auto lane_consts = [&]() {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int hf = ln >> 5;
        c.l31 = ln & 31; c.half = hf;
        const int tr_u = ln & 15, tr_g = (ln >> 4) & 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int dcol = dt * 32 + tr_g * 16 + (tr_u & 3) * 4;
            const int keyA = hf * 4 + (tr_u >> 2), keyB = keyA + 8;
            c.vra[dt] = keyA * 128 + (((dcol >> 3) ^ (((keyA >> 1) & 1) << 2)) << 4) + ((dcol & 7) << 1);
            c.vrb[dt] = keyB * 128 + (((dcol >> 3) ^ (((keyB >> 1) & 1) << 2)) << 4) + ((dcol & 7) << 1);
        }
        x.l31 = ln & 31; x.half = hf; x.lane = ln; x.wave = wave;
        x.vsw = (ln & 7) ^ ((((ln >> 3) >> 1) & 1) << 2);
    };
    lane_consts();
    const int hpb = P.hpb;
    const int n_items = P.n_problems * (int)P.fd_heads.d;
    const int n_my = ((int)blockIdx.x < n_items) ? (n_items - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int G = n_my * hpb;
    if (G == 0) return;
    auto item_of = [&](int g, uint32_t& prob, int64_t& hoff) {
        const int it = g / hpb, hh = g - it * hpb;
        const uint32_t item = blockIdx.x + (uint32_t)it * gridDim.x;
        prob = fdiv(item, P.fd_heads);
        hoff = ((int64_t)(item - prob * P.fd_heads.d) * hpb + hh) * 64;
    };
    auto build_tab = [&](int32_t* tab, int32_t* ot, uint32_t prob) {
        const int64_t base0 = seg0_base(P.rm, (int)prob);
        int lp0 = (int)threadIdx.x;                             // (re-read, opaque: a kernel-lifetime copy of the thread index is the first thing
        asm volatile("" : "+v"(lp0));                           //  the register allocator spills across the head loop)
        for (int lp = lp0; lp < Lp; lp += NW * 64) {
            const int l = lp < L ? lp : L - 1;
            const int64_t r0 = l < L0 ? seg0_row(P.rm, base0, l) : 0;
            if (tab != nullptr) tab[lp] = (int32_t)((l < L0 ? r0 * P.ld0 : ((int64_t)prob * P.L1 + (l - L0)) * P.ld1 + (FAR ? 0 : P.seg1_delta)) >> 3);
            if (ot != nullptr) ot[lp] = (int32_t)((l < L0 ? r0 * P.ldo0 : ((int64_t)prob * P.L1 + (l - L0)) * P.ldo1 + (FAR ? 0 : P.oseg1_delta)) >> 3);
        }
    };
    const int n = c.nsub;
    const bool force_safe = P.safe_softmax != 0;
    const float n_pad = (float)(Lp - L);
    const float scale_log2 = P.scale_log2;

    x.k0 = (gbf16p)P.k0; x.v0 = (gbf16p)P.v0; x.q0 = (gbf16p)P.q0;
    x.L = L; x.L0 = L0; x.n = n; x.d1 = P.seg1_delta; x.vp0 = vp0;

    // this lane's Q row of tile t of a head (rows past the last query: the last one - same Q, same output, same bytes stored)
    typedef const __attribute__((address_space(3))) int32_t* ltab_t;          // row tables: LDS reads (ds_read), never flat ones
    auto q_ptr = [&](ltab_t tab, int64_t ho, int t) -> gbf16p {
        int lq = (t0 + t) * 32 + l31;
        lq = lq < P.qend ? lq : P.qend - 1;
        asm volatile("" : "+v"(lq));               // (opaque: the row's kernel-invariant address parts are NOT to be kept across the tile loop)
        gbf16p row = (x.q0 + ho) + ((int64_t)(tab[lq] + half) << 3);
        if constexpr (FAR) row += lq >= L0 ? P.seg1_delta : (int64_t)0;
        return row;
    };
    // development aid (-DDWM_ATTN_TRACE): shader-clock stamps of the 4 waves of workgroups 0-7 at 8 points of every head, written to the
    // (otherwise unused) lse buffer as int64 [8 workgroups][4 waves][64 heads][8] (scripts/experiments/attn_trace_stream.py)
#ifdef DWM_ATTN_TRACE
    uint32_t trs[8] = {0, 0, 0, 0, 0, 0, 0, 0};             // (scalar registers: the stamps are wave-uniform; written once per head, below)
#define DWM_TRS(slot_) do { trs[slot_] = (uint32_t)__builtin_readcyclecounter(); } while (0)
#else
#define DWM_TRS(slot_) do {} while (0)
#endif
    StRegs<NT> r;
    // Q of a head: plain (compiler-tracked) loads into arch VGPRs, issued as soon as the tile loop is over; the fragments move into the
    // accumulator file (and are scaled, where the producer did not fold the scale into Q) behind the head's last wait.  (An in-place form -
    // `global_load_dwordx4 a[..]` from inline asm straight into the S MFMAs' operand registers - was built first and is NOT safe: the
    // compiler believes an asm output written at once, and under register pressure it spilled / copied those registers right behind
    // the load, before the bytes had landed: wrong results in some tile-count variants, found by scripts/dev/check_stream_asm.py's audit.)
    auto load_q = [&](bf16x8 (&qn)[NT][4], ltab_t tab, int64_t ho) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#if ST_QLOAD64
            // four loads of 16 rows x 64 contiguous bytes instead of four of 32 rows x 32 bytes (half the lines per instruction, as
            // st_store_tile64): rows (l31 & 15) and (l31 & 15) + 16, this lane's chunk ch of each 64-byte half; take_q undoes the
            // lane exchange with one v_permlane16_swap per dword pair
            int ra = l31 & 15;
            asm volatile("" : "+v"(ra));
            int la = (t0 + t) * 32 + ra, lb = la + 16;
            la = la < P.qend ? la : P.qend - 1;
            lb = lb < P.qend ? lb : P.qend - 1;
            asm volatile("" : "+v"(la), "+v"(lb));
            const int ch = ((l31 >> 4) << 1) + half;
            gbf16p pa = (x.q0 + ho) + ((int64_t)(tab[la] + ch) << 3), pb = (x.q0 + ho) + ((int64_t)(tab[lb] + ch) << 3);
            if constexpr (FAR) { pa += la >= L0 ? P.seg1_delta : (int64_t)0; pb += lb >= L0 ? P.seg1_delta : (int64_t)0; }
            qn[t][0] = *(const __attribute__((address_space(1))) bf16x8*)(pa);
            qn[t][1] = *(const __attribute__((address_space(1))) bf16x8*)(pb);
            qn[t][2] = *(const __attribute__((address_space(1))) bf16x8*)(pa + 32);
            qn[t][3] = *(const __attribute__((address_space(1))) bf16x8*)(pb + 32);
#else
            const gbf16p qp = q_ptr(tab, ho, t);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qn[t][ks] = *(const __attribute__((address_space(1))) bf16x8*)(qp + ks * 16);
#endif
        }
    };
    // (behind an asm `s_waitcnt vmcnt(0)` the compiler still counts the K fragments requested before it as pending loads; merged with
    //  the loop's own state at the head of the tile loop that made it drain every request - vmcnt(0) - once per two key steps in the
    //  5-tile form.  The empty asm makes the fragments the results of this statement: nothing pending.)
    auto launder_k = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(r.kf[i][0]), "+v"(r.kf[i][1]), "+v"(r.kf[i][2]), "+v"(r.kf[i][3]));
    };
    // (the empty asm DEFINES the fragments as accumulator-class registers: without it the compiler keeps them in arch VGPRs and copies
    //  each one into a scratch AGPR tuple in front of every S MFMA - 4 extra instructions per MFMA and, worse, a v_accvgpr_write -> MFMA
    //  read without the wait states the hazard needs, which it cannot see inside the asm: wrong scores.  check_stream_asm.py audits it.)
    auto take_q = [&](bf16x8 (&qn)[NT][4]) {
#if ST_QLOAD64
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                u32x4 a = *reinterpret_cast<u32x4*>(&qn[t][2 * pr]), b = *reinterpret_cast<u32x4*>(&qn[t][2 * pr + 1]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(a[i], b[i], false, false);
                    a[i] = sw[0]; b[i] = sw[1];
                }
                qn[t][2 * pr] = *reinterpret_cast<bf16x8*>(&a);
                qn[t][2 * pr + 1] = *reinterpret_cast<bf16x8*>(&b);
            }
#endif
        if (scale_log2 != 1.f) {                              // (one wave-uniform branch, not one per fragment)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qn[t][ks] = scale_frag(qn[t][ks], scale_log2);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) r.qf[t][ks] = qn[t][ks];
            asm volatile("" : "+a"(r.qf[t][0]), "+a"(r.qf[t][1]), "+a"(r.qf[t][2]), "+a"(r.qf[t][3]));
        }
    };

    // ---- start of the workgroup: tables of the first head's item and of the second head's (if it is another one), the first head's
    //      V rows, Q and first K fragments - nothing to hide these under
    uint32_t prob; int64_t hoff;
    item_of(0, prob, hoff);
    build_tab(tabs, otab, prob);
    if (G > 1 && hpb == 1) {                                  // (hpb > 1: head 1 belongs to the same item)
        uint32_t p1; int64_t h1;
        item_of(1, p1, h1);
        build_tab(tabs + Lt, nullptr, p1);
    }
    __syncthreads();
    {
        x.tab = tabs; x.ntab = tabs; x.hoff = hoff; x.nhoff = hoff; x.has_next = true; x.nv_lds = lds0;
        for (int k = 0; k < n; ++k) st_vdma<FAR>(x, k, st_vtab(x, k));
        bf16x8 q0[NT][4];
        load_q(q0, (ltab_t)tabs, hoff);
        const gbf16p kp0 = st_kptr<FAR>(x, 0, st_ktab(x, 0)), kp1 = st_kptr<FAR>(x, 1, st_ktab(x, 1));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            st_kload(r.kf[0], kp0, m);
            st_kload(r.kf[1], kp1, m);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0), as the builtin: the compiler's own bookkeeping sees the drain
        launder_k();
        take_q(q0);
    }
    __syncthreads();

    for (int g = 0; g < G; ++g) {
        const int it = g / hpb;
        const int32_t* const tab = tabs + (it & 1) * Lt;
        item_of(g, prob, hoff);
        const bool has_next = g + 1 < G;
        uint32_t nprob = prob; int64_t nhoff = hoff;
        const int32_t* ntab = tab;
        if (has_next) {
            item_of(g + 1, nprob, nhoff);
            if ((g + 1) / hpb != it) ntab = tabs + ((it + 1) & 1) * Lt;
        }
        c.rowtab = tab;
        c.vimg = vimg0 + (g & 1) * (Lp * 128);
        x.tab = tab; x.ntab = ntab; x.hoff = hoff; x.nhoff = nhoff; x.has_next = has_next;
        x.nv_lds = lds0 + (uint32_t)((g + 1) & 1) * (uint32_t)(Lp * 128);

        // ---- the unit: this wave's NT tiles against all keys of the head
        DWM_TRS(0);
#ifdef DWM_ATTN_TRACE
        trs[6] = 0;
#endif
        lane_consts();
#pragma unroll
        for (int t = 0; t < NT; ++t) r.ls[t] = 0.f;
        {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            r.ot[NT - 1][0] = zero;                                // (the other tiles' first PV MFMAs take C = 0)
            r.ot[NT - 1][1] = zero;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) r.vf[i >> 1][i & 1] = st_vread(c, c.vimg, i >> 1, i & 1);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m == 0) st_mfma_s_first(r.s[0], r.kf[0][m], r.qf[0][m]);
            else st_mfma_s(r.s[0], r.kf[0][m], r.qf[0][m]);
        }
        int32_t tk = st_ktab(x, 2);
        int32_t tv[ND > 0 ? ND : 1];
#pragma unroll
        for (int i = 0; i < ND; ++i) tv[i] = st_vtab_p(x, 0, x.vp0 + i);
        asm volatile("s_nop 15" : "+v"(r.s[0]));              // E(0, 0) follows at once: the wait states the compiler cannot know about
        st_block<NT, 0, true, false, FAR, ND>(r, c, x, 0, tk, tv);
        DWM_TRS(1);
        int k = 1;
        for (; k + 2 < n; k += 2) {
            st_block<NT, 1, false, false, FAR, ND>(r, c, x, k, tk, tv);
            st_block<NT, 0, false, false, FAR, ND>(r, c, x, k + 1, tk, tv);
        }
        DWM_TRS(7);
        if constexpr (NODD) {                                   // two steps left: k (odd), k + 1 = n - 1
            st_block<NT, 1, false, false, FAR, ND>(r, c, x, k, tk, tv);
            st_block<NT, 0, false, true, FAR, ND>(r, c, x, k + 1, tk, tv);
        } else {
            st_block<NT, 1, false, true, FAR, ND>(r, c, x, k, tk, tv);
        }
        DWM_TRS(2);
        // ---- the next head's Q rows.  (The table pointers are made opaque here: left alone the compiler computes the five Q
        //      and five output row addresses BEFORE the tile loop and keeps them in scratch memory across it.)
        uint32_t ntab_a = (uint32_t)(uintptr_t)(ltab_t)ntab, otab_a = (uint32_t)(uintptr_t)(ltab_t)otab;
        asm volatile("" : "+s"(ntab_a), "+s"(otab_a));
        const ltab_t ntab_l = (ltab_t)(uintptr_t)ntab_a, otab_l = (ltab_t)(uintptr_t)otab_a;
        // row sums: the two lanes of a query, minus the pad keys' contribution (exactly 1 each); acceptance test of the fast path
        bool ok = !force_safe;
        const float lmin = n_pad > 0.f ? 0.015625f : 5.421010862e-20f;
        float l_tot[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float l_half = r.ls[t];
            const auto lsw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_half), __float_as_uint(l_half), false, false);
            l_tot[t] = (__uint_as_float(lsw[0]) + __uint_as_float(lsw[1])) - n_pad;
            ok = ok && (l_tot[t] >= lmin) && (l_tot[t] <= 1.8446744e19f);
        }
        auto out_row = [&](int t, int row) -> bf16_t* {            // output row `row` (0..31) of tile t (rows past the last query: the last one)
            asm volatile("" : "+v"(row));                        // (opaque BEFORE the clamp: the clamped row numbers are loop invariants otherwise - ten
                                                                 //  values hoisted out of the head loop and kept across it in scratch memory at 5 tiles)
            int lq = (t0 + t) * 32 + row;
            lq = lq < P.qend ? lq : P.qend - 1;
            asm volatile("" : "+v"(lq));
            bf16_t* orow = (P.o0 + hoff) + ((int64_t)otab_l[lq] << 3);
            if constexpr (FAR) orow += lq >= L0 ? P.oseg1_delta : (int64_t)0;
            return orow;
        };
        auto out_ptr = [&](int t) -> bf16_t* { return out_row(t, l31); };
#ifdef ST_X_NO_FALLBACK
        ok = true;
#endif
        if (__all(ok)) {
            // tile by tile (the order is pinned: all accumulators at once would need 160 arch registers)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x16 o[2] = {r.ot[t][0], r.ot[t][1]};
                uint4 pkd[4];
                st_pack_tile(o, l_tot[t], pkd);
                // (the 64-byte-run form measured +3..4 % on the whole kernel with 2..4 tiles per wave - L = 448: 664-695 against 648-668
                //  TFLOP/s - and -2 % with 5, where its second row pointer and eight swaps per tile sit in the one place three waves are
                //  serialised on the store path anyway: profiles/r6g5_*)
                if constexpr (NT <= ST_STORE64_MAX_NT) st_store_tile64(pkd, out_row(t, l31 & 15), out_row(t, (l31 & 15) + 16), l31, half);
                else st_store_tile(pkd, out_ptr(t), half);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {                                                // wave-uniform: redo the unit by the online softmax (st_fallback_tile)
#ifdef DWM_ATTN_TRACE
            trs[6] = 0x7fffffffu;                               // (trace builds: slot 6 flags a unit that took the fallback; l_tot of tile 0 in slot 7... see below)
#endif
#pragma unroll 1
            for (int t = 0; t < NT; ++t)
                st_fallback_tile((const bf16_t*)q_ptr((ltab_t)tab, hoff, t), out_ptr(t), P.k0 + hoff, P.v0 + hoff, tab, FAR ? P.seg1_delta : (int64_t)0, L, L0, n, scale_log2);
        }
        // ---- the next head's Q rows: requested behind the stores (requested in front of them, their 80 registers are parked in the
        //      accumulator file under the normalisation - which needs the data, i.e. waits for it before the first store is issued).
        //      Unconditional - the last head of a workgroup requests its own rows once more: a conditional hand-over would keep the
        //      old fragments live beside the new ones.
        bf16x8 qn[NT][4];
        load_q(qn, ntab_l, nhoff);
        // ---- head seam: everything this wave requested has landed (its V requests of the next head, Q, the first K fragments; the
        //      stores too - vmcnt counts them), the tables of the head after the next, one barrier
        DWM_TRS(3);
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0), as the builtin: the compiler's own bookkeeping sees the drain
        DWM_TRS(4);
        launder_k();
        take_q(qn);
        if constexpr (NODD) {                                   // the next head's steps 0 / 1 were requested into sets 1 / 0
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const bf16x8 tmp = r.kf[0][m];
                r.kf[0][m] = r.kf[1][m];
                r.kf[1][m] = tmp;
            }
        }
        __syncthreads();
        // tables (behind the barrier: the slower waves were still reading the ones these replace): the next item's output rows, and
        // the input rows of the head AFTER the next when it opens an item (the next head requests them under its tile loop)
        const bool new_out = has_next && (g + 1) / hpb != it;
        const bool new_in = g + 2 < G && (g + 2) / hpb != (g + 1) / hpb;
        if (new_out) build_tab(nullptr, otab, nprob);
        if (new_in) {
            uint32_t p2; int64_t h2;
            item_of(g + 2, p2, h2);
            build_tab(tabs + (((g + 2) / hpb) & 1) * Lt, nullptr, p2);
        }
        if (new_out || new_in) __syncthreads();
        DWM_TRS(5);
#ifdef DWM_ATTN_TRACE
        if (P.lse != nullptr && blockIdx.x < 8 && lane < 8 && g < 64) {
            uint32_t v = trs[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) v = lane == i ? trs[i] : v;
            ((long long*)P.lse)[(((int)blockIdx.x * NW + wave) * 64 + g) * 8 + lane] = (long long)v;
        }
#endif
    }
}

__global__ void __launch_bounds__(256, 1)
attn_stream_kernel(const AttnParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#ifdef ST_X_STAGGER
    {   // (experiment builds: workgroup classes start ST_X_STAGGER_TICKS of the 100 MHz counter apart - the seams of the classes then fall apart)
        const uint64_t t_in = __builtin_readcyclecounter();
        const uint64_t d = (uint64_t)((blockIdx.x >> 3) % ST_X_STAGGER) * ST_X_STAGGER_TICKS;
        while (__builtin_readcyclecounter() - t_in < d) __builtin_amdgcn_s_sleep(16);
    }
#endif
    // this wave's query tiles of every head: nqt / 4 (+ 1 for the first nqt % 4 waves) adjacent tiles; the host side launches
    // this kernel for 8 <= nqt <= 20 only (2..5 tiles per wave)
    const int nqt = (P.qend + 31) >> 5;
    const int q4 = nqt >> 2, x4 = nqt & 3;
    const int cnt = q4 + (wave < x4 ? 1 : 0);
    const int t0 = wave * q4 + (wave < x4 ? wave : x4);
    const bool nodd = (((P.L + 31) >> 5) & 1) != 0;
    // the V requests of a key step (4 pieces of 8 rows) go to the waves with FEWER query tiles where the tile counts differ by the
    // pattern of the model's two lengths: 19 tiles = 5 / 5 / 5 / 4 (the 4-tile wave requests all four pieces), 14 = 4 / 4 / 3 / 3 (the
    // 3-tile waves two each); everywhere else every wave requests its own piece
    const int n = (P.L + 31) >> 5;
    const int pat = (P.stream_far == 0 && nqt == n) ? (nqt == 19 ? 19 : nqt == 14 ? 14 : 0) : 0;
    if (pat == 19) {
        if (cnt == 5) st_heads<5, true, false, 0>(P, smem, t0, 0);
        else st_heads<4, true, false, 4>(P, smem, t0, 0);
    } else if (pat == 14) {
        if (cnt == 4) st_heads<4, false, false, 0>(P, smem, t0, 0);
        else st_heads<3, false, false, 2>(P, smem, t0, (wave - 2) * 2);
    } else if (P.stream_far == 0) {
        switch (cnt * 2 + (nodd ? 1 : 0)) {
            case 4: st_heads<2, false, false, 1>(P, smem, t0, wave); break;
            case 5: st_heads<2, true, false, 1>(P, smem, t0, wave); break;
            case 6: st_heads<3, false, false, 1>(P, smem, t0, wave); break;
            case 7: st_heads<3, true, false, 1>(P, smem, t0, wave); break;
            case 8: st_heads<4, false, false, 1>(P, smem, t0, wave); break;
            case 9: st_heads<4, true, false, 1>(P, smem, t0, wave); break;
            case 10: st_heads<5, false, false, 1>(P, smem, t0, wave); break;
            default: st_heads<5, true, false, 1>(P, smem, t0, wave); break;
        }
    } else {                                              // segments further apart than the folded 32-bit entries reach
        switch (cnt * 2 + (nodd ? 1 : 0)) {
            case 4: st_heads<2, false, true, 1>(P, smem, t0, wave); break;
            case 5: st_heads<2, true, true, 1>(P, smem, t0, wave); break;
            case 6: st_heads<3, false, true, 1>(P, smem, t0, wave); break;
            case 7: st_heads<3, true, true, 1>(P, smem, t0, wave); break;
            case 8: st_heads<4, false, true, 1>(P, smem, t0, wave); break;
            case 9: st_heads<4, true, true, 1>(P, smem, t0, wave); break;
            case 10: st_heads<5, false, true, 1>(P, smem, t0, wave); break;
            default: st_heads<5, true, true, 1>(P, smem, t0, wave); break;
        }
    }
}

}  // namespace

// Called by dwm_attention_fwd (attention.hip) for the launches this kernel covers: unmasked self-attention whose V rows of a head
// fit the LDS twice, 8 <= query tiles <= 20 (225 <= L <= 608: two to five tiles per wave).
static std::atomic<int64_t> g_stream_launches{0};           // launches served (dwm_attn_stream_launches: diagnostics, relaxed)
extern "C" int64_t dwm_attn_stream_launches(void) { return g_stream_launches.load(std::memory_order_relaxed); }

int dwm_attn_stream_launch(const dwm_attn::AttnParams& P, unsigned nblk, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    // the row tables hold offsets from q0 / k0 / v0 / o0 in 16-byte units as int32: both segment displacements must be whole units
    // (otherwise: -1, the caller keeps attn_res_kernel).  Within +-16 GiB the displacement is folded into the segment-1 entries (no
    // select on the way from an entry to an address); two segments in SEPARATE allocations may lie further apart - a caching
    // allocator on a 288-GB device hands out such pairs - and run the FAR instantiation: entries relative to each segment, the
    // displacement added per row.  Same arithmetic on the same values either way: the results are bit-identical
    // (tests/test_round6_gpu.py places the segments 20 GiB apart).
    if (P.seg1_delta % 8 != 0 || P.oseg1_delta % 8 != 0) return -1;
    const int64_t lim = 1ll << 33;
    dwm_attn::AttnParams Pk = P;
    Pk.stream_far = (P.seg1_delta <= -lim || P.seg1_delta >= lim || P.oseg1_delta <= -lim || P.oseg1_delta >= lim) ? 1 : 0;
    const int Lp = (P.L + 31) & ~31;
    const size_t lds = (size_t)2 * Lp * 128 + (size_t)3 * Lp * sizeof(int32_t);
    if (lds > 160 * 1024) return -1;
#ifdef ST_X_NBLK
    if (nblk > ST_X_NBLK) nblk = ST_X_NBLK;                 // (experiment builds: fewer workgroups - is the seam's store time a per-CU or a chip-wide limit?)
#endif
    hipLaunchKernelGGL(attn_stream_kernel, dim3(nblk), dim3(256), lds, s, Pk);
    g_stream_launches.fetch_add(1, std::memory_order_relaxed);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DWM_OK : (int)e;
}
