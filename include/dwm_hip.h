/*
 * dwm_hip.h — C ABI of libdwm_hip.so: the MI355X (gfx950) kernels under the
 * CTSD SD-3.5 MMDiT denoising hot path of OpenDWM.
 *
 * The reference (SenseTime-FVG/OpenDWM @ 2025-07-04) has NO native / FFI
 * boundary on this path: every op below is, in the
 * reference, a PyTorch/diffusers call made from Python.  Each entry point
 * therefore cites the reference *call site(s)* it replaces; the Python-level
 * drop-in boundary (the JSON "_class_name" model class, src/dwm/common.py:133-179)
 * is mirrored by opendwm_amd/dit.py, which is the only caller of this ABI.
 * INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers (HBM);
 *   - "bf16" = raw uint16 bfloat16 storage; accumulation / statistics in fp32;
 *   - nothing allocates, nothing synchronises, everything is enqueued on the
 *     hipStream_t passed (void* here so the header needs no HIP include);
 *   - every function returns 0 on success, a negative DWM_E* code on invalid
 *     arguments (surfaced as RuntimeError by the Python shim) and the positive
 *     hipError_t value if the launch itself failed;
 *   - re-entrant: no global mutable state that a result depends on, and no
 *     environment reads - kernel selection comes from the argument structs only
 *     (what is process-wide: per-kernel "attribute set" flags and the device's CU
 *     count, both idempotent, and two relaxed-atomic launch counters for diagnostics).
 */
#ifndef DWM_HIP_H
#define DWM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DWM_OK 0
#define DWM_EINVAL (-1)     /* bad shape / null pointer */
#define DWM_EALIGN (-2)     /* pointer or leading dimension not 16-byte aligned */
#define DWM_EUNSUPPORTED (-3)

/* ABI version; bump on any struct change. */
#define DWM_ABI_VERSION 17
int dwm_abi_version(void);
/* SHA-256 (hex) of the sources this library was built from (csrc .hip and .h files + this header, in sorted order), as
 * computed by opendwm_amd/build.py; the Python binding compares it with the sources it finds next to itself and refuses a
 * stale binary. */
const char* dwm_source_hash(void);

/* ------------------------------------------------------------------------
 * GEMM:  C[M,Nout] = epilogue( A[M,K] · W[N,K]^T )
 * bf16 in, fp32 accumulate on v_mfma_f32_32x32x16_bf16, bf16 out.  Optionally an implicit-GEMM
 * convolution (a_map / c_map / tap_shift below).
 * W is a torch.nn.Linear weight ([out,in], K contiguous).  K % 64 == 0,
 * N % 8 == 0, lda/ldc/... % 8 == 0 (16-byte rows).
 *
 * Replaces every torch.nn.Linear on the path:
 *   diffusers JointTransformerBlock (called crossview_temporal_dit.py:517-521):
 *     attn.to_q/k/v, add_{q,k,v}_proj, to_out.0, to_add_out, ff.net.*, norm1.linear ...
 *   VTSelfAttentionBlock (crossview_temporal.py:562-582): ff_in, attn1.*, ff
 *   pos_embed.proj (as im2col GEMM), context_embedder, time_text_embed,
 *   view/time index MLPs (crossview_temporal_dit.py:421-439,528-568), proj_out (:600).
 * ---------------------------------------------------------------------- */
enum {
    DWM_EPI_PLAIN = 0,   /* v = act(acc + bias)                                        */
    DWM_EPI_GEGLU = 1,   /* W/bias packed in 64-row groups [32 value rows | 32 gate rows];
                            out[:, j] = (acc_v + b_v) * gelu_erf(acc_g + b_g); Nout = N/2
                            (diffusers FeedForward "geglu", crossview_temporal.py:548-560) */
    DWM_EPI_RESID = 2,   /* v = act(acc + bias); v *= gate; v += res; v = a*blend + (1-a)*v
                            (gated residual of JointTransformerBlock; residual adds of
                            VTSelfAttentionBlock; AlphaBlender crossview_temporal.py:68-72) */
    DWM_EPI_RMSHEAD = 3  /* v = acc + bias; per 64-column head RMSNorm (affine) on columns
                            < rms_ncols (diffusers Attention qk_norm="rms_norm")          */
};
enum { DWM_ACT_NONE = 0, DWM_ACT_GELU_TANH = 1, DWM_ACT_SILU = 2, DWM_ACT_RELU = 3 };

/* Row map "compact pixel index -> row of a zero-padded [I, rh+2, rw+2] token grid":
 *   row(m) = (m / (rw*rh)) * ipitch + ((m / rw) % rh) * rpitch + (m % rw) * xstep + origin
 * rw == 0 means identity (row(m) = m); xstep == 0 is read as 1.  xstep = 2 with a doubled rpitch
 * addresses every second pixel (stride-2 Downsample2D of the VAE encoder).  Used to run the 3x3 convolutions of the layout
 * ImageAdapter (diffusers AdapterResnetBlock.block1, src/dwm/models/adapters.py:20-22) as
 * implicit GEMM on token-major activations without materialising im2col or padding copies. */
typedef struct dwm_rowmap2d {
    int64_t rw, rh;          /* interior width / height (pixels)                 */
    int64_t rpitch, ipitch;  /* padded row pitch (rw+2) and image pitch, in rows  */
    int64_t origin;          /* row of interior pixel (0,0): rpitch + 1           */
    int64_t xstep;           /* column step (0 or 1 = dense, 2 = stride-2 conv)   */
} dwm_rowmap2d;

typedef struct dwm_gemm_args {
    const void* A;  int64_t lda;          /* bf16 [M,K], row stride lda elements          */
    const void* W;                        /* bf16 [N,K] contiguous                        */
    const void* bias;                     /* bf16 [N] or NULL                             */
    void* C;        int64_t ldc;          /* bf16 [M,Nout]                                */
    int64_t M, N, K;
    int32_t epilogue;                     /* DWM_EPI_*                                    */
    int32_t act;                          /* DWM_ACT_* (PLAIN / RESID)                    */
    /* RESID */
    const void* gate; int64_t ld_gate; int64_t rows_per_gate;   /* bf16 gate[row/rows_per_gate][n] or NULL */
    const void* res;  int64_t ld_res;  int64_t res_mod;         /* bf16 res[res_mod > 0 ? row % res_mod : res_mod < 0 ? row / -res_mod : row][n] or NULL */
    const void* blend; int64_t ld_blend;                        /* bf16 blend[row][n] or NULL (not together with gate) */
    const float* alpha; int64_t rows_per_alpha;                 /* fp32 alpha[row/rows_per_alpha]           */
    /* RMSHEAD */
    const void* rms_w; int64_t rms_ncols; float rms_eps;        /* bf16 rms_w[rms_ncols]                    */
    int32_t reserved;                                           /* ignored (ablation knobs of -DDWM_DEV_HOOKS builds) */
    /* implicit-GEMM convolution (all zero / rw == 0 for a plain GEMM):
     * K = ntaps * k_per_tap; K index (t, c) reads A[a_map(m) + tap_shift[t]][c].  W is [N, ntaps*k_per_tap]. */
    dwm_rowmap2d a_map;                                         /* A rows                                    */
    dwm_rowmap2d c_map;                                         /* C, res and blend rows                     */
    int32_t ntaps; int32_t k_per_tap;
    int64_t tap_shift[27];                                      /* in rows; up to 3x3x3 taps (causal Conv3d) */
    /* split-K (PLAIN / RESID epilogues): when the tile grid fills less than half of the GPU and K is long, the K
     * axis is cut into ranges (one workgroup each), fp32 partial tiles go to `workspace` and a second kernel reduces
     * them in a fixed order and applies the epilogue.  workspace: 16-byte aligned device scratch of workspace_bytes
     * (>= 2 * M * N * 4 to be usable), owned by the caller, one per stream; NULL = never split.
     * split_k: 0 = automatic, 1 = never, > 1 = exactly this many ranges (DWM_EUNSUPPORTED if impossible). */
    void* workspace; int64_t workspace_bytes; int32_t split_k;
    /* fp32 residual stream (RESID, res_mod == 0): when C32 != NULL, `res` and `blend` are fp32 matrices (ld_res / ld_blend in
     * fp32 elements, rows through c_map like C), the result v is written to C32 in fp32 (in place over `res` or `blend` allowed)
     * and - unless C is NULL, which is allowed only here - rounded to the bf16 C as well.  Keeps a chain of residual blocks from
     * accumulating one bf16 storage rounding per block: the layout ImageAdapter (src/dwm/models/adapters.py:40-60: its input -
     * and so its error - is the same at every denoise step) and the hidden / context streams of the MMDiT forward
     * (src/dwm/models/crossview_temporal_dit.py:486-598: ~130 residual adds per forward). */
    void* C32; int64_t ldc32;
    /* tile configuration: 0 = automatic, 1 = 256 x 256 x 64 tiles (one 8-wave workgroup per CU), 2 = 256 x 128 x 32 tiles
     * (two 4-wave workgroups per CU; chosen automatically where it cuts the padded columns, e.g. N = 320 / 640; not with
     * C32; split-K grids keep the 256 x 256 tile), 3 = automatic and the 4-wave kernels (gemm_bf16_4w.hip) may serve the launch,
     * 4 = as 3, their fast form only (A/B measurements) */
    int32_t tile;
} dwm_gemm_args;

int dwm_gemm_bf16(const dwm_gemm_args* args, void* stream);
/* args->tile == 3 ("automatic, 4-wave kernels allowed": what the MMDiT inference forward passes): launches without row maps / taps /
 * split-K and with M % 256 == N % 256 == 0, K % 64 == 0, K >= 128 run the same epilogues on a 4-wave main loop (gemm_bf16_4w.hip; 412
 * against 434 ms per denoise step, profiles/README.md); every other caller keeps the 8-wave kernels, on which the whole GPU suite
 * has run.  (The library reads no environment: the Python host side maps DWM_GEMM4W=1 / 0 / f to tile 3 / 0 / 4 per call.)  This counts
 * the launches served (a relaxed atomic: diagnostics, the library's only process-wide mutable state besides lazily set kernel attributes). */
int64_t dwm_gemm4w_launches(void);
/* ... and how many of them ran the general form of those kernels (ragged M / N, A row map, taps, per-image residual row). */
int64_t dwm_gemm4w_launches_general(void);

/* ------------------------------------------------------------------------
 * Weight-gradient GEMM, both operands row-major with the contraction index as their ROW (gemm_tn.hip):
 *     out[n, t*C + c] = sum_{m < M} A[m, n] * B[clamp(m + tap_shift[t], 0, b_rows - 1), c]       (bf16, fp32 accumulate)
 * A = dY [M, N], B = X: dW = dY^T X of torch.nn.Linear (ntaps = 0) - autograd's weight gradient under loss.backward(),
 * src/dwm/pipelines/ctsd.py:1401-1404 - and, with dY and X on the same zero-bordered padded token grid, the weight gradient of a
 * 3x3 / (3,1,1) / 3x3x3 convolution in one launch (tap t reads the rows shifted by tap_shift[t]; the border rows of A are zero).
 * M % 64 == 0, N % 8 == 0, C % 8 == 0; out is [N, max(ntaps,1)*C] with leading dimension ldo.  The contraction is cut into K
 * ranges (split_k: 0 = automatic, > 0 = exactly this many), fp32 partials go to `workspace` (>= split_k * N * ntaps*C * 4 bytes,
 * 16-byte aligned, owned by the caller, one per stream) and are reduced in range order.
 * ---------------------------------------------------------------------- */
typedef struct dwm_gemm_tn_args {
    const void* A; int64_t lda;
    const void* B; int64_t ldb; int64_t b_rows;
    void* out; int64_t ldo;
    int64_t M, N, C;
    int32_t ntaps; int32_t split_k;
    int64_t tap_shift[27];
    void* workspace; int64_t workspace_bytes;
} dwm_gemm_tn_args;
int dwm_gemm_tn(const dwm_gemm_tn_args* args, void* stream);

/* ------------------------------------------------------------------------
 * Fused multi-head attention forward, head_dim 64, bf16, flash-style online
 * softmax, S^T = K·Q^T on MFMA so each lane owns one query row.
 *
 * One "problem" p is one (batch-like index, head) softmax(QK^T*scale [+mask])V
 * over L = L0 + L1 tokens: segment 0 (L0 "sample" tokens, rows addressed through
 * the row map below) followed by segment 1 (L1 "context" tokens, dense rows
 * p*L1 + l).  Q/K/V/O element (row, head h, d) lives at ptr[row*ld + h*64 + d].
 *
 * Row map of segment 0 (folds the einops.rearrange of
 * crossview_temporal_dit.py:307-315,336-361 into addressing; no copies):
 *   row(p, l) = sum_{i<3} ((p / pdiv[i]) % pmod[i]) * pstride[i]
 *             + (l % ldiv[0]) * lstride[0] + ((l / ldiv[0]) % ldiv[1]) * lstride[1]
 *             + (l / (ldiv[0] * ldiv[1])) * lstride[2]
 *
 * Mask (True/1 = attend), crossview_temporal_dit.py:301-305:
 *   mode 0: none
 *   mode 1: groups — allowed(q,k) = mask[(p/p_per_mask)*G*G + gq*G + gk],
 *           g(l) = (l / group_size) % G      (mask is the [B,V,V] bool tensor)
 *   mode 2: dense uint8 mask[p][Lq][Lk]      (generic VTSelfAttentionBlock API)
 *
 * Replaces F.scaled_dot_product_attention inside diffusers JointAttnProcessor2_0
 * (JointTransformerBlock attn / attn2) and AttnProcessor2_0
 * (VTSelfAttentionBlock.attn1, crossview_temporal.py:572-574).
 * ---------------------------------------------------------------------- */
typedef struct dwm_attn_args {
    const void *q0, *k0, *v0; int64_t ld0;     /* segment 0 (bf16)            */
    const void *q1, *k1, *v1; int64_t ld1;     /* segment 1 or NULL (L1 = 0)  */
    void* o0; int64_t ldo0;
    void* o1; int64_t ldo1;
    int64_t L0, L1;
    int64_t n_problems;                        /* batch-like count (excl. heads) */
    int32_t heads; int32_t head_dim;           /* head_dim must be 64         */
    float scale;
    int32_t mask_mode;
    int64_t pdiv[3], pmod[3], pstride[3];
    int64_t ldiv[2], lstride[3];
    const uint8_t* mask; int64_t mask_G; int64_t group_size; int64_t p_per_mask;
    int32_t variant;                           /* 0 = auto.  Kernel selection (attention.hip): bits 0-3 tiled kernel: 2 = 64
                                                * queries per wave, 1 = 32 (auto: 64 from L = 1024 on); resident kernel: number of
                                                * compute waves (1-12, the other waves of its 12 only copy), bit 4 online softmax with a
                                                * running maximum for every unit of the resident kernels (default: their maximum-free
                                                * fast path with a checked fallback), bit 5 keep the tiled kernel (default for L <= 32:
                                                * the packed short-sequence kernel; for unmasked self-attention with 64 <= L <= 608:
                                                * the resident kernel), bit 7 per-wave form of the group-masked kernel, bits 8-11 heads
                                                * per workgroup / item.  Unmasked self-attention with 225 <= L <= 608 runs the one-wave-per-
                                                * SIMD streaming form of the resident kernel (attention_stream.hip) by default; bit 13 keeps
                                                * the 12-wave form (bit 12: the streaming form, as in round 5's opt-in).  Bit 15 is not a
                                                * kernel choice: q arrives with scale * log2(e) folded in by its producer (`scale` is then
                                                * ignored; forward kernels) */
    int32_t cross;                             /* 1: cross-attention - queries = segment 0 only, keys / values =
                                                * segment 1 only (q1, k0, v0, o1 unused: pass q1 = q0, k0 = k1, v0 = v1);
                                                * diffusers BasicTransformerBlock.attn2 (text conditioning of the SD 2.1 UNet) */
    float* lse;                                /* optional out fp32 [n_problems, heads, L0+L1]: NEGATIVE
                                                * log2-domain log-sum-exp of scale*log2(e)*q.k, i.e.
                                                * P = exp2(scale*log2(e)*q.k + lse) (saved for the backward) */
} dwm_attn_args;

int dwm_attention_fwd(const dwm_attn_args* args, void* stream);
/* Launches of dwm_attention_fwd served by the streaming kernel (attention_stream.hip) in this process (a relaxed atomic: diagnostics).
 * Two-segment launches: while (q1 - q0) and (o1 - o0) lie within +-16 GiB the kernel folds them into its 32-bit row offsets; pairs
 * further apart (separate allocations on a 288-GB device can be) run its FAR instantiation, which adds the displacement per row -
 * the same arithmetic on the same values, bit-identical results (tests/test_round6_gpu.py).  Only displacements that are not
 * multiples of 16 bytes are left to attn_res_kernel. */
int64_t dwm_attn_stream_launches(void);

/* Backward of dwm_attention_fwd (F.scaled_dot_product_attention inside JointAttnProcessor2_0 /
 * AttnProcessor2_0 under autograd).  `fwd` repeats the forward call: q/k/v, the forward OUTPUTS
 * o0/o1 (read here), the row map, masks and fwd.lse as written by the forward.  do0/do1 are laid out
 * like o0/o1 (ldo0/ldo1); dq/dk/dv are addressed like q/k/v through the same row map with leading
 * dimensions ld_d0/ld_d1 and must satisfy dq1-dq0 == dk1-dk0 == dv1-dv0.  delta = caller scratch,
 * fp32 [n_problems, heads, L0+L1].  All 16-byte aligned, ldo % 8 == 0. */
typedef struct dwm_attn_bwd_args {
    dwm_attn_args fwd;
    const void *do0, *do1;
    void *dq0, *dk0, *dv0; int64_t ld_d0;
    void *dq1, *dk1, *dv1; int64_t ld_d1;
    float* delta;
} dwm_attn_bwd_args;
int dwm_attention_bwd(const dwm_attn_bwd_args* args, void* stream);

/* Diagnostic (tests only): one wave issues ds_read_b64_tr_b16 at LDS byte offset offs[lane]
 * (8-byte aligned, < 8192) of an image with img16[i] = i; out[lane*4 + j] = element j. */
int dwm_debug_tr_probe(const int32_t* offs, int16_t* out, void* stream);

/* ------------------------------------------------------------------------
 * LayerNorm family over the last dim D (D % 8 == 0, D <= 8192), one wave per row:
 *   x' = x + addvec[row / rows_per_add]            (optional; x' also written to xsum)
 *   n  = (x' - mean) * rsqrt(var + eps)            (biased variance, fp32 statistics)
 *   y  = n * weight + bias                         (affine, optional)
 *   y  = y * (1 + scale[row/rows_per_mod]) + shift[row/rows_per_mod]     (optional)
 *   y2 = n * (1 + scale2[...]) + shift2[...]       (optional second output)
 * Replaces: AdaLayerNormZero / SD35AdaLayerNormZeroX / AdaLayerNormContinuous
 * normalise+modulate, JointTransformerBlock.norm2 + modulate, torch.nn.LayerNorm
 * norm_in/norm1/norm3 of VTSelfAttentionBlock (crossview_temporal.py:545-558)
 * and the "hidden_states + view_emb / sequence_emb" adds at
 * crossview_temporal_dit.py:229-230,334.
 * ---------------------------------------------------------------------- */
typedef struct dwm_layernorm_args {
    const void* x; int64_t ldx;                /* bf16 [rows, D]                      */
    void* y;       int64_t ldy;                /* bf16                                */
    void* y2;      int64_t ldy2;               /* bf16 or NULL                        */
    void* xsum;    int64_t ldxsum;             /* bf16 or NULL                        */
    int64_t rows; int32_t D; float eps;
    const void* weight; const void* bias;      /* bf16 [D] or NULL                    */
    const void* scale;  const void* shift;  int64_t ld_mod;  int64_t rows_per_mod;
    const void* scale2; const void* shift2;    /* share ld_mod / rows_per_mod          */
    const void* addvec; int64_t ld_add; int64_t rows_per_add;
} dwm_layernorm_args;

int dwm_layernorm(const dwm_layernorm_args* args, void* stream);
/* The same with x (and xsum) in fp32 - ldx / ldxsum in fp32 elements - and everything else as above: the LayerNorms of
 * the bf16 forward when the hidden state is kept as an fp32 residual stream (dwm_gemm_args.C32); the reference keeps the
 * stream in the module dtype (diffusers JointTransformerBlock / crossview_temporal.py:562-582), fp32 on its CPU path. */
int dwm_layernorm_x32(const dwm_layernorm_args* args, void* stream);

/* Stand-alone per-head RMSNorm (in place) over 64-wide heads of x[rows, ncols]:
 * x[r, c] = x[r, c] * rsqrt(mean_head(x^2) + eps) * w[c]; (fallback for shapes the
 * GEMM RMSHEAD epilogue does not cover).  diffusers RMSNorm. */
int dwm_rmsnorm_heads(void* x, int64_t ldx, int64_t rows, int64_t ncols,
                      const void* w, float eps, void* stream);

/* ------------------------------------------------------------------------
 * Small element-wise kernels (HBM-bound glue)
 * ---------------------------------------------------------------------- */
/* y = silu(x), bf16, n elements (n % 8 == 0).  AdaLN "linear(silu(temb))" prologue. */
int dwm_silu(const void* x, void* y, int64_t n, void* stream);

/* diffusers Timesteps(C, flip_sin_to_cos=True, downscale_freq_shift=0):
 * out[i, :C/2] = cos(t_i * f), out[i, C/2:] = sin(t_i * f), f_j = exp(-ln(1e4) * j / (C/2)).
 * t fp32 [n]; out bf16 [n, C].  (crossview_temporal_dit.py:153-154,163-164,435,531,563) */
int dwm_timestep_sinusoid(const float* t, int64_t n, int32_t C, void* out, void* stream);

/* im2col for the p x p / stride p patch conv (SD3 PatchEmbed.proj):
 * x [I, C, H, W] (fp32 if x_is_f32 else bf16) -> out bf16 [I*(H/p)*(W/p), ldo>=C*p*p],
 * column = (c*p + py)*p + px; columns [C*p*p, ldo) are zero-filled. */
int dwm_patchify(const void* x, int32_t x_is_f32, int64_t I, int32_t C, int32_t H, int32_t W,
                 int32_t p, void* out, int64_t ldo, void* stream);

/* inverse of the final einsum "nhwpqc->nchpwq" (crossview_temporal_dit.py:603-621):
 * x bf16 [I*h*w, ldx] with column (py*p + px)*C + c  ->  out bf16 [I, C, h*p, w*p]. */
int dwm_unpatchify(const void* x, int64_t ldx, int64_t I, int32_t C, int32_t h, int32_t w,
                   int32_t p, void* out, void* stream);

/* classifier-free guidance + FlowMatchEuler step (ctsd.py:1548-1575):
 * pred bf16 [2, n] (uncond ; cond), latents fp32 [n] in/out:
 *   latents += dsigma * (u + guidance * (c - u));  model_in bf16 [2, n] (optional) receives
 *   the updated latents duplicated for the next step (ctsd.py:1528,1536-1538). */
int dwm_cfg_euler_step(const void* pred, float* latents, void* model_in, int64_t n,
                       float guidance, float dsigma, void* stream);

/* the same with a per-group step dsigma[i / group_elems] (group = one [C,H,W] frame latent): the
 * per-frame scheduler of the diffusion-forcing mode, FlowMatchEulerDiscreteScheduler.step_by_indices
 * (src/dwm/schedulers/temporal_independent.py:176-197) fused with the `torch.where(in_schedule_range, ...)`
 * of ctsd.py:1565-1572 (pass dsigma = 0 for frames outside the schedule range). */
int dwm_cfg_euler_step_grouped(const void* pred, float* latents, void* model_in, int64_t n, float guidance,
                               const float* dsigma, int64_t group_elems, void* stream);

/* torch.nn.PixelUnshuffle(r) of x [I, C, H, W] (fp32 if x_is_f32 else bf16), written token-major:
 * out bf16 [I*(H/r)*(W/r), ldo >= C*r*r], column = (c*r + dy)*r + dx, zero padded to ldo
 * (src/dwm/models/adapters.py:42). */
int dwm_unshuffle_tokens(const void* x, int32_t x_is_f32, int64_t I, int32_t C, int32_t H, int32_t W,
                         int32_t r, void* out, int64_t ldo, void* stream);

/* torch.nn.AvgPool2d(2, 2) on token-major bf16 [I, h, w, C] -> [I, h/2, w/2, C] (diffusers
 * AdapterBlock.downsample); h, w even, C % 8 == 0. */
int dwm_avgpool2_tokens(const void* x, int64_t I, int32_t h, int32_t w, int32_t C, void* out, void* stream);

/* y += x, bf16, n % 8 == 0 (hidden_states + condition_residual, crossview_temporal_dit.py:491-494). */
int dwm_add_inplace(void* y, const void* x, int64_t n, void* stream);
/* y (bf16) += x (fp32), the sum rounded once: layout residuals kept in fp32 across denoise steps
 * (crossview_temporal_dit.py:491-494 with the ImageAdapter output cached) */
int dwm_add_f32_inplace(void* y, const float* x, int64_t n, void* stream);
/* y (fp32) += x (fp32), n % 4 == 0: cached layout residuals onto the fp32 hidden stream (crossview_temporal_dit.py:491-494) */
int dwm_add_f32_f32_inplace(float* y, const float* x, int64_t n, void* stream);

/* ------------------------------------------------------------------------
 * VAE blocks (diffusers AutoencoderKL, called at src/dwm/pipelines/ctsd.py:1213-1218,1634-1640)
 * on token-major activations x [I, P, C] (P = pixels per image).
 * ---------------------------------------------------------------------- */
/* torch.nn.GroupNorm(G, C, eps) [+ SiLU]: statistics over (P, C/G) per image and group (fp32,
 * `stats` = caller scratch of dwm_groupnorm_stats_floats(I, P, G) floats), y = (x - mean) * rstd * gamma + beta [then x*sigmoid(x)].
 * If out_map (rw > 0) is given, y is written into the zero-padded token grid that feeds a 3x3
 * implicit-GEMM convolution (borders must have been zeroed once by the caller).  C/G % 4 == 0. */
int dwm_groupnorm_silu(const void* x, void* y, int64_t I, int64_t P, int32_t C, int32_t G, float eps,
                       const void* gamma, const void* beta, int32_t silu, float* stats,
                       const dwm_rowmap2d* out_map, void* stream);

/* Same with an (image, pixel) -> row map: image i, pixel p lives in token row
 * (i / iv) * s_ihi + (i % iv) * s_ilo + (p / pn) * s_phi + p % pn  (iv = 0 / NULL: row = i*P + p).
 * diffusers TemporalResnetBlock's GroupNorm over [B*V, C, T, H, W] on the [(b t v), (h w), C] layout:
 * I = B*V, P = T*h*w, iv = V, pn = h*w, s_ihi = T*V*h*w, s_ilo = h*w, s_phi = V*h*w.  out_map then maps the
 * token row (not the image-local pixel).  C/G == 4 or >= 8. */
typedef struct dwm_gn_imgmap { int64_t iv, pn, s_ihi, s_ilo, s_phi; } dwm_gn_imgmap;
/* fp32 elements the `stats` scratch of the dwm_groupnorm_* entry points needs for I images of P pixels and G groups: the
 * final (sum, sumsq) per image and group plus the per-chunk partial sums that are added in a fixed order (the
 * statistics, and with them every GroupNorm output, are bit-reproducible from run to run). */
int64_t dwm_groupnorm_stats_floats(int64_t I, int64_t P, int32_t G);
int dwm_groupnorm_silu_mapped(const void* x, void* y, int64_t I, int64_t P, int32_t C, int32_t G, float eps,
                              const void* gamma, const void* beta, int32_t silu, float* stats,
                              const dwm_rowmap2d* out_map, const dwm_gn_imgmap* img_map, void* stream);

/* CogVideoXSpatialNorm3D (diffusers autoencoder_kl_cogvideox; decoder norms of AutoencoderKLCogVideoX, selected by
 * src/dwm/pipelines/ctsd.py:953-964):  y = silu?( GroupNorm(f) * conv_y(zq) + conv_b(zq) ) with zq nearest-resized to f.
 * f rows are ordered (frame t, video b, y, x), I = videos, P = frames*h*w; the 1x1x1 convolutions are evaluated at
 * latent resolution by the caller (one GEMM, N = 2C) and gathered here: f pixel (t, b, y, x) reads mod row
 * ((zt[t]*videos + b)*hz + (y >> shift))*wz + (x >> shift), columns [0, C) = conv_y, [C, 2C) = conv_b. */
typedef struct dwm_gn_zmap {
    const void* mod; int64_t ld_mod;
    int32_t frames, videos, h, w, shift;
    int32_t zt[32];                 /* f frame -> zq frame (nearest, incl. the separate first frame of odd clips) */
} dwm_gn_zmap;
int dwm_groupnorm_spatial(const void* x, void* y, int64_t I, int64_t P, int32_t C, int32_t G, float eps,
                          const void* gamma, const void* beta, int32_t silu, float* stats,
                          const dwm_rowmap2d* out_map, const dwm_gn_imgmap* img_map, const dwm_gn_zmap* zmap, void* stream);
/* the same in fp32 (x, y, gamma, beta, zmap->mod fp32; ld_mod % 4 == 0; libm SiLU): the fp32 accuracy path of the temporal VAE */
int dwm_groupnorm_spatial_f32(const float* x, float* y, int64_t I, int64_t P, int32_t C, int32_t G, float eps,
                              const float* gamma, const float* beta, int32_t silu, float* stats,
                              const dwm_rowmap2d* out_map, const dwm_gn_imgmap* img_map, const dwm_gn_zmap* zmap, void* stream);

/* out frame j = w0[j] * x[f0[j]] + w1[j] * x[f1[j]] on frames of `frame_elems` contiguous bf16 elements: the temporal
 * average pooling of CogVideoXDownsample3D (0.5 / 0.5, first frame of an odd clip kept) and the temporal nearest
 * upsampling of CogVideoXUpsample3D (1 / 0 with repeated sources).  n_out <= 64. */
typedef struct dwm_frame_mix { int32_t n_out; int32_t f0[64], f1[64]; float w0[64], w1[64]; } dwm_frame_mix;
int dwm_frame_mix_bf16(const void* x, void* y, int64_t frame_elems, const dwm_frame_mix* mix, void* stream);
int dwm_frame_mix_f32(const float* x, float* y, int64_t frame_elems, const dwm_frame_mix* mix, void* stream);   /* fp32 elements */

/* F.interpolate(scale_factor=2, mode="nearest") of token-major x [I, h, w, C], written into the
 * padded grid y [I, 2h+2, 2w+2, C] (diffusers Upsample2D before its 3x3 conv). */
int dwm_upsample2_padded(const void* x, void* y, int64_t I, int32_t h, int32_t w, int32_t C, void* stream);

/* copy compact token rows x [I*rh*rw, C] into the interior of the zero-bordered padded grid y
 * described by map (input staging of a 3x3 implicit-GEMM convolution). */
int dwm_pad_tokens(const void* x, void* y, int64_t rows, int32_t C, const dwm_rowmap2d* map, void* stream);

/* y[r, :L] = softmax(scale * x[r, :L]) (fp32 math, bf16 storage; single-head mid-block attention). */
int dwm_softmax_rows(const void* x, void* y, int64_t rows, int32_t L, int64_t ld, float scale, void* stream);

/* Explicit perspective modelling (RayEncoder / get_rays, src/dwm/models/crossview_temporal_dit.py:11-102): the 72 inputs of
 * RayEncoder.proj for every latent token of I images of h x w tokens.  cam fp32 [I, 21] = { inverse of the token-resolution
 * intrinsics (9, row-major), camera->reference-ego rotation (9), camera origin (3) };
 * out bf16 [I*h*w, ldo >= 72]: [ sin(o_d 2^k pi), d-major, k < 8 | cos | sin(ray_d 2^k pi), k < 4 | cos ], rest zero. */
int dwm_ray_features(const float* cam, int64_t I, int32_t h, int32_t w, void* out, int64_t ldo, void* stream);
/* the same features in fp32 (the fp32 accuracy path: RayEncoder.proj then runs through dwm_gemm_f32) */
int dwm_ray_features_f32(const float* cam, int64_t I, int32_t h, int32_t w, float* out, int64_t ldo, void* stream);

/* out = coef[g][0] * x + coef[g][1] * y, fp32, one coefficient pair per group of `group_elems` consecutive elements
 * (g = element / group_elems); `out` (fp32) and / or `out_bf16` receive the result.  DDPMScheduler.add_noise / get_velocity
 * with one timestep per (sample, frame, view) (src/dwm/schedulers/temporal_independent.py:8-45). */
int dwm_frame_affine(const float* x, const float* y, const float* coef, float* out, void* out_bf16, int64_t n,
                     int64_t group_elems, void* stream);

/* Classifier-free guidance + the tensor-timestep DDIM update (src/dwm/schedulers/temporal_independent.py:67-170) in one pass.
 * pred: model output, bf16 (pred_is_f32 = 0) or fp32, [2n] = (unconditional ; conditional) when cfg != 0, else [n];
 * latents fp32 [n] updated in place; coef fp32 [n / group_elems, 6] =
 *   { sqrt(a_t), sqrt(1 - a_t), sqrt(a_prev), sqrt(1 - a_prev - std^2), std, 0 } per (sample, frame, view) timestep;
 * prediction_type 0 epsilon | 1 sample | 2 v_prediction; clip_range > 0 clamps the predicted x0 (config.clip_sample);
 * use_clipped_model_output re-derives epsilon from the clipped x0; noise (fp32 [n], may be NULL) is the eta > 0 variance
 * noise; x0_out (may be NULL) receives pred_original_sample; model_in (may be NULL) the bf16 next model input (CFG-doubled
 * when cfg != 0). */
int dwm_cfg_ddim_step(const void* pred, int32_t pred_is_f32, int32_t cfg, float* latents, void* model_in, float* x0_out,
                      const float* noise, const float* coef, int64_t n, int64_t group_elems, float guidance,
                      int32_t prediction_type, float clip_range, int32_t use_clipped_model_output, void* stream);

/* classifier-free guidance + one step of a linear multistep scheduler on fp32 latents (SD 2.1 configs:
 * diffusers DPMSolverMultistepScheduler, dpmsolver++ / midpoint, called at ctsd.py:1548-1575):
 *   out = u + g (c - u);  x0 = kx * x + ko * out  (epsilon: kx = 1/alpha_t, ko = -sigma_t/alpha_t; v: kx = alpha_t, ko = -sigma_t)
 *   x <- A * x + B * x0 + C * x0_prev;  x0_prev <- x0;  model_in (optional, bf16 [2, n]) <- x
 * The scalar coefficients are the scheduler's per-step constants, computed on the host. */
int dwm_cfg_multistep(const void* pred, float* latents, float* x0_prev, void* model_in, int64_t n, float guidance,
                      float kx, float ko, float A, float B, float C, void* stream);

/* dst bf16 <- src fp32 (n % 4 == 0) */
int dwm_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);

/* ------------------------------------------------------------------------
 * Training: backward / optimizer kernels of the CTSD train step
 * (src/dwm/pipelines/ctsd.py:1195-1437: forward under autocast, loss.backward(), optimizer.step()).
 * They are the hand-written derivatives of the forward entry points above; reductions accumulate
 * with fp32 atomics into buffers the caller has zeroed.
 * ---------------------------------------------------------------------- */
/* out[c, r] = in[r, c], bf16; out is [cols, ld_out] with columns [rows, rows_pad) zero-filled.
 * Feeds the weight-gradient GEMM dW = dY^T X (both operands K-major = row-contiguous over tokens)
 * and the input-gradient GEMM dX = dY W (W^T refreshed once per optimizer step). */
int dwm_transpose_bf16(const void* in, int64_t ld_in, int64_t rows, int64_t cols,
                       void* out, int64_t ld_out, int64_t rows_pad, void* stream);

/* out[g, n] += sum over rows r with r / rows_per_group == g of a[r, n] * (b ? b[r, n] : 1)
 * (bias gradients, AdaLN shift / scale / gate gradients per image, mixer alpha gradient). */
int dwm_segsum(const void* a, int64_t lda, const void* b, int64_t ldb, int64_t rows, int64_t ncols,
               int64_t rows_per_group, float* out, int64_t ld_out, void* stream);
/* out[g, n] += sum of a[r, n] * (b[r, n] - b2[r, n]), the difference taken in fp32 before the product: the AlphaBlender
 * gradient d(alpha) = <dy, x_spatial - x_temporal> (crossview_temporal.py:68-72) in one pass over the three tensors. */
int dwm_segsum_diff(const void* a, int64_t lda, const void* b, int64_t ldb, const void* b2, int64_t ldb2, int64_t rows,
                    int64_t ncols, int64_t rows_per_group, float* out, int64_t ld_out, void* stream);

/* y = act(x) / dx = dy * act'(x); act = DWM_ACT_GELU_TANH | DWM_ACT_SILU; n % 8 == 0 */
int dwm_act_fwd(const void* x, void* y, int64_t n, int32_t act, void* stream);
int dwm_act_bwd(const void* x, const void* dy, void* dx, int64_t n, int32_t act, void* stream);

/* diffusers GEGLU on the un-packed projection u [rows, 2*inner] = [value | gate]:
 * g = value * gelu(gate) (erf form);  du = [dg * gelu(gate) | dg * value * gelu'(gate)] */
int dwm_geglu_fwd(const void* u, int64_t ldu, int64_t rows, int64_t inner, void* g, int64_t ldg, void* stream);
int dwm_geglu_bwd(const void* u, int64_t ldu, const void* dg, int64_t lddg, int64_t rows, int64_t inner,
                  void* du, int64_t lddu, void* stream);

/* out[r, n] = a[r, n] * (gate_a ? gate_a[r / rows_per_gate_a, n] : 1) * (coef_a ? coef_a[r / rows_per_coef_a] : 1)
 *           + (b ? b[r, n] * (coef_b ? coef_b[r / rows_per_coef_b] : 1) : 0)
 * = gated residual add (forward), gate * dy (its backward), AlphaBlender forward / backward
 * (crossview_temporal.py:68-72), plain residual add. */
typedef struct dwm_rowcombine_args {
    const void* a; int64_t lda;
    const void* gate_a; int64_t ld_gate_a; int64_t rows_per_gate_a;
    const float* coef_a; int64_t rows_per_coef_a;
    const void* b; int64_t ldb;
    const float* coef_b; int64_t rows_per_coef_b;
    void* out; int64_t ldo;
    int64_t rows; int64_t ncols;
} dwm_rowcombine_args;
int dwm_rowcombine(const dwm_rowcombine_args* args, void* stream);

/* Backward of dwm_layernorm.  xhat = normalise(x [+ addvec]); output 1 = xhat * gamma1 + ..,
 * gamma1 = weight * (1 + scale[g]) (either factor optional), output 2 = xhat * (1 + scale2[g]) + ..
 *   dx (+)= rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)),  dxhat = dy*gamma1 + dy2*gamma2
 *   dgamma[gg, :] += sum dy * xhat,  dbeta[gg, :] += sum dy   (gg = g if grad_per_group else 0:
 *   per-image AdaLN scale / shift gradients vs. LayerNorm weight / bias gradients); same for 2. */
typedef struct dwm_layernorm_bwd_args {
    const void* x; int64_t ldx;
    const void* addvec; int64_t ld_add; int64_t rows_per_add;
    const void* dy; int64_t lddy;
    const void* dy2; int64_t lddy2;            /* or NULL */
    void* dx; int64_t lddx; int32_t accumulate; /* accumulate != 0: dx += */
    int64_t rows; int32_t D; float eps;
    const void* weight;                        /* bf16 [D] or NULL */
    const void* scale; const void* scale2; int64_t ld_mod; int64_t rows_per_mod;
    float* dgamma; float* dbeta; float* dgamma2; float* dbeta2; int64_t ld_grad; int32_t grad_per_group;
} dwm_layernorm_bwd_args;
int dwm_layernorm_bwd(const dwm_layernorm_bwd_args* args, void* stream);

/* dwm_rmsnorm_heads that also returns rinv [rows, ncols/64] (fp32), and its backward, in place on
 * dy (-> dx): y is the normalised output of the forward (xhat * w; w must be non-zero),
 * dw[c] += sum_rows dy * xhat. */
int dwm_rmsnorm_heads_train(void* x, int64_t ldx, int64_t rows, int64_t ncols, const void* w, float eps,
                            float* rinv, void* stream);
int dwm_rmsnorm_heads_bwd(const void* y, int64_t ldy, const float* rinv, const void* w, void* dy, int64_t lddy,
                          int64_t rows, int64_t ncols, float* dw, void* stream);

/* torch.optim.AdamW step on fp32 master parameters (g is multiplied by grad_scale first);
 * p_bf16 (optional) receives the refreshed bf16 compute copy.  bias_corr{1,2} = 1 - beta^t. */
int dwm_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
              float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, float grad_scale,
              void* stream);

/* The same step for many tensors in ONE launch.  items / block_item / block_start are DEVICE arrays: block b updates elements
 * [block_start[b], min(block_start[b] + chunk, items[block_item[b]].n)) of tensor block_item[b]; every tensor needs
 * ceil(n / chunk) consecutive-start blocks.  All tensors share the hyper-parameters and the step count (bias corrections). */
typedef struct dwm_adamw_item {
    float* p; const float* g; float* m; float* v; void* p_bf16; int64_t n;
} dwm_adamw_item;
int dwm_adamw_multi(const dwm_adamw_item* items, const int32_t* block_item, const int64_t* block_start, int64_t n_blocks,
                    int64_t chunk, float lr, float beta1, float beta2, float eps, float weight_decay, float bias_corr1,
                    float bias_corr2, float grad_scale, void* stream);

/* Backward of dwm_groupnorm_silu / dwm_groupnorm_silu_mapped (the UNet's ResnetBlock2D / TemporalResnetBlock / TransformerModel
 * norms in the SD 2.1 training branch, src/dwm/pipelines/ctsd.py:1240-1253): x = the forward input (compact rows, through
 * img_map if given), dz = gradient of the forward OUTPUT read through dz_map (the padded grid the forward wrote, or NULL for
 * compact rows), dx [rows, C] bf16 = gradient of x (accumulate != 0: added to what dx holds), dgamma / dbeta fp32 [C] +=.
 * `stats`: caller scratch of 2 * dwm_groupnorm_stats_floats(I, P, G) floats (forward statistics are recomputed). */
int dwm_groupnorm_bwd(const void* x, const void* dz, void* dx, int64_t I, int64_t P, int32_t C, int32_t G, float eps,
                      const void* gamma, const void* beta, int32_t silu, int32_t accumulate, float* stats,
                      float* dgamma, float* dbeta, const dwm_rowmap2d* dz_map, const dwm_gn_imgmap* img_map, void* stream);

/* Block permutation: dst block (i0, i1, i2, i3) <- src block at i0*sstride[0] + i1*sstride[1] + i2*sstride[2] + i3*sstride[3]
 * (strides in blocks), blocks of `block_bytes` contiguous bytes (a multiple of 16; both buffers 16-byte aligned), dst dense in the
 * order (i0, i1, i2, i3).  The pack / unpack around the frame-shard all-to-all (opendwm_amd/sharding.py: "my frames, all token rows" <->
 * "all frames, my token rows" - no reference counterpart, the reference never shards a sample): one launch per direction instead of a
 * torch permute + contiguous / copy_ pair. */
typedef struct dwm_block_permute_args {
    const void* src;
    void* dst;
    int64_t block_bytes;
    int64_t n[4];
    int64_t sstride[4];
} dwm_block_permute_args;
int dwm_block_permute(const dwm_block_permute_args* args, void* stream);

/* y fp32 [rows, ldy] (+)= x bf16 [rows, ldx] */
int dwm_cast_bf16_to_f32(const void* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t cols,
                         int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------
 * fp32 accuracy path (BASELINE.json north_star: outputs "within 1e-3 rel fp32" of the reference's fp32 CPU path; the
 * reference runs the same graph in whatever dtype the caller / autocast selects, src/dwm/pipelines/ctsd.py:1189-1193).
 * Same argument structs and semantics as the bf16 entry points above with EVERY tensor pointer fp32 (masks stay bytes,
 * alpha stays fp32).  Not a throughput path: exact-fp32 MFMA for the attention contractions, libm-accurate
 * transcendental functions, and for the GEMM three bf16 MFMA products of a two-plane split of both operands.
 * ---------------------------------------------------------------------- */
/* C fp32 = epilogue(A fp32 [M, K] x W^T).  W: the PRE-SPLIT bf16 weight [N, 3K] = [ hi | lo | hi ] with hi = bf16(w),
 * lo = bf16(w - hi) (opendwm_amd.ops.split_weight); bias / gate / res / blend / rms_w fp32.
 * Implicit convolution as in dwm_gemm_bf16 (a_map / c_map / tap_shift, up to 27 taps): W = [ hi_t | lo_t | hi_t ] per tap t, taps in
 * groups of 9 (group g = [N, 9 * 3 * k_per_tap] contiguous, groups one after the other; <= 9 taps: one group = the plain [N, 3K]).
 * workspace (16-byte aligned): the two bf16 planes of A (4 * A rows * k_per_tap bytes, + 256) followed by the fp32 partial sums,
 * 4*M*N bytes per group of taps (more, if given, lets the call split K over ranges when the tile grid is small). */
int dwm_gemm_f32(const dwm_gemm_args* args, void* stream);
/* fp32-I/O forms of the token-major glue kernels of the SD 2.1 UNet / the 2-D VAE / the layout ImageAdapter (same argument meaning
 * as the bf16 entry points above; gamma / beta fp32): GroupNorm(+SiLU) incl. the row-mapped form of TemporalResnetBlock
 * (img_map may be NULL), nearest 2x upsample into a padded grid, padded-grid scatter, row softmax of the VAE's single-head
 * mid-block attention, pixel-unshuffle, 2x2 average pooling.  The reference runs these in whatever dtype the pipeline selects
 * (src/dwm/pipelines/ctsd.py:1189-1193); its CPU path - BASELINE.json configs[0] - is fp32. */
int dwm_groupnorm_silu_f32(const float* x, float* y, int64_t I, int64_t P, int32_t C, int32_t G, float eps,
                           const float* gamma, const float* beta, int32_t silu, float* stats,
                           const dwm_rowmap2d* out_map, const dwm_gn_imgmap* img_map, void* stream);
int dwm_upsample2_padded_f32(const float* x, float* y, int64_t I, int32_t h, int32_t w, int32_t C, void* stream);
int dwm_pad_tokens_f32(const float* x, float* y, int64_t rows, int32_t C, const dwm_rowmap2d* map, void* stream);
int dwm_softmax_rows_f32(const float* x, float* y, int64_t rows, int32_t L, int64_t ld, float scale, void* stream);
int dwm_unshuffle_tokens_f32(const float* x, int64_t I, int32_t C, int32_t H, int32_t W, int32_t r, float* out, int64_t ldo,
                             void* stream);
int dwm_avgpool2_tokens_f32(const float* x, int64_t I, int32_t h, int32_t w, int32_t C, float* out, void* stream);
/* dwm_cfg_multistep with an fp32 prediction and an fp32 model input: the guided DPM-Solver++ update of the SD 2.1 denoise loop on
 * the fp32 path (src/dwm/pipelines/ctsd.py:1536-1575 with diffusers DPMSolverMultistepScheduler.step) */
int dwm_cfg_multistep_f32(const float* pred, float* latents, float* x0_prev, float* model_in, int64_t n, float guidance,
                          float kx, float ko, float A, float B, float C, void* stream);
int dwm_layernorm_f32(const dwm_layernorm_args* args, void* stream);
/* strides in fp32 elements; head_dim 64; every mask / row-map / segment mode of dwm_attention_fwd; optional lse */
int dwm_attention_f32(const dwm_attn_args* args, void* stream);
int dwm_silu_f32(const float* x, float* y, int64_t n, void* stream);
int dwm_timestep_sinusoid_f32(const float* t, int64_t n, int32_t C, float* out, void* stream);
int dwm_patchify_f32(const float* x, int64_t I, int32_t C, int32_t H, int32_t W, int32_t p, float* out, int64_t ldo, void* stream);
int dwm_unpatchify_f32(const float* x, int64_t ldx, int64_t I, int32_t C, int32_t h, int32_t w, int32_t p, float* out, void* stream);
/* latents += dsigma * (u + g (c - u)) with pred = [uncond; cond] fp32; dsigma_group (or NULL): one step per group of
 * group_elems elements (diffusion forcing); model_in (or NULL): fp32 [2n], the next CFG-doubled model input */
int dwm_cfg_euler_step_f32(const float* pred, float* latents, float* model_in, int64_t n, float guidance, float dsigma,
                           const float* dsigma_group, int64_t group_elems, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DWM_HIP_H */
